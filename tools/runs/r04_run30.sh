# Round 4, run 30: Cartesian role of chains of 257-512 residues in Gram form (one stored pair's registers at a time): parity tests at L=300/400, config 4 timing
O=gpurun_out/r04_run30
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_cartesian.py tests/test_gpu_selfcheck.py "tests/test_gpu_configs.py::test_config4_L400_B32_all_channels" "tests/test_gpu_configs.py::test_config5_eight_targets_B32_on_one_gpu" -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-250
run 300 python3 tools/percall.py . 4 2 4 >> $O/percall.txt 2>&1; tail -1 $O/percall.txt
run 300 python3 tools/percall.py . 4 2 4 >> $O/percall.txt 2>&1; tail -1 $O/percall.txt
run 300 python3 tools/diag/step_time.py . 400 16 100000 >> $O/step.txt 2>&1; tail -1 $O/step.txt
run 300 python3 tools/diag/step_time.py . 300 16 100000 >> $O/step.txt 2>&1; tail -1 $O/step.txt
