# Round 4, run 20: topology tests with their limits; 1024-decoy outcome sample of the shipped constants (both protocols)
O=gpurun_out/r04_run20
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_topologies.py tests/test_gpu_boundary.py -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt | cut -c1-200
echo "## shipped build, default protocol (build_runs(fastrelax=True))" > $O/outcome.txt
run 600 python3 tools/outcome_sample.py . 16 1000 --fastrelax >> $O/outcome.txt 2>> $O/err.txt || exit 1
echo "## shipped build, --no-fastrelax" >> $O/outcome.txt
run 600 python3 tools/outcome_sample.py . 16 1000 >> $O/outcome.txt 2>> $O/err.txt || exit 1
cat $O/outcome.txt | cut -c1-300
