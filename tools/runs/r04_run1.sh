# Round 4, run 1: GPU suite with the default protocol switched to --fastrelax + the new iteration-phase parity test; default bench line.
O=gpurun_out/r04_run1
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_iteration_parity.py tests/test_gpu_cartesian.py tests/test_gpu_relax.py -m gpu -q -s > $O/pytest_new.txt 2>&1; echo "pytest(new) rc=$?"; tail -5 $O/pytest_new.txt
run 900 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_iteration_parity.py --deselect tests/test_gpu_cartesian.py --deselect tests/test_gpu_relax.py > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
run 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
