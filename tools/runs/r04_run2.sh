# Round 4, run 2: shared launches -- bitwise tests, the iteration-parity limits, batch mode on one GPU
O=gpurun_out/r04_run2
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_shared_launch.py -m gpu -q -s -x > $O/pytest_shared.txt 2>&1; echo "pytest(shared) rc=$?"; tail -5 $O/pytest_shared.txt
run 600 python3 -m pytest tests/test_gpu_iteration_parity.py tests/test_gpu_cartesian.py tests/test_gpu_boundary.py -m gpu -q -s > $O/pytest_new.txt 2>&1; echo "pytest(new) rc=$?"; tail -5 $O/pytest_new.txt
run 600 python3 tools/e2e_batch.py . 150 8 80 8 > $O/batch8.txt 2>&1; echo "batch rc=$?"; tail -5 $O/batch8.txt
run 600 python3 tools/e2e_batch.py . 150 16 80 16 > $O/batch16.txt 2>&1; echo "batch rc=$?"; tail -5 $O/batch16.txt
TRX2_ENGINE_STREAMS=1 run 600 python3 tools/e2e_batch.py . 150 8 80 8 > $O/batch8_1eng.txt 2>&1; echo "batch rc=$?"; tail -5 $O/batch8_1eng.txt
