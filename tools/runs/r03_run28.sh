O=gpurun_out/r28
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1100 python3 -m pytest tests/test_gpu_bench.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
