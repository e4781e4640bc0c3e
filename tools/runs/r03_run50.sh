O=gpurun_out/r50
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "low_register" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
