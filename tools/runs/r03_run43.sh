O=gpurun_out/r43
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
run 900 python3 tools/e2e_batch.py $PWD 150 4 80 1 2 4 > $O/e2e_batch.txt 2>&1; cat $O/e2e_batch.txt
