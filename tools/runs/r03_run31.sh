O=gpurun_out/r31
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "full_fold_outcome" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E 'final energy|passed|failed|Error|assert' $O/pytest.txt | cut -c1-400
