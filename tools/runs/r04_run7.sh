# Round 4, run 7: XCD-grouped layout of shared pair launches (one fold = one XCD from eight folds on), per-context single-decoy shape
O=gpurun_out/r04_run7
mkdir -p $O
R=$PWD
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_shared_launch.py -m gpu -q -x > $O/pytest_shared.txt 2>&1; echo "pytest(shared) rc=$?"; tail -3 $O/pytest_shared.txt
for x in 0 1; do
  TRX2_XCD_GROUPS=$x run 300 python3 tools/e2e_batch.py . 150 16 40 16 > $O/batch16_xcd$x.txt 2>&1; echo "xcd=$x batch rc=$?"; tail -1 $O/batch16_xcd$x.txt
done
run 300 python3 tools/e2e_batch.py . 150 32 40 32 > $O/batch32.txt 2>&1; echo "batch32 rc=$?"; tail -1 $O/batch32.txt
run 300 python3 tools/e2e_batch.py . 150 2 40 1 > $O/batch1.txt 2>&1; echo "one target at a time rc=$?"; tail -1 $O/batch1.txt
