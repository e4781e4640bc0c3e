# Round 4, run 9: segment cache for single-decoy folds only (fixed block <-> entry mapping): bitwise tests, then A/B cache off / on
O=gpurun_out/r04_run9
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_shared_launch.py tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for v in 0 1; do
  export TRX2_SEG_CACHE=$v
  run 300 python3 tools/e2e_single.py . 150 60 >> $O/single_c$v.txt 2>&1; echo "cache=$v single rc=$?"; tail -1 $O/single_c$v.txt
  run 300 python3 tools/e2e_single.py . 90 60 >> $O/single_c$v.txt 2>&1; tail -1 $O/single_c$v.txt
  run 300 python3 tools/e2e_batch.py . 150 16 40 16 > $O/batch16_c$v.txt 2>&1; echo "cache=$v batch rc=$?"; tail -1 $O/batch16_c$v.txt
  run 300 python3 tools/e2e_batch.py . 150 32 40 32 > $O/batch32_c$v.txt 2>&1; echo "cache=$v batch32 rc=$?"; tail -1 $O/batch32_c$v.txt
  run 300 python3 tools/shared_scaling.py . 150 1500 1 2 32 > $O/scaling_c$v.txt 2>&1; cat $O/scaling_c$v.txt
done
unset TRX2_SEG_CACHE
run 300 python3 tools/percall.py . 3 1 4 > $O/percall3.txt 2>&1; tail -1 $O/percall3.txt
