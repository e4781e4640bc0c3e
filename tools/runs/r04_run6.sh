# Round 4, run 6: one-wave (k_pair1) against four-wave workgroups for single-decoy folds: alone, and in shared launches of a batch job
O=gpurun_out/r04_run6
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for v in 0 1; do
  if [ $v -eq 1 ]; then export TRX2_PAIR1_WG4=1; tag=wg4; else unset TRX2_PAIR1_WG4; tag=wave1; fi
  run 300 python3 tools/shared_scaling.py . 150 1500 1 2 8 32 64 > $O/scaling_$tag.txt 2>&1; echo "$tag scaling rc=$?"; cat $O/scaling_$tag.txt
  run 300 python3 tools/e2e_batch.py . 150 16 40 16 > $O/batch16_$tag.txt 2>&1; echo "$tag batch rc=$?"; tail -1 $O/batch16_$tag.txt
  run 300 python3 tools/e2e_batch.py . 150 2 40 1 > $O/batch1_$tag.txt 2>&1; echo "$tag single-target rc=$?"; tail -1 $O/batch1_$tag.txt
done
unset TRX2_PAIR1_WG4
run 300 python3 -m pytest tests/test_gpu_boundary.py -m gpu -q -s -k "summary or layout" > $O/pytest_boundary.txt 2>&1; echo "pytest rc=$?"; grep "best apo\|passed\|failed" $O/pytest_boundary.txt
