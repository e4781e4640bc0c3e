# Round 4, run 8: segment cache of the pair kernel: parity suite, then A/B (cache off / on / on with two waves per SIMD for the all-channel 64- and 1-decoy shapes)
O=gpurun_out/r04_run8
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1000 python3 -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for v in off on w2; do
  unset TRX2_SEG_CACHE TRX2FOLD_LIB
  if [ $v = off ]; then export TRX2_SEG_CACHE=0; fi
  if [ $v = w2 ]; then export TRX2FOLD_LIB=$PWD/trrosettax2-dynamics_amd/libtrx2fold_w2.so; fi
  for c in 2 3 4; do
    l=2; if [ $c = 3 ]; then l=1; fi
    run 300 python3 tools/percall.py . $c $l 4 >> $O/percall_$v.txt 2>&1; echo "$v config $c rc=$?"; tail -1 $O/percall_$v.txt
  done
  run 300 python3 tools/shared_scaling.py . 150 1500 1 32 >> $O/scaling_$v.txt 2>&1; tail -2 $O/scaling_$v.txt
  run 300 python3 tools/e2e_batch.py . 150 16 40 16 > $O/batch16_$v.txt 2>&1; echo "$v batch rc=$?"; tail -1 $O/batch16_$v.txt
done
