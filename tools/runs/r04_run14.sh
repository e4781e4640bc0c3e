# Round 4, run 14: contact walk with the next contact's residue requested ahead: at three waves per SIMD (spills), at two, and without
O=gpurun_out/r04_run14
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for v in nopf pf3 pf2; do
  export TRX2FOLD_LIB=$PWD/trrosettax2-dynamics_amd/_ab/libtrx2fold_$v.so
  for c in 2 3 4; do l=2; if [ $c = 3 ]; then l=1; fi; run 300 python3 tools/percall.py . $c $l 4 >> $O/percall_$v.txt 2>&1; echo "$v $(tail -1 $O/percall_$v.txt)"; done
  run 300 python3 tools/pool_sweep.py . 2 1280 640 >> $O/pool_$v.txt 2>&1; echo "$v $(tail -1 $O/pool_$v.txt)"
  run 300 python3 tools/pool_sweep.py . 3 640 320 >> $O/pool_$v.txt 2>&1; echo "$v $(tail -1 $O/pool_$v.txt)"
  run 300 python3 tools/e2e_single.py . 150 60 >> $O/single_$v.txt 2>&1; echo "$v $(tail -1 $O/single_$v.txt)"
  run 300 python3 tools/e2e_batch.py . 150 16 40 16 > $O/batch16_$v.txt 2>&1; echo "$v $(tail -1 $O/batch16_$v.txt | cut -c1-140)"
done
