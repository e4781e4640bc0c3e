O=gpurun_out/r25
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for lib in $X/libtrx2fold_stamp.so $X/libtrx2fold_stamp_nosb.so; do
  echo "== $lib"; TRX2FOLD_LIB=$lib run 300 python3 tools/stamp_single_decoy.py $PWD 150 2>&1 | tail -12
done > $O/stamps.txt 2>&1; cat $O/stamps.txt
for lib in "" $X/libtrx2fold_nosb.so "" $X/libtrx2fold_nosb.so; do
  echo "== lib=$lib"
  for cfg in "2 2" "3 1"; do
    TRX2FOLD_LIB=$lib run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-150
  done
done > $O/ab.txt 2>&1; cat $O/ab.txt
