O=gpurun_out/r20
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for lib in "" $X/libtrx2fold_r17.so $X/libtrx2fold_base.so; do
  TRX2FOLD_LIB=$lib run 500 python3 tests/tools/track_L400.py $PWD 2>&1 | tail -5
done > $O/track.txt 2>&1; cat $O/track.txt
