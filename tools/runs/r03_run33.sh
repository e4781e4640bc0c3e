O=gpurun_out/r33
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for a in "2 32" "2 32" "2 16" "2 64" "3 32"; do
  TRX2FOLD_LIB=$X/libtrx2fold_stamp.so run 200 python3 tools/stamp_pair.py $PWD $a 2>&1 | grep -v ' 0 cycles'
done > $O/stamp_pair.txt 2>&1; cat $O/stamp_pair.txt
