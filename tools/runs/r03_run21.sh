O=gpurun_out/r21
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for a in "2 32" "2 64" "2 1" "3 64" "3 1" "4 16"; do
  TRX2FOLD_LIB=$X/libtrx2fold_stamp.so run 200 python3 tools/stamp_pair.py $PWD $a 2>&1 | tail -14
done > $O/stamp_pair.txt 2>&1; cat $O/stamp_pair.txt
