# step kernel: DPP scans, early loads, paired slices; pair epilogue DPP: full GPU suite, stamps, A/B against the previous build
O=gpurun_out/r18
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
TRX2FOLD_LIB=$X/libtrx2fold_stamp.so run 300 python3 tools/stamp_single_decoy.py $PWD 150 > $O/stamp150.txt 2>&1; echo "stamp rc=$?"; cat $O/stamp150.txt
for lib in "" $X/libtrx2fold_r17.so; do
  echo "== lib=$lib"
  for cfg in "2 2" "3 1" "4 2"; do
    TRX2FOLD_LIB=$lib run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1
  done
  TRX2FOLD_LIB=$lib run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1
done > $O/ab.txt 2>&1; cat $O/ab.txt
