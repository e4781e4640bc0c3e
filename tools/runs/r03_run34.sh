# contact scan's first batch requested in the prologue: parity subset, stamps, A/B
O=gpurun_out/r34
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for a in "2 32" "2 64" "3 64"; do
  TRX2FOLD_LIB=$X/libtrx2fold_stamp.so run 200 python3 tools/stamp_pair.py $PWD $a 2>&1 | grep -v ' 0 cycles'
done > $O/stamp_pair.txt 2>&1; cat $O/stamp_pair.txt
for lib in "" $X/libtrx2fold_r32.so "" $X/libtrx2fold_r32.so; do
  echo "== lib=$lib"
  for cfg in "2 2" "3 1" "4 2"; do
    TRX2FOLD_LIB=$lib run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-150
  done
  TRX2FOLD_LIB=$lib run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1
done > $O/ab.txt 2>&1; cat $O/ab.txt
