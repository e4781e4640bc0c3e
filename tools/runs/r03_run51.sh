# Cartesian role's arrays in dynamic LDS (static 35 -> 9 KB): suite, A/B against the previous build at small and large shapes
O=gpurun_out/r51
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for lib in "" $X/libtrx2fold_r50.so; do
  echo "== lib=$lib"
  for cfg in "2 2" "3 1" "4 2"; do TRX2FOLD_LIB=$lib run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-140; done
  TRX2FOLD_LIB=$lib run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1
  TRX2FOLD_LIB=$lib run 600 python3 tools/pool_sweep.py $PWD 2 1280 192 640
  TRX2FOLD_LIB=$lib run 600 python3 tools/pool_sweep.py $PWD 3 1280 640
done > $O/ab.txt 2>&1; cat $O/ab.txt
