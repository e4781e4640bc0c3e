# config-4 tracking ratio across builds (is 0.948 a regression or the scatter of a 16-decoy ratio?)
O=gpurun_out/r19
mkdir -p $O
X=$PWD/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for lib in "" $X/libtrx2fold_r17.so $X/libtrx2fold_twoloop.so $X/libtrx2fold_base.so; do
  echo "== lib=$lib"
  TRX2FOLD_LIB=$lib run 400 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -s -k "config4 or config3 or config2" 2>&1 | grep -E 'tracking|passed|failed|Error' | cut -c1-220
done > $O/tracking.txt 2>&1; cat $O/tracking.txt
