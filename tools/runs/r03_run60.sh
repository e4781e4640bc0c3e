O=gpurun_out/r60
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for v in 9999 1; do
  echo "== TRX2_STEP_LOWREG_MIN=$v"
  TRX2_STEP_LOWREG_MIN=$v run 600 python3 tools/pool_sweep.py $PWD 2 1280 64 96 128 160 192 256
  TRX2_STEP_LOWREG_MIN=$v run 600 python3 tools/pool_sweep.py $PWD 3 1280 64 128 192
done > $O/lowreg_min.txt 2>&1; cat $O/lowreg_min.txt
