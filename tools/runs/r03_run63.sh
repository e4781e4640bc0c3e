O=gpurun_out/r63
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for v in 9999 1; do
  echo "== TRX2_STEP_LOWREG_MIN=$v"
  POOL_L=90 TRX2_STEP_LOWREG_MIN=$v run 600 python3 tools/pool_sweep.py $PWD 3 1280 16 32 64 96 128
  POOL_L=90 TRX2_STEP_LOWREG_MIN=$v run 600 python3 tools/pool_sweep.py $PWD 2 1280 32 64 128
done > $O/short_small.txt 2>&1; cat $O/short_small.txt
