O=gpurun_out/r56
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for L in 220 256; do
  POOL_L=$L run 600 python3 tools/pool_sweep.py $PWD 3 1280 640
  POOL_L=$L TRX2_STEP_ONE_PER_CU=1 run 600 python3 tools/pool_sweep.py $PWD 3 1280 640
done > $O/long.txt 2>&1; cat $O/long.txt
