O=gpurun_out/r62
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_cartesian.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
for v in 9999 1; do
  echo "== TRX2_STEP_LOWREG_MIN=$v"
  POOL_L=90 TRX2_STEP_LOWREG_MIN=$v run 600 python3 tools/pool_sweep.py $PWD 3 2560 128 192 256 320 512 640 1280
  POOL_L=128 TRX2_STEP_LOWREG_MIN=$v run 600 python3 tools/pool_sweep.py $PWD 2 1280 256 640
done > $O/short.txt 2>&1; cat $O/short.txt
