O=gpurun_out/r59
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for v in "" 1 "" 1; do
  echo "== TRX2_STEP_LOWREG_ALWAYS=$v"
  if [ -z "$v" ]; then
    for cfg in "2 2" "3 1"; do run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-140; done; run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1; run 300 python3 tools/pool_sweep.py $PWD 2 1280 192
  else
    for cfg in "2 2" "3 1"; do TRX2_STEP_LOWREG_ALWAYS=1 run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-140; done; TRX2_STEP_LOWREG_ALWAYS=1 run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1; TRX2_STEP_LOWREG_ALWAYS=1 run 300 python3 tools/pool_sweep.py $PWD 2 1280 192
  fi
done > $O/lowreg_always.txt 2>&1; cat $O/lowreg_always.txt
