# LDS reserve of the step launch at small shapes, with the Cartesian role in Gram form
O=gpurun_out/r15
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for res in "" 0 29696; do
  echo "== TRX2_STEP_LDS_RESERVE=$res"
  for cfg in "2 2" "3 1" "4 2"; do
    if [ -z "$res" ]; then run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1; else TRX2_STEP_LDS_RESERVE=$res run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1; fi
  done
done > $O/reserve.txt 2>&1; cat $O/reserve.txt
