O=gpurun_out/r45
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for t in 18 12 9 6 4; do
  echo "== TRX2_ROW_TARGET=$t"
  for cfg in "2 2" "3 1" "4 2"; do TRX2_ROW_TARGET=$t run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-140; done
  TRX2_ROW_TARGET=$t run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1
done > $O/row_target.txt 2>&1; cat $O/row_target.txt
