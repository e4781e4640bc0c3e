# two chunks in flight: full GPU suite, A/B with TRX2_NO_PIPELINE=1
O=gpurun_out/r32
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for np in "" 1 "" 1; do
  echo "== TRX2_NO_PIPELINE=$np"
  for cfg in "2 2" "3 1" "4 2"; do
    if [ -z "$np" ]; then run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-150; else TRX2_NO_PIPELINE=1 run 300 python3 tools/percall.py $PWD $cfg 6 2>&1 | tail -1 | cut -c1-150; fi
  done
  if [ -z "$np" ]; then run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1; else TRX2_NO_PIPELINE=1 run 300 python3 tools/single_decoy_trace.py $PWD 150 1 8 2>&1 | tail -1; fi
done > $O/ab.txt 2>&1; cat $O/ab.txt
