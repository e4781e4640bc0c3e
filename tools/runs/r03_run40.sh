# torsion history in rows L long; two step workgroups per CU for launches of more than 256 slots: GPU suite, A/B
O=gpurun_out/r40
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for one in "" 1 "" 1; do
  echo "== TRX2_STEP_ONE_PER_CU=$one"
  if [ -z "$one" ]; then
    run 600 python3 tools/pool_sweep.py $PWD 2 1280 320 640; run 600 python3 tools/pool_sweep.py $PWD 3 1280 640; run 300 python3 tools/percall.py $PWD 2 2 6 | cut -c1-140
  else
    TRX2_STEP_ONE_PER_CU=1 run 600 python3 tools/pool_sweep.py $PWD 2 1280 320 640; TRX2_STEP_ONE_PER_CU=1 run 600 python3 tools/pool_sweep.py $PWD 3 1280 640; TRX2_STEP_ONE_PER_CU=1 run 300 python3 tools/percall.py $PWD 2 2 6 | cut -c1-140
  fi
done > $O/ab.txt 2>&1; cat $O/ab.txt
