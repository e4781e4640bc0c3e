# Round 4, run 19: persistent second-lane host thread, non-bundle synthetic targets (tests/test_gpu_topologies.py), batch mode against targets in flight
O=gpurun_out/r04_run19
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_topologies.py -m gpu -q -s > $O/topologies.txt 2>&1; echo "topologies rc=$?"; grep -E "L=|passed|failed|Error|assert" $O/topologies.txt | cut -c1-400 | tail -20
run 1100 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_topologies.py > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-200
run 600 python3 tools/e2e_batch.py . 150 32 40 8 16 32 > $O/batch.txt 2> $O/batch.err; echo "batch rc=$?"; cut -c1-170 $O/batch.txt
