R=$PWD
O=gpurun_out/r6
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
# 1. relax stage tests first (new code), then the whole suite
run 600 python3 -m pytest tests/test_gpu_relax.py -m gpu -x -q -s > $O/relax.txt 2>&1; echo "relax rc=$?"; tail -12 $O/relax.txt
run 1150 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
# 2. outcome with and without the relax-lite stage, 1024 decoys per map
run 300 python3 tools/outcome_sample.py $R 16 1000 > $O/outcome_plain.txt 2>&1
run 400 python3 tools/outcome_sample.py $R 16 1000 fastrelax > $O/outcome_relax.txt 2>&1
cat $O/outcome_plain.txt $O/outcome_relax.txt
# 3. does an iteration's fold get longer as the feedback reshapes the map?  300 iterations of one chain
run 400 python3 tools/e2e_chain_profile.py $R 150 10 300 > $O/e2e_chain300.txt 2>&1; tail -c 600 $O/e2e_chain300.txt
# 4. bench
run 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.json
