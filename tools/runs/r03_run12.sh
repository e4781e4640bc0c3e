R=$PWD
O=gpurun_out/r12
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 400 python3 tests/tools/bench_slowdown_probe.py $R > $O/probe.txt 2>&1; cat $O/probe.txt
run 300 python3 tools/queue_collision.py $R 150 64 > $O/collision.txt 2>&1; cat $O/collision.txt
run 1150 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
run 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench rc=$?"; tail -c 200 $O/bench_s20.json
