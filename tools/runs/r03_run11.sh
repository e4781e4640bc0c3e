R=$PWD
O=gpurun_out/r11
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 400 python3 tests/tools/bench_slowdown_probe.py $R > $O/probe.txt 2>&1; cat $O/probe.txt
OMP_WAIT_POLICY=passive run 400 python3 tests/tools/bench_slowdown_probe.py $R > $O/probe_passive.txt 2>&1; cat $O/probe_passive.txt
