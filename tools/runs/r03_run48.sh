O=gpurun_out/r48
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 tools/pool_sweep.py $PWD 2 1280 192 256 320 448 640 > $O/p1280.txt 2>&1; cat $O/p1280.txt
run 900 python3 tools/pool_sweep.py $PWD 2 5120 640 960 1280 1920 2560 > $O/p5120.txt 2>&1; cat $O/p5120.txt
run 600 python3 tools/pool_sweep.py $PWD 3 2560 640 1280 > $O/p3.txt 2>&1; cat $O/p3.txt
