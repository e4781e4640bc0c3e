O=gpurun_out/r37
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 tools/pool_sweep.py $PWD 2 5120 640 960 1280 2560 > $O/pool2_5120.txt 2>&1; cat $O/pool2_5120.txt
run 600 python3 tools/pool_sweep.py $PWD 2 2560 960 1280 > $O/pool2_2560.txt 2>&1; cat $O/pool2_2560.txt
run 600 python3 tools/pool_sweep.py $PWD 3 2560 640 1280 > $O/pool3.txt 2>&1; cat $O/pool3.txt
