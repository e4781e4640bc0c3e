O=gpurun_out/r36
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 tools/pool_sweep.py $PWD 2 2560 320 384 448 512 640 > $O/pool2_2560.txt 2>&1; cat $O/pool2_2560.txt
run 600 python3 tools/pool_sweep.py $PWD 2 1280 320 448 512 640 > $O/pool2_1280.txt 2>&1; cat $O/pool2_1280.txt
run 600 python3 tools/pool_sweep.py $PWD 2 640 128 192 256 320 > $O/pool2_640.txt 2>&1; cat $O/pool2_640.txt
run 600 python3 tools/pool_sweep.py $PWD 3 1280 320 384 448 > $O/pool3.txt 2>&1; cat $O/pool3.txt
