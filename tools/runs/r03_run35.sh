O=gpurun_out/r35
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 tools/pool_sweep.py $PWD 2 1280 64 128 192 256 320 384 > $O/pool2.txt 2>&1; cat $O/pool2.txt
run 600 python3 tools/pool_sweep.py $PWD 2 2560 192 256 320 > $O/pool2_2560.txt 2>&1; cat $O/pool2_2560.txt
run 600 python3 tools/pool_sweep.py $PWD 3 1280 64 128 192 256 > $O/pool3.txt 2>&1; cat $O/pool3.txt
