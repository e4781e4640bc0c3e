R=$PWD
O=gpurun_out/r10
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 300 python3 tools/queue_collision.py $R 150 64 > $O/collision.txt 2>&1; cat $O/collision.txt
