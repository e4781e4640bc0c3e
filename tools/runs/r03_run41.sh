O=gpurun_out/r41
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for n in 1 2 3 4 6 8; do run 300 python3 tools/two_single.py $PWD 150 $n 6 2>&1 | tail -1; done > $O/chains.txt 2>&1; cat $O/chains.txt
for n in 4 8; do GPU_MAX_HW_QUEUES=8 run 300 python3 tools/two_single.py $PWD 150 $n 6 2>&1 | tail -1; done > $O/chains_hwq8.txt 2>&1; cat $O/chains_hwq8.txt
