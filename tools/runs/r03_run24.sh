O=gpurun_out/r24
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for cfg in 2 3 4; do for n in 1 2 3 4; do run 300 python3 tools/multi_ctx.py $PWD $cfg $n 6 2>&1 | tail -1; done; done > $O/multi.txt 2>&1; cat $O/multi.txt
for cfg in 2 4; do for n in 4 8; do GPU_MAX_HW_QUEUES=8 run 300 python3 tools/multi_ctx.py $PWD $cfg $n 6 2>&1 | tail -1; done; done > $O/multi_hwq8.txt 2>&1; cat $O/multi_hwq8.txt
