O=gpurun_out/r58
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for n in 1 2 3 4; do MULTI_B=1920 run 600 python3 tools/multi_ctx.py $PWD 2 $n 2 2>&1 | tail -1; done > $O/multi_big.txt 2>&1; cat $O/multi_big.txt
