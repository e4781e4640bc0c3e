O=gpurun_out/r65
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1000 python3 tools/soak_shapes.py $PWD 25 7 big > $O/soak_big.txt 2>&1; echo "soak rc=$?"; tail -28 $O/soak_big.txt
