# Round 4, run 25: long shape soak on the closing build (two seeds) -- every fold status 0 / 2, finite, bit for bit repeatable
O=gpurun_out/r04_run25
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1100 python3 tools/soak_shapes.py . 100 21 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -2 $O/soak.txt | cut -c1-200
