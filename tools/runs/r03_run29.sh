O=gpurun_out/r29
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1000 python3 tools/soak_shapes.py $PWD 60 1 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -70 $O/soak.txt
run 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
