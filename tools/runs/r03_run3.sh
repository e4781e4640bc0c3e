R=$PWD
O=gpurun_out/r3
mkdir -p $O
E=$R/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
# 1. the whole GPU suite on the default build (list kernel, uniform energy totals, one-sum path everywhere)
run 1150 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
# 2. restraint-loop forms: straight loop (list33, earlier build), visit lambda (lam = default), two entries per trip (u2)
for lib in list33 lam u2; do
  for shape in "2 32" "2 64" "2 192" "3 64" "3 128" "4 16" "4 32"; do
    TRX2FOLD_LIB=$E/libtrx2fold_$lib.so run 200 python3 tools/pair_ab.py $R $shape >> $O/pair_ab.txt 2>&1
  done
done
cat $O/pair_ab.txt
# 3. slices per row (the split of a row's list and of its b-range over workgroups), default build
for shape in "2 32" "2 64" "2 192" "3 64" "3 128" "4 16" "4 32"; do
  for ns in 1 2 3 4 6; do
    echo "NSPLIT=$ns" >> $O/nsplit.txt
    TRX2_NSPLIT=$ns run 200 python3 tools/pair_ab.py $R $shape >> $O/nsplit.txt 2>&1
  done
done
cat $O/nsplit.txt
# 4. bench
run 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
