R=$PWD
O=gpurun_out/r2
mkdir -p $O
E=$R/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
# 1. parity of the list kernel (the new default build): the whole GPU suite
run 1100 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
# 2. pair-kernel A/B: round-2 kernel (base) against the list kernel at three occupancy targets
for lib in base list22 list24 list33; do
  for shape in "2 32" "2 64" "2 192" "3 64" "3 128" "4 16" "4 32"; do
    TRX2FOLD_LIB=$E/libtrx2fold_$lib.so run 200 python3 tools/pair_ab.py $R $shape >> $O/pair_ab.txt 2>&1
  done
done
cat $O/pair_ab.txt
# 3. single-decoy folds (the iteration phase of run_inference): base vs list
for lib in base list24; do
  TRX2FOLD_LIB=$E/libtrx2fold_$lib.so run 200 python3 tools/single_decoy_trace.py $R 150 1 8 >> $O/single.txt 2>&1
  TRX2FOLD_LIB=$E/libtrx2fold_$lib.so run 200 python3 tools/single_decoy_trace.py $R 90 1 8 >> $O/single.txt 2>&1
done
cat $O/single.txt
cd /tmp; export TMPDIR=/tmp
TRX2FOLD_LIB=$E/libtrx2fold_list24.so run 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sd1 -- python3 $R/tools/single_decoy_trace.py $R 150 1 4 > $R/$O/single_prof.log 2>&1
f=$(find /tmp/sd1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $R/$O/single_decoy_L150_kernel_stats.csv && cut -c1-160 $R/$O/single_decoy_L150_kernel_stats.csv
cd $R
# 4. the one-sum question: decoy 0's line-search record, nine sums against one sum
TRX2FOLD_LIB=$E/libtrx2fold_dbg9.so run 120 python3 tools/dbg_linesearch.py $R 400 70 > $O/dbg9.txt 2>&1
TRX2FOLD_LIB=$E/libtrx2fold_dbg1.so run 120 python3 tools/dbg_linesearch.py $R 400 70 > $O/dbg1.txt 2>&1
head -30 $O/dbg9.txt; head -60 $O/dbg1.txt
