# Round 4, run 12: -ffp-contract=on (contraction decided by the source, the same in every instantiation of a kernel body): bitwise diagnostics, suite, timings
O=gpurun_out/r04_run12
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for ev in 300 0; do for w in 4 1; do for sh in 0 1; do for c in 0 1; do
  echo "evals=$ev waves=$w shared=$sh cache=$c"; TRX2_SEG_CACHE=$c timeout -k 5 120 python3 tools/diag_segcache.py . $sh $w $ev 2>&1 | grep "^(" | head -2 | tr '\n' ' '; echo
done; done; done; done > $O/diag.txt 2>&1
cat $O/diag.txt
run 1100 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
for c in 2 3 4; do l=2; if [ $c = 3 ]; then l=1; fi; run 300 python3 tools/percall.py . $c $l 4 >> $O/percall.txt 2>&1; tail -1 $O/percall.txt; done
for w in 4 1; do run 300 python3 tools/e2e_single.py . 150 60 10 $w >> $O/single.txt 2>&1; tail -1 $O/single.txt; done
run 300 python3 tools/e2e_batch.py . 150 16 40 16 > $O/batch16.txt 2>&1; tail -1 $O/batch16.txt
run 300 python3 tools/e2e_batch.py . 150 32 40 32 > $O/batch32.txt 2>&1; tail -1 $O/batch32.txt
