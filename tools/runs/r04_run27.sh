# Round 4, run 27: the closing unrestrained minimisation of the relax stage shortened or dropped -- global and torsion-level outcome
O=gpurun_out/r04_run27
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for it in 0 10 30; do
  echo "## closing run: $it iterations (0 = dropped)" >> $O/outcome.txt
  OUTCOME_CLOSING_ITERS=$it run 900 python3 tools/outcome_sample.py . 8 1000 --fastrelax >> $O/outcome.txt 2>> $O/err.txt || exit 1
done
cut -c1-330 $O/outcome.txt
