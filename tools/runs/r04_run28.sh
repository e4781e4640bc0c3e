# Round 4, run 28: the closing run's torsion-potential weight scaled (its omega / bonded / repulsion / hydrogen-bond terms kept)
O=gpurun_out/r04_run28
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for sc in 0 0.3; do
  echo "## closing run: rama weight x $sc" >> $O/outcome.txt
  OUTCOME_CLOSING_RAMA=$sc run 900 python3 tools/outcome_sample.py . 8 1000 --fastrelax >> $O/outcome.txt 2>> $O/err.txt || exit 1
done
cut -c1-330 $O/outcome.txt
