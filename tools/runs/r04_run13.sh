# Round 4, run 13: the re-calibrated surrogate constants (omega 0.02, bonded x 0.4, guard offset -1.3): GPU suite, then timings
O=gpurun_out/r04_run13
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 1150 python3 -m pytest tests -m gpu -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.txt | cut -c1-300
for c in 2 3 4; do l=2; if [ $c = 3 ]; then l=1; fi; run 300 python3 tools/percall.py . $c $l 4 >> $O/percall.txt 2>&1; tail -1 $O/percall.txt; done
run 300 python3 tools/outcome_sample.py . 16 5000 > $O/outcome_norelax.txt 2>&1; cat $O/outcome_norelax.txt
