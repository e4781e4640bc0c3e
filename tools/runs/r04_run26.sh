# Round 4, run 26: outcome sample with the two-sample reading (spread among our draws against the spread of the reference's two)
O=gpurun_out/r04_run26
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
echo "## shipped build, default protocol (build_runs(fastrelax=True))" > $O/outcome.txt
run 900 python3 tools/outcome_sample.py . 16 1000 --fastrelax >> $O/outcome.txt 2>> $O/err.txt || exit 1
echo "## shipped build, --no-fastrelax" >> $O/outcome.txt
run 900 python3 tools/outcome_sample.py . 16 1000 >> $O/outcome.txt 2>> $O/err.txt || exit 1
cut -c1-400 $O/outcome.txt
