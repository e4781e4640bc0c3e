O=gpurun_out/r30
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 tools/outcome_sample.py $PWD 16 1000 > $O/outcome.txt 2>&1; echo "rc=$?"; cat $O/outcome.txt
run 600 python3 tools/outcome_sample.py $PWD 16 1000 --fastrelax > $O/outcome_relax.txt 2>&1; echo "rc=$?"; cat $O/outcome_relax.txt
