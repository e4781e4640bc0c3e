# Round 4, run 3: where does the shared-launch batch spend its time?  kernel trace of the batch job, engine statistics, cost model samples
O=gpurun_out/r04_run3
mkdir -p $O
R=$PWD
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 300 python3 tools/e2e_batch.py . 150 8 40 8 > $O/batch8.txt 2>&1; echo "batch rc=$?"; tail -2 $O/batch8.txt
cd /tmp; export TMPDIR=/tmp
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/tools/e2e_batch.py $R 150 8 40 8 > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/batch8_kernel_stats.csv; head -12 $O/batch8_kernel_stats.csv; rm -rf $O/prof
run 600 python3 tools/fit_cost_model.py . $O/cost_model.json > $O/cost_model.txt 2>&1; echo "fit rc=$?"; tail -3 $O/cost_model.txt
