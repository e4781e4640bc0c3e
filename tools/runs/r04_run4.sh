# Round 4, run 4: one-wave-per-row pair kernel for single-decoy folds (k_pair1): bitwise shared-launch test, GPU suite, batch scaling, kernel trace
O=gpurun_out/r04_run4
mkdir -p $O
R=$PWD
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 600 python3 -m pytest tests/test_gpu_shared_launch.py -m gpu -q -x > $O/pytest_shared.txt 2>&1; echo "pytest(shared) rc=$?"; tail -3 $O/pytest_shared.txt
run 300 python3 tools/e2e_batch.py . 150 8 40 8 > $O/batch8.txt 2>&1; echo "batch rc=$?"; tail -1 $O/batch8.txt
run 300 python3 tools/e2e_batch.py . 150 16 40 16 > $O/batch16.txt 2>&1; echo "batch rc=$?"; tail -1 $O/batch16.txt
run 300 python3 tools/e2e_batch.py . 150 32 40 32 > $O/batch32.txt 2>&1; echo "batch rc=$?"; tail -1 $O/batch32.txt
cd /tmp; export TMPDIR=/tmp
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/tools/e2e_batch.py $R 150 16 40 16 > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/batch16_kernel_stats.csv; head -8 $O/batch16_kernel_stats.csv; rm -rf $O/prof
run 1000 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_shared_launch.py > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
