# Round 4, run 24: who launches a single-decoy fold follows the number of live contexts: one target, batch mode at 2 / 3 / 16 targets in flight, shared-launch tests
O=gpurun_out/r04_run24
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 900 python3 -m pytest tests/test_gpu_shared_launch.py tests/test_gpu_boundary.py tests/test_restraint_variants.py -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt | cut -c1-200
run 300 python3 tools/e2e_single.py . 150 80 >> $O/single.txt 2>> $O/err.txt; run 300 python3 tools/e2e_single.py . 150 80 >> $O/single.txt 2>> $O/err.txt; cut -c1-200 $O/single.txt
run 600 python3 tools/e2e_batch.py . 150 16 40 2 3 16 > $O/batch.txt 2>> $O/err.txt; echo "batch rc=$?"; cut -c1-330 $O/batch.txt
TRX2_SHARED_LAUNCH=1 run 600 python3 tools/e2e_batch.py . 150 16 40 2 3 > $O/batch_forced.txt 2>> $O/err.txt; echo "batch forced rc=$?"; cut -c1-170 $O/batch_forced.txt
