R=$PWD
O=gpurun_out/r4
mkdir -p $O
E=$R/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
# 1. the self-checking build first (the one-sum path at L=400), then the whole GPU suite
run 600 python3 -m pytest tests/test_gpu_selfcheck.py -m gpu -x -q -s > $O/selfcheck.txt 2>&1; echo "selfcheck rc=$?"; tail -12 $O/selfcheck.txt
run 1150 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
# 2. two concurrent single-decoy chains (the iteration phase of run_inference): does one slow the other, and is it the host?
run 200 python3 tools/two_single.py $R 150 1 6 >> $O/two_single.txt 2>&1
run 200 python3 tools/two_single.py $R 150 2 6 >> $O/two_single.txt 2>&1
TRX2_GRAPH=1 run 200 python3 tools/two_single.py $R 150 1 6 >> $O/two_single.txt 2>&1
TRX2_GRAPH=1 run 200 python3 tools/two_single.py $R 150 2 6 >> $O/two_single.txt 2>&1
GPU_MAX_HW_QUEUES=8 run 200 python3 tools/two_single.py $R 150 2 6 >> $O/two_single.txt 2>&1
cat $O/two_single.txt
# 3. remove_clash guard (DESIGN.md deviation 5): evaluations saved against outcome, 1024 decoys per map
run 300 python3 tools/outcome_sample.py $R 16 1000 > $O/outcome_default.txt 2>&1
TRX2FOLD_LIB=$E/libtrx2fold_guard.so run 300 python3 tools/outcome_sample.py $R 16 1000 > $O/outcome_guard.txt 2>&1
cat $O/outcome_default.txt $O/outcome_guard.txt
