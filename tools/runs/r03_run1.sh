R=$PWD
O=gpurun_out/r1
mkdir -p $O
# a step that times out or is killed ends the whole call (no further GPU step after a hang)
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
# 1. the one-sum question: default build, one-sum forced for two residues per thread, the same with the self-check
for lib in "" trrosettax2-dynamics_amd/csrc/_exp/libtrx2fold_onesum.so trrosettax2-dynamics_amd/csrc/_exp/libtrx2fold_onesum_check.so trrosettax2-dynamics_amd/libtrx2fold_check.so; do
  echo "== lib=$lib" >> $O/onesum.txt
  TRX2FOLD_LIB=${lib:+$R/$lib} run 120 python3 tools/selfcheck_fold.py $R 400 8 60 >> $O/onesum.txt 2>&1
  TRX2FOLD_LIB=${lib:+$R/$lib} run 120 python3 tools/selfcheck_fold.py $R 400 8 400 >> $O/onesum.txt 2>&1
done
TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libtrx2fold_check.so run 120 python3 tools/selfcheck_fold.py $R 150 64 100000 0 >> $O/onesum.txt 2>&1
cat $O/onesum.txt
# 2. library defaults for one call: lanes, compaction, hardware queues
for cfg in 2 3 4; do for lanes in 1 2; do
  run 200 python3 tools/percall.py $R $cfg $lanes 6 >> $O/percall.txt 2>&1
done; done
run 200 python3 tools/percall.py $R 2 2 6 2 >> $O/percall.txt 2>&1
run 200 python3 tools/percall.py $R 2 2 6 0 >> $O/percall.txt 2>&1
GPU_MAX_HW_QUEUES=8 run 200 python3 tools/percall.py $R 2 2 6 >> $O/percall.txt 2>&1
GPU_MAX_HW_QUEUES=8 run 200 python3 tools/percall.py $R 3 1 6 >> $O/percall.txt 2>&1
GPU_MAX_HW_QUEUES=8 run 200 python3 tools/percall.py $R 3 2 6 >> $O/percall.txt 2>&1
cat $O/percall.txt
# 3. tests
run 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
# 4. bench
run 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
