R=$PWD
O=gpurun_out/r5
mkdir -p $O
E=$R/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
# 1. the whole GPU suite: row plans, 512-thread step kernel for 256 < L <= 512
run 1150 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
# 2. row plan target (entries per slice and partner residue of a wave step)
for shape in "2 32" "2 64" "2 192" "3 64" "3 128" "4 16" "4 32"; do
  for t in 8 12 18 26 40; do
    echo "TARGET=$t" >> $O/target.txt
    TRX2_ROW_TARGET=$t run 200 python3 tools/pair_ab.py $R $shape >> $O/target.txt 2>&1
  done
done
cat $O/target.txt
# 3. one chain of run_inference, per-iteration timers
run 300 python3 tools/e2e_chain_profile.py $R 150 10 30 > $O/e2e_chain.txt 2>&1; cat $O/e2e_chain.txt
run 300 python3 tools/e2e_chain_profile.py $R 90 10 30 >> $O/e2e_chain.txt 2>&1; tail -1 $O/e2e_chain.txt
