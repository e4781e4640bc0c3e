#!/bin/bash
# Round 5: every gpurun command sequence of the round, one function per call.  usage (on the GPU box): bash tools/r05_runs.sh runN
R=$PWD
run1() {  # tolerance sweep with the relax stage on + step-kernel phase stamps at 1 and 32 decoys per launch (item 3)
  O=$R/gpurun_out/r05_run1; mkdir -p $O
  TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libtrx2fold_stamp.so timeout -k 10 120 python3 tools/stamp_single_decoy.py $R 150 > $O/stamp_step_single_L150.txt 2>&1; echo "stamp single rc=$?"
  TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libtrx2fold_stamp.so timeout -k 10 120 python3 tools/stamp_chain.py $R 2 32 > $O/stamp_step_c2_32.txt 2>&1; echo "stamp c2 rc=$?"
  TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libtrx2fold_stamp.so timeout -k 10 120 python3 tools/stamp_chain.py $R 4 16 > $O/stamp_step_c4_16.txt 2>&1; echo "stamp c4 rc=$?"
  timeout -k 10 900 python3 tools/tol_sweep_relax.py $R 1024 1000 all > $O/tol_sweep.txt 2>&1; echo "sweep rc=$?"
  tail -3 $O/tol_sweep.txt | cut -c1-300
}
run2() {  # where the evaluations go: per-run counts of the default protocol
  O=$R/gpurun_out/r05_run2; mkdir -p $O
  timeout -k 10 300 python3 tools/run_profile.py $R 256 90 > $O/run_profile_L90.txt 2>&1; echo "rc=$?"
  timeout -k 10 300 python3 tools/run_profile.py $R 64 150 > $O/run_profile_c2.txt 2>&1; echo "rc=$?"
  timeout -k 10 300 python3 tools/run_profile.py $R 64 151 > $O/run_profile_c3.txt 2>&1; echo "rc=$?"
}
run3() {  # warm first step (TRX2_WARM_START): parity tests, per-run profile, short tolerance sweep on 2 x 1024 decoys
  O=$R/gpurun_out/r05_run3; mkdir -p $O
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cartesian.py tests/test_gpu_relax.py tests/test_gpu_selfcheck.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
  timeout -k 10 300 python3 tools/run_profile.py $R 256 90 > $O/run_profile_L90.txt 2>&1; echo "rc=$?"
  timeout -k 10 300 python3 tools/run_profile.py $R 64 150 > $O/run_profile_c2.txt 2>&1; echo "rc=$?"
  timeout -k 10 600 python3 tools/tol_sweep_relax.py $R 2048 1000 short > $O/tol_sweep_short.txt 2>&1; echo "sweep rc=$?"
}
run4() {  # which runs may start warm
  O=$R/gpurun_out/r05_run4; mkdir -p $O
  timeout -k 10 900 python3 tools/warm_sweep.py $R 2048 > $O/warm_sweep.txt 2>&1; echo "rc=$?"
}
run5() {  # relax tolerance scale, fine steps, warm starts on; 4096 decoys per cell and map
  O=$R/gpurun_out/r05_run5; mkdir -p $O
  timeout -k 10 900 python3 tools/tol_sweep_relax.py $R 4096 1000 fine > $O/tol_sweep_fine.txt 2>&1; echo "rc=$?"
}
"$@"
