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
run6() {  # fitted rama / omega terms: device against oracle, then outcome A/B against the rounds-1-4 terms (2 x 2048 decoys per cell)
  O=$R/gpurun_out/r05_run6; mkdir -p $O
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cartesian.py tests/test_gpu_relax.py -q -m gpu -k "eval or tracks or short or relax" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
  for sc in 1.0,0.4 0.4,0.4 1.0,1.0; do echo "== fitted terms, TRX2_SF_FA_SCALE=$sc"; TRX2_SF_FA_SCALE=$sc timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 2048 1000 model; done > $O/model_ab.txt 2>&1
  echo "== rounds 1-4 terms (TRX2_BACKBONE_FIT=0), TRX2_SF_FA_SCALE=0.4,0.4" >> $O/model_ab.txt
  TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libtrx2fold_nofit.so TRX2_SF_FA_SCALE=0.4,0.4 timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 2048 1000 model >> $O/model_ab.txt 2>&1
  echo "model rc=$?"
}
run7() {  # model scan: rama fit alone, fitted omega at stiffness 2 / 3 / 4 / 6, relax-stage scale 0.4 / 1.0 (2 x 2048 decoys per cell)
  O=$R/gpurun_out/r05_run7; mkdir -p $O; : > $O/model_scan.txt
  for v in nofit rama o2 o3 o4 o6 omegaonly4; do for sc in 0.4,0.4 1.0,0.4; do
    lib=$R/trrosettax2-dynamics_amd/libv_$v.so; [ $v = nofit ] && lib=$R/trrosettax2-dynamics_amd/libtrx2fold_nofit.so
    echo "== $v TRX2_SF_FA_SCALE=$sc" >> $O/model_scan.txt
    TRX2FOLD_LIB=$lib TRX2_SF_FA_SCALE=$sc timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 2048 1000 model 2>&1 | grep -v "^#" >> $O/model_scan.txt
  done; done
  echo "rc=$?"
}
run8() {  # model scan 2: shrinkage of the fitted rama surface / helix term (TRX2_RAMA_SCAN), omega stiffness 1 / 2 / 4; twisted peptides counted without the terminus
  O=$R/gpurun_out/r05_run8; mkdir -p $O; : > $O/model_scan2.txt
  for v in o1 o2 o4; do echo "== $v (rama fit as fitted), TRX2_SF_FA_SCALE=1.0,0.4" >> $O/model_scan2.txt
    TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libv_$v.so TRX2_SF_FA_SCALE=1.0,0.4 timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 2048 1000 model 2>&1 | grep -v "^#" >> $O/model_scan2.txt; done
  for sc in 0,0 0,1 0.25,1 0.5,1 0.5,0.5 1,0 0.25,0.25; do echo "== rama-only lib (rounds 1-4 omega), TRX2_RAMA_SCAN=$sc" >> $O/model_scan2.txt
    TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libv_rama.so TRX2_RAMA_SCAN=$sc TRX2_SF_FA_SCALE=0.4,0.4 timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 2048 1000 model 2>&1 | grep -v "^#" >> $O/model_scan2.txt; done
  echo "rc=$?"
}
run9() {  # model scan 3: fitted omega (every peptide tethered) at stiffness 1 / 2 / 3 / 4 with the rama surface shrunk to one half
  O=$R/gpurun_out/r05_run9; mkdir -p $O; : > $O/model_scan3.txt
  for v in rama o1 o2 o3 o4; do for sc in 1.0,0.4 0.4,0.4; do echo "== $v, TRX2_RAMA_SCAN=0.5,1 TRX2_SF_FA_SCALE=$sc" >> $O/model_scan3.txt
    TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libv_$v.so TRX2_RAMA_SCAN=0.5,1 TRX2_SF_FA_SCALE=$sc timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 2048 1000 model 2>&1 | grep -v "^#" >> $O/model_scan3.txt; done; done
  echo "rc=$?"
}
run10() {  # the shipped model (two-pass rama fit, surface at one half, omega stiffness 3): parity tests + outcome on 2 x 4096 decoys, against the rounds-1-4 terms
  O=$R/gpurun_out/r05_run10; mkdir -p $O
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cartesian.py tests/test_gpu_relax.py -q -m gpu -k "eval or tracks or short or relax" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
  echo "== shipped: fitted rama (surface x 0.5, two-pass constants / propensities), fitted omega x 3, TRX2_SF_FA_SCALE=1.0,0.4" > $O/model_final.txt
  timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 4096 1000 model 2>&1 | grep -v "^#" >> $O/model_final.txt
  echo "== rounds 1-4 terms, TRX2_SF_FA_SCALE=0.4,0.4" >> $O/model_final.txt
  TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/libtrx2fold_nofit.so TRX2_SF_FA_SCALE=0.4,0.4 timeout -k 10 300 python3 tools/tol_sweep_relax.py $R 4096 1000 model 2>&1 | grep -v "^#" >> $O/model_final.txt
  echo "rc=$?"
}
run11() {  # the whole GPU suite (no -x: every failure shown), prints kept for calibrating the new bounds
  O=$R/gpurun_out/r05_run11; mkdir -p $O
  timeout -k 10 1100 python3 -m pytest tests -m gpu -q -s > $O/pytest_full.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_full.txt | cut -c1-300
}
run12() {  # the tests run11 left red, after their fixes
  O=$R/gpurun_out/r05_run12; mkdir -p $O
  timeout -k 10 1100 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_iteration_parity.py tests/test_gpu_outcome_vs_oracle.py tests/test_gpu_parity.py tests/test_gpu_topologies.py tests/test_gpu_zz_open_findings.py -m gpu -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.txt | cut -c1-300
}
run13() {
  O=$R/gpurun_out/r05_run13; mkdir -p $O
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "backbone_terms" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.txt | cut -c1-300
}
run14() {  # step-kernel A/B: live launch durations for every libstep_*.so variant beside the shipped build
  O=$R/gpurun_out/r05_run14; mkdir -p $O; : > $O/step_ab.txt
  for lib in $R/trrosettax2-dynamics_amd/libtrx2fold.so $R/trrosettax2-dynamics_amd/libstep_*.so; do [ -f $lib ] || continue
    TRX2FOLD_LIB=$lib timeout -k 10 300 python3 tools/step_ab.py $R 3 >> $O/step_ab.txt 2>&1; done
  cat $O/step_ab.txt
}
run15() {  # step-kernel A/B (speculative record prefetch + fused energy / Gram reduction) and the parity / bitwise tests
  O=$R/gpurun_out/r05_run15; mkdir -p $O; : > $O/step_ab.txt
  for lib in $R/trrosettax2-dynamics_amd/libstep_*.so $R/trrosettax2-dynamics_amd/libtrx2fold.so; do [ -f $lib ] || continue
    TRX2FOLD_LIB=$lib timeout -k 10 300 python3 tools/step_ab.py $R 3 >> $O/step_ab.txt 2>&1; done
  cat $O/step_ab.txt
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shared_launch.py tests/test_gpu_selfcheck.py tests/test_gpu_cartesian.py tests/test_gpu_relax.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt | cut -c1-300
}
run16() {  # the whole GPU suite again
  O=$R/gpurun_out/r05_run16; mkdir -p $O
  timeout -k 10 1150 python3 -m pytest tests -m gpu -q > $O/pytest_full.txt 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_full.txt | cut -c1-300
}
"$@"
