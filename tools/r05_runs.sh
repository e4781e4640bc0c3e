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
run17() {  # interim bench line (default flags) with the round's model and minimiser changes in
  O=$R/gpurun_out/r05_run17; mkdir -p $O
  timeout -k 10 1100 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
}
run18() {  # engine stream priorities (TRX2_ENGINE_PRIORITY): batch mode with 16 and 8 targets in flight, with and without
  O=$R/gpurun_out/r05_run18; mkdir -p $O; : > $O/engine_prio.txt
  for p in 0 1 0 1; do
    echo "== TRX2_ENGINE_PRIORITY=$p" >> $O/engine_prio.txt
    TRX2_ENGINE_PRIORITY=$p timeout -k 10 300 python3 tools/e2e_batch.py $R 150 16 40 16 2>&1 | cut -c1-400 >> $O/engine_prio.txt
    TRX2_ENGINE_PRIORITY=$p timeout -k 10 300 python3 tools/e2e_batch.py $R 150 8 80 8 2>&1 | cut -c1-400 >> $O/engine_prio.txt
  done
  cat $O/engine_prio.txt | cut -c1-330
}
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
profiles() {  # Round-5 profile records: kernel traces of the three single-GPU configs at the shape bench.py's `value` times, of run_inference on
  # one target (the metric's own job) and of batch mode with sixteen targets in flight; then the PMC passes (separate rocprofv3 --pmc runs)
  # of the pair and step kernels at those shapes; results under gpurun_out/r05_profiles/, copied to profiles/ afterwards.
  O=$R/gpurun_out/r05_profiles; rm -rf $O; mkdir -p $O
  export TMPDIR=/tmp
  for cfg in 2 3 4; do
    cd /tmp; rm -rf /tmp/kt$cfg
    run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$cfg -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-sub-records --no-legs --no-e2e > $O/bench_c${cfg}_under_profiler.json 2> $O/bench_c${cfg}.err
    f=$(find /tmp/kt$cfg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r05_c${cfg}_kernel_stats.csv && cut -c1-150 $O/r05_c${cfg}_kernel_stats.csv | head -6
    cd $R
  done
  cd /tmp; rm -rf /tmp/kt_e2e
  run 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_e2e -- python3 $R/tools/e2e_single.py $R 150 60 > $O/e2e_single.log 2>&1
  f=$(find /tmp/kt_e2e -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r05_e2e_single_kernel_stats.csv && cut -d, -f1-5 $O/r05_e2e_single_kernel_stats.csv | head -6
  rm -rf /tmp/ktb
  run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -- python3 $R/tools/e2e_batch.py $R 150 16 40 16 > $O/batch16_under_profiler.txt 2>&1
  f=$(find /tmp/ktb -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r05_batch16_kernel_stats.csv && cut -c1-150 $O/r05_batch16_kernel_stats.csv | head -6
  cd $R
  for spec in "2 32" "2 640" "3 64" "4 16"; do
    set -- $spec
    for k in pair step; do
      bash tools/pmc_run.sh $1 $2 $k r05_profiles/pmc 20 || exit $?
    done
  done
  cd /tmp
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmcs
    SCALING_WAVES=1 TRX2_ENGINE_STREAMS=1 TRX2_ENGINE_HALF=0 run 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs -- python3 $R/tools/shared_scaling.py $R 150 800 16 > $O/shared16_pmc_$tag.log 2>&1
    f=$(find /tmp/pmcs -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f 200 k_pair1_multi 48 > $O/shared16_pair_$tag.json; python3 $R/tools/pmc_report.py $f 200 k_step_multi 48 > $O/shared16_step_$tag.json; fi
    rm -rf /tmp/pmcs   # the same sixteen folds in half-evaluation form: one kernel, eight folds in either role per launch
    SCALING_WAVES=1 TRX2_ENGINE_STREAMS=1 TRX2_ENGINE_HALF=1 run 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs -- python3 $R/tools/shared_scaling.py $R 150 800 16 > $O/shared16_half_pmc_$tag.log 2>&1
    f=$(find /tmp/pmcs -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f 200 k_half_multi 99 > $O/shared16_half_$tag.json; fi
  done
  cd $R
  python3 tools/make_traffic_json.py $O/pmc $O/r05_traffic.json
}
final() {  # Round-5 closing sequence on the GPU box: the GPU suite, smoke, the profile records, the bench lines (with the fresh traffic records)
  O=$R/gpurun_out/r05_final; mkdir -p $O
  run 1150 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-200
  run 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-200
}
benches() {
  O=$R/gpurun_out/r05_final; mkdir -p $O
  [ -f $R/gpurun_out/r05_profiles/r05_traffic.json ] && cp $R/gpurun_out/r05_profiles/r05_traffic.json $R/profiles/r05_traffic.json
  run 1000 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
  run 1000 python3 bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench rc=$?"; tail -c 300 $O/bench_s20.json
}
run19() {  # four lanes of 16 against two of 32 for the call of 64 (probe with independent contexts)
  O=$R/gpurun_out/r05_run19; mkdir -p $O
  TRX2_SHARED_LAUNCH=0 timeout -k 10 400 python3 tools/lanes4_probe.py $R 5 2 > $O/lanes4_c2.txt 2>&1; cat $O/lanes4_c2.txt
  TRX2_SHARED_LAUNCH=0 timeout -k 10 400 python3 tools/lanes4_probe.py $R 5 3 > $O/lanes4_c3.txt 2>&1; cat $O/lanes4_c3.txt
}
run20() {  # engine streams confined to disjoint CU sets (TRX2_ENGINE_CUMASK): batch mode with 16 and 8 targets in flight
  O=$R/gpurun_out/r05_run20; mkdir -p $O; : > $O/engine_cumask.txt
  for m in 0 1 2 0 1 2; do
    echo "== TRX2_ENGINE_CUMASK=$m" >> $O/engine_cumask.txt
    TRX2_ENGINE_CUMASK=$m timeout -k 10 300 python3 tools/e2e_batch.py $R 150 16 40 16 2>&1 | cut -c1-330 >> $O/engine_cumask.txt
    TRX2_ENGINE_CUMASK=$m timeout -k 10 300 python3 tools/e2e_batch.py $R 150 8 80 8 2>&1 | cut -c1-330 >> $O/engine_cumask.txt
  done
  cut -c1-170 $O/engine_cumask.txt
}
run21() {  # Cartesian role: energy + Gram products in one reduction, state loads before it -- A/B and the bitwise / parity tests
  O=$R/gpurun_out/r05_run21; mkdir -p $O; : > $O/step_ab.txt
  for lib in $R/trrosettax2-dynamics_amd/libstep_*.so $R/trrosettax2-dynamics_amd/libtrx2fold.so; do [ -f $lib ] || continue
    TRX2FOLD_LIB=$lib timeout -k 10 300 python3 tools/step_ab.py $R 3 >> $O/step_ab.txt 2>&1; done
  cat $O/step_ab.txt
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shared_launch.py tests/test_gpu_selfcheck.py tests/test_gpu_cartesian.py tests/test_gpu_relax.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt | cut -c1-300
}
run22() {
  O=$R/gpurun_out/r05_run22; mkdir -p $O
  timeout -k 10 300 python3 tools/straggler_profile.py $R > $O/stragglers_c2.txt 2>&1; cat $O/stragglers_c2.txt
}
run23() {  # do two engines overlap? 28 single-decoy folds in flight on two engines / on one: wall time without the profiler, then the kernel-trace timeline
  O=$R/gpurun_out/r05_run23; mkdir -p $O; rm -f $O/overlap.txt
  cd /tmp && export TMPDIR=/tmp
  for ne in 2 1; do
    echo "== $ne engine(s), 28 folds in flight, 3000 evaluations each; without the profiler:" >> $O/overlap.txt
    SCALING_WAVES=1 TRX2_ENGINE_STREAMS=$ne timeout -k 10 300 python3 $R/tools/shared_scaling.py $R 150 3000 28 2>&1 | grep '^{' >> $O/overlap.txt || return $?
    rm -rf /tmp/kt_ov$ne
    SCALING_WAVES=1 TRX2_ENGINE_STREAMS=$ne timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_ov$ne -- python3 $R/tools/shared_scaling.py $R 150 3000 28 > $O/scaling_engines$ne.txt 2>&1 || return $?
    f=$(find /tmp/kt_ov$ne -name '*kernel_trace.csv' | head -1)
    echo "under rocprofv3 --kernel-trace:" >> $O/overlap.txt; grep '^{' $O/scaling_engines$ne.txt >> $O/overlap.txt
    python3 $R/tools/overlap_timeline.py $f 0.6 >> $O/overlap.txt 2>&1
  done
  cat $O/overlap.txt
}
run24() {  # half-evaluation launches (k_half_multi): bitwise tests, then A/B against pair | step launches -- folds in flight, batch mode
  O=$R/gpurun_out/r05_run24; mkdir -p $O
  timeout -k 10 600 python3 -m pytest tests/test_gpu_shared_launch.py -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt; [ $rc -eq 0 ] || return $rc
  for half in 0 1; do for ne in 2 1; do
    echo "== TRX2_ENGINE_HALF=$half TRX2_ENGINE_STREAMS=$ne" >> $O/scaling.txt
    SCALING_WAVES=1 TRX2_ENGINE_HALF=$half TRX2_ENGINE_STREAMS=$ne timeout -k 10 300 python3 tools/shared_scaling.py $R 150 1500 8 16 28 32 64 2>&1 | grep '^{' | cut -c1-400 >> $O/scaling.txt || return $?
  done; done
  cat $O/scaling.txt
  for half in 0 1; do for ne in 2 1; do
    echo "== TRX2_ENGINE_HALF=$half TRX2_ENGINE_STREAMS=$ne" >> $O/batch.txt
    TRX2_ENGINE_HALF=$half TRX2_ENGINE_STREAMS=$ne timeout -k 10 400 python3 tools/e2e_batch.py $R 150 16 40 16 8 2>&1 | grep '^{' | cut -c1-330 >> $O/batch.txt || return $?
  done; done
  cat $O/batch.txt
}
run25() {  # half-evaluation launches in batch mode: from how many folds per launch class on?  two shapes, every cell twice
  O=$R/gpurun_out/r05_run25; mkdir -p $O; rm -f $O/batch.txt
  for rep in 1 2; do for cell in "0 0" "1 0" "1 8" "1 12" "1 1000"; do
    set -- $cell
    echo "== TRX2_ENGINE_HALF=$1 TRX2_ENGINE_HALF_MIN=$2 (two engines)" >> $O/batch.txt
    TRX2_ENGINE_HALF=$1 TRX2_ENGINE_HALF_MIN=$2 timeout -k 10 400 python3 tools/e2e_batch.py $R 150 16 40 16 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
    TRX2_ENGINE_HALF=$1 TRX2_ENGINE_HALF_MIN=$2 timeout -k 10 400 python3 tools/e2e_batch.py $R 150 8 80 8 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  done; done
  cat $O/batch.txt
}
run26() {  # half-evaluation launches at the ends of the range: 3 and 32 targets in flight; three engines
  O=$R/gpurun_out/r05_run26; mkdir -p $O; rm -f $O/batch.txt
  for rep in 1 2; do for half in 0 1; do
    echo "== TRX2_ENGINE_HALF=$half (two engines)" >> $O/batch.txt
    TRX2_ENGINE_HALF=$half timeout -k 10 400 python3 tools/e2e_batch.py $R 150 6 40 3 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
    TRX2_ENGINE_HALF=$half timeout -k 10 400 python3 tools/e2e_batch.py $R 150 32 20 32 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  done; done
  for ne in 3 1; do
    echo "== TRX2_ENGINE_HALF=1 TRX2_ENGINE_STREAMS=$ne" >> $O/batch.txt
    TRX2_ENGINE_STREAMS=$ne timeout -k 10 400 python3 tools/e2e_batch.py $R 150 32 20 32 16 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  done
  cat $O/batch.txt
}
run27() {  # half-evaluation launches under the library's rule (from twelve live contexts on): tests, then batch mode at 3 / 8 / 16 targets in flight
  O=$R/gpurun_out/r05_run27; mkdir -p $O; rm -f $O/batch.txt
  timeout -k 10 900 python3 -m pytest tests/test_gpu_shared_launch.py tests/test_abi.py -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt; [ $rc -eq 0 ] || return $rc
  timeout -k 10 400 python3 tools/e2e_batch.py $R 150 16 40 16 8 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  timeout -k 10 400 python3 tools/e2e_batch.py $R 150 6 40 3 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  timeout -k 10 400 python3 tools/e2e_batch.py $R 150 8 80 8 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  cat $O/batch.txt
}
run28() {  # half-evaluation kernel on 512-thread workgroups (a step workgroup keeps its CU to itself): bitwise tests, batch mode at 16 / 8 / 32 / 3 targets in flight
  O=$R/gpurun_out/r05_run28; mkdir -p $O; rm -f $O/batch.txt
  timeout -k 10 900 python3 -m pytest tests/test_gpu_shared_launch.py -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt; [ $rc -eq 0 ] || return $rc
  for rep in 1 2; do
    timeout -k 10 400 python3 tools/e2e_batch.py $R 150 16 40 16 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
    timeout -k 10 400 python3 tools/e2e_batch.py $R 150 8 80 8 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
    timeout -k 10 400 python3 tools/e2e_batch.py $R 150 32 20 32 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  done
  TRX2_ENGINE_HALF=1 timeout -k 10 400 python3 tools/e2e_batch.py $R 150 6 40 3 2>&1 | grep '^{' | cut -c1-250 >> $O/batch.txt || return $?
  cat $O/batch.txt
}
run29() {  # more streams than the default four hardware queues: TRX2_POOL_STREAMS x GPU_MAX_HW_QUEUES, batch mode (16 x Nmax 40, 8 x Nmax 80), one target, config 2
  O=$R/gpurun_out/r05_run29; mkdir -p $O; rm -f $O/pool.txt
  for cell in "4 -" "8 -" "8 8" "6 8" "4 8"; do
    set -- $cell
    echo "== TRX2_POOL_STREAMS=$1 GPU_MAX_HW_QUEUES=$2" >> $O/pool.txt
    ( [ "$2" != "-" ] && export GPU_MAX_HW_QUEUES=$2; export TRX2_POOL_STREAMS=$1
      timeout -k 10 400 python3 tools/e2e_batch.py $R 150 16 40 16 2>&1 | grep '^{' | cut -c1-170
      timeout -k 10 400 python3 tools/e2e_batch.py $R 150 8 80 8 2>&1 | grep '^{' | cut -c1-170
      timeout -k 10 400 python3 tools/e2e_single.py $R 150 60 2>&1 | grep -i "decoys\|^{" | tail -2 | cut -c1-200
      timeout -k 10 400 python3 bench.py --config 2 --steps 5 --warmup 1 --no-cpu-baseline --no-sub-records --no-legs --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2 value', round(d['value'],1))"
    ) >> $O/pool.txt 2>&1 || return $?
  done
  cat $O/pool.txt
}
run30() {  # the step kernel's LDS reserve for the other lane's pair workgroups at the headline shape (2 x 32 slots): TRX2_STEP_LDS_RESERVE=0 / 29696 / default
  O=$R/gpurun_out/r05_run30; mkdir -p $O; rm -f $O/reserve.txt
  for rep in 1 2; do for rs in default 0 29696; do
    echo "== TRX2_STEP_LDS_RESERVE=$rs" >> $O/reserve.txt
    for cfg in 2 3; do
      ( [ "$rs" != "default" ] && export TRX2_STEP_LDS_RESERVE=$rs
        timeout -k 10 400 python3 bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-sub-records --no-legs --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $cfg value', round(d['value'],1), 'step ms', round(d.get('roofline_step',{}).get('avg_launch_ms',0)*1e3,2), 'us')" ) >> $O/reserve.txt 2>&1 || return $?
    done
  done; done
  cat $O/reserve.txt
}
run31() {  # the one-wave pair kernel with the segment cache at THREE waves per SIMD (libtrx2fold_w3.so, -DTRX2_PAIR1_W3): pair | step form, saturated shapes
  O=$R/gpurun_out/r05_run31; mkdir -p $O; rm -f $O/w3.txt
  for lib in libtrx2fold.so libtrx2fold_w3.so; do for half in 0 1; do
    echo "== $lib TRX2_ENGINE_HALF=$half" >> $O/w3.txt
    ( export TRX2FOLD_LIB=$R/trrosettax2-dynamics_amd/$lib TRX2_ENGINE_HALF=$half
      SCALING_WAVES=1 timeout -k 10 300 python3 tools/shared_scaling.py $R 150 1500 28 64 2>&1 | grep '^{' | cut -c60-330
      timeout -k 10 400 python3 tools/e2e_batch.py $R 150 32 20 32 2>&1 | grep '^{' | cut -c1-170
      timeout -k 10 400 python3 tools/e2e_batch.py $R 150 16 40 16 2>&1 | grep '^{' | cut -c1-170 ) >> $O/w3.txt 2>&1 || return $?
  done; done
  cat $O/w3.txt
}
"$@"
