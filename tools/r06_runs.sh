#!/bin/bash
# Round 6: the gpurun command sequences behind the records under profiles/r06_*.  usage (on the GPU box): bash tools/r06_runs.sh <function>
R=$PWD
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
traces() {  # kernel traces (rocprofv3 --kernel-trace --stats) of bench.py's `value` -- the metric's job, Nmax shortened to 40 so that the trace stays
  # below a million dispatches (named in the line's workload) -- of configs 2-4 as `value`, and of batch mode with sixteen targets in flight
  O=$R/gpurun_out/r06_profiles; mkdir -p $O; export TMPDIR=/tmp
  cd /tmp; rm -rf /tmp/kt0
  run 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt0 -- python3 $R/bench.py --steps 4 --warmup 0 --nmax 40 --no-cpu-baseline --no-sub-records --no-e2e > $O/bench_metric_job_under_profiler.json 2> $O/bench_metric_job.err
  f=$(find /tmp/kt0 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_metric_job_kernel_stats.csv && cut -c1-160 $O/r06_metric_job_kernel_stats.csv | head -6
  for cfg in 2 3 4; do
    cd /tmp; rm -rf /tmp/kt$cfg
    run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$cfg -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-legs > $O/bench_c${cfg}_under_profiler.json 2> $O/bench_c${cfg}.err
    f=$(find /tmp/kt$cfg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_c${cfg}_kernel_stats.csv && cut -c1-150 $O/r06_c${cfg}_kernel_stats.csv | head -4
  done
  rm -rf /tmp/ktb
  run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -- python3 $R/tools/e2e_batch.py $R 150 16 40 16 > $O/batch16_under_profiler.txt 2>&1
  f=$(find /tmp/ktb -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r06_batch16_kernel_stats.csv && cut -c1-150 $O/r06_batch16_kernel_stats.csv | head -4
  cd $R
}
pmc() {  # PMC passes (separate rocprofv3 --pmc runs, tools/pmc_run.sh) for the launch shapes given as "config decoys" pairs, e.g. pmc "e 1" "2 32"
  for spec in "$@"; do
    set -- $spec
    for k in pair step; do
      bash tools/pmc_run.sh $1 $2 $k r06_profiles/pmc 20 || exit $?
    done
  done
}
pmc_a() { pmc "e 1" "2 32" "2 640"; }
pmc_b() { pmc "3 64" "3 640" "4 16"; }
shared16() {  # sixteen single-decoy folds in one engine's launches: pair | step form and half-evaluation form
  O=$R/gpurun_out/r06_profiles; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmcs
    SCALING_WAVES=1 TRX2_ENGINE_STREAMS=1 TRX2_ENGINE_HALF=0 run 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs -- python3 $R/tools/shared_scaling.py $R 150 800 16 > $O/shared16_pmc_$tag.log 2>&1
    f=$(find /tmp/pmcs -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f 200 k_pair1_multi 48 > $O/shared16_pair_$tag.json; python3 $R/tools/pmc_report.py $f 200 k_step_multi 48 > $O/shared16_step_$tag.json; fi
    rm -rf /tmp/pmcs
    SCALING_WAVES=1 TRX2_ENGINE_STREAMS=1 TRX2_ENGINE_HALF=1 run 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs -- python3 $R/tools/shared_scaling.py $R 150 800 16 > $O/shared16_half_pmc_$tag.log 2>&1
    f=$(find /tmp/pmcs -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f 200 k_half_multi 99 > $O/shared16_half_$tag.json; fi
  done
  cd $R
}
final() {  # closing sequence: the GPU suite, smoke, the bench lines
  O=$R/gpurun_out/r06_final; mkdir -p $O
  run 1150 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt | cut -c1-200
  run 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt | cut -c1-200
}
benches() {
  O=$R/gpurun_out/r06_final; mkdir -p $O
  run 1000 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json
  run 1000 python3 bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; echo "bench rc=$?"; tail -c 300 $O/bench_s20.json
}
"$@"
