# Round-4 profile records (run on the GPU box from the repo root): kernel traces of the three single-GPU configs at the shape
# bench.py's `value` times, then the PMC passes (separate rocprofv3 --pmc runs) of the pair and step kernels at those shapes and
# at the pooled leg's; results under gpurun_out/r04_profiles/, to be copied to profiles/.
R=$PWD
O=$R/gpurun_out/r04_profiles
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
export TMPDIR=/tmp
for cfg in 2 3 4; do
  cd /tmp; rm -rf /tmp/kt$cfg
  run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$cfg -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-sub-records --no-legs --no-e2e > $O/bench_c${cfg}_under_profiler.json 2> $O/bench_c${cfg}.err
  f=$(find /tmp/kt$cfg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r04_c${cfg}_kernel_stats.csv && cut -c1-150 $O/r04_c${cfg}_kernel_stats.csv | head -8
  cd $R
done
for spec in "2 32" "2 640" "3 64" "4 16"; do
  set -- $spec
  for k in pair step; do
    bash tools/pmc_run.sh $1 $2 $k r04_profiles/pmc 20 || exit $?
  done
done
# the shared-launch shape (batch mode's iteration phase): kernel trace of sixteen targets in flight, and the PMC passes of the shared
# kernels with sixteen folds in one engine's launches
cd /tmp; rm -rf /tmp/ktb
run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktb -- python3 $R/tools/e2e_batch.py $R 150 16 40 16 > $O/batch16_under_profiler.txt 2>&1
f=$(find /tmp/ktb -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r04_batch16_kernel_stats.csv && cut -c1-150 $O/r04_batch16_kernel_stats.csv | head -6
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmcs
  SCALING_WAVES=1 TRX2_ENGINE_STREAMS=1 run 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcs -- python3 $R/tools/shared_scaling.py $R 150 400 16 > $O/shared16_pmc_$tag.log 2>&1
  f=$(find /tmp/pmcs -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f 200 k_pair1_multi > $O/shared16_pair_$tag.json; python3 $R/tools/pmc_report.py $f 200 k_step_multi > $O/shared16_step_$tag.json; cat $O/shared16_pair_$tag.json; echo; fi
done
cd $R
python3 tools/make_traffic_json.py $O/pmc $O/r04_traffic.json
