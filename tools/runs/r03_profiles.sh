# Round-3 profile records (run on the GPU box from the repo root): kernel traces of the three single-GPU configs at the shape
# bench.py's `value` times, then the PMC passes (separate rocprofv3 --pmc runs) of the pair and step kernels at those shapes and
# at the pooled leg's; results under gpurun_out/r03_profiles/, to be copied to profiles/.
R=$PWD
O=$R/gpurun_out/r03_profiles
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
export TMPDIR=/tmp
for cfg in 2 3 4; do
  cd /tmp; rm -rf /tmp/kt$cfg
  run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$cfg -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-sub-records --no-legs --no-e2e > $O/bench_c${cfg}_under_profiler.json 2> $O/bench_c${cfg}.err
  f=$(find /tmp/kt$cfg -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r03_c${cfg}_kernel_stats.csv && cut -c1-150 $O/r03_c${cfg}_kernel_stats.csv | head -8
  cd $R
done
for spec in "2 32" "2 640" "3 64" "4 16"; do
  set -- $spec
  for k in pair step; do
    bash tools/pmc_run.sh $1 $2 $k r03_profiles/pmc 20 || exit $?
  done
done
python3 tools/make_traffic_json.py $O/pmc $O/r03_traffic.json
