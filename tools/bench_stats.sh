# rocprofv3 --kernel-trace --stats of the default bench run, the headline queue only (no legs, no sub-records, no CPU baseline).  Run on the GPU box from the repo root.
set -e
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/bstats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats -- python3 $R/bench.py --no-cpu-baseline --no-sub-records --no-legs > $R/gpurun_out/b_stats.json 2> $R/gpurun_out/b_stats.err
f=$(find /tmp/bstats -name '*kernel_stats.csv' | head -1)
cp $f $R/gpurun_out/r02_c2_kernel_stats.csv
cat $R/gpurun_out/r02_c2_kernel_stats.csv | cut -c1-150
