# Round 4, run 31: kernel trace of run_inference on ONE target (two chains, single-decoy folds launching for themselves): the metric's own configuration
O=$PWD/gpurun_out/r04_run31
mkdir -p $O
R=$PWD
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_e2e
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_e2e -- python3 $R/tools/e2e_single.py $R 150 60 > $O/prof.log 2>&1; echo "rc=$?"; tail -1 $O/prof.log | cut -c1-200
f=$(find /tmp/kt_e2e -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r04_e2e_single_kernel_stats.csv && cut -d, -f1-5 $O/r04_e2e_single_kernel_stats.csv | head -8
