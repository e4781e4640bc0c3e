set -e
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/sdh
timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/sdh -- python3 $R/tools/single_decoy_hist.py $R > $R/gpurun_out/sdh.txt 2>&1
f=$(find /tmp/sdh -name '*kernel_trace.csv' | head -1)
python3 $R/tools/single_decoy_hist.py --report $f >> $R/gpurun_out/sdh.txt
grep -v "^[WEI]2026" $R/gpurun_out/sdh.txt
