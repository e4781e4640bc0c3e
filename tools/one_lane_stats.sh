R=$PWD; export TMPDIR=/tmp; cd /tmp
for cfg in "160 160 1" "320 160 2" "64 64 1"; do
  rm -rf /tmp/ol; timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ol -- python3 $R/tools/one_lane.py $R $cfg 2>/dev/null | grep lane
  f=$(find /tmp/ol -name '*kernel_stats.csv' | head -1); grep "k_pair\|k_step" $f | cut -d, -f1-4 | cut -c1-90
done
