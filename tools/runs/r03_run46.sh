# instruction-cache counters of the step and pair kernels (is the step kernel, 38 KB of straight-line code per role, fetch-bound?)
R=$PWD
O=$R/gpurun_out/r46
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E 'ICACHE|IFETCH|SQ_WAIT_INST|SQC_' | head -40 > $O/avail.txt; head -40 $O/avail.txt
for spec in "2 1 step" "2 32 step" "2 32 pair" "3 1 pair"; do
  set -- $spec
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"; do
    tag=ic_c$1_$3_B$2_$(echo $grp | cut -c1-9)
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag -- python3 $R/tools/pmc_kernel.py $R $1 $2 $3 20 > $O/$tag.log 2>&1
    rc=$?; echo "$tag rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
    f=$(ls $O/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f 20 k_$3 | tee $O/$tag.json; echo; rm -rf $O/$tag; else tail -3 $O/$tag.log; fi
  done
done
