#!/bin/bash
# PMC passes over ONE kernel at ONE launch shape of a bench config (separate rocprofv3 --pmc runs, kernel trace only;
# MI355X_MICROARCH.md, HBM section):
#   usage: tools/pmc_run.sh <config 2|3|4|e (the metric's job: one decoy on a fed-back map)> <decoys per launch> <pair|step> <out dir under gpurun_out/> [replays]
# tools/pmc_kernel.py drives the launches; tools/pmc_report.py averages the last <replays> dispatches of the kernel.
# Results: $GRAFT_REPO_ROOT/gpurun_out/<out dir>/c<config>_<kernel>_B<decoys>_pass<i>.json (assembled by tools/make_traffic_json.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=$1; B=$2; K=$3; OUT=$R/gpurun_out/$4; N=${5:-20}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  tag=c${CFG}_${K}_B${B}_pass$i
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/pmc_kernel.py $R $CFG $B $K $N > $OUT/$tag.log 2>&1
  rc=$?; echo "$tag ($grp) rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  f=$(ls $OUT/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f $N k_$K | tee $OUT/$tag.json; echo; rm -rf $OUT/$tag; fi  # keep the summaries, drop the raw traces
done
