#!/bin/bash
# PMC passes over the pair kernel of one bench config (separate rocprofv3 --pmc runs, kernel trace only; MI355X_MICROARCH.md):
#   usage: tools/pmc_run.sh <config 2|3|4> <out dir under gpurun_out/> [replays]
# Each pass folds one batch and replays the pair kernel on the final coordinates (tools/pmc_pair.py); tools/pmc_report.py
# averages the last replays.  Run from anywhere on the GPU box; results land in $GRAFT_REPO_ROOT/gpurun_out/<out dir>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=$1; OUT=$R/gpurun_out/$2; N=${3:-20}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/c${CFG}_pass$i -- python3 $R/tools/pmc_pair.py $R $CFG $N > $OUT/c${CFG}_pass$i.log 2>&1
  rc=$?; echo "pass $i ($grp) rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  f=$(ls $OUT/c${CFG}_pass$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f $N $OUT/c${CFG}_pass$i.csv | tee $OUT/c${CFG}_pass$i.json; echo; rm -rf $OUT/c${CFG}_pass$i; fi  # keep the summaries, drop the raw traces
done
