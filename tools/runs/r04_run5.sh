# Round 4, run 5: how does a shared launch scale with the folds it holds, and what does the pair kernel fetch?
O=gpurun_out/r04_run5
mkdir -p $O
R=$PWD
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
run 300 python3 tools/shared_scaling.py . 150 1500 1 2 4 8 16 32 64 > $O/scaling_L150.txt 2>&1; echo "scaling rc=$?"; cat $O/scaling_L150.txt
TRX2_ENGINE_STREAMS=1 run 300 python3 tools/shared_scaling.py . 150 1500 1 4 16 32 > $O/scaling_L150_1eng.txt 2>&1; echo "scaling rc=$?"; cat $O/scaling_L150_1eng.txt
TRX2_ENGINE_STREAMS=3 run 300 python3 tools/shared_scaling.py . 150 1500 16 32 64 > $O/scaling_L150_3eng.txt 2>&1; echo "scaling rc=$?"; cat $O/scaling_L150_3eng.txt
run 300 python3 tools/shared_scaling.py . 90 1500 1 16 64 > $O/scaling_L90.txt 2>&1; echo "scaling rc=$?"; cat $O/scaling_L90.txt
cd /tmp; export TMPDIR=/tmp
for N in 1 16; do
  TRX2_ENGINE_STREAMS=1 run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$N -- python3 $R/tools/shared_scaling.py $R 150 1500 $N > $R/$O/prof$N.log 2>&1; echo "prof rc=$?"
  f=$(ls $R/$O/prof$N/*/*kernel_stats.csv | head -1); cp $f $R/$O/N${N}_kernel_stats.csv; head -4 $R/$O/N${N}_kernel_stats.csv; rm -rf $R/$O/prof$N
done
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  TRX2_ENGINE_STREAMS=1 run 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$O/pmc_$tag -- python3 $R/tools/shared_scaling.py $R 150 400 16 > $R/$O/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
  f=$(ls $R/$O/pmc_$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_report.py $f 200 k_pair1_multi | tee $R/$O/pmc_pair_$tag.json; echo; python3 $R/tools/pmc_report.py $f 200 k_step_multi | tee $R/$O/pmc_step_$tag.json; echo; rm -rf $R/$O/pmc_$tag; fi
done
