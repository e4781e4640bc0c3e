# Round 4, run 18: launch pair duration of ONE engine against folds in flight (how long pair and step take alone, for the
# question whether a launch that carries one group's step beside the other group's pair would pay)
O=gpurun_out/r04_run18
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for st in 1 2; do
  echo "streams=$st" >> $O/scaling.txt
  SCALING_WAVES=1 TRX2_ENGINE_STREAMS=$st run 300 python3 tools/shared_scaling.py . 150 1200 4 8 14 28 56 >> $O/scaling.txt 2>> $O/err.txt || exit 1
done
cut -c1-260 $O/scaling.txt
