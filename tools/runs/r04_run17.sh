# Round 4, run 17: batch mode with the engine's step launches in the 256-register instantiation (A/B), 1-3 engine streams
O=gpurun_out/r04_run17
mkdir -p $O
R=$PWD
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for cfg in "0 2" "1 2" "0 2" "1 2" "1 3" "1 1" "0 1"; do
  set -- $cfg
  echo "lowreg=$1 streams=$2" >> $O/batch.txt
  TRX2_ENGINE_STEP_LOWREG=$1 TRX2_ENGINE_STREAMS=$2 run 300 python3 tools/e2e_batch.py . 90 16 40 16 >> $O/batch.txt 2>> $O/batch.err || exit 1
done
cut -c1-200 $O/batch.txt
