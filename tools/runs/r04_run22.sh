# Round 4, run 22: one target end to end (two chains) with and without the launch engine; 1 / 2 / 3 engine streams
O=gpurun_out/r04_run22
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for cfg in "1 2" "0 2" "1 2" "0 2" "1 1"; do
  set -- $cfg
  echo "shared=$1 streams=$2" >> $O/single.txt
  TRX2_SHARED_LAUNCH=$1 TRX2_ENGINE_STREAMS=$2 run 300 python3 tools/e2e_single.py . 150 80 >> $O/single.txt 2>> $O/err.txt || exit 1
done
cut -c1-250 $O/single.txt
