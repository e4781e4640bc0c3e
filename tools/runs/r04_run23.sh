# Round 4, run 23: single-decoy folds in flight, launched by the folds themselves (TRX2_SHARED_LAUNCH=0) against the launch engine, 1-12 folds
O=gpurun_out/r04_run23
mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
for sh in 0 1; do for w in 4 1; do
  echo "shared=$sh waves=$w" >> $O/scaling.txt
  SCALING_WAVES=$w TRX2_SHARED_LAUNCH=$sh run 300 python3 tools/shared_scaling.py . 150 1600 1 2 3 4 6 8 12 >> $O/scaling.txt 2>> $O/err.txt || exit 1
done; done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_run23/scaling.txt"):
    if l.startswith("shared"): print(l.strip())
    else:
        d=json.loads(l); print("  folds %2d  us/fold-eval %6.2f  wall %.4f" % (d["folds"], d["us_per_fold_eval"], d["wall_s"]))
PY
