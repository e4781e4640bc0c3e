# what bounds the step launch at 32 decoys: kernel traces with / without the Cartesian role, one decoy and 2 x 32
O=gpurun_out/r16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
X=$R/trrosettax2-dynamics_amd/csrc/_exp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT/KILL rc=$rc: $*"; exit $rc; fi; return $rc; }
prof() { # name, then command
  local n=$1; shift
  rm -rf /tmp/prof_$n
  run 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o t -- "$@" > $R/$O/$n.log 2>&1
  local f=$(find /tmp/prof_$n -name '*kernel_stats.csv' | head -1)
  echo "== $n"; if [ -z "$f" ]; then echo "no stats file"; tail -3 $R/$O/$n.log; return 1; fi; head -4 "$f" | cut -c1-150
  cp $f $R/$O/${n}_kernel_stats.csv
  local tr=$(find /tmp/prof_$n -name '*kernel_trace.csv' | head -1)
  if [ -z "$tr" ]; then echo "no trace file"; return 1; fi
  python3 - $tr <<'PY'
import csv,sys,collections
import numpy as np
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    if len(v)>500:
        v=np.array(v)/1e3
        print("   %-40s n=%6d mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us"%(k[:40],len(v),v.mean(),*np.percentile(v,[10,50,90,99]),v.max()))
PY
}
prof single_gram python3 $R/tools/single_decoy_trace.py $R 150 1 6
TRX2FOLD_LIB=$X/libtrx2fold_twoloop.so prof single_twoloop python3 $R/tools/single_decoy_trace.py $R 150 1 6
prof c2_gram python3 $R/tools/percall.py $R 2 2 2
TRX2FOLD_LIB=$X/libtrx2fold_twoloop.so prof c2_twoloop python3 $R/tools/percall.py $R 2 2 2
PERCALL_NOCART=1 prof c2_nocart python3 $R/tools/percall.py $R 2 2 2
prof c2_onelane python3 $R/tools/percall.py $R 2 1 2
