"""Which role sets the length of a step launch?  Folds 64 decoys (L=150, distances only) with the Cartesian stage (fused k_step:
torsion role + Cartesian role) and without it (k_chain only); run under `rocprofv3 --kernel-trace` and read the trace with
--report <kernel_trace.csv>.  usage: step_roles.py <repo> | step_roles.py --report <csv>"""
import importlib, sys
import numpy as np
if sys.argv[1] == "--report":
    import csv
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    by = {}
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        by.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in by.items():
        if len(v) > 100:
            v = np.array(v)
            print(f"{n:28s} n={len(v):6d}  mean {v.mean():5.1f}  median {np.median(v):5.1f}  p10 {np.percentile(v,10):5.1f}  p90 {np.percentile(v,90):5.1f}  max {v.max():5.1f} us")
    sys.exit(0)
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = 150; m = S.make_map(L); ctx = T.Context(0); ctx.set_map(m["dist"], seq=m["seq"])
for cart in (True, False):
    runs = T.protocol.build_runs(L, 2, cartesian_stage=cart)
    ctx.fold_batch(64, runs, seed=1)
    r = ctx.fold_batch(64, runs, seed=2)
    print(f"cartesian stage {cart}: {r['seconds']*1e3:.0f} ms, {r['launches']} launches, {r['seconds']/r['launches']*1e6:.1f} us per evaluation")
ctx.close()
