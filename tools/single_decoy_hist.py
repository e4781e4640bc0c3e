"""Launch-length histogram of the step kernel when ONE decoy is folded (L=150, distances only, no Cartesian stage -> k_chain): a launch
is then one decoy's step, and the modes show what a two-loop recursion / a rejected trial cost.  Run under rocprofv3 --kernel-trace;
usage: single_decoy_hist.py <repo> | single_decoy_hist.py --report <kernel_trace.csv>"""
import importlib, sys
import numpy as np
if sys.argv[1] == "--report":
    import csv
    by = {}
    for r in csv.DictReader(open(sys.argv[2])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        by.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in by.items():
        if len(v) > 100:
            v = np.array(v)
            h, e = np.histogram(v, bins=np.arange(np.floor(v.min()), np.ceil(np.percentile(v, 99.5)) + 0.5, 0.5))
            print(f"{n}  n={len(v)}  mean {v.mean():.2f} us")
            print("   " + "  ".join(f"{a:.1f}:{c}" for a, c in zip(e[:-1], h) if c > len(v) // 200))
    sys.exit(0)
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = 150; m = S.make_map(L, seed=L); ctx = T.Context(0); ctx.set_map(m["dist"], seq=m["seq"])
runs = T.protocol.build_runs(L, 2, cartesian_stage=False)
for i in range(4):
    r = ctx.fold_batch(1, runs, seed=5 + i)
print(f"{r['seconds']*1e3:.0f} ms, {r['launches']} launches, evals {r['n_evals']}, iters {r['n_iters']}")
ctx.close()
