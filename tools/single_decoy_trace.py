"""Single-decoy folds (what every feedback iteration of run_inference folds): n folds of one decoy, for kernel traces.
usage: single_decoy_trace.py <repo> <L> <orient 0|1> <n>"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, orient, n = int(sys.argv[2]), sys.argv[3] != "0", int(sys.argv[4])
m = S.make_map(L, seed=L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
runs = T.protocol.build_runs(L, 2)
ctx.fold_batch(1, runs, seed=1)
t0 = time.perf_counter(); rs = [ctx.fold_batch(1, runs, seed=2, decoy0=k) for k in range(n)]; el = time.perf_counter() - t0
ev = sum(int(r["n_evals"][0]) for r in rs); ln = sum(r["launches"] for r in rs)
print(json.dumps(dict(lib=os.path.basename(os.environ.get("TRX2FOLD_LIB", "default")), L=L, orient=orient, folds=n, ms_per_fold=round(1e3 * el / n, 2), evals_per_fold=ev / n,
                      us_per_eval=round(1e6 * el / ev, 2), launch_pairs_per_fold=ln / n)))
ctx.close()
