"""One decoy at a time (what every feedback iteration folds): L=90 example map, all channels.  Run under rocprofv3 --kernel-trace
and read with step_roles.py --report.  usage: single_decoy.py <repo> [n=10]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden"); m = np.load(os.path.join(g, "seq_NMR.npz"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"]); runs = T.protocol.build_runs(90, 2)
ctx.fold_batch(1, runs, seed=1)
sec, ev, la = 0.0, 0, 0
for i in range(n):
    r = ctx.fold_batch(1, runs, seed=2 + i); sec += r["seconds"]; ev += int(r["n_evals"][0]); la += r["launches"]
print(f"{n} single-decoy folds: {sec/n*1e3:.1f} ms each, {ev/n:.0f} evaluations, {la/n:.0f} launches, {sec/la*1e6:.1f} us per launch pair")
ctx.close()
