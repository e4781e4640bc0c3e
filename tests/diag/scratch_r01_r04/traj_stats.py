"""Summed accepted iterations, device / oracle, at fixed evaluation budgets, for several decoy sets (noise of the statistic).
usage: traj_stats.py <repo> [B]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
T = importlib.import_module("trrosettax2-dynamics_amd")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
g = os.path.join(sys.argv[1], "tests", "golden"); m = np.load(os.path.join(g, "seq_NMR.npz"))
ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"]); Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
runs = T.protocol.build_runs(90, 2)
for seed in (99, 7, 2024):
    t0 = np.stack([O.random_torsions(90, seed, d) for d in range(B)]).astype(np.float32)
    out = []
    for n in (20, 80, 160, 400):
        r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
        orc = [O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=n)[2]["n_iters"] for d in range(B)]
        out.append((n, r["n_iters"].sum() / sum(orc), int((np.array(orc) == r["n_iters"]).sum())))
    print(f"seed {seed:5d} B={B}: " + "  ".join(f"{n}: ratio {q:.2f} (identical counts {k}/{B})" for n, q, k in out))
ctx.close()
