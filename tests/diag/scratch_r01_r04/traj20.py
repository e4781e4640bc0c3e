"""per-decoy agreement with the oracle at 20 evaluations (calibration of the trajectory test).  usage: traj20.py <repo>"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden"); m = np.load(os.path.join(g, "seq_NMR.npz"))
ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"]); Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
runs = T.protocol.build_runs(90, 2); B = 12
for seed in (99, 7, 2024):
    t0 = np.stack([O.random_torsions(90, seed, d) for d in range(B)]).astype(np.float32)
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=20)
    orc = [O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=20)[2] for d in range(B)]
    rel = np.array([abs(r["f"][d] - orc[d]["f_final"]) / abs(orc[d]["f_final"]) for d in range(B)])
    same = int(sum(int(r["n_iters"][d]) == orc[d]["n_iters"] for d in range(B)))
    print(f"seed {seed:5d}: identical iteration counts {same}/12   rel |df| sorted: " + " ".join(f"{v:.0e}" for v in np.sort(rel)) + f"   n(<1e-3)={int((rel<1e-3).sum())} n(<1e-2)={int((rel<1e-2).sum())}")
ctx.close()
