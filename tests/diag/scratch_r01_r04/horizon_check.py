"""Device vs oracle after n evaluations from the same start, for growing n: where do the trajectories separate?
usage: horizon_check.py <repo> <L> [noise=0.15]"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = int(sys.argv[2]); noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15; B = 3
m = S.make_map(L, seed=L, n_moves=150); ctx = T.Context(0)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
runs = T.protocol.build_runs(L, 2, cartesian_stage=False)
rng = np.random.default_rng(7)
t0 = np.stack([m["tors"] + rng.normal(size=(L, 3)) * noise for _ in range(B)]).astype(np.float32)
print("  n decoy        f_device        f_oracle   rel.diff  iters dev/orc  run dev")
for n in (1, 2, 3, 4, 6, 8, 12, 16, 30):
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
    for d in range(B):
        to, xo, st = O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=n)
        print("%3d %4d  %14.4f  %14.4f  %9.2e   %3d / %3d" % (n, d, r["f"][d], st["f_final"], abs(r["f"][d] - st["f_final"]) / abs(st["f_final"]), r["n_iters"][d], st["n_iters"]))
ctx.close()
