"""As cart_det.py, 16 decoys, one fold: n_iters per decoy and how many decoys lost iterations.  usage: cart_det2.py <repo> L evals"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth"); P = T.protocol
L = int(sys.argv[2]); n = int(sys.argv[3]); B = 16
m = S.make_map(L, seed=L, n_moves=150); ctx = T.Context(0)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
rng = np.random.default_rng(L)
t0 = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.08 for _ in range(B)]).astype(np.float32)
runs = [dict(w=P.SF_CART, max_iter=1000, sep_lo=1, sep_hi=L, precheck=0, skip_to=0, cartesian=1)]
r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
print(r["n_iters"], "rejected trials in total:", int((n - 1) * B - r["n_iters"].sum()), " mean f", r["f"].mean())
ctx.close()
