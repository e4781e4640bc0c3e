"""Cartesian-only run from near-native starts, device against oracle (the scenario of tests/test_gpu_cartesian.py).  usage: cart_track.py <repo> L"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth"); P = T.protocol
L = int(sys.argv[2]); B = 4
m = S.make_map(L, seed=L, n_moves=150); ctx = T.Context(0)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
rng = np.random.default_rng(L)
t0 = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.08 for _ in range(B)]).astype(np.float32)
runs = [dict(w=P.SF_CART, max_iter=1000, sep_lo=1, sep_hi=L, precheck=0, skip_to=0, cartesian=1)]
for n in (12, 20, 30, 40):
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
    out = []
    for d in range(B):
        to, xo, st = O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=n)
        out.append(f"{int(r['n_iters'][d])}/{st['n_iters']} {abs(r['f'][d]-st['f_final'])/abs(st['f_final']):.1e} {kabsch_rmsd(r['xyz'][d].reshape(-1, 3), xo.reshape(-1, 3)):.3f}")
    print(f"L={L} evals {n}: " + " | ".join(out))
ctx.close()
