"""Scatter of the 20-evaluation tracking ratio at L=400 (tests/test_gpu_configs.py, configuration 4): accepted iterations device /
oracle from the reference's random starts, for several seeds and batch sizes.  TRX2FOLD_LIB selects the build.
usage: track_L400.py <repo>"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = 400
m = S.make_map(L, seed=L); ctx = T.Context(0)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"]); Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
runs = T.protocol.build_runs(L, 2)
tot_g = tot_o = 0
for seed in (400, 401, 402, 403):
    t0 = np.stack([O.random_torsions(L, seed, d) for d in range(32)]).astype(np.float32)
    r = ctx.fold_batch(32, runs, tors0=t0, max_evals=20)
    _, _, st, _ = O.fold_batch(Tb, t0.astype(np.float64), runs, max_evals=20)
    oi = np.array([s["n_iters"] for s in st])
    tot_g += r["n_iters"].sum(); tot_o += oi.sum()
    print(f"{os.path.basename(os.environ.get('TRX2FOLD_LIB', 'default')):24s} seed {seed}: first 16: {r['n_iters'][:16].sum() / oi[:16].sum():.3f}  last 16: {r['n_iters'][16:].sum() / oi[16:].sum():.3f}  "
          f"all 32: {r['n_iters'].sum() / oi.sum():.3f}  identical counts {(oi == r['n_iters']).sum()}/32", flush=True)
print(f"   128 decoys: {tot_g / tot_o:.4f}")
ctx.close()
