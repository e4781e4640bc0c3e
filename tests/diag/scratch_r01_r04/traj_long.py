"""Minimiser lockstep with the oracle for a chain longer than 256 residues (2 residues per thread in k_chain).
usage: traj_long.py <repo> [L]"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300; B = 6
m = S.make_map(L, seed=L, n_moves=150); ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"]); runs = T.protocol.build_runs(L, 2)
t0 = np.stack([O.random_torsions(L, 5, d) for d in range(B)]).astype(np.float32)
for n in (20, 60, 150):
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
    orc = [O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=n)[2] for d in range(B)]
    rel = [abs(r["f"][d] - orc[d]["f_final"]) / abs(orc[d]["f_final"]) for d in range(B)]
    print(f"L={L} evals {n:4d}: iters dev {list(map(int, r['n_iters']))} orc {[o['n_iters'] for o in orc]}  ratio {r['n_iters'].sum()/sum(o['n_iters'] for o in orc):.2f}  rel|df| median {np.median(rel):.1e} max {max(rel):.1e}")
ctx.close()
