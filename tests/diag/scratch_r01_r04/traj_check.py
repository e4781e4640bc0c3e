"""Short-horizon agreement of the device minimiser with the oracle: same start torsions, same protocol, same evaluation
budget.  Prints energies after N evaluations for both.  usage: traj_check.py <repo>"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden"); m = np.load(os.path.join(g, "seq_NMR.npz"))
ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"])
Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
runs = T.protocol.build_runs(90, 2)
B = 4
t0 = np.stack([O.random_torsions(90, 99, d) for d in range(B)]).astype(np.float32)
print("evals  decoy   f_device       f_oracle      rel.diff   max|dtors| rad   (dev evals/iters | orc evals/iters)")
for n in (8, 20, 40, 80, 160, 400):
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
    for d in range(B):
        to, xo, st = O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=n)
        dt = np.abs((r["tors"][d] - to + np.pi) % (2 * np.pi) - np.pi).max()
        print(f"{n:5d}  {d:3d}  {r['f'][d]:12.2f}  {st['f_final']:12.2f}  {abs(r['f'][d]-st['f_final'])/abs(st['f_final']):9.2e}  {dt:10.2e}      ({r['n_evals'][d]}/{r['n_iters'][d]} | {st['n_evals']}/{st['n_iters']})")
ctx.close()
