"""20 evaluations of 8 decoys at L=400 from the random start, with and without a Cartesian run in the protocol"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
m = S.make_map(L, seed=L); ctx = T.Context(0)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
for cart in (True, False):
    runs = T.protocol.build_runs(L, 2, cartesian_stage=cart)
    for ne in (20, 60):
        r = ctx.fold_batch(8, runs, seed=3, max_evals=ne)
        print(f"cart {cart} max_evals {ne}: n_evals {r['n_evals']} n_iters {r['n_iters']} status {r['status']} f {np.round(r['f'], 1)}")
ctx.close()
