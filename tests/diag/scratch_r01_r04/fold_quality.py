"""GPU: does a bench config actually FOLD?  Folds one batch from random starts and reports, against the synthetic map's own target
structure: C-alpha RMSD (and to the mirror image), distance-restraint energy relative to the target's, evaluations.
usage: fold_quality.py <repo> <config 2|3|4> [B] [kind]"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle.kabsch import kabsch_rmsd
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, orient = {2: (150, 64, False), 3: (150, 64, True), 4: (400, 32, True)}[int(sys.argv[2])]
if len(sys.argv) > 3: B = int(sys.argv[3])
kw = dict(kind=sys.argv[4]) if len(sys.argv) > 4 else {}
m = S.make_map(L, **kw); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
w = np.array(T.protocol.SF, np.float64)
_, e_t, _, _ = ctx.eval_batch(np.asarray(m["tors"], np.float32)[None], w)
r = ctx.fold_batch(B, T.protocol.build_runs(L, 2), seed=150)
ca = S.nerf_backbone(m["tors"])[1]
rm = np.array([kabsch_rmsd(r["xyz"][i, :, 1], ca) for i in range(B)])
mir = np.array([kabsch_rmsd(r["xyz"][i, :, 1] * np.array([1, 1, -1.0]), ca) for i in range(B)])
depth = r["e_terms"][:, 0] / e_t[0, 0]
print(f"config {sys.argv[2]} L={L} B={B}: RMSD to target median {np.median(rm):.2f} (min {rm.min():.2f}, <2A: {(rm<2).sum()}), to mirror image median {np.median(mir):.2f} "
      f"(<2A: {(mir<2).sum()}); dist-energy depth median {np.median(depth):.3f} min {depth.min():.3f} max {depth.max():.3f}; target E_dist {e_t[0,0]:.0f}; "
      f"evals median {int(np.median(r['n_evals']))}; status!=0: {(r['status']!=0).sum()}; {r['seconds']:.2f} s")
ctx.close()
