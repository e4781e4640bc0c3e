"""CPU (oracle, OpenMP): does a synthetic map fold from random starts?  usage: oracle_fold_quality.py L n orient(0/1) [kind] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, n, orient = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else None
seed = int(sys.argv[5]) if len(sys.argv) > 5 else None
m = S.make_map(L, seed=seed, kind=kind)
Tb = O.Tables(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else [None, None, None]))
t0 = np.stack([O.random_torsions(L, 150, d) for d in range(n)])
t = time.time()
tors, xyz, st, used = O.fold_batch(Tb, t0, T.protocol.build_runs(L, 2))
ca = S.nerf_backbone(m["tors"])[1]
rm = np.array([kabsch_rmsd(xyz[i, :, 1], ca) for i in range(n)])
mir = np.array([kabsch_rmsd(xyz[i, :, 1] * np.array([1, 1, -1.0]), ca) for i in range(n)])
_, e_t, _, _ = O.evaluate(Tb, m["tors"], np.array(T.protocol.SF, float), grad=False)
depth = np.array([s["e_final"][0] for s in st]) / e_t[0]
print(f"L={L} orient={orient} kind={kind}: RMSD to target {np.round(np.sort(rm),1)}\n mirror {np.round(np.sort(mir),1)}\n depth median {np.median(depth):.3f} min {depth.min():.3f}; "
      f"evals median {int(np.median([s['n_evals'] for s in st]))}; {time.time()-t:.0f} s on {used} threads")
