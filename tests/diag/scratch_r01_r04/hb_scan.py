"""CPU: scan the weight of the backbone hydrogen-bond surrogate with the ORACLE (OpenMP over decoys) on the reference's two
example maps: median C-alpha RMSD to the closer of the reference's two initial PyRosetta decoys, trapped starts, evaluations.
usage: python tools/hb_scan.py <decoys per cell> <hb weights in the torsion stages, comma separated> [cart weight factor]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd

T = importlib.import_module("trrosettax2-dynamics_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
weights = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,5").split(",")]
cart_factor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6   # hbond_*_bb 3.0 against cen_hb 5.0
g = os.path.join(ROOT, "tests", "golden")
seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
dec = np.load(os.path.join(g, "ref_decoys.npz"))
for tag, refs in (("NMR", ("conf_2_1", "conf_2_2")), ("Xray", ("conf_1_1", "conf_1_2"))):
    m = np.load(os.path.join(g, f"seq_{tag}.npz"))
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    t0 = np.stack([O.random_torsions(90, 777, d) for d in range(n)])
    for w in weights:
        runs = T.protocol.build_runs(90, 2)
        for r in runs:
            r["w"] = list(r["w"])
            if r["w"][0] != 0:  # stages with restraints: sf / sf1 carry cen_hb, sf_cart carries hbond_sr_bb + hbond_lr_bb
                r["w"][7] = w * (cart_factor if r["cartesian"] else 1.0)
        tors, xyz, st, used = O.fold_batch(Tb, t0, runs)
        best = np.array([min(kabsch_rmsd(xyz[i, :, 1], dec[k][:, 1]) for k in refs) for i in range(n)])
        hb = np.array([s["e_final"][8] for s in st])
        ev = np.array([s["n_evals"] for s in st])
        good = best[best < 3]
        print(f"{tag:5s} w_hb {w:5.2f}: median {np.median(best):.3f} (good only {np.median(good):.3f}) q25-75 {np.percentile(best,25):.2f}-{np.percentile(best,75):.2f} "
              f"<=0.5: {(best<=0.5).mean():.2f} >3A: {(best>3).sum()}/{n}  E_hb median {np.median(hb):.1f}  evals {int(np.median(ev))}", flush=True)
