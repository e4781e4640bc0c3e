"""Oracle-only scan of energy-model variants on the reference's example maps: how often do decoys end twisted or in the
mirror topology?  usage: model_scan.py <label> [n_decoys]   (constants come from include/trx2_model.h at build time)"""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd
spec = importlib.util.spec_from_file_location("protocol", os.path.join(ROOT, "trrosettax2-dynamics_amd", "protocol.py")); P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
G = os.path.join(ROOT, "tests", "golden"); dec = np.load(os.path.join(G, "ref_decoys.npz"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cart = len(sys.argv) > 3 and sys.argv[3] == "cart"
runs = P.build_runs(90, 2, cartesian_stage=cart)
out = []
for tag, refs in (("NMR", ("conf_2_1", "conf_2_2")), ("Xray", ("conf_1_1", "conf_1_2"))):
    m = np.load(os.path.join(G, f"seq_{tag}.npz")); T = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    t0 = time.time(); rows = []
    for d in range(n):
        tors, xyz, st = O.fold(T, O.random_torsions(90, 4242, d), runs)
        ca = xyz[:, 1]
        rm = min(kabsch_rmsd(ca, dec[k][:, 1]) for k in refs); mir = min(kabsch_rmsd(ca * [1, 1, -1], dec[k][:, 1]) for k in refs)
        dw = np.degrees(np.abs((tors[:-1, 2] % (2 * np.pi)) - np.pi))
        _, geom = O.extract_internal(xyz)
        rows.append((rm, mir, dw.max(), (dw > 30).sum(), (tors[1:-1, 0] <= 0).mean(), st["n_evals"], st["e_final"][0],
                     np.degrees(geom[:, 3]).std(), dw.std(), geom[:, 1].std(), np.degrees(geom[:-1, 5]).std()))
    r = np.array(rows); good = r[:, 0] < 3
    print(f"{sys.argv[1]:28s} {tag:4s} n={n}  median rmsd {np.median(r[:,0]):.2f} (good only {np.median(r[good,0]):.2f})  mirror-trapped {int(((r[:,0]>3)&(r[:,1]<r[:,0])).sum())}  other>3A {int(((r[:,0]>3)&(r[:,1]>=r[:,0])).sum())}  "
          f"twisted>60deg {int((r[:,2]>60).sum())}  mean n(|dw|>30) {r[:,3].mean():.1f}  phi<=0 {r[:,4].mean():.2f}  evals {np.median(r[:,5]):.0f}  dist {np.median(r[:,6]):.0f}  N-CA-C sd {r[:,7].mean():.1f}  C-N-CA sd {r[:,10].mean():.1f}  CA-C bond sd {r[:,9].mean():.3f}  omega sd {r[:,8].mean():.1f}  [{time.time()-t0:.0f}s]")
