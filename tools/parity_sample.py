"""Large-sample outcome parity on the reference's example maps: C-alpha RMSD of folded decoys to the reference's PyRosetta
decoys (closest of the two initial decoys of the same map), for the full protocol and the torsion-only one.
usage: parity_sample.py <repo> [n_batches of 64] [first seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden"); dec = np.load(os.path.join(g, "ref_decoys.npz"))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
ctx = T.Context(0)
for tag, refs in (("NMR", ("conf_2_1", "conf_2_2")), ("Xray", ("conf_1_1", "conf_1_2"))):
    m = np.load(os.path.join(g, f"seq_{tag}.npz")); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"])
    for label, cart in (("full protocol (Cartesian stage)", True), ("torsion-only", False)):
        rm, mir, tw, bsd, asd, ev, sec = [], [], [], [], [], [], 0.0
        for b in range(nb):
            r = ctx.fold_batch(64, T.protocol.build_runs(90, 2, cartesian_stage=cart), seed=seed0 + b)
            assert np.all(r["status"] == 0)
            sec += r["seconds"]; ev += list(r["n_evals"])
            for i in range(64):
                ca = r["xyz"][i, :, 1]
                rm.append(min(kabsch_rmsd(ca, dec[k][:, 1]) for k in refs)); mir.append(min(kabsch_rmsd(ca * [1, 1, -1], dec[k][:, 1]) for k in refs))
                dw = np.degrees(np.abs((r["tors"][i, :-1, 2] % (2 * np.pi)) - np.pi)); tw.append(dw.max() > 60)
                gg = O.extract_internal(r["xyz"][i].astype(np.float64))[1]; bsd.append(gg[:, 1].std()); asd.append(np.degrees(gg[:, 3]).std())
        rm, mir = np.array(rm), np.array(mir); n = len(rm); gross = rm > 3
        print(f"{tag:4s} {label:32s} n={n}: RMSD median {np.median(rm):.2f}  quartiles {np.percentile(rm,25):.2f}-{np.percentile(rm,75):.2f}  "
              f"<=0.5A {100*(rm<=0.5).mean():.0f}%  <=1A {100*(rm<=1).mean():.0f}%  >3A {100*gross.mean():.1f}% (mirror {100*(gross&(mir<rm)).mean():.1f}%)  "
              f"twisted>60 {100*np.mean(tw):.0f}%  CA-C sd {np.mean(bsd):.3f}  N-CA-C sd {np.mean(asd):.1f}  evals {np.median(ev):.0f}  {n/sec:.0f} decoys/s")
ctx.close()
