#!/usr/bin/env python
"""VERDICT r5 item 6, oracle side: evaluations per fold against the L-BFGS history length m (TRX2_LBFGS_M; the device's Gram form is written
for 8).  The oracle is compiled once per m into a scratch directory; every variant folds the SAME seeded starts under the default protocol on
  (a) the synthetic L=150 all-channel map as the network gives it (config 3's map, what the initial batches fold), and
  (b) that map fed back once from a folded decoy (what every iteration of the metric's job folds: about twice the selected restraints).
Prints per m: evaluations (median, mean, max), accepted iterations, final energy median, C-alpha RMSD to the map's target (median).
Usage: python tests/diag/history_sweep.py [n_decoys=32] [m ...]      (test infrastructure: imports oracle/)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(m, n):
    import importlib
    import numpy as np
    from oracle import oracle as O
    from oracle.kabsch import kabsch_rmsd
    so = os.path.join(tempfile.gettempdir(), f"libtrx2oracle_m{m}.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-shared", f"-DTRX2_LBFGS_M={m}", "-o", so, os.path.join(ROOT, "oracle", "trx2_oracle.c"), "-lm"])
    O.build = lambda force=False: so
    T = importlib.import_module("trrosettax2-dynamics_amd")
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
    PD = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
    L = 150
    mp = S.make_map(L, seed=L)
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    ca = S.nerf_backbone(mp["tors"])[1]
    host = {k: mp[k] for k in ("dist", "theta", "omega", "phi")}
    # the fed-back map: one decoy folded with the SHIPPED history (m = 8 library of the repo), written and read back as the pipeline does
    import ctypes
    ref_so = O.__dict__.get("_ref_so")
    maps = {"initial": host}
    pdb = os.path.join(tempfile.gettempdir(), "hs_seed.pdb")
    if not os.path.exists(pdb + ".npz"):
        raise SystemExit("seed decoy missing")
    maps["fed_back"] = dict(np.load(pdb + ".npz"))
    for tag, h in maps.items():
        Tb = O.Tables(h["dist"], h["omega"], h["theta"], h["phi"], seq=mp["seq"])
        t0 = np.stack([O.random_torsions(L, 4242, d) for d in range(n)])
        _, xyz, st, _ = O.fold_batch(Tb, t0, runs, nthreads=min(n, O.usable_cores()))
        ev = np.array([s["n_evals"] for s in st]); it = np.array([s["n_iters"] for s in st]); f = np.array([s["f_final"] for s in st])
        rm = np.array([kabsch_rmsd(xyz[i][:, 1], ca) for i in range(n)])
        print(f"m={m:2d} {tag:9s} n={n}: evaluations median {np.median(ev):7.0f} mean {ev.mean():7.0f} max {ev.max():6d} | iterations mean {it.mean():7.0f} | "
              f"f median {np.median(f):12.1f} | RMSD to target median {np.median(rm):.3f} A, within 2 A {(rm < 2).mean():.2f}", flush=True)


def seed_decoy():
    """one decoy of the initial map folded with the repo's oracle (m = 8) and the map fed back from it (host mirror of the feedback step)"""
    import importlib
    import numpy as np
    from oracle import oracle as O
    T = importlib.import_module("trrosettax2-dynamics_amd")
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
    PD = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
    L = 150
    mp = S.make_map(L, seed=L)
    host = {k: mp[k] for k in ("dist", "theta", "omega", "phi")}
    Tb = O.Tables(host["dist"], host["omega"], host["theta"], host["phi"], seq=mp["seq"])
    _, xyz, _, _ = O.fold_batch(Tb, O.random_torsions(L, 99, 0)[None], T.protocol.build_runs(L, 2, fastrelax=True), nthreads=1)
    pdb = os.path.join(tempfile.gettempdir(), "hs_seed.pdb")
    PD.write_pdb(pdb, mp["seq"], xyz[0])
    fb = FB.feedback_labels(host, pdb, 1.0, True)
    np.savez(pdb + ".npz", **{k: fb[k] for k in ("dist", "theta", "omega", "phi")})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    elif len(sys.argv) > 1 and sys.argv[1] == "--seed":
        seed_decoy()
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
        ms = [int(x) for x in sys.argv[2:]] or [8, 12, 16]
        subprocess.check_call([sys.executable, __file__, "--seed"])
        for m in ms:
            subprocess.check_call([sys.executable, __file__, "--child", str(m), str(n)])
