"""Churn soak of batch mode: many targets of mixed chain lengths (all three step-kernel classes and both pair-kernel channel sets
share the engines' launches, folds join and leave all the time), folded twice with different numbers of targets in flight.
Every target must succeed and every PDB must be byte for byte the same in both passes (the files do not depend on scheduling).
usage: soak_batch.py <repo> [n targets = 24] [Nmax = 6] [in flight A = 16] [in flight B = 5]"""
import contextlib, hashlib, importlib, io, json, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, sys.argv[1])
S = importlib.import_module("trrosettax2-dynamics_amd.synth"); PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
LB = importlib.import_module("trrosettax2-dynamics_amd._lib")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 6
flights = [int(sys.argv[4]) if len(sys.argv) > 4 else 16, int(sys.argv[5]) if len(sys.argv) > 5 else 5]
Ls = [60, 90, 130, 150, 220, 300]
work = tempfile.mkdtemp(prefix="trx2_soakb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    fdir = os.path.join(work, "fasta"); os.makedirs(fdir)
    names, pred = [], {}
    for k in range(n):
        L = Ls[k % len(Ls)]
        nm = f"s{k:02d}_L{L}"
        names.append(nm)
        open(os.path.join(fdir, nm + ".fasta"), "w").write(f">{nm}\n{'A' * L}\n")
    # run_batch reads {save_dir}/{name}/pred_npz/{name}_{NMR,Xray}.npz when no npz is given: one pair of synthetic maps per length
    digests = []
    for k_pass, tif in enumerate(flights):
        save = os.path.join(work, f"out{k_pass}")
        for nm in names:
            L = int(nm.split("_L")[1])
            d = os.path.join(save, nm, "pred_npz"); os.makedirs(d, exist_ok=True)
            for c, tag in enumerate(("NMR", "Xray")):
                m = S.make_map(L, seed=L + c, n_moves=150)
                np.savez(os.path.join(d, f"{nm}_{tag}.npz"), dist=m["dist"], omega=m["omega"], theta=m["theta"], phi=m["phi"])
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = PL.run_batch(names, fdir, save, targets_in_flight=tif, init_num=2, Nmax=nmax, angle=True, mult_two_models=True, seed=11)
        el = time.perf_counter() - t0
        files = {}
        for nm in names:
            pd = os.path.join(save, nm, "pred_pdb")
            for f in sorted(os.listdir(pd)):
                files[(nm, f)] = hashlib.sha256(open(os.path.join(pd, f), "rb").read()).hexdigest()
        digests.append(files)
        print(json.dumps(dict(targets=n, targets_in_flight=tif, Nmax=nmax, decoys=res["decoys"], failed=res["failed"], errors=res["errors"][:3], files=len(files),
                              seconds=round(el, 1), engines={k: round(v, 1) for k, v in LB.shared_launch_stats(0).items() if k in ("chunks", "folds_per_launch", "folds")})), flush=True)
        assert res["failed"] == 0, res["errors"]
    same = digests[0].keys() == digests[1].keys() and all(digests[0][k] == digests[1][k] for k in digests[0])
    print("files identical between the two passes:", same, f"({len(digests[0])} files)")
    sys.exit(0 if same else 1)
finally:
    shutil.rmtree(work, ignore_errors=True)
