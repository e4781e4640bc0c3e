"""Batch mode of run_inference on ONE GPU: n targets (the same synthetic pair of maps under n names), folded 1 / 2 / .. at a time.
usage: e2e_batch.py <repo> <L> <n targets> <Nmax> <targets in flight ...>   (0 = run_batch's default)"""
import contextlib, importlib, io, json, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, sys.argv[1])
S = importlib.import_module("trrosettax2-dynamics_amd.synth"); PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
L, n, nmax = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ks = [int(x) for x in sys.argv[5:]] or [1, 2]
work = tempfile.mkdtemp(prefix="trx2_e2eb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    maps = [S.make_map(L, seed=L + c) for c in range(2)]
    paths = []
    for tag, m in zip(("NMR", "Xray"), maps):
        q = os.path.join(work, f"m_{tag}.npz"); np.savez(q, dist=m["dist"], omega=m["omega"], theta=m["theta"], phi=m["phi"]); paths.append(q)
    names = [f"t{i}" for i in range(n)]
    fdir = os.path.join(work, "fasta"); os.makedirs(fdir)
    for nm in names:
        open(os.path.join(fdir, nm + ".fasta"), "w").write(f">{nm}\n{maps[0]['seq']}\n")
    for k in ks:
        save = os.path.join(work, f"out{k}")
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = PL.run_batch(names, fdir, save, targets_in_flight=(k or None), init_num=10, Nmax=nmax, angle=True, mult_two_models=True, seed=3,
                               npz_nmr=paths[0], npz_xray=paths[1])
        el = time.perf_counter() - t0
        print(json.dumps(dict(L=L, targets=n, Nmax=nmax, targets_in_flight=k, decoys=res["decoys"], failed=res["failed"], seconds=round(el, 2),
                              decoys_per_sec=round(res["decoys"] / el, 1), engines=importlib.import_module("trrosettax2-dynamics_amd._lib").shared_launch_stats(0))), flush=True)
        shutil.rmtree(save, ignore_errors=True)
finally:
    shutil.rmtree(work, ignore_errors=True)
