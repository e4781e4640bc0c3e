"""run_inference on ONE target (pipeline.run_single: two chains, single-decoy folds with four waves per row): decoys/sec.
usage: e2e_single.py <repo> <L> <Nmax> [init_num=10] [single-decoy waves per row = 4]"""
import contextlib, importlib, io, json, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, sys.argv[1])
S = importlib.import_module("trrosettax2-dynamics_amd.synth"); PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
L, nmax = int(sys.argv[2]), int(sys.argv[3]); init = int(sys.argv[4]) if len(sys.argv) > 4 else 10
waves = int(sys.argv[5]) if len(sys.argv) > 5 else 4
work = tempfile.mkdtemp(prefix="trx2_e2es_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    maps = [S.make_map(L, seed=L + c) for c in range(2)]
    paths = []
    for tag, m in zip(("NMR", "Xray"), maps):
        q = os.path.join(work, f"m_{tag}.npz"); np.savez(q, dist=m["dist"], omega=m["omega"], theta=m["theta"], phi=m["phi"]); paths.append(q)
    fa = os.path.join(work, "t.fasta"); open(fa, "w").write(f">t\n{maps[0]['seq']}\n")
    ph = {}
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        n = PL.run_single("t", fa, os.path.join(work, "out"), init_num=init, Nmax=nmax, npz_nmr=paths[0], npz_xray=paths[1], seed=7, phase_times=ph, single_decoy_waves=waves)
    el = time.perf_counter() - t0
    its = sum(v["iterations"] for v in ph.values())
    print(json.dumps(dict(L=L, waves=waves, Nmax=nmax, decoys=n, seconds=round(el, 2), decoys_per_sec=round(n / el, 1),
                          ms_per_iteration=round(1e3 * sum(v["iteration_s"] for v in ph.values()) / max(its, 1), 1))))
finally:
    shutil.rmtree(work, ignore_errors=True)
