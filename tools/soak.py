"""Soak: many folds, maps, lanes, feedback steps and context create/destroy cycles; device memory must come back.
usage: soak.py <repo> [cycles=30]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); P = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
g = os.path.join(sys.argv[1], "tests", "golden"); seq = P.read_fasta(os.path.join(g, "seq.fasta"))
mn, mx = (dict(np.load(os.path.join(g, f"seq_{t}.npz"))) for t in ("NMR", "Xray"))
runs = T.protocol.build_runs(90, 2)
free0 = torch.cuda.mem_get_info(0)[0]
t0 = time.time(); n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for c in range(n):
    ctx = T.Context(0, lanes=1 + c % 2)
    m = mn if c % 3 else mx
    if c % 4 == 0: ctx.set_map(m["dist"], seq=seq)
    else: ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    r = ctx.fold_batch((1, 7, 40, 64)[c % 4], runs, seed=c)
    assert np.all(r["status"] == 0)
    for k in range(3):
        xyz, s = P.as_read_from_pdb(seq, r["xyz"][0])
        d = ctx.feedback_step(xyz, s, 1.0, angle=bool(c % 4))
        r = ctx.fold_batch(1, runs, seed=100 + c * 10 + k)
    if c % 5 == 0: ctx.glocon_matrix(np.stack([P.as_read_from_pdb(seq, r["xyz"][0])[0]] * 3), [seq] * 3)
    ctx.close()
    if c == 9:
        free0 = torch.cuda.mem_get_info(0)[0]   # after every code path has run once: runtime pools, code objects, scratch exist
    if c % 10 == 9:
        print(f"cycle {c + 1}: device memory in use by this process vs start: {(free0 - torch.cuda.mem_get_info(0)[0]) / 2**20:.1f} MiB, {time.time() - t0:.0f} s", flush=True)
leak = (free0 - torch.cuda.mem_get_info(0)[0]) / 2**20
print(f"after {n} cycles: {leak:.1f} MiB more than after cycle 10")
sys.exit(1 if leak > 16 else 0)
