import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
T = importlib.import_module("trrosettax2-dynamics_amd"); LB = importlib.import_module("trrosettax2-dynamics_amd._lib")
g = os.path.join(sys.argv[1], "tests", "golden")
seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
m = np.load(os.path.join(g, "seq_NMR.npz"))
LB.set_shared_launches(int(sys.argv[2]))
waves = int(sys.argv[3])
runs = T.protocol.build_runs(90, 2, fastrelax=True)
c = T.Context(0); c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq); c.set_single_decoy_waves(waves)
out = []
for rep in range(3):
    for seed in (40, 41):
        r = c.fold_batch(1, runs, seed=seed, max_evals=int(sys.argv[4]))
        out.append((seed, int(r["n_evals"][0]), float(r["f"][0]), float(np.abs(r["xyz"]).sum())))
for o in out: print(o)
