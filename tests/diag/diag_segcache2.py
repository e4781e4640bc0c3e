"""diagnostic: one evaluation (eval_batch, B=1) at several torsion sets, repeated so that the second pass hits the segment cache; saves e_terms / grad"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
T = importlib.import_module("trrosettax2-dynamics_amd")
from oracle import oracle as O
g = os.path.join(sys.argv[1], "tests", "golden")
seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
m = np.load(os.path.join(g, "seq_NMR.npz"))
c = T.Context(0); c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq); c.set_single_decoy_waves(int(sys.argv[2]))
w = np.array(T.protocol.SF, np.float32)
rng = np.random.default_rng(0)
base = np.stack([O.random_torsions(90, 5, d) for d in range(4)]).astype(np.float32)
out = []
for rep in range(2):
    for d in range(4):
        for eps in (0.0, 1e-3):
            t = (base[d] + eps)[None].astype(np.float32)
            f, e, gr, xyz = c.eval_batch(t, w)
            out.append(np.concatenate([e.ravel(), gr.ravel().astype(np.float64)]))
np.save(sys.argv[3], np.stack(out))
