"""How unbalanced is the per-lane contact walk of the pair kernel?  For the final coordinates of a config-2 fold: per wave
(residue a, b-range split, wave) the number of contacts (|CA-CA| < 8.5 A, |a-b| >= 3) of each of the 64 decoys.
The walk takes max-over-lanes steps; a perfectly shared walk would take ceil(total / 64).  usage: vdw_walk_stats.py <repo>"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, nsplit = 150, 64, 3
m = S.make_map(L); ctx = T.Context(0); ctx.set_map(m["dist"], seq=m["seq"])
r = ctx.fold_batch(B, T.protocol.build_runs(L, 2), seed=150); ctx.close()
ca = r["xyz"][:, :, 1]                                            # [B, L, 3]
d2 = ((ca[:, :, None] - ca[:, None]) ** 2).sum(-1)                # [B, L, L]
sep = np.abs(np.arange(L)[:, None] - np.arange(L)[None])
con = (d2 < 8.5 ** 2) & (sep >= 3)[None]                          # [B, a, b]
chunk = (L + nsplit - 1) // nsplit
mx, tot, anyv = [], [], []
for a in range(L):
    for s in range(nsplit):
        lo, hi = s * chunk, min(L, (s + 1) * chunk)
        for w in range(4):
            bs = np.arange(lo + w, hi, 4)
            c = con[:, a, bs]                                     # [B, visits]
            per = c.sum(1)
            mx.append(per.max()); tot.append(per.sum()); anyv.append(int(c.any(0).sum()))
mx, tot, anyv = np.array(mx), np.array(tot), np.array(anyv)
print(f"waves {len(mx)}; visits per wave {chunk/4:.1f}")
print(f"lockstep (old): visits with any contact            mean {anyv.mean():.2f} per wave")
print(f"per-lane walk (now): max contacts over 64 lanes    mean {mx.mean():.2f}")
print(f"perfectly shared: ceil(total / 64)                 mean {np.ceil(tot / 64).mean():.2f}   (total contacts per wave {tot.mean():.1f})")
