#!/usr/bin/env python
"""Derive the ideal backbone geometry and the soft-sphere contact radii used by the fold's backbone prior
from the reference's committed PyRosetta decoys (tests/golden/ref_decoys.npz = example/output/seq/pred_pdb).
The Rosetta database is not in the reference tree, so these data are the only pin."""
import numpy as np, os, sys
g = np.load(os.path.join(os.path.dirname(__file__), "../tests/golden/ref_decoys.npz"))
names = [k for k in g.files if k.startswith("conf")]
X = np.stack([g[k] for k in names]).astype(np.float64)  # (8, L, 5, 3) N CA C O CB
seq = "".join(l.strip() for l in open(os.path.join(os.path.dirname(__file__), "../tests/golden/seq.fasta")) if l[0] != ">")
N, CA, C, O, CB = [X[:, :, i] for i in range(5)]
def ang(a, b, c):
    v, w = a - b, c - b
    return np.degrees(np.arccos((v * w).sum(-1) / np.linalg.norm(v, axis=-1) / np.linalg.norm(w, axis=-1)))
def dih(a, b, c, d):
    b0, b1, b2 = a - b, c - b, d - c
    b1n = b1 / np.linalg.norm(b1, axis=-1, keepdims=True)
    v = b0 - (b0 * b1n).sum(-1, keepdims=True) * b1n
    w = b2 - (b2 * b1n).sum(-1, keepdims=True) * b1n
    return np.degrees(np.arctan2((np.cross(b1n, v) * w).sum(-1), (v * w).sum(-1)))
def st(n, v):
    v = v[~np.isnan(v)]
    print(f"{n:14s} mean {v.mean():9.4f} sd {v.std():7.4f} median {np.median(v):9.4f}")
st("N-CA", np.linalg.norm(N - CA, axis=-1)); st("CA-C", np.linalg.norm(CA - C, axis=-1))
st("C-N+1", np.linalg.norm(C[:, :-1] - N[:, 1:], axis=-1)); st("C-O", np.linalg.norm(C - O, axis=-1))
st("CA-CB", np.linalg.norm(CA - CB, axis=-1))
st("N-CA-C", ang(N, CA, C)); st("CA-C-N", ang(CA[:, :-1], C[:, :-1], N[:, 1:])); st("C-N-CA", ang(C[:, :-1], N[:, 1:], CA[:, 1:]))
st("CA-C-O", ang(CA, C, O)); st("N-CA-CB", ang(N, CA, CB)); st("C-CA-CB", ang(C, CA, CB))
st("O-C-N+1", ang(O[:, :-1], C[:, :-1], N[:, 1:]))
st("imp C-N-CA-CB", dih(C, N, CA, CB)); st("dih N-C-CA-CB", dih(N, C, CA, CB))
st("N+1-CA-C-O", dih(N[:, 1:], CA[:, :-1], C[:, :-1], O[:, :-1]))
om = dih(CA[:, :-1], C[:, :-1], N[:, 1:], CA[:, 1:]); st("|omega|", np.abs(om)); print("cis:", (np.abs(om) < 90).sum())
# CB in the local frame used by the reference's virtual-CB formula: b=CA-N, c=C-CA, a=b x c
b = CA - N; c = C - CA; a = np.cross(b, c)
M = np.stack([a, b, c], -1)  # (..,3,3) columns
rhs = (CB - CA)
ok = ~np.isnan(rhs).any(-1)
coef = np.linalg.solve(M[ok], rhs[ok][..., None])[..., 0]
print("CB = ka*a + kb*b + kc*c + CA  fitted:", coef.mean(0), "sd", coef.std(0), " (reference virtual: -0.58273431 0.56802827 -0.54067466)")
phi = dih(C[:, :-1], N[:, 1:], CA[:, 1:], C[:, 1:]); psi = dih(N[:, :-1], CA[:, :-1], C[:, :-1], N[:, 1:])
print("frac phi<=0:", (phi <= 0).mean(axis=1))
# closest approaches by atom type pair and sequence separation
an = ["N", "CA", "C", "O", "CB"]
L = X.shape[1]
for sep_lo, sep_hi in ((2, 2), (3, 3), (4, 9999)):
    print(f"--- |i-j| in [{sep_lo},{sep_hi}]  (min / 0.1% / 1%)")
    for p in range(5):
        row = []
        for q in range(5):
            ds = []
            for i in range(L):
                for j in range(i + sep_lo, min(L, i + sep_hi + 1)):
                    ds.append(np.linalg.norm(X[:, i, p] - X[:, j, q], axis=-1)); ds.append(np.linalg.norm(X[:, j, p] - X[:, i, q], axis=-1))
            ds = np.concatenate(ds); ds = ds[~np.isnan(ds)]
            row.append(f"{an[p]}-{an[q]} {ds.min():.2f}/{np.percentile(ds,0.1):.2f}/{np.percentile(ds,1):.2f}")
        print("  ".join(row))
