"""phi / psi of a stretch of residues: the reference's initial decoys of a map against this build's draws (default protocol).
usage: loop_torsions.py <repo> <NMR|Xray> <first residue> <last residue> [n = 512]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
tag, lo, hi = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 512
g = os.path.join(sys.argv[1], "tests", "golden")
ref = np.load(os.path.join(g, "ref_decoys.npz"))
seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
names = {"NMR": ("conf_2_1", "conf_2_2", "conf_1_3", "conf_1_4"), "Xray": ("conf_1_1", "conf_1_2", "conf_2_3", "conf_2_4")}[tag]


def dihedral(a, b, c, d):
    b0, b1, b2 = a - b, c - b, d - c
    b1 = b1 / np.linalg.norm(b1)
    v, w = b0 - (b0 @ b1) * b1, b2 - (b2 @ b1) * b1
    return np.arctan2(np.cross(b1, v) @ w, v @ w)


def phipsi(x):
    N, CA, C = x[:, 0], x[:, 1], x[:, 2]
    L = len(x); ph = np.full(L, np.nan); ps = np.full(L, np.nan)
    for i in range(L):
        if i > 0: ph[i] = dihedral(C[i - 1], N[i], CA[i], C[i])
        if i + 1 < L: ps[i] = dihedral(N[i], CA[i], C[i], N[i + 1])
    return np.degrees(ph), np.degrees(ps)


m = np.load(os.path.join(g, f"seq_{tag}.npz"))
ctx = T.Context(0, lanes=2)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
r = ctx.fold_batch(n, T.protocol.build_runs(90, 2, fastrelax=True), seed=77)
ctx.close()
ours = np.array([phipsi(r["xyz"][d].astype(np.float64)) for d in range(n)])      # [n, 2, L]
refs = {k: phipsi(np.nan_to_num(ref[k]).astype(np.float64)) for k in names}
print(f"{tag} map, residues {lo}..{hi}: phi / psi of the reference decoys {names} | this build's {n} draws: share with phi > 0, circular median phi / psi of either sign group")
for i in range(lo - 1, hi):
    rr = "  ".join("%5.0f/%5.0f" % (refs[k][0][i], refs[k][1][i]) for k in names)
    ph, ps = ours[:, 0, i], ours[:, 1, i]
    pos = ph > 0
    cm = lambda v: np.degrees(np.angle(np.exp(1j * np.radians(v)).mean())) if len(v) else float("nan")
    print(f"  {i + 1:3d} {seq[i]}  ref {rr} | phi>0 {100 * pos.mean():3.0f} %: {cm(ph[pos]):5.0f}/{cm(ps[pos]):5.0f}   phi<0: {cm(ph[~pos]):5.0f}/{cm(ps[~pos]):5.0f}  (psi spread {np.degrees(np.sqrt(-2 * np.log(np.abs(np.exp(1j * np.radians(ps)).mean())))):.0f})")
