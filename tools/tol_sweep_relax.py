"""Round 5: evaluations per decoy against outcome, WITH the relax stage on (the default protocol, 35 runs).
Cells: tolerance of the 14 centroid-stage runs (include/trx2_model.h TRX2_MIN_TOL; the reference hands MinMover 1e-4, folding.py:91)
x scale of the FastRelax scripts' tolerances (protocol.RELAX_TOL_SCALE; 1relax_round1.txt / 2relax_round2.txt carry 0.01 and 0.00001),
plus mixed schedules.  Per cell and map, n decoys: median C-alpha RMSD to the closer of the reference's two initial decoys of the map,
fractions within 0.5 / 1 A, beyond 3 A, peptides twisted beyond 60 degrees, mean |dphi|,|dpsi| to the closer reference decoy and between
two draws of ours, CA-C / N-CA-C spread, evaluations (median, mean, max).
usage: tol_sweep_relax.py <repo> [decoys per cell and map = 1024] [first seed = 1000] [part: grid | mixed | all]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
P = T.protocol
g = os.path.join(sys.argv[1], "tests", "golden"); dec = np.load(os.path.join(g, "ref_decoys.npz"))
n_dec = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
part = sys.argv[4] if len(sys.argv) > 4 else "all"
seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
MAPS = (("NMR", ("conf_2_1", "conf_2_2")), ("Xray", ("conf_1_1", "conf_1_2")))


def rmsd_many(X, q):
    """X[n,L,3] against q[L,3] -> rmsd[n] (Kabsch, proper rotations)"""
    X = X - X.mean(1, keepdims=True); q = q - q.mean(0)
    H = np.einsum("nli,lj->nij", X, q)
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(U @ Vt))
    e0 = (X ** 2).sum((1, 2)) + (q ** 2).sum()
    return np.sqrt(np.maximum(0.0, (e0 - 2 * (S[:, 0] + S[:, 1] + d * S[:, 2])) / X.shape[1]))


def dih(a, b, c, d_):
    b0, b1, b2 = a - b, c - b, d_ - c
    b1 = b1 / np.linalg.norm(b1, axis=-1, keepdims=True)
    v = b0 - (b0 * b1).sum(-1, keepdims=True) * b1; w = b2 - (b2 * b1).sum(-1, keepdims=True) * b1
    return np.arctan2((np.cross(b1, v) * w).sum(-1), (v * w).sum(-1))


def phipsi(x):  # x[..., L, 5, 3] -> [..., L-2, 2]
    N_, CA_, C_ = x[..., 0, :].astype(np.float64), x[..., 1, :].astype(np.float64), x[..., 2, :].astype(np.float64)
    return np.stack([dih(C_[..., :-2, :], N_[..., 1:-1, :], CA_[..., 1:-1, :], C_[..., 1:-1, :]),
                     dih(N_[..., 1:-1, :], CA_[..., 1:-1, :], C_[..., 1:-1, :], N_[..., 2:, :])], -1)


def cdiff(a, b):  # mean |circular difference| in degrees over residues, phi and psi separately -> [..., 2]
    return np.degrees(np.abs((a - b + np.pi) % (2 * np.pi) - np.pi)).mean(-2)


ctx = T.Context(0, lanes=2)
maps = {t: np.load(os.path.join(g, f"seq_{t}.npz")) for t, _ in MAPS}


def sample(tag, refs, runs):
    m = maps[tag]; ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    xs, ts, ev, sec = [], [], [], 0.0
    left, b = n_dec, 0
    while left > 0:
        nb = min(512, left)
        r = ctx.fold_batch(nb, runs, seed=seed0 + b)
        assert np.all(r["status"] == 0), r["status"]
        xs.append(r["xyz"]); ts.append(r["tors"]); ev.append(r["n_evals"]); sec += r["seconds"]; left -= nb; b += 1
    x = np.concatenate(xs); t = np.concatenate(ts); ev = np.concatenate(ev).astype(float)
    ca = x[:, :, 1].astype(np.float64)
    rr = np.stack([rmsd_many(ca, dec[k][:, 1]) for k in refs], 1); rm = rr.min(1)
    dw_ = np.degrees(np.abs((t[:, :-1, 2] % (2 * np.pi)) - np.pi))
    tw = dw_[:, 1:].max(1) > 60            # the first peptide apart: Rosetta does not tether a terminus (conf_1_1's is cis)
    tw0 = dw_[:, 0] > 60
    ok = rm < 3.0
    pp = phipsi(x[ok]); ppr = [phipsi(dec[k]) for k in refs]
    to = np.stack([cdiff(pp, pr) for pr in ppr], 1)                 # [n, 2 refs, 2 angles]
    to = to[np.arange(len(to)), to.sum(-1).argmin(1)]               # the closer reference draw
    rng = np.random.default_rng(0); ij = rng.integers(0, len(pp), size=(4000, 2)); ij = ij[ij[:, 0] != ij[:, 1]]
    own = cdiff(pp[ij[:, 0]], pp[ij[:, 1]])
    xd = x.astype(np.float64)
    cac = np.linalg.norm(xd[:, :, 2] - xd[:, :, 1], axis=-1).std(1).mean()
    u, v = xd[:, :, 0] - xd[:, :, 1], xd[:, :, 2] - xd[:, :, 1]
    ang = np.degrees(np.arccos((u * v).sum(-1) / np.linalg.norm(u, axis=-1) / np.linalg.norm(v, axis=-1))).std(1).mean()
    return (f"{tag} med {np.median(rm):.3f} q {np.percentile(rm,25):.2f}-{np.percentile(rm,75):.2f} <=.5 {100*(rm<=0.5).mean():4.1f}% <=1 {100*(rm<=1).mean():4.1f}% "
            f">3 {100*(rm>3).mean():3.1f}% tw {100*tw.mean():3.1f}% (first peptide {100*tw0.mean():3.1f}%) dphi/dpsi ref {np.median(to[:,0]):4.1f}/{np.median(to[:,1]):4.1f} own {np.median(own[:,0]):4.1f}/{np.median(own[:,1]):4.1f} "
            f"CA-C sd {cac:.4f} NCAC sd {ang:.2f} ev med {np.median(ev):.0f} mean {ev.mean():.0f} max {ev.max():.0f} [{sec:.1f}s]")


def protocol(c_tol, r_scale, closing_tol=None, closing_iter=None, first_declash=None, last_declash=None):
    runs = P.build_runs(90, 2, fastrelax=True)
    assert len(runs) == 35
    for i, r in enumerate(runs):
        if i < 14:
            r["tol"] = c_tol
        else:
            r["tol"] = r["tol"] / P.RELAX_TOL_SCALE * r_scale
    if first_declash is not None:
        for r in runs[:5]: r["tol"] = first_declash
    if last_declash is not None:
        for r in runs[9:14]: r["tol"] = last_declash
    if closing_tol is not None: runs[-1]["tol"] = closing_tol
    if closing_iter is not None: runs[-1]["max_iter"] = closing_iter
    return runs


def cell(name, runs):
    t0 = time.time()
    line = f"{name:44s}"
    for tag, refs in MAPS:
        line += " | " + sample(tag, refs, runs)
    print(line + f"  ({time.time()-t0:.0f}s wall)", flush=True)


print(f"# {n_dec} decoys per cell and map, seeds from {seed0}; shipped: centroid {float(getattr(P, 'CENTROID_TOL', 1e-6)):g}, relax scale {P.RELAX_TOL_SCALE:g}", flush=True)
if part in ("grid", "all"):
    for c in (1e-6, 3e-6, 1e-5, 3e-5, 1e-4):
        for s in (0.01, 0.1, 1.0):
            cell(f"centroid {c:g} relax x{s:g}", protocol(c, s))
if part == "model":
    cell("default protocol", P.build_runs(90, 2, fastrelax=True))
    cell("--no-fastrelax", P.build_runs(90, 2))
if part == "fine":
    for sc in (0.01, 0.02, 0.03, 0.05, 0.1):
        cell(f"centroid 1e-6 relax x{sc:g}", protocol(1e-6, sc))
    cell("centroid 1e-6 relax: ramp steps x1, tight steps x0.01", [dict(r, tol=(r["tol"] * 100 if (i >= 14 and r["tol"] > 1e-5) else r["tol"])) for i, r in enumerate(protocol(1e-6, 0.01))])
    cell("centroid 2e-6 relax x0.01", protocol(2e-6, 0.01))
    cell("no-fastrelax, 1e-6", [dict(r, tol=1e-6) for r in P.build_runs(90, 2)])
    cell("no-fastrelax, 1e-6, cold", [dict(r, tol=1e-6, warm=0) for r in P.build_runs(90, 2)])
if part == "short":
    cell("centroid 1e-6 relax x0.01 (shipped)", protocol(1e-6, 0.01))
    cell("centroid 1e-6 relax x0.1", protocol(1e-6, 0.1))
    cell("c 1e-6 x0.1, first+last declash 1e-4", protocol(1e-6, 0.1, first_declash=1e-4, last_declash=1e-4))
    cell("centroid 3e-6 relax x0.01", protocol(3e-6, 0.01))
    cell("centroid 3e-6 relax x0.1", protocol(3e-6, 0.1))
if part in ("mixed", "all"):
    cell("c 1e-6 x0.01, closing 1e-5", protocol(1e-6, 0.01, closing_tol=1e-5))
    cell("c 1e-6 x0.01, closing 1e-5, 30 it", protocol(1e-6, 0.01, closing_tol=1e-5, closing_iter=30))
    cell("c 1e-6 x0.01, closing 10 it", protocol(1e-6, 0.01, closing_iter=10))
    cell("c 1e-5 x0.1, closing 1e-5", protocol(1e-5, 0.1, closing_tol=1e-5))
    cell("c 1e-5 x0.1, first declash 1e-4", protocol(1e-5, 0.1, first_declash=1e-4))
    cell("c 1e-5 x0.1, first+last declash 1e-4", protocol(1e-5, 0.1, first_declash=1e-4, last_declash=1e-4))
    cell("c 1e-5 x1, first+last declash 1e-4", protocol(1e-5, 1.0, first_declash=1e-4, last_declash=1e-4))
    cell("c 3e-5 x0.1, first+last declash 1e-4", protocol(3e-5, 0.1, first_declash=1e-4, last_declash=1e-4))
    cell("c 1e-4 x0.01 (relax polishes)", protocol(1e-4, 0.01))
    cell("c 1e-6 x0.1, first+last declash 1e-4", protocol(1e-6, 0.1, first_declash=1e-4, last_declash=1e-4))
ctx.close()
