"""Synthetic distograms for chain lengths the reference ships no data for (its only example is L=90,
BASELINE.json's L=150 / L=400 configs need synthetic maps).  numpy only, deterministic in (L, seed).

Two kinds of target structure:
  "bundle" (default): a protein-like helical bundle -- rigid alpha-helices of 12-24 residues joined by loops of 3-6, compacted by
      pivot Monte-Carlo on the loop torsions only (simulated annealing on radius of gyration + C-alpha clashes).  Rigid rods
      joined by short loops cannot thread through each other, so the target is free of the knots and entanglements that made
      the round-1 random-coil targets unfoldable from random starts (measured: DESIGN.md section 2).
  "meander", "mixed": the same construction with extended strands (phi, psi = -120, 130) of 5-9 residues and turns of 2-4 in place
      of the helices, or with helices and strands in turn: extended chains whose residues sit in the beta basin of the torsion
      potential and whose restraints are mostly long-range (a helical bundle's are half local).  The strands are NOT paired into
      hydrogen-bonded sheets (the pivot Monte-Carlo does not find a registry; tried with a pairing reward, 0.5 units of the
      model's hydrogen-bond energy against 33 in a bundle of the same length).
  "coil": the round-1 recipe below, kept for the evaluation-parity tests (any structure will do there).

Recipe ("coil"): a backbone is built by NeRF from torsions drawn from the reference's start table
(folding/utils_ros/utils_ros.py:667-696) and compacted by a short Monte-Carlo over that table (radius of gyration +
CA clash count), so the target is exactly realisable by the fold's ideal geometry.  Its C-beta 6-D geometry
(formulae of utils_trX2dy/utils.py:97-182) is one-hot binned on the reference's bin edges (utils.py:191,203,215,227,
with the TRUE phi -- the reference's phi-from-theta label bug is a property of its trained network, not of geometry),
blurred along bins (Gaussian, sigma 1.5 bins), mixed 0.96 blur + 0.04 uniform, renormalised, cast to float32;
dist and omega are symmetrised.  (SURVEY.md 8d proposed 0.1 uniform.  That puts 0.086 of contact probability on every pair
that is NOT in contact in the target, above gen_rst's 0.05 threshold, so every far pair got a restraint whose only shape is
the background term -- a mild attraction towards short distances, on half of all pairs; distance-only folds then collapsed
into wrong globules, 0 of 16 within 7 A of the target or its mirror image.  With 0.04, far pairs get no restraint, as in
the reference's own maps, and distance-only folds reach the target or its mirror image: DESIGN.md section 2.)
"""
import os
import tempfile

import numpy as np

# ideal geometry: include/trx2_model.h
B_N_CA, B_CA_C, B_C_N = 1.458, 1.524, 1.334
A_N_CA_C, A_CA_C_N, A_C_N_CA = np.radians(111.4), np.radians(117.0), np.radians(121.0)
CB_K = (-0.58433326, 0.57201293, -0.53795593)
BASINS = np.radians(np.array([(-140, 153), (-72, 145), (-122, 117), (-82, -14), (-61, -41), (57, 39)], float))
BASIN_P = np.array([0.135, 0.155, 0.073, 0.122, 0.497, 0.018])


def _place(a, b, c, length, ang, tor):
    bc = (c - b) / np.linalg.norm(c - b)
    n = np.cross(b - a, bc)
    n /= np.linalg.norm(n)
    m = np.cross(n, bc)
    return c + bc * (-length * np.cos(ang)) + m * (length * np.sin(ang) * np.cos(tor)) + n * (length * np.sin(ang) * np.sin(tor))


def nerf_backbone(tors):
    """tors[L,3] (phi, psi, omega) -> N, CA, C, CB arrays [L,3]; same construction as the device NeRF."""
    L = len(tors)
    N = np.zeros((L, 3)); CA = np.zeros((L, 3)); C = np.zeros((L, 3))
    CA[0] = (B_N_CA, 0, 0)
    C[0] = (B_N_CA - B_CA_C * np.cos(A_N_CA_C), B_CA_C * np.sin(A_N_CA_C), 0)
    for i in range(L - 1):
        N[i + 1] = _place(N[i], CA[i], C[i], B_C_N, A_CA_C_N, tors[i, 1])
        CA[i + 1] = _place(CA[i], C[i], N[i + 1], B_N_CA, A_C_N_CA, tors[i, 2])
        C[i + 1] = _place(C[i], N[i + 1], CA[i + 1], B_CA_C, A_N_CA_C, tors[i + 1, 0])
    b, c = CA - N, C - CA
    CB = CA + CB_K[0] * np.cross(b, c) + CB_K[1] * b + CB_K[2] * c
    return N, CA, C, CB


def compact_torsions(L, seed, n_moves=None):
    """Monte-Carlo over basin assignments minimising Rg^2 + 10 * (number of CA pairs |i-j|>=3 closer than 4 A)."""
    rng = np.random.default_rng(seed)
    basin = rng.choice(len(BASIN_P), size=L, p=BASIN_P)
    n_moves = n_moves if n_moves is not None else 12 * L
    # With omega fixed at pi the transform from residue i's frame to residue i+1's depends only on
    # (psi of basin_i, phi of basin_{i+1}): 36 matrices, built once with the same atom placement as nerf_backbone.
    nb = len(BASIN_P)
    lCA, lC = np.zeros(3), np.array([B_CA_C, 0, 0])
    lN = B_N_CA * np.array([np.cos(A_N_CA_C), np.sin(A_N_CA_C), 0])
    M = np.zeros((nb, nb, 4, 4))
    for bi in range(nb):
        Nn = _place(lN, lCA, lC, B_C_N, A_CA_C_N, BASINS[bi, 1])
        CAn = _place(lCA, lC, Nn, B_N_CA, A_C_N_CA, np.pi)
        for bj in range(nb):
            Cn = _place(lC, Nn, CAn, B_CA_C, A_N_CA_C, BASINS[bj, 0])
            ex = (Cn - CAn) / np.linalg.norm(Cn - CAn)
            v = Nn - CAn
            ey = v - ex * (v @ ex)
            ey /= np.linalg.norm(ey)
            M[bi, bj, :3, 0], M[bi, bj, :3, 1], M[bi, bj, :3, 2], M[bi, bj, :3, 3] = ex, ey, np.cross(ex, ey), CAn
            M[bi, bj, 3, 3] = 1.0
    iu = np.triu_indices(L, 3)

    def score(bs):
        F = np.eye(4)
        ca = np.zeros((L, 3))
        for i in range(L - 1):
            F = F @ M[bs[i], bs[i + 1]]
            ca[i + 1] = F[:3, 3]
        d2 = ((ca[:, None] - ca[None]) ** 2).sum(-1)
        return ((ca - ca.mean(0)) ** 2).sum(1).mean() + 10.0 * (d2[iu] < 16.0).sum()

    cur = score(basin)
    temp = 2.0
    for _ in range(n_moves):
        i = rng.integers(0, L)
        old = basin[i]
        basin[i] = rng.choice(len(BASIN_P), p=BASIN_P)
        new = score(basin)
        if new <= cur or rng.random() < np.exp((cur - new) / temp):
            cur = new
        else:
            basin[i] = old
    return np.concatenate([BASINS[basin], np.full((L, 1), np.pi)], axis=1), cur


def _rot(points, origin, axis, ang):
    """rotate points about the line origin + t * axis (Rodrigues), vectorised"""
    k = axis / np.linalg.norm(axis)
    v = points - origin
    c, s_ = np.cos(ang), np.sin(ang)
    return origin + v * c + np.cross(k, v) * s_ + np.outer(v @ k, k) * (1 - c)


def bundle_torsions(L, seed, n_moves=None, kind="bundle"):
    """Secondary-structure layout from the seed, then pivot Monte-Carlo over the loop torsions.  kind: "bundle" rigid helices of
    12-24 residues joined by loops of 3-6; "meander" extended strands of 5-9 residues joined by turns of 2-4; "mixed" helices and
    strands in turn."""
    rng = np.random.default_rng(seed + 7919 + {"bundle": 0, "meander": 101, "mixed": 211}[kind])
    ss = np.zeros(L, bool)                      # True = rigid element
    strand = np.zeros(L, bool)                  # ... which is an extended strand
    i = int(rng.integers(1, 4))
    k = 0
    while i < L - 8:
        is_strand = kind == "meander" or (kind == "mixed" and k % 2 == 1)
        n = int(rng.integers(5, 10)) if is_strand else int(rng.integers(12, 25))
        ss[i:min(L - 2, i + n)] = True
        strand[i:min(L - 2, i + n)] = is_strand
        i += n + (int(rng.integers(2, 5)) if is_strand else int(rng.integers(3, 7)))
        k += 1
    loops = np.nonzero(~ss)[0]
    loops = loops[(loops > 0) & (loops < L - 1)]
    loop_basins = BASINS[[0, 1, 2, 3]]          # beta / PPII / bridge regions of the reference's start table
    tors = np.empty((L, 3))
    tors[:, 2] = np.pi
    tors[:, :2] = BASINS[4]                     # (-61, -41): alpha helix
    tors[strand, :2] = np.radians((-120.0, 130.0))   # antiparallel-sheet region
    for r in np.nonzero(~ss)[0]:
        tors[r, :2] = loop_basins[rng.integers(0, 4)] + rng.normal(size=2) * np.radians(10)
    N, CA, C, _ = nerf_backbone(tors)
    iu = np.triu_indices(L, 3)

    def score(ca):
        d2 = ((ca[:, None] - ca[None]) ** 2).sum(-1)[iu]
        clash = np.clip(4.6 - np.sqrt(d2), 0, None)
        return ((ca - ca.mean(0)) ** 2).sum(1).mean() + 30.0 * (clash ** 2).sum()

    cur = score(CA)
    n_moves = n_moves if n_moves is not None else 40 * len(loops) + 2000
    for it in range(n_moves):
        temp = 4.0 * (1 - it / n_moves) + 0.05
        r = int(loops[rng.integers(0, len(loops))])
        new = loop_basins[rng.integers(0, 4)] + rng.normal(size=2) * np.radians(12) if rng.random() < 0.5 else tors[r, :2] + rng.normal(size=2) * np.radians(8)
        dphi, dpsi = new[0] - tors[r, 0], new[1] - tors[r, 1]
        N2, CA2, C2 = N.copy(), CA.copy(), C.copy()
        # phi_r: axis N_r -> CA_r moves C_r and everything after; psi_r: axis CA_r -> C_r moves every later residue
        C2[r:] = _rot(C2[r:], N2[r], CA2[r] - N2[r], dphi)
        N2[r + 1:] = _rot(N2[r + 1:], N2[r], CA2[r] - N2[r], dphi)
        CA2[r + 1:] = _rot(CA2[r + 1:], N2[r], CA2[r] - N2[r], dphi)
        ax, org = C2[r] - CA2[r], CA2[r].copy()
        N2[r + 1:] = _rot(N2[r + 1:], org, ax, dpsi)
        CA2[r + 1:] = _rot(CA2[r + 1:], org, ax, dpsi)
        C2[r + 1:] = _rot(C2[r + 1:], org, ax, dpsi)
        sc = score(CA2)
        if sc <= cur or rng.random() < np.exp((cur - sc) / temp):
            cur, N, CA, C = sc, N2, CA2, C2
            tors[r, :2] = new
    return tors, cur


def _dihedral(a, b, c, d):
    b0, b1, b2 = a - b, c - b, d - c
    b1 = b1 / np.linalg.norm(b1, axis=-1, keepdims=True)
    v = b0 - (b0 * b1).sum(-1, keepdims=True) * b1
    w = b2 - (b2 * b1).sum(-1, keepdims=True) * b1
    return np.arctan2((np.cross(b1, v) * w).sum(-1), (v * w).sum(-1))


def _angle(a, b, c):
    v, w = a - b, c - b
    v = v / np.linalg.norm(v, axis=-1, keepdims=True)
    w = w / np.linalg.norm(w, axis=-1, keepdims=True)
    return np.arccos(np.clip((v * w).sum(-1), -1, 1))


def _blur_mix(onehot, sigma=1.5, mix=None):
    mix = float(os.environ.get("TRX2_SYNTH_MIX", "0.04")) if mix is None else mix
    K = onehot.shape[-1]
    k = np.arange(K)
    G = np.exp(-0.5 * ((k[:, None] - k[None]) / sigma) ** 2)
    G /= G.sum(1, keepdims=True)
    p = (1 - mix) * (onehot @ G.T) + mix / K
    return (p / p.sum(-1, keepdims=True)).astype(np.float32)


def make_map(L, seed=None, n_moves=None, kind=None):
    """-> dict(dist[L,L,37], omega[L,L,25], theta[L,L,25], phi[L,L,13], tors[L,3], seq).  kind: "bundle" (default), "meander",
    "mixed" (bundle_torsions) or "coil" (default when n_moves is given: the short-Monte-Carlo coils of the evaluation-parity tests)."""
    seed = L if seed is None else seed
    kind = kind or ("coil" if n_moves is not None else "bundle")
    cache = os.path.join(os.environ.get("TRX2_SYNTH_CACHE", os.path.join(tempfile.gettempdir(), "trx2_synth")),
                         f"map_{kind}_L{L}_s{seed}_m{n_moves}.npz")
    if os.path.exists(cache):
        with np.load(cache) as z:
            return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}
    out = _make_map(L, seed, n_moves, kind)
    try:
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        tmp = f"{cache}.{os.getpid()}.tmp.npz"
        np.savez(tmp, **out)
        os.replace(tmp, cache)
    except OSError:
        pass  # the cache is an optimisation only
    return out


def _make_map(L, seed, n_moves, kind="bundle"):
    tors, _ = bundle_torsions(L, seed, n_moves, kind) if kind in ("bundle", "meander", "mixed") else compact_torsions(L, seed, n_moves)
    N, CA, C, CB = nerf_backbone(tors)
    i, j = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.linalg.norm(CB[i] - CB[j], axis=-1)
        om = _dihedral(CA[i], CB[i], CB[j], CA[j])
        th = _dihedral(N[i], CA[i], CB[i], CB[j])
        ph = _angle(CA[i], CB[i], CB[j])
    contact = (d > 2.0) & (d <= 20.0) & (i != j)
    # bin k in 1..36 <=> d in (2+0.5(k-1), 2+0.5k]; bin 0 = no contact (utils.py:191-196)
    jd = np.where(contact, np.clip(np.ceil((d - 2.0) / 0.5), 1, 36), 0).astype(int)
    ja = lambda x: np.where(contact, np.clip(np.ceil((np.nan_to_num(x) + np.pi) / (np.pi / 12)), 1, 24), 0).astype(int)
    jp = np.where(contact, np.clip(np.ceil(np.nan_to_num(ph) / (np.pi / 12)), 1, 12), 0).astype(int)
    out = dict(dist=_blur_mix(np.eye(37)[jd]), omega=_blur_mix(np.eye(25)[ja(om)]), theta=_blur_mix(np.eye(25)[ja(th)]),
               phi=_blur_mix(np.eye(13)[jp]))
    for k in ("dist", "omega"):
        s = 0.5 * (out[k] + out[k].transpose(1, 0, 2))
        out[k] = (s / s.sum(-1, keepdims=True)).astype(np.float32)
    out["tors"] = tors
    out["seq"] = "A" * L
    out["contact_fraction"] = float(contact.sum() / (L * (L - 1)))
    return out
