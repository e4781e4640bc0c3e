"""Synthetic distograms for chain lengths the reference ships no data for (its only example is L=90,
BASELINE.json's L=150 / L=400 configs need synthetic maps).  numpy only, deterministic in (L, seed).

Recipe: a backbone is built by NeRF from torsions drawn from the reference's start table
(folding/utils_ros/utils_ros.py:667-696) and compacted by a short Monte-Carlo over that table (radius of gyration +
CA clash count), so the target is exactly realisable by the fold's ideal geometry.  Its C-beta 6-D geometry
(formulae of utils_trX2dy/utils.py:97-182) is one-hot binned on the reference's bin edges (utils.py:191,203,215,227,
with the TRUE phi -- the reference's phi-from-theta label bug is a property of its trained network, not of geometry),
blurred along bins (Gaussian, sigma 1.5 bins), mixed 0.9 blur + 0.1 uniform, renormalised, cast to float32;
dist and omega are symmetrised.
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


def _blur_mix(onehot, sigma=1.5, mix=0.1):
    K = onehot.shape[-1]
    k = np.arange(K)
    G = np.exp(-0.5 * ((k[:, None] - k[None]) / sigma) ** 2)
    G /= G.sum(1, keepdims=True)
    p = (1 - mix) * (onehot @ G.T) + mix / K
    return (p / p.sum(-1, keepdims=True)).astype(np.float32)


def make_map(L, seed=None, n_moves=None):
    """-> dict(dist[L,L,37], omega[L,L,25], theta[L,L,25], phi[L,L,13], tors[L,3], seq)"""
    seed = L if seed is None else seed
    cache = os.path.join(os.environ.get("TRX2_SYNTH_CACHE", os.path.join(tempfile.gettempdir(), "trx2_synth")),
                         f"map_L{L}_s{seed}_m{n_moves}.npz")
    if os.path.exists(cache):
        with np.load(cache) as z:
            return {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}
    out = _make_map(L, seed, n_moves)
    try:
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        tmp = f"{cache}.{os.getpid()}.tmp.npz"
        np.savez(tmp, **out)
        os.replace(tmp, cache)
    except OSError:
        pass  # the cache is an optimisation only
    return out


def _make_map(L, seed, n_moves):
    tors, _ = compact_torsions(L, seed, n_moves)
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
