"""Feedback step between folds: decoy -> realised 6-D geometry -> re-weighted distograms.

Host-side (numpy/scipy) mirror of /root/reference/utils_trX2dy/utils.py:97-475, the caller on both sides of the fold
(SURVEY.md 8f1).  Same function names and argument meaning as the reference so that run_inference reads alike.
Reference quirks reproduced ON PURPOSE:
  R1  the phi one-hot is binned from THETA values on phi's edges            (utils.py:226)
  R2  angle one-hots are forced to bin 0 wherever the distance bin is 0     (utils.py:207-208,219-220,231-232)
  R3  the decay is a no-op when the realised bin is the last one            (utils.py:392-396)
  R4  the pair mask max_k p < 0.5 is taken from the channel being processed (utils.py:386)
  R10 the reliability score compares RADIANS with -180/0/180 => "fraction of residues with phi <= 0" (utils.py:352-362)
  R11 the convergence array `tmp` is un-normalised and un-smoothed; first iteration falls back to `dist` (utils.py:425-435)
Pinned bit-for-bit by tests/golden/feedback_{NMR,Xray}.npz (SHA-256 of the reference's full outputs).
"""
import numpy as np
from scipy.ndimage import gaussian_filter1d

from .pdbio import read_backbone

# params("0HD"), utils.py:331 -- the only flag the pipeline uses (utils.py:385)
P_MASK, PCUT, DECAY = 0.5, 0.05, 0.50
VCB = (-0.58273431, 0.56802827, -0.54067466)  # virtual C-beta, utils.py:135
DMAX = 20.0


def get_dihedrals(a, b, c, d):
    """IUPAC dihedral, arithmetic in the dtype of the inputs (utils.py:97-110)"""
    b0 = -1.0 * (b - a)
    b1 = c - b
    b2 = d - c
    b1 = b1 / np.linalg.norm(b1, axis=-1)[:, None]
    v = b0 - np.sum(b0 * b1, axis=-1)[:, None] * b1
    w = b2 - np.sum(b2 * b1, axis=-1)[:, None] * b1
    return np.arctan2(np.sum(np.cross(b1, v) * w, axis=-1), np.sum(v * w, axis=-1))


def get_angles(a, b, c):
    """planar angle at b (utils.py:113-122)"""
    v = a - b
    v = v / np.linalg.norm(v, axis=-1)[:, None]
    w = c - b
    w = w / np.linalg.norm(w, axis=-1)[:, None]
    return np.arccos(np.sum(v * w, axis=1))


def get_neighbors(xyz, seq, dmax=DMAX):
    """xyz[L,5,3] (N CA C O CB) -> dense dist6d, omega6d, theta6d, phi6d [L,L], 0 outside dmax (utils.py:125-182).
    C-beta: the real atom for non-Gly residues, the virtual one (utils.py:132-135) for Gly or when the atom is absent
    (the reference drops such rows from its KD-tree and mis-indexes; this package's decoys always carry CB).
    Arithmetic in float64: the reference's reader fills float64 arrays (np.nan * np.zeros, utils.py:270) with Biopython's
    float32 coordinates, so its geometry runs in float64 on float32-valued inputs -- read_backbone's float32 is widened here."""
    xyz = np.asarray(xyz).astype(np.float64)
    N, Ca, C = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    L = len(Ca)
    if L != len(seq):
        raise ValueError("sequence length does not match the structure")
    b = Ca - N
    c = C - Ca
    Cb = VCB[0] * np.cross(b, c) + VCB[1] * b + VCB[2] * c + Ca
    real = xyz[:, 4]
    use = np.array([s != "G" for s in seq]) & np.all(np.isfinite(real), axis=1)
    Cb = np.where(use[:, None], real, Cb).astype(xyz.dtype)
    d2 = ((Cb[:, None].astype(np.float64) - Cb[None].astype(np.float64)) ** 2).sum(-1)
    idx0, idx1 = np.nonzero((d2 <= dmax * dmax) & ~np.eye(L, dtype=bool))  # cKDTree.query_ball_tree(r) is inclusive
    out = [np.zeros((L, L)) for _ in range(4)]
    out[0][idx0, idx1] = np.linalg.norm(Cb[idx1] - Cb[idx0], axis=-1)
    out[1][idx0, idx1] = get_dihedrals(Ca[idx0], Cb[idx0], Cb[idx1], Ca[idx1])
    out[2][idx0, idx1] = get_dihedrals(N[idx0], Ca[idx0], Cb[idx0], Cb[idx1])
    out[3][idx0, idx1] = get_angles(Ca[idx0], Cb[idx0], Cb[idx1])
    return tuple(out)


def bin_geometry(dist6d, omega6d, theta6d, phi6d=None):
    """integer bins per pair (the argmax of the reference's one-hots, utils.py:185-235) -> jd, jo, jt, jp"""
    count = lambda x, edges: (edges[None, None, :] < x[:, :, None]).sum(-1)
    jd = count(dist6d, np.arange(2, 20.5, 0.5))
    jd = np.where(jd >= 37, 0, jd)
    nocontact = jd == 0
    a_edges = np.arange(-np.pi, np.pi, np.pi / 12)
    jo = np.where(nocontact, 0, count(omega6d, a_edges))            # R2
    jt = np.where(nocontact, 0, count(theta6d, a_edges))
    jp = np.where(nocontact, 0, count(theta6d, np.arange(0, np.pi, np.pi / 12)))  # R1: theta on phi's edges
    return jd, jo, jt, jp


def get_distribution_from_pdb(pdb_path):
    """-> realised bins (jd, jt, jo, jp) in the reference's return order dist, theta, omega, phi (utils.py:294-316)"""
    xyz, seq = read_backbone(pdb_path)
    jd, jo, jt, jp = bin_geometry(*get_neighbors(xyz, seq))
    return jd, jt, jo, jp


def process_distribution_with_pred_distribution(unprocessed, fact_bins, norm=True, smooth=True, sigma=1.0):
    """utils.py:379-403 with the realised distribution given as integer bins [L,L] instead of one-hots."""
    K = unprocessed.shape[-1]
    tmp = np.copy(unprocessed)
    processed = np.copy(unprocessed)
    ii, jj = np.nonzero(unprocessed.max(axis=-1) < P_MASK)                    # R4
    idx = fact_bins[ii, jj]
    hit = idx < K - 1                                                         # R3: last bin -> empty slice
    v = tmp[ii[hit], jj[hit], idx[hit]]
    tmp[ii[hit], jj[hit], idx[hit]] = np.where(v < PCUT, v, v * DECAY)
    rows = tmp[ii, jj]
    rows = rows / np.sum(rows, axis=-1)[:, None]
    if smooth:
        rows = gaussian_filter1d(rows, sigma, axis=-1, mode="reflect")       # gaussian_filter on each 1-D row
    processed[ii, jj] = rows
    return processed if norm else tmp


def get_npz_from_pred_pdb(unprocessed_npz_dir, pred_pdb_dir, tmp=False, simga=1.0, angle=True):
    """utils.py:406-475 (the keyword really is spelled `simga` there; kept so call sites read alike)"""
    npz = np.load(unprocessed_npz_dir)
    jd, jt, jo, jp = get_distribution_from_pdb(pred_pdb_dir)
    if tmp:
        base = npz["tmp"] if "tmp" in npz.files else npz["dist"]              # R11
        return process_distribution_with_pred_distribution(base, jd, norm=False)
    out = process_distribution_with_pred_distribution(npz["dist"], jd, True, True, simga)
    if not angle:
        return out
    return (out,
            process_distribution_with_pred_distribution(npz["omega"], jo, True, True, simga),
            process_distribution_with_pred_distribution(npz["theta"], jt, True, True, simga),
            process_distribution_with_pred_distribution(npz["phi"], jp, True, True, simga))


def feedback_labels(arrays, pred_pdb_dir, sigma=1.0, angle=True):
    """What run_inference.py:75-88,116-131 computes per iteration with two get_npz_from_pred_pdb calls (processed channels,
    then tmp=True), from arrays already in memory and ONE parse of the decoy.  -> dict dist[/theta/omega/phi]/tmp"""
    jd, jt, jo, jp = get_distribution_from_pdb(pred_pdb_dir)
    labels = {"dist": process_distribution_with_pred_distribution(arrays["dist"], jd, True, True, sigma)}
    if angle:
        labels["theta"] = process_distribution_with_pred_distribution(arrays["theta"], jt, True, True, sigma)
        labels["omega"] = process_distribution_with_pred_distribution(arrays["omega"], jo, True, True, sigma)
        labels["phi"] = process_distribution_with_pred_distribution(arrays["phi"], jp, True, True, sigma)
    base = arrays["tmp"] if "tmp" in arrays else arrays["dist"]                  # R11
    labels["tmp"] = process_distribution_with_pred_distribution(base, jd, norm=False)
    return labels


def backbone_phi_psi(xyz):
    """(phi, psi) pairs as Biopython's PPBuilder reports them (utils.py:337-349): chains are split where the C-N
    peptide distance exceeds 1.8 A; residues lacking either angle (segment ends) are dropped.  All residues at once: the
    per-element arithmetic of get_dihedrals does not depend on how many rows it is given."""
    N, CA, C = (xyz[:, k].astype(np.float64) for k in range(3))
    L = len(CA)
    if L < 3:
        return []
    link = np.linalg.norm(C[:-1] - N[1:], axis=-1) < 1.8
    i = np.nonzero(link[:-1] & link[1:])[0] + 1          # residues 1..L-2 bonded on both sides
    if len(i) == 0:
        return []
    phi = get_dihedrals(C[i - 1], N[i], CA[i], C[i])
    psi = get_dihedrals(N[i], CA[i], C[i], N[i + 1])
    return list(zip(phi.tolist(), psi.tolist()))


def calculate_reliability_score(pdb_file):
    """utils.py:352-372.  R10: the angles are radians, so the test -180 <= phi <= 0 is "phi <= 0"."""
    xyz, _ = read_backbone(pdb_file)
    pp = backbone_phi_psi(xyz)
    if not pp:
        return 0
    return sum(1 for phi, psi in pp if (-180 <= phi <= 0) and (-180 <= psi <= 180)) / len(pp)
