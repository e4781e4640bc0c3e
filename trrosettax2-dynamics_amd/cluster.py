"""Clustering of generated structures: GloCon / RMSD matrices + KMeans, the reference's cluster.py.

Mirrors /root/reference/utils_trX2dy/utils.py:514-616 (get_glocon_matrix, kmeans_clustering, save_cluster_result) and
/root/reference/cluster.py.  SURVEY.md 8f4.  Differences, deliberate:
  D  files are taken in sorted order (the reference uses os.listdir order, so its matrix rows depend on the file system)
  D  modes "rmsd" and "tmscore": the reference runs its prebuilt ./bin/TMscore ELF once per pair of files (utils.py:514-541; no
     source in the tree) and parses "RMSD of the common residues" and "TM-score".  Here both matrices come from this package's
     superposition (evaluate.rmsd_common / evaluate.tm_score: the TM-score program's seeded search), computed for all pairs at
     once on the GPU when device is given (trx2_superpose_matrix; tested equal to the host functions to 1e-9)
  D  selected files are copied with shutil instead of `os.system("cp ...")`
The GloCon matrix is O(n^2 L^2) numpy passes on the host; device=<gpu index> computes it with trx2_glocon_matrix, summing in
numpy's pairwise order (bitwise the same matrix in the tests).
"""
import argparse
import os
import shutil

import numpy as np

from .feedback import get_neighbors
from .pdbio import read_backbone


def _pdb_files(pdb_dir):
    return sorted(f for f in os.listdir(pdb_dir) if f.endswith(".pdb"))


def glocon_score(dist1, dist2):
    """utils.py:560-563 for one pair of dist6d matrices"""
    dist_diff = np.abs(dist1 - dist2)
    dist_diff[dist_diff <= 3] = 0
    return np.sum(np.triu(dist_diff)) / (len(dist_diff) * (len(dist_diff) - 1) / 2)


def get_glocon_matrix(pdb_dir, device=None):
    """-> (matrix[n,n] float64, pdb_files); utils.py:543-569"""
    pdb_files = _pdb_files(pdb_dir)
    parsed = [read_backbone(os.path.join(pdb_dir, f)) for f in pdb_files]
    n = len(pdb_files)
    if device is not None:
        from ._lib import Context
        lens = {len(s) for _, s in parsed}
        if len(lens) != 1:
            raise ValueError("the device GloCon matrix needs structures of one length")
        ctx = Context(int(device))
        try:
            return ctx.glocon_matrix(np.stack([x for x, _ in parsed]), [s for _, s in parsed]), pdb_files
        finally:
            ctx.close()
    dist = [get_neighbors(x, s)[0] for x, s in parsed]
    m = np.zeros((n, n))
    for i in range(n):
        for j in range(i):
            m[i][j] = glocon_score(dist[i], dist[j])
    return m + m.T, pdb_files


def _ca_stack(pdb_dir):
    pdb_files = _pdb_files(pdb_dir)
    ca = [read_backbone(os.path.join(pdb_dir, f))[0][:, 1] for f in pdb_files]
    if len({len(c) for c in ca}) != 1 or not all(np.isfinite(c).all() for c in ca):
        raise ValueError("the superposition matrices need structures of one length with every C-alpha present")
    return np.stack(ca), pdb_files


def get_tmscore_and_rmsd_matrix(pdb_dir, device=None):
    """-> (tmscore[n,n], rmsd[n,n], pdb_files): utils.py:524-541 without the TMscore subprocess per pair.  C-alpha RMSD after
    optimal superposition and TM-score normalised by the chain length, for every pair; zero diagonal as in the reference."""
    ca, pdb_files = _ca_stack(pdb_dir)
    n = len(ca)
    if device is not None:
        from ._lib import Context
        ctx = Context(int(device))
        try:
            rm, tm = ctx.superpose_matrix(ca)
        finally:
            ctx.close()
    else:
        from .evaluate import rmsd_common, tm_score
        rm, tm = np.zeros((n, n)), np.zeros((n, n))
        for i in range(n):
            for j in range(i):
                x, y = ca[i].astype(np.float64), ca[j].astype(np.float64)
                rm[i, j] = rm[j, i] = rmsd_common(x, y)
                tm[i, j] = tm[j, i] = tm_score(x, y)
    rm[np.diag_indices(n)] = 0.0
    tm[np.diag_indices(n)] = 0.0           # the reference never runs a file against itself: its diagonals stay 0
    return tm, rm, pdb_files


def get_rmsd_matrix(pdb_dir, device=None):
    """C-alpha RMSD after optimal superposition for every pair (TMscore's "RMSD of the common residues")"""
    _, rm, pdb_files = get_tmscore_and_rmsd_matrix(pdb_dir, device=device) if device is not None else (None, *_host_rmsd(pdb_dir))
    return rm, pdb_files


def _host_rmsd(pdb_dir):
    ca, pdb_files = _ca_stack(pdb_dir)
    ca = ca.astype(np.float64)
    n = len(ca)
    m = np.zeros((n, n))
    for i in range(n):
        for j in range(i):
            a, b = ca[i] - ca[i].mean(0), ca[j] - ca[j].mean(0)
            u, s_, vt = np.linalg.svd(a.T @ b)
            s_[-1] *= np.sign(np.linalg.det(u @ vt))
            m[i][j] = np.sqrt(max(0.0, (np.sum(a * a) + np.sum(b * b) - 2.0 * s_.sum()) / len(a)))
    return m + m.T, pdb_files


def kmeans_clustering(matrix, pdb_files, n_clusters=10):
    """utils.py:572-592 (without the plot)"""
    from sklearn.cluster import KMeans
    labels = KMeans(n_clusters=n_clusters, n_init=10, random_state=0).fit(matrix).labels_
    clusters = {}
    for i, label in enumerate(labels):
        clusters.setdefault(int(label), []).append(pdb_files[i])
    return clusters


def save_cluster_result(pdb_dir, n_clusters=10, n_files=5, output_dir=None, mode="glocon", device=None):
    """utils.py:595-616: cluster on the rows of the chosen matrix, keep the first n_files of every cluster"""
    if mode == "glocon":
        matrix, pdb_files = get_glocon_matrix(pdb_dir, device=device)
    elif mode == "rmsd":
        matrix, pdb_files = get_rmsd_matrix(pdb_dir, device=device)
    elif mode == "tmscore":
        matrix, _, pdb_files = get_tmscore_and_rmsd_matrix(pdb_dir, device=device)
    else:
        raise ValueError(f"unknown mode {mode!r}")
    if output_dir is None:
        output_dir = os.path.join(pdb_dir, "clusters_result")
    os.makedirs(output_dir, exist_ok=True)
    try:
        clusters = kmeans_clustering(matrix, pdb_files, n_clusters=n_clusters)
    except ValueError:
        return "no_cluster"
    for files in clusters.values():
        for f in files[:n_files]:
            shutil.copyfile(os.path.join(pdb_dir, f), os.path.join(output_dir, f))
    return clusters


def main(argv=None):
    """command line of the reference's cluster.py: -d/--pdb_dir, -m/--mode, -o/--output_dir, --n_clusters, --n_files"""
    ap = argparse.ArgumentParser(prog="cluster.py", description="Group generated models by KMeans on a pairwise matrix.")
    ap.add_argument("-d", "--pdb_dir", required=True, help="folder holding the .pdb models")
    ap.add_argument("-m", "--mode", default="glocon", choices=("glocon", "tmscore", "rmsd"), help="matrix to cluster on")
    ap.add_argument("-o", "--output_dir", default=None, help="where the kept models go (<pdb_dir>/clusters_result if omitted)")
    ap.add_argument("--n_clusters", type=int, default=10, help="KMeans k")
    ap.add_argument("--n_files", type=int, default=5, help="models kept from every cluster")
    ap.add_argument("--device", type=int, default=None, help="compute the matrix on this GPU (extension; numpy if omitted)")
    a = ap.parse_args(argv)
    target = a.output_dir if a.output_dir else os.path.join(a.pdb_dir, "clusters_result")
    res = save_cluster_result(a.pdb_dir, n_clusters=a.n_clusters, n_files=a.n_files, output_dir=target, mode=a.mode, device=a.device)
    print("Clustering failed or not possible." if res == "no_cluster" else f"Clustering completed. Results saved in {target}.")
    return 0
