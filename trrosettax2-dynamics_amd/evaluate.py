"""Evaluation of predicted against native structures: best RMSD and best TM-score per native, summary.txt.

Mirrors /root/reference/evaluate.py and utils_trX2dy/evaluate_utils.py:33-100 (run_score).  The reference shells out to a
prebuilt `bin/TMscore` (no source in the tree) for every (native, predicted) pair and parses its text.  Here the two numbers
it parses are computed directly: "RMSD of the common residues" = C-alpha RMSD after optimal superposition of the residues
present in both files (matched by residue number), and the TM-score by the published search of the TM-score program
(Zhang & Skolnick 2004): seeds = every fragment of length L, L/2, L/4, .. 4 of the aligned residues; from each seed the
superposition is refined on the pairs closer than a cutoff until the set stops changing; the best
sum 1 / (1 + (d_i / d0)^2) / L_norm wins, d0 = 1.24 (L_norm - 15)^(1/3) - 1.8, L_norm = length of the second structure.
Pinned by the reference's committed example summary (tests): apo 3.018 A / 0.6661, holo 3.931 A / 0.6269 -- the four printed
decimals of two proteins; the TM-score program's exact heuristics (its extra seeds from secondary structure, its d0 schedule)
are not in the tree, so other inputs may differ from the binary in the 3rd-4th decimal.  `device=` computes all pairs of a
native at once on the GPU (trx2_superpose_matrix: the same search, one wave per pair and seed; equal to tm_score() to 1e-9),
instead of ~50 k numpy SVDs per pair.
`--align` (evaluate_utils.py:56-58: TMscore's -seq) establishes the residue correspondence by sequence alignment instead of
residue numbers: nw_align() restates the program's alignment -- Needleman-Wunsch with BLOSUM62, affine gaps (open -11,
extend -1), gaps beside the termini free, a gap preferred to a substitution on ties -- and is pinned residue pair for residue
pair to the binary's printed alignments (tests/golden/align_tmscore.json, captured by tests/golden/make_golden_align.py);
the superposition then runs on the aligned pairs (host or device).
"""
import argparse
import os
import shutil

import numpy as np


def read_ca(path):
    """-> {residue number: CA xyz} of the first model / first alternate location"""
    out = {}
    with open(path) as f:
        for line in f:
            if line.startswith("ENDMDL"):
                break
            if line.startswith("ATOM") and line[12:16].strip() == "CA" and line[16] in (" ", "A"):
                out.setdefault(int(line[22:26]), (float(line[30:38]), float(line[38:46]), float(line[46:54])))
    return out


_AA = "ARNDCQEGHILKMFPSTWYV"
_AA3 = dict(ALA="A", ARG="R", ASN="N", ASP="D", CYS="C", GLN="Q", GLU="E", GLY="G", HIS="H", ILE="I", LEU="L", LYS="K", MET="M", PHE="F",
            PRO="P", SER="S", THR="T", TRP="W", TYR="Y", VAL="V")
# BLOSUM62 (Henikoff & Henikoff 1992), rows and columns in the order of _AA
_B62 = np.array([
    [4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0], [-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3],
    [-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3], [-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3],
    [0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1], [-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2],
    [-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2], [0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3],
    [-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3], [-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3],
    [-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1], [-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2],
    [-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1], [-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1],
    [-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2], [1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2],
    [0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0], [-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3],
    [-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1], [0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4]])
_AA_IDX = {a: i for i, a in enumerate(_AA)}
_ALIGN_CACHE = {}


def nw_align(seq_a, seq_b, gap_open=-11, gap_ext=-1):
    """Residue correspondence of `TMscore a b -seq`: global alignment of the two sequences with BLOSUM62 and affine gaps, gaps
    beside either terminus free (the first row / column of the score matrix is zero, a gap in the last row / column costs
    nothing), a gap preferred to a substitution where both are optimal (horizontal before vertical).  -> list of (i, j) index
    pairs.  Letters outside the twenty standard ones score as the worst substitution (-4) against everything."""
    key = (seq_a, seq_b, gap_open, gap_ext)
    if key in _ALIGN_CACHE:
        return _ALIGN_CACHE[key]
    n, m = len(seq_a), len(seq_b)
    ia = [_AA_IDX.get(c, -1) for c in seq_a]
    ib = [_AA_IDX.get(c, -1) for c in seq_b]
    neg = -(10 ** 6)
    S = [[0] * (m + 1) for _ in range(n + 1)]
    H = [[neg] * (m + 1) for _ in range(n + 1)]
    V = [[neg] * (m + 1) for _ in range(n + 1)]
    JH = [[0] * (m + 1) for _ in range(n + 1)]
    JV = [[0] * (m + 1) for _ in range(n + 1)]
    P = [[0] * (m + 1) for _ in range(n + 1)]     # bit 1: diagonal optimal, 2: horizontal gap, 4: vertical gap
    for i in range(1, n + 1):
        V[i][0], JV[i][0], P[i][0] = 0, i, 4
    for j in range(1, m + 1):
        H[0][j], JH[0][j], P[0][j] = 0, j, 2
    for i in range(1, n + 1):
        Si, Sp, Hi, Vi, Vp, JHi, JVi, JVp, Pi = S[i], S[i - 1], H[i], V[i], V[i - 1], JH[i], JV[i], JV[i - 1], P[i]
        row = _B62[ia[i - 1]] if ia[i - 1] >= 0 else None
        oh, eh = (0, 0) if i == n else (gap_open, gap_ext)
        for j in range(1, m + 1):
            a, b = Si[j - 1] + oh, Hi[j - 1] + eh
            if b >= a:
                h, JHi[j] = b, JHi[j - 1] + 1
            else:
                h, JHi[j] = a, 1
            ov, ev = (0, 0) if j == m else (gap_open, gap_ext)
            a, b = Sp[j] + ov, Vp[j] + ev
            if b >= a:
                v, JVi[j] = b, JVp[j] + 1
            else:
                v, JVi[j] = a, 1
            Hi[j], Vi[j] = h, v
            d = Sp[j - 1] + (int(row[ib[j - 1]]) if row is not None and ib[j - 1] >= 0 else -4)
            best = d if d >= h and d >= v else (h if h >= v else v)
            Si[j] = best
            Pi[j] = (1 if d == best else 0) | (2 if h == best else 0) | (4 if v == best else 0)
    i, j, pairs = n, m, []
    while i + j:
        p = P[i][j]
        if p & 2:
            j -= JH[i][j]
        elif p & 4:
            i -= JV[i][j]
        else:
            pairs.append((i - 1, j - 1))
            i -= 1
            j -= 1
    pairs.reverse()
    _ALIGN_CACHE[key] = pairs
    return pairs


def read_ca_seq(path):
    """-> (residue numbers, CA xyz [n,3], one-letter sequence) of the first model, in file order"""
    nums, xyz, seq = [], [], []
    with open(path) as f:
        for line in f:
            if line.startswith("ENDMDL"):
                break
            if line.startswith("ATOM") and line[12:16].strip() == "CA" and line[16] in (" ", "A"):
                k = int(line[22:26])
                if k in nums:
                    continue
                nums.append(k)
                xyz.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
                seq.append(_AA3.get(line[17:20], "X"))
    return nums, np.array(xyz), "".join(seq)


def _matched(native_pdb, pred_pdb, align):
    """-> (x, y, length of the second structure): CA coordinates of the residue pairs the TM-score program would compare"""
    if align:
        _, xa, sa = read_ca_seq(native_pdb)
        _, xb, sb = read_ca_seq(pred_pdb)
        pairs = nw_align(sa, sb)
        if len(pairs) < 3:
            raise ValueError(f"{native_pdb} and {pred_pdb} align on fewer than three residues")
        return xa[[i for i, _ in pairs]], xb[[j for _, j in pairs]], len(sb)
    a, b = read_ca(native_pdb), read_ca(pred_pdb)
    common = sorted(set(a) & set(b))
    if len(common) < 3:
        raise ValueError(f"{native_pdb} and {pred_pdb} share fewer than three residues")
    return np.array([a[k] for k in common]), np.array([b[k] for k in common]), len(b)


def _kabsch(x, y):
    """rotation R and translation t minimising |R x + t - y| (rows are points)"""
    cx, cy = x.mean(0), y.mean(0)
    u, s, vt = np.linalg.svd((x - cx).T @ (y - cy))
    d = np.sign(np.linalg.det(u @ vt))
    r = (u * np.array([1.0, 1.0, d])) @ vt
    return r, cy - cx @ r


def rmsd_common(x, y):
    r, t = _kabsch(x, y)
    return float(np.sqrt(((x @ r + t - y) ** 2).sum(1).mean()))


def tm_score(x, y, l_norm=None):
    """TM-score of the aligned CA sets x, y [n,3], normalised by l_norm (default: n): the TM-score program's search"""
    n = len(x)
    l_norm = l_norm or n
    d0 = 1.24 * (l_norm - 15) ** (1.0 / 3.0) - 1.8 if l_norm > 21 else 0.5
    d0 = max(d0, 0.5)
    d0_search = min(max(d0, 4.5), 8.0)

    def score(r, t):
        d2 = ((x @ r + t - y) ** 2).sum(1)
        return (1.0 / (1.0 + d2 / (d0 * d0))).sum() / l_norm, d2

    best = 0.0
    frag, lens = n, []
    while frag >= 4 and len(lens) < 6:
        lens.append(frag)
        frag //= 2
    if lens and lens[-1] > 4:
        lens.append(4)
    for lf in lens:
        for start in range(0, n - lf + 1):
            sel = np.zeros(n, bool)
            sel[start:start + lf] = True
            for it in range(21):  # the seed superposition, then up to 20 refinements
                r, t = _kabsch(x[sel], y[sel])
                s, d2 = score(r, t)
                best = max(best, s)
                d = d0_search - 1.0 if it == 0 else d0_search + 1.0   # the program's first cut is tighter
                new = d2 < d * d
                while new.sum() < 3 and n > 3:
                    d += 0.5
                    new = d2 < d * d
                if it > 0 and np.array_equal(new, sel):
                    break
                sel = new
    return float(best)


def compare(native_pdb, pred_pdb, align=False):
    """-> (rmsd, tm_score) as the TM-score program reports them for `TMscore native pred` (align: `-seq`)"""
    x, y, lb = _matched(native_pdb, pred_pdb, align)
    return rmsd_common(x, y), tm_score(x, y, l_norm=lb)


def compare_many(native_pdb, pred_pdbs, device, align=False):
    """compare(native, p) for every p on the GPU: the models are grouped by (number of matched residues, own length, the
    native's matched coordinates), each group is one trx2_superpose_matrix call"""
    from ._lib import Context
    groups, out = {}, [None] * len(pred_pdbs)
    for k, p in enumerate(pred_pdbs):
        x, y, lb = _matched(native_pdb, p, align)
        groups.setdefault((lb, x.astype(np.float32).tobytes()), []).append((k, x.astype(np.float32), y.astype(np.float32)))
    ctx = Context(int(device))
    try:
        for (lb, _), items in groups.items():
            rm, tm = ctx.superpose_matrix(items[0][1][None], np.stack([y for _, _, y in items]), l_norm=lb)
            for (k, _, _), r, t in zip(items, rm[0], tm[0]):
                out[k] = (float(r), float(t))
    finally:
        ctx.close()
    return out


def run_score(native_pdb_dir, pred_pdb_dir, align=False, save_summary=False, save_dir=None, device=None):
    """evaluate_utils.py:33-100: -> (min_rmsd, max_tmscore, mean_rmsd, mean_tmscore); summary.txt in the reference's format
    (values rounded to the three / four decimals the TM-score program prints)"""
    lines, rmsds, tms = [], [], []
    for native in sorted(f for f in os.listdir(native_pdb_dir) if f.endswith(".pdb")):
        best_r, best_t = None, None
        preds = sorted(f for f in os.listdir(pred_pdb_dir) if f.endswith(".pdb")) if os.path.exists(pred_pdb_dir) else []
        pairs = compare_many(os.path.join(native_pdb_dir, native), [os.path.join(pred_pdb_dir, p) for p in preds], device, align) \
            if device is not None and preds else None
        for k, pred in enumerate(preds):
            r, t = pairs[k] if pairs else compare(os.path.join(native_pdb_dir, native), os.path.join(pred_pdb_dir, pred), align)
            r, t = round(r, 3), round(t, 4)
            if best_r is None or r < best_r[0]:
                best_r = (r, pred[:-4])
            if best_t is None or t > best_t[0]:
                best_t = (t, pred[:-4])
        if best_r is None:
            continue
        lines.append(f"{native[:-4]} best_RMSD: {best_r[0]} model: {best_r[1]} best_TM_score: {best_t[0]} model: {best_t[1]}\n")
        rmsds.append(best_r[0]); tms.append(best_t[0])
    if not lines:
        raise ValueError("no (native, predicted) pair of .pdb files found")
    out = (float(np.min(rmsds)), float(np.max(tms)), float(np.mean(rmsds)), float(np.mean(tms)))
    lines += [f"Mean RMSD: {round(out[2], 2)}\n", f"Mean TM-score: {round(out[3], 2)}\n", f"Min RMSD: {round(out[0], 2)}\n",
              f"Max TM-score: {round(out[1], 2)}\n"]
    if save_summary:
        d = save_dir or pred_pdb_dir
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "summary.txt"), "w") as f:
            f.write("".join(lines))
    return out


def main(argv=None):
    """command line of the reference's evaluate.py: -n/--native_dir, -p/--pred_dir, -o/--output (file or folder), --align"""
    ap = argparse.ArgumentParser(prog="evaluate.py", description="Best RMSD and TM-score of a set of models against every native structure.")
    ap.add_argument("-n", "--native_dir", required=True, help="folder of native .pdb files")
    ap.add_argument("-p", "--pred_dir", required=True, help="folder of model .pdb files")
    ap.add_argument("-o", "--output", default=None, help="summary file (*.txt) or folder; the model folder if omitted")
    ap.add_argument("--align", action="store_true", help="establish the residue correspondence by sequence alignment (TM-score's -seq)")
    ap.add_argument("--device", type=int, default=None, help="superpose on this GPU (extension; numpy if omitted)")
    a = ap.parse_args(argv)
    folder, name = a.pred_dir, "summary.txt"
    if a.output:
        folder, name = (os.path.dirname(a.output) or os.getcwd(), os.path.basename(a.output)) if a.output.endswith(".txt") else (a.output, name)
    stats = run_score(a.native_dir, a.pred_dir, align=a.align, save_summary=True, save_dir=folder, device=a.device)
    if name != "summary.txt":
        shutil.move(os.path.join(folder, "summary.txt"), os.path.join(folder, name))
    print("Evaluation Summary:")
    for label, v in zip(("Min RMSD", "Max TM-score", "Mean RMSD", "Mean TM-score"), stats):
        print(f"  {label}: {round(v, 3)}")
    print(f"Full summary saved to: {os.path.join(folder, name)}")
    return 0
