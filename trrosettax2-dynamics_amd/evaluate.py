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
instead of ~50 k numpy SVDs per pair.  Not mirrored: `--align` (TM-score's -seq sequence alignment) -> NotImplementedError.
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


def compare(native_pdb, pred_pdb):
    """-> (rmsd, tm_score) as the TM-score program reports them for `TMscore native pred`"""
    a, b = read_ca(native_pdb), read_ca(pred_pdb)
    common = sorted(set(a) & set(b))
    if len(common) < 3:
        raise ValueError(f"{native_pdb} and {pred_pdb} share fewer than three residues")
    x = np.array([a[k] for k in common])
    y = np.array([b[k] for k in common])
    return rmsd_common(x, y), tm_score(x, y, l_norm=len(b))


def compare_many(native_pdb, pred_pdbs, device):
    """compare(native, p) for every p on the GPU: the models are grouped by (residues shared with the native, own length), each
    group is one trx2_superpose_matrix call"""
    from ._lib import Context
    a = read_ca(native_pdb)
    groups, out = {}, [None] * len(pred_pdbs)
    for k, p in enumerate(pred_pdbs):
        b = read_ca(p)
        common = tuple(sorted(set(a) & set(b)))
        if len(common) < 3:
            raise ValueError(f"{native_pdb} and {p} share fewer than three residues")
        groups.setdefault((common, len(b)), []).append((k, np.array([b[r] for r in common], np.float32)))
    ctx = Context(int(device))
    try:
        for (common, lb), items in groups.items():
            x = np.array([[a[r] for r in common]], np.float32)
            rm, tm = ctx.superpose_matrix(x, np.stack([y for _, y in items]), l_norm=lb)
            for (k, _), r, t in zip(items, rm[0], tm[0]):
                out[k] = (float(r), float(t))
    finally:
        ctx.close()
    return out


def run_score(native_pdb_dir, pred_pdb_dir, align=False, save_summary=False, save_dir=None, device=None):
    """evaluate_utils.py:33-100: -> (min_rmsd, max_tmscore, mean_rmsd, mean_tmscore); summary.txt in the reference's format
    (values rounded to the three / four decimals the TM-score program prints)"""
    if align:
        raise NotImplementedError("--align (TM-score -seq) is not implemented: residues are matched by number")
    lines, rmsds, tms = [], [], []
    for native in sorted(f for f in os.listdir(native_pdb_dir) if f.endswith(".pdb")):
        best_r, best_t = None, None
        preds = sorted(f for f in os.listdir(pred_pdb_dir) if f.endswith(".pdb")) if os.path.exists(pred_pdb_dir) else []
        pairs = compare_many(os.path.join(native_pdb_dir, native), [os.path.join(pred_pdb_dir, p) for p in preds], device) \
            if device is not None and preds else None
        for k, pred in enumerate(preds):
            r, t = pairs[k] if pairs else compare(os.path.join(native_pdb_dir, native), os.path.join(pred_pdb_dir, pred))
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
    ap.add_argument("--align", action="store_true", help="TM-score's -seq alignment: not implemented, raises")
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
