"""Fit of the rama / omega backbone terms to the only Rosetta energies the reference tree holds: the per-residue POSE_ENERGIES_TABLE of
its eight committed decoys (tests/golden/pose_energies.json, ref2015_cart).  Prints the constants of
include/trx2_model.h (TRX2_RAMA_FIT_*, TRX2_OMEGA_FIT_*); tests/test_pose_energies.py re-runs the fit and compares.

What the table holds.  rama_prepro is a TWO-BODY energy in Rosetta (residue i's term needs the identity of i+1) and the table splits every
two-body energy half / half between its residues: T_i = (e_{i-1} + e_i) / 2 with e_0 = 0.  The recursion e_i = 2 T_i - e_{i-1} recovers the
per-residue terms; it closes on e_{L-1} = 0 (the last residue has no psi) to 1e-4 in all eight decoys, which confirms the reading.
omega is a one-body term.

Model (same arithmetic in oracle/trx2_oracle.c and csrc/kernel_step.h):
  rama_i  = prior(phi, psi) + c_class + sum_k a_class,k f_k(phi, psi) + h_class r_alpha(phi, psi)   [+ dev_aa r_alpha with --per-aa: diagnostic]
            prior = the six-basin mixture of rounds 1-4 (the reference's own start table, utils_ros.py:667-696);
            f = cos psi, sin psi, cos(phi - psi), sin(phi - psi), cos phi, sin phi, cos(phi + psi), sin(phi + psi);
            classes: general / glycine (own surface), proline / residue before a proline (a constant: 8 samples at one place each);
            r_alpha = the mixture's posterior weight of the two right-handed helical basins, h_class a helix constant per class (general, glycine).
  omega_i = A(psi_i) + B(psi_i) x + C(psi_i) x^2,  x = (omega_i - 180 deg) / 10 deg,  A, B, C = q0 + q1 cos psi + q2 sin psi
            (ref2015's tether has a psi / phi-dependent centre and width; psi_i alone carries most of it: leave-one-decoy-out rank
            correlation 0.80 against 0.85 with phi_{i+1} as well and 0.68 for one global quadratic).
Ridge regression (constants unpenalised); validated leaving one decoy out -- an optimistic validation (all eight decoys are folds of ONE
sequence, so a held-out decoy shares its residue types and most of its torsions with the training set): it guards against a fit of noise,
not against what another protein would show.
usage: fit_backbone_terms.py <repo> [--quiet] [--per-aa] [--decoys=conf_1_1,conf_1_2,..] [--flags]
  --decoys: fit on these decoys only, report the rank correlations on the others (leave-one-chain-out: tests/diag/fit_generalisation.py)
  --flags:  print the constants as -D compiler flags (variant builds of the library) instead of as header lines"""
import json, os, sys
import numpy as np

ROOT = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AA = "ACDEFGHIKLMNPQRSTVWY"
BASINS = np.array([(-140, 153, .135), (-72, 145, .155), (-122, 117, .073), (-82, -14, .122), (-61, -41, .497), (57, 39, .018)])  # trx2_model.h TRX2_RAMA_INIT
KAPPA, FLOOR, PREF = 8.0, 1e-3, 0.497
LAM_RAMA, LAM_OMEGA = 3.0, 1.0
SHRINK = 0.5                     # include/trx2_model.h TRX2_RAMA_FIT_SHRINK: the class surfaces enter at one half (outcome scans, profiles/r05_model_scan*.txt)
W_RAMA, W_OMEGA = 0.45, 0.4      # ref2015_cart's weights: the table holds weighted energies


def dih(a, b, c, d_):
    b0, b1, b2 = a - b, c - b, d_ - c
    b1 = b1 / np.linalg.norm(b1, axis=-1, keepdims=True)
    v = b0 - (b0 * b1).sum(-1, keepdims=True) * b1; w = b2 - (b2 * b1).sum(-1, keepdims=True) * b1
    return np.arctan2((np.cross(b1, v) * w).sum(-1), (v * w).sum(-1))


def load():
    g = os.path.join(ROOT, "tests", "golden")
    tab = json.load(open(os.path.join(g, "pose_energies.json"))); ref = np.load(os.path.join(g, "ref_decoys.npz"))
    seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
    phi, psi, om, e, w_ = [], [], [], [], []
    for name in sorted(tab):
        x = ref[name].astype(np.float64); N, CA, C = x[:, 0], x[:, 1], x[:, 2]; L = len(x)
        p, s, o = np.full(L, np.nan), np.full(L, np.nan), np.full(L, np.nan)
        p[1:] = dih(C[:-1], N[1:], CA[1:], C[1:]); s[:-1] = dih(N[:-1], CA[:-1], C[:-1], N[1:]); o[:-1] = dih(CA[:-1], C[:-1], N[1:], CA[1:])
        T = np.array(tab[name]["per_residue"]["rama_prepro"])
        ei = np.zeros(L)
        for i in range(L):
            ei[i] = 2 * T[i] - (ei[i - 1] if i else 0.0)
        assert abs(ei[-1]) < 5e-4, (name, ei[-1])           # the half / half reading closes
        phi.append(p); psi.append(s); om.append(o); e.append(ei / W_RAMA); w_.append(np.array(tab[name]["per_residue"]["omega"]) / W_OMEGA)
    return seq, np.array(phi), np.array(psi), np.array(om), np.array(e), np.array(w_)


def prior(phi, psi):
    t = np.stack([w * np.exp(KAPPA * (np.cos(phi - np.radians(p0)) + np.cos(psi - np.radians(s0)) - 2)) for p0, s0, w in BASINS])
    s = t.sum(0)
    return -np.log((s + FLOOR) / PREF), (t[3] + t[4]) / (s + FLOOR)


def rama_features(phi, psi):
    return np.stack([np.cos(psi), np.sin(psi), np.cos(phi - psi), np.sin(phi - psi), np.cos(phi), np.sin(phi), np.cos(phi + psi), np.sin(phi + psi)], -1)


def classes(seq):
    cls = np.zeros(len(seq), int)            # 0 general, 1 glycine, 2 proline, 3 before a proline
    for i, a in enumerate(seq):
        if a == "G": cls[i] = 1
        elif a == "P": cls[i] = 2
        elif i + 1 < len(seq) and seq[i + 1] == "P": cls[i] = 3
    return cls


def ridge(X, y, lam, pen):
    return np.linalg.solve(X.T @ X + lam * np.diag(pen), X.T @ y)


def fit_rama(seq, phi, psi, E, decoys=None, per_aa=False):
    """Two passes: (1) ridge fit of class surfaces + class constants + helix terms together; (2) with the surfaces fixed at SHRINK
    times their fitted coefficients, the constants and helix terms are fitted again (they absorb what the shrunk surfaces leave).
    Helix terms: one constant per class (general, glycine; unpenalised like the class constants) and -- per_aa only, a diagnostic since round 6
    -- a ridge-penalised deviation per residue type on top of it (round 5 shipped those per-type numbers as the default).
    -> (w[44]: general surface 8 (as fitted, unshrunk), glycine surface 8, constants 4, class helix 4, per-type deviations 20; model energies [decoys, L])"""
    L = len(seq); idx = np.arange(1, L - 1); cls = classes(seq)
    d = list(range(len(phi))) if decoys is None else decoys
    E0, ra = prior(phi, psi); F = rama_features(phi, psi)
    X = np.zeros(phi.shape + (8 + 8 + 4 + 4 + 20,))
    X[..., 0:8] = F * (cls == 0)[None, :, None]; X[..., 8:16] = F * (cls == 1)[None, :, None]
    for c in range(4): X[..., 16 + c] = (cls == c)[None, :]
    for c in range(2): X[..., 20 + c] = ra * (cls == c)[None, :]
    if per_aa:
        for i, a in enumerate(seq):
            if a in AA: X[:, i, 24 + AA.index(a)] = ra[:, i]
    X = np.nan_to_num(X); y = np.nan_to_num(E - E0)
    pen = np.ones(X.shape[-1]); pen[16:20] = 1e-6      # class constants unpenalised; the helix constants are shape terms: penalised like the surfaces
    w = ridge(X[d][:, idx].reshape(-1, X.shape[-1]), y[d][:, idx].reshape(-1), LAM_RAMA, pen)
    surf = X[..., :16] @ (SHRINK * w[:16])
    w2 = ridge(X[d][:, idx][..., 16:].reshape(-1, 28), (y - surf)[d][:, idx].reshape(-1), LAM_RAMA, pen[16:])
    w[16:] = w2
    w[24:][~np.array([a in seq for a in AA])] = 0.0
    return w, E0 + surf + X[..., 16:] @ w[16:]


def fit_omega(psi, om, E, decoys=None):
    L = psi.shape[1]; idx = np.arange(1, L - 1)     # Rosetta scores no omega tether on a terminus (the table's first entry is ~0 whatever the angle)
    d = list(range(len(psi))) if decoys is None else decoys
    x = np.degrees(om); x = (np.where(x < 0, x + 360, x) - 180.0) / 10.0
    F = np.stack([np.ones_like(psi), np.cos(psi), np.sin(psi)], -1)
    X = np.concatenate([F, F * x[..., None], F * (x ** 2)[..., None]], -1)
    pen = np.ones(9); pen[[0, 3, 6]] = 0.0
    Xs = np.nan_to_num(X)
    w = ridge(Xs[d][:, idx].reshape(-1, 9), np.nan_to_num(E)[d][:, idx].reshape(-1), LAM_OMEGA, pen)
    return w, Xs @ w


def spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)).astype(float), np.argsort(np.argsort(b)).astype(float)
    return float(np.corrcoef(ra, rb)[0, 1])


def main(quiet=False, per_aa=False, decoys=None, flags=False):
    seq, phi, psi, om, E, W = load()
    names = sorted(json.load(open(os.path.join(ROOT, "tests", "golden", "pose_energies.json"))))
    sel = None if decoys is None else [names.index(n) for n in decoys]
    L = len(seq); ir = np.arange(1, L - 1); io = np.arange(1, L - 1)
    wr, pr = fit_rama(seq, phi, psi, E, sel, per_aa)
    wo, po = fit_omega(psi, om, W, sel)
    assert wo[6] - np.hypot(wo[7], wo[8]) > 0, "the omega term's curvature must stay positive for every psi"
    if not quiet:
        held = [a for a in range(len(phi)) if sel is None or a not in sel]
        if sel is None:
            lodo_r, lodo_o, lodo_aa = [], [], []
            for a in range(len(phi)):
                tr = [b for b in range(len(phi)) if b != a]
                lodo_r.append(spearman(fit_rama(seq, phi, psi, E, tr, per_aa)[1][a, ir], E[a, ir]))
                lodo_aa.append(spearman(fit_rama(seq, phi, psi, E, tr, True)[1][a, ir], E[a, ir]))
                lodo_o.append(spearman(fit_omega(psi, om, W, tr)[1][a, io], W[a, io]))
        else:       # fitted on a subset: the other decoys are the held-out set
            lodo_r = [spearman(pr[a, ir], E[a, ir]) for a in held]; lodo_o = [spearman(po[a, io], W[a, io]) for a in held]; lodo_aa = lodo_r
        E0 = prior(phi, psi)[0]
        x = np.degrees(om); x = np.where(x < 0, x + 360, x) - 180.0
        fitted = list(range(len(phi))) if sel is None else sel
        print("rama : rank correlation over residues, median over decoys: six-basin prior %.2f | fit, in sample %.2f | fit, held out %.2f (min %.2f)%s"
              % (np.median([spearman(E0[a, ir], E[a, ir]) for a in range(8)]), np.median([spearman(pr[a, ir], E[a, ir]) for a in fitted]), np.median(lodo_r), min(lodo_r),
                 "" if per_aa or sel is not None else " | with per-type deviations, held out %.2f (min %.2f)" % (np.median(lodo_aa), min(lodo_aa))))
        print("omega: quadratic tether at 180 deg %.2f | fit, in sample %.2f | fit, held out %.2f (min %.2f)"
              % (np.median([spearman((x ** 2)[a, io], W[a, io]) for a in range(8)]), np.median([spearman(po[a, io], W[a, io]) for a in fitted]), np.median(lodo_o), min(lodo_o)))
    f = lambda v: "{" + ", ".join("%.4ff" % t for t in v) + "}"
    defs = [("TRX2_RAMA_FIT_GENERAL", f(wr[0:8]), ""), ("TRX2_RAMA_FIT_GLY", f(wr[8:16]), ""),
            ("TRX2_RAMA_FIT_CONST", f(wr[16:20]), " /* general, glycine, proline, before a proline */"),
            ("TRX2_RAMA_FIT_HELIX_CLASS", f(wr[20:24]), " /* general, glycine, proline, before a proline */")]
    if per_aa:
        defs.append(("TRX2_RAMA_FIT_HELIX_AA", f(wr[24:44]), " /* " + AA + ": deviation from the class constant (diagnostic builds) */"))
    defs.append(("TRX2_OMEGA_FIT", f(wo), " /* A(psi), B(psi), C(psi): q0 + q1 cos psi + q2 sin psi each; x = (omega - 180 deg) / 10 deg */"))
    out = ["#define TRX2_RAMA_FIT_SHRINK %.2f" % SHRINK] + ["#define %s %s%s" % d for d in defs]
    if flags:       # the same constants as compiler flags (variant builds: make -C csrc variant VARIANT=<name> VFLAGS="...")
        print(" ".join(["-D%s='%s'" % (n, v) for n, v, _ in defs] + (["-DTRX2_RAMA_FIT_AA=1"] if per_aa else [])))
    else:
        print("\n".join(out))
    return out


if __name__ == "__main__":
    dec = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--decoys=")]
    main("--quiet" in sys.argv, "--per-aa" in sys.argv, dec[0] if dec else None, "--flags" in sys.argv)
