"""the open finding of tests/test_gpu_zz_open_findings.py as a MEASUREMENT.

The reference's stage-1 iteration decoy (the first fold of the map fed back from its initial decoy 0) lies near the 90th percentile of this
build's draws of that map, on both example maps.  DESIGN section 2 said "the reference's stage-1 decoys drift 0.72 / 0.39 A (core) from its
own initial decoys and this build's draws follow the map, not that drift" -- a statement about the RESPONSE of the ensemble to the feedback
step.  Measured here, per map (n decoys per ensemble, default protocol, C-alpha, everything superposed on the chain's 80-residue core as the
reference's two initial decoys define it):

  reference   d_ref  = stage-k decoy - mean of its two initial decoys           (one draw minus the mean of two draws)
  this build  d_ours = ensemble mean on the fed-back map - ensemble mean on the initial map   (the systematic part of the response)
  null        d_sim  = one draw of ours on the fed-back map - mean of two draws of ours on the initial map, 4000 times:
              what the reference's statistic looks like when the three structures ARE draws of this build's distributions

and then: |d_ref| against the distribution of |d_sim| (percentile), cos(d_ref, d_ours) against cos(d_sim, d_ours), the projection
beta = <d_ref, d_ours> / <d_ours, d_ours> (1 = the reference moved along this build's mean response by the same amount), the split of
<|d_sim|^2> into mean shift and spread, a per-residue table.  With variant names on the command line the fed-back folds are repeated under
modified protocols (restraint weights scaled, the minimiser's tolerance) and the test's percentile statistic is printed for each.

usage: python tests/diag/iteration_drift.py <repo> [n = 1024] [variant ...]     variants: w0.5 w0.75 w1.5 w2 tol1e-5 tol1e-4 norelax
(needs a GPU; no oracle: the fed-back maps come from the host mirror of the feedback step, tests/golden/make_oracle_outcomes.maps_and_targets)"""
import importlib
import importlib.util
import os
import sys

import numpy as np

repo = sys.argv[1]
sys.path.insert(0, repo)
T = importlib.import_module("trrosettax2-dynamics_amd")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
variants = sys.argv[3:]
g = os.path.join(repo, "tests", "golden")
spec = importlib.util.spec_from_file_location("make_oracle_outcomes", os.path.join(g, "make_oracle_outcomes.py"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
G = importlib.util.module_from_spec(spec)
# (the generator module imports the oracle at its top for its own main(); only maps_and_targets -- numpy + the package's host feedback -- is used)
spec.loader.exec_module(G)
seq, ref, cases = G.maps_and_targets(g)
cases = {k: (a, nm) for k, a, nm in cases}
rng = np.random.default_rng(6)
ALL = np.arange(90)


def fit(P, Q, idx):
    """P superposed onto Q over the residues idx (Kabsch, proper rotation); returns all of P transformed"""
    pc, qc = P[idx].mean(0), Q[idx].mean(0)
    U, S, Vt = np.linalg.svd((P[idx] - pc).T @ (Q[idx] - qc))
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1, 1, d]) @ U.T
    return (P - pc) @ R.T + qc


def rms(d, idx):
    return float(np.sqrt((d[idx] ** 2).sum(1).mean()))


def cosine(a, b, idx):
    a, b = a[idx].ravel(), b[idx].ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))


def variant_runs(name):
    runs = T.protocol.build_runs(90, 2, fastrelax=(name != "norelax"))
    if name.startswith("w"):
        s = float(name[1:])
        for r in runs:
            r["w"] = [r["w"][0] * s, r["w"][1] * s, r["w"][2] * s] + list(r["w"][3:])
    elif name.startswith("tol"):
        t = float(name[3:])
        for r in runs[:14]:
            r["tol"] = t
    return runs


def ensemble(ctx, arrs, runs, seed, frame, core):
    ctx.set_map(arrs["dist"], arrs["omega"], arrs["theta"], arrs["phi"], seq=seq)
    r = ctx.fold_batch(n, runs, seed=seed)
    ca = r["xyz"][:, :, 1].astype(np.float64)
    X = np.stack([fit(ca[i], frame, core) for i in range(n)])
    ok = np.sqrt(((X[:, core] - frame[core]) ** 2).sum(2).mean(1)) < 3.0      # mirror-image topologies aside
    return X[ok], int(np.median(r["n_evals"]))


def percentile_stat(X, target, k=120):
    """the statistic of tests/test_gpu_iteration_parity.py: the target's median RMSD to k draws, ranked among the draws' own medians"""
    from itertools import combinations
    Y = X[:k]
    m = len(Y)

    def rm(P, Q):
        return rms(fit(P, Q, ALL) - Q, ALL)
    D = np.zeros((m, m))
    for i, j in combinations(range(m), 2):
        D[i, j] = D[j, i] = rm(Y[i], Y[j])
    own = np.array([np.median(np.delete(D[i], i)) for i in range(m)])
    d_t = np.median([rm(Y[i], target) for i in range(m)])
    return 100.0 * float((own < d_t).mean()), float(d_t), float(np.median(D[np.triu_indices(m, 1)]))


ctx = T.Context(0, lanes=2)
base = T.protocol.build_runs(90, 2, fastrelax=True)
for tag, (i0, i1, s1, s2) in G.CHAINS.items():
    R0, R1, S1, S2 = (ref[k][:, 1].astype(np.float64) for k in (i0, i1, s1, s2))
    pair = np.sqrt(((fit(R0, R1, ALL) - R1) ** 2).sum(1))
    core = np.sort(np.argsort(pair)[:80])
    Mref = 0.5 * (R0 + fit(R1, R0, core))
    E = {}
    for key, sd in (("initial", 61), ("stage1", 62), ("stage2", 63)):
        E[key], ev = ensemble(ctx, cases[f"{tag}/{key}"][0], base, sd, Mref, core)
        print(f"== {tag}/{key}: {len(E[key])} of {n} draws kept (core RMSD to the reference's initial mean < 3 A), evaluations median {ev}")
    mean0 = E["initial"].mean(0)
    for stage, S in ((1, S1), (2, S2)):
        Xs = E[f"stage{stage}"]
        d_ref = fit(S, Mref, core) - Mref
        d_ours = Xs.mean(0) - mean0
        # the reference's statistic on draws of this build
        mags, coss, betas = [], [], []
        for _ in range(4000):
            a, b = E["initial"][rng.choice(len(E["initial"]), 2, replace=False)]
            c = Xs[rng.integers(len(Xs))]
            M = 0.5 * (a + fit(b, a, core))
            d = fit(c, M, core) - M
            mags.append(rms(d, core)); coss.append(cosine(d, d_ours, core)); betas.append(float((d[core] * d_ours[core]).sum() / (d_ours[core] ** 2).sum()))
        mags, coss, betas = np.array(mags), np.array(coss), np.array(betas)
        m_ref, c_ref = rms(d_ref, core), cosine(d_ref, d_ours, core)
        b_ref = float((d_ref[core] * d_ours[core]).sum() / (d_ours[core] ** 2).sum())
        sp0 = float(np.sqrt(((E["initial"][:, core] - mean0[core]) ** 2).sum(2).mean()))
        sp1 = float(np.sqrt(((Xs[:, core] - Xs.mean(0)[core]) ** 2).sum(2).mean()))
        print(f"-- {tag} stage {stage} (core of 80 residues; all 90 in brackets)")
        print(f"   reference: |d_ref| = {m_ref:.3f} A ({rms(d_ref, ALL):.3f})   [stage decoy minus the mean of the two initial decoys]")
        print(f"   this build: |d_ours| = {rms(d_ours, core):.3f} A ({rms(d_ours, ALL):.3f})   [shift of the ensemble mean, initial map -> fed-back map]; "
              f"spread of a draw about its ensemble mean: {sp0:.3f} A (initial map), {sp1:.3f} A (fed-back map)")
        print(f"   null |d_sim| (one draw minus mean of two draws, 4000 x): median {np.median(mags):.3f}, 5-95 % {np.percentile(mags, 5):.3f}-{np.percentile(mags, 95):.3f};  "
              f"the reference's {m_ref:.3f} A sits at percentile {100 * (mags < m_ref).mean():.0f}")
        print(f"   expected from mean shift + spread: sqrt(|d_ours|^2 + s1^2 + s0^2 / 2) = {np.sqrt(rms(d_ours, core) ** 2 + sp1 ** 2 + sp0 ** 2 / 2):.3f} A")
        print(f"   direction: cos(d_ref, d_ours) = {c_ref:+.3f}; null cos(d_sim, d_ours): median {np.median(coss):+.3f}, 5-95 % {np.percentile(coss, 5):+.3f}..{np.percentile(coss, 95):+.3f} "
              f"(percentile of the reference's: {100 * (coss < c_ref).mean():.0f})")
        print(f"   projection beta = <d_ref, d_ours> / |d_ours|^2 = {b_ref:+.2f}; null: median {np.median(betas):+.2f}, 5-95 % {np.percentile(betas, 5):+.2f}..{np.percentile(betas, 95):+.2f}")
        pct, d_t, pw = percentile_stat(Xs, fit(S, Mref, core))
        print(f"   the test's statistic (120 draws): the reference decoy's median distance to our draws {d_t:.3f} A = percentile {pct:.0f} of the draws' own medians (two draws of ours: {pw:.3f} A)")
        if stage == 1:
            print("   per residue (core-superposed):  residue type  |d_ref|  |d_ours|  spread(fed-back)  in core")
            sres = np.sqrt(((Xs - Xs.mean(0)) ** 2).sum(2).mean(0))
            for i in range(90):
                print(f"      {i + 1:3d} {seq[i]}  {np.linalg.norm(d_ref[i]):6.2f}  {np.linalg.norm(d_ours[i]):6.2f}  {sres[i]:6.2f}  {'*' if i in core else ' '}")
    for v in variants:
        runs = variant_runs(v)
        out = []
        for stage, S in ((1, S1), (2, S2)):
            Xv, ev = ensemble(ctx, cases[f"{tag}/stage{stage}"][0], runs, 61 + stage, Mref, core)
            pct, d_t, pw = percentile_stat(Xv, fit(S, Mref, core))
            sp = float(np.sqrt(((Xv[:, core] - Xv.mean(0)[core]) ** 2).sum(2).mean()))
            out.append(f"stage {stage}: percentile {pct:3.0f} (reference decoy {d_t:.3f} A from our draws, two of ours {pw:.3f} A, spread {sp:.3f} A, mean shift {rms(Xv.mean(0) - mean0, core):.3f} A, {len(Xv)} kept, evals {ev})")
        print(f"   variant {v:8s} {tag}: " + " | ".join(out))
ctx.close()
