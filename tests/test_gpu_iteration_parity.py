"""GPU: outcome parity on the ITERATION phase -- the half of the reference's eight example decoys that are folds of fed-back maps
(conf_1_3 / conf_1_4 = NMR/seq1, seq2; conf_2_3 / conf_2_4 = Xray/seq3, seq4).

The maps those decoys were folded from are rebuilt with the DEVICE feedback on the distograms resident in the fold context
(trx2_feedback_step; pinned bit for bit to the reference's functions): npz1 = feedback(seq_{tag}.npz, reference initial0),
npz2 = feedback(npz1, reference seq{k}).  Each is then folded 256 times with the default protocol (the reference's: --fastrelax on)
and the C-alpha RMSD distribution to the reference's decoy OF THAT MAP is asserted.  One reference decoy is one draw of the
reference's own distribution for its map (its initial pairs differ by 0.86 A on the NMR map and 0.62 A on the X-ray map), so
the comparison's floor is the reference's own spread; thresholds are measured values + sampling margin (docstring of the test).
tests/test_iteration_provenance.py holds the CPU half (scores that establish which decoy came from which map).
"""
import importlib
import os

import numpy as np
import pytest
import torch  # noqa: F401  -- before libtrx2fold.so (see test_gpu_boundary.py)

pytestmark = pytest.mark.gpu

from oracle.kabsch import kabsch_rmsd

T = importlib.import_module("trrosettax2-dynamics_amd")
FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
PD = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
P = T.protocol

CHAINS = {"NMR": ("conf_2_1", "conf_2_2", "conf_1_3", "conf_1_4"), "Xray": ("conf_1_1", "conf_1_2", "conf_2_3", "conf_2_4")}
B = 256


def fold_stats(ctx, ref, target, others, seed):
    r = ctx.fold_batch(B, P.build_runs(90, 2, fastrelax=True), seed=seed)
    assert np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"]))
    ca = r["xyz"][:, :, 1]
    d_t = np.array([kabsch_rmsd(ca[i], ref[target][:, 1]) for i in range(B)])
    d_o = {k: np.array([kabsch_rmsd(ca[i], ref[k][:, 1]) for i in range(B)]) for k in others}
    return r, d_t, d_o


# Measured on MI355X, 256 decoys per map and stage, seeds as below (profiles/r04_iteration_parity.txt): C-alpha RMSD to the reference's
# decoy of that map -- NMR stage 1 median 1.15 A (quartiles 1.07-1.22; 98 % within 1.5 A, 1.2 % beyond 3 A), NMR stage 2 1.10 A
# (0.97-1.35; 89 %, 1.6 %), X-ray stage 1 0.82 A (0.44-1.23; 84 %, 8.2 %), X-ray stage 2 0.76 A (0.54-1.26; 91 %, 8.2 %).  The reference's own decoys of one map differ by 0.81-1.51 A (NMR) and 0.34-0.72 A (X-ray) among themselves
# (SURVEY.md section 4): these folds sit inside the NMR spread and at the upper end of the X-ray one.  Limits = measured + margin
# (sampling error of a median at n = 256: ~0.03 A on the unimodal NMR map, ~0.1 A on the bimodal X-ray map).
# Closing build of round 4 (relax-stage scale, guard offset): 1.146 / 1.076 / 0.807 / 0.774 A; beyond 3 A 2.3 / 5.1 / 3.9 / 8.2 % (sd of a fraction of 5 % at n = 256: 1.4 %).
LIMITS = {("NMR", 1): dict(med=1.25, f15=0.90, far=0.06), ("NMR", 2): dict(med=1.22, f15=0.80, far=0.08),
          ("Xray", 1): dict(med=0.97, f15=0.76, far=0.13), ("Xray", 2): dict(med=0.92, f15=0.82, far=0.13)}


TWO_SAMPLE = {}


@pytest.mark.parametrize("tag", ["NMR", "Xray"])
def test_folds_of_the_fed_back_maps_reach_the_reference_iteration_decoys(golden_dir, seq, tmp_path, tag):
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    m = dict(np.load(os.path.join(golden_dir, f"seq_{tag}.npz")))
    i0, i1, s1, s2 = CHAINS[tag]
    ctx = T.Context(0, lanes=2)
    try:
        ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
        host = {k: m[k] for k in ("dist", "theta", "omega", "phi")}
        rows = []
        for stage, (fed, target) in enumerate(((i0, s1), (s1, s2)), start=1):
            # the decoy as the reference reads it back from its PDB file (the fixture holds exactly those float32 values; NaN = no CB)
            delta = ctx.feedback_step(ref[fed], seq, 1.0, True)
            path = str(tmp_path / f"{fed}.pdb")
            PD.write_pdb(path, seq, np.nan_to_num(ref[fed]))
            host = FB.feedback_labels(host, path, 1.0, True)
            for ch in ("dist", "omega", "theta", "phi", "tmp"):       # the resident maps ARE the reference's iteration npz
                assert np.array_equal(ctx.get_map(ch), host[ch]), (tag, stage, ch)
            assert delta > 0.01                                          # the reference's chain did not stop here either
            r, d_t, d_o = fold_stats(ctx, ref, target, (i0, i1), seed=7000 + 10 * stage)
            lim = LIMITS[(tag, stage)]
            far = d_t > 3.0
            rows.append((stage, target, np.median(d_t), np.percentile(d_t, 25), np.percentile(d_t, 75), 100 * (d_t <= 1.0).mean(), 100 * (d_t <= 1.5).mean(), 100 * far.mean(),
                         np.median(d_o[i0]), np.median(d_o[i1]), int(np.median(r["n_evals"]))))
            print("\n%s stage %d -> %s: median %.3f A (quartiles %.2f-%.2f), <=1 A %.0f %%, <=1.5 A %.0f %%, >3 A %.1f %%; to initial0 / initial1 of the map: %.2f / %.2f; evaluations %d"
                  % ((tag,) + rows[-1]))
            # Two-sample reading: ONE reference draw exists for this map; if it is a draw of our distribution, our decoys are as far
            # from it as from one another (non-mirror decoys; 120 of them -> 7 140 pairs)
            ok = [r["xyz"][i, :, 1].astype(np.float64) for i in np.nonzero(~far)[0][:120]]
            n_ok = len(ok)
            D = np.zeros((n_ok, n_ok))
            for i in range(n_ok):
                for j in range(i + 1, n_ok):
                    D[i, j] = D[j, i] = kabsch_rmsd(ok[i], ok[j])
            pw = D[np.triu_indices(n_ok, 1)]
            # every draw of ours has its own median distance to the other draws (central draws small, peripheral ones large): the
            # reference's draw, measured the same way, must be one of them
            own = np.array([np.median(np.delete(D[i], i)) for i in range(n_ok)])
            d_ok = np.median(d_t[~far][:n_ok])
            pct = 100.0 * (own < d_ok).mean()
            print("   two draws of ours: median %.3f A (5-95 %%: %.2f-%.2f); a draw's median distance to the others: %.2f-%.2f (5-95 %%); the reference's draw: %.3f A = percentile %.0f"
                  % (np.median(pw), np.percentile(pw, 5), np.percentile(pw, 95), np.percentile(own, 5), np.percentile(own, 95), d_ok, pct))
            TWO_SAMPLE[(tag, stage)] = (float(np.median(pw)), float(d_ok), float(pct))
            assert 1.0 <= pct <= 99.0, (pct, d_ok, np.sort(own)[-5:])     # two-sided: neither outside the cloud nor implausibly central; the joint statement is below
            assert np.median(d_t) <= lim["med"], np.sort(d_t)[::16]
            assert (d_t <= 1.5).mean() >= lim["f15"] and far.mean() <= lim["far"], ((d_t <= 1.5).mean(), far.mean())
            # no twisted peptides with the relax stage on (the reference's decoys: |omega| 177.6 deg mean, one cis in eight)
            dw = np.degrees(np.abs((r["tors"][:, :-1, 2] % (2 * np.pi)) - np.pi))
            assert (dw.max(1) > 60).mean() <= 0.03, (dw.max(1) > 60).mean()
    finally:
        ctx.close()

