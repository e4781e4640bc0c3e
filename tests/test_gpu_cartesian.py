"""GPU: the Cartesian-space run (MinMover.cartesian(True) on sf_cart, folding.py:83-84,100-102) against the oracle."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd

T = importlib.import_module("trrosettax2-dynamics_amd")
P = T.protocol


@pytest.fixture(scope="module")
def ctx():
    c = T.Context(0)
    yield c
    c.close()


PCT_INITIAL = {}


def cart_only(L, max_iter=1000):
    return [dict(w=P.SF_CART, max_iter=max_iter, sep_lo=1, sep_hi=L, precheck=0, skip_to=0, cartesian=1)]


def test_cartesian_run_tracks_oracle_over_a_short_horizon(ctx, golden_dir, seq):
    """k_cart alone: same start, same budget.  Energies, accepted iterations and (superposed) coordinates agree while the
    float32 and float64 trajectories are still together; the relaxed geometry comes back through the internal coordinates."""
    m = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    rng = np.random.default_rng(5)
    B = 6
    t0 = np.stack([O.random_torsions(90, 31, d) + rng.normal(size=(90, 3)) * 0.05 for d in range(B)]).astype(np.float32)
    runs = cart_only(90)
    rows = []
    for n in (12, 40):
        r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
        assert np.all(np.isfinite(r["xyz"]))
        for d in range(B):
            to, xo, st = O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=n)
            rows.append((n, d, r["f"][d], st["f_final"], int(r["n_iters"][d]), st["n_iters"], kabsch_rmsd(r["xyz"][d].reshape(-1, 3), xo.reshape(-1, 3))))
    print("\nevals decoy      f_device      f_oracle   iters dev/orc   all-atom RMSD dev-vs-orc")
    for q in rows:
        print("%5d %4d  %12.2f  %12.2f   %3d / %3d        %.4f" % q)
    # At 12 evaluations about half the decoys are still in exact lockstep (energy to 1e-7, all-atom RMSD < 0.002 A); in the
    # others one line-search decision has flipped between float32 and float64, which in this steep first phase moves the
    # energy by 0.1-1.3 % (measured: median relative difference 5.7e-4, worst 1.3e-2 in round 3; 2.5e-3 / 1.1e-2 with round 4's
    # build, whose multiply-adds are fused as the source writes them: another rounding, other decoys flip).
    short = [q for q in rows if q[0] == 12]
    rel = np.array([abs(q[2] - q[3]) / abs(q[3]) for q in short])
    assert sum(q[4] == q[5] for q in short) >= B - 2
    assert np.median(rel) <= 5e-3 and rel.max() <= 3e-2, rel
    assert all(q[6] < 0.1 for q in short), short
    late = [q for q in rows if q[0] == 40]
    assert max(abs(q[2] - q[3]) / abs(q[3]) for q in late) <= 5e-2 and all(q[6] < 0.5 for q in late), late
    # The bonded term on the device holds the chain together as it does in the oracle.  This start is deliberately harsh
    # (full-weight restraints on an unfolded chain; the real protocol reaches the Cartesian run only after torsion-space
    # folding, where the spread is 0.011 A -- next test), so the bound is the oracle's own under the same run, not a constant.
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=200)
    dev = lambda x: max(np.abs(np.linalg.norm(x[:, 1] - x[:, 0], axis=-1) - 1.458).max(), np.abs(np.linalg.norm(x[1:, 0] - x[:-1, 2], axis=-1) - 1.334).max())
    dmax = max(dev(r["xyz"][d]) for d in range(B))
    omax = max(dev(O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=200)[1]) for d in range(B))
    print("largest N-CA / C-N bond deviation after 200 evaluations: device %.3f A, oracle %.3f A" % (dmax, omax))
    assert dmax <= 1.5 * omax + 0.02, (dmax, omax)


# relax: the default protocol (--fastrelax, the reference's default) -- measured on 1024 decoys per map (profiles/r03_outcome_n1024.txt):
# X-ray median 0.480 A, 58 % within 0.5 A, 83 % within 1 A; NMR 0.746 A, 6 % / 86 %; NO peptide twisted beyond 60 degrees in either map
# (5 % / 9 % without the stage).  plain: --no-fastrelax (rounds 1-3's protocol), thresholds as before.
@pytest.mark.parametrize("relax", [True, False], ids=["fastrelax", "no-fastrelax"])
@pytest.mark.parametrize("tag,refs,med_max,trap_max,f05_min,f10_min", [("Xray", ("conf_1_1", "conf_1_2"), 0.75, 0.11, 0.40, 0.74),
                                                                         ("NMR", ("conf_2_1", "conf_2_2"), 0.90, 0.08, 0.0, 0.74)])
def test_full_protocol_with_cartesian_stage(ctx, golden_dir, seq, tag, refs, med_max, trap_max, f05_min, f10_min, relax):
    """Mode-2 protocol with the Cartesian run on (the default for L <= 512): OUTCOME parity with the reference's PyRosetta decoys
    (the energy model itself is unpinned, DESIGN.md section 2, so this is the only anchor it has).
    Thresholds = measured + margin, not slack.  Measured on 1024 decoys per map
    (profiles/r02_outcome_parity_n1024.txt): median RMSD to the closer initial reference decoy X-ray 0.48 A (quartiles 0.39-0.89),
    NMR 0.76 A (0.63-0.90); decoys > 3 A away 6.2 % / 3.6 %, nearly all of them the mirror-image topology (6.0 % / 3.3 %).
    The X-ray distribution is BIMODAL -- a cluster at 0.3-0.5 A (half of the decoys) and one at 0.65-1.2 A, almost nothing in
    between -- so its median jumps across the gap with the sample (0.48 on 1024 decoys; 0.53 and 0.64 on two 256-decoy
    samples of round 3 whose builds differ by rounding only) and cannot carry a tight bound: for that map the cluster
    POPULATIONS are asserted (>= 40 % within 0.5 A: measured 53 % / 46-47 %; >= 74 % within 1 A: measured 81 % / 79-81 %; sampling
    sd of a fraction at n = 256: 3 %) and the median only loosely (<= 0.75).  NMR is unimodal: median <= 0.90 A (sampling error
    ~0.03), >= 74 % within 1 A (measured 83 %).  Trapped starts <= 11 % / 8 % (sd of the count 3.9 / 3.0 decoys: a doubling fails),
    and the far decoys ARE mirror topologies (>= 70 % of them closer to the mirrored reference than to the reference, the
    criterion of that table)."""
    m = np.load(os.path.join(golden_dir, f"seq_{tag}.npz"))
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    dec = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    B = 256  # mirror trapping is decided by the random start (3-6 % of starts): a rate needs a large batch
    runs = P.build_runs(90, 2, fastrelax=relax)
    assert any(q["cartesian"] for q in runs)
    if relax and tag == "Xray":
        f05_min = 0.55            # measured 64.8 % of 4096 with round 5's fitted rama term (58 % before; sampling sd of a fraction at n = 256: 3 %)
    r = ctx.fold_batch(B, runs, seed=4242)
    assert np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"]))
    best = np.array([min(kabsch_rmsd(r["xyz"][i, :, 1], dec[k][:, 1]) for k in refs) for i in range(B)])
    geo = [O.extract_internal(r["xyz"][i].astype(np.float64))[1] for i in range(B)]
    bond_sd = np.mean([g[:, 1].std() for g in geo]); ang_sd = np.mean([np.degrees(g[:, 3]).std() for g in geo])
    dw = np.degrees(np.abs((r["tors"][:, :-1, 2] % (2 * np.pi)) - np.pi))
    twisted = int((dw.max(1) > 60).sum())
    print("\ncart protocol, " + tag + " map, %d decoys: median RMSD %.2f (good only %.2f), >3 A: %d, twisted>60: %d, CA-C sd %.3f, N-CA-C sd %.1f, evals median %d, %.3f s"
          % (B, np.median(best), np.median(best[best < 3]), int((best > 3).sum()), twisted, bond_sd, ang_sd, np.median(r["n_evals"]), r["seconds"]))
    far = np.nonzero(best > 3.0)[0]
    mir = np.array([min(kabsch_rmsd(r["xyz"][i, :, 1] * np.array([1.0, 1.0, -1.0]), dec[k][:, 1]) for k in refs) for i in far])
    n_mirror = int((mir < best[far]).sum())   # profiles/r02_outcome_parity_n1024.txt's criterion: closer to the mirrored reference
    print("   <= 0.5 A: %.0f %%, <= 1 A: %.0f %%; > 3 A: %d of %d, of which mirror topologies (closer to the mirrored reference): %d"
          % (100 * (best <= 0.5).mean(), 100 * (best <= 1.0).mean(), len(far), B, n_mirror))
    assert np.median(best) <= med_max, np.sort(best)[::16]
    assert (best <= 0.5).mean() >= f05_min and (best <= 1.0).mean() >= f10_min, ((best <= 0.5).mean(), (best <= 1.0).mean())
    assert len(far) <= trap_max * B, (len(far), B)
    assert n_mirror >= 0.7 * len(far) - 1, (n_mirror, len(far))
    # Two-sample reading (profiles/r04_outcome_n1024.txt).  The reference's two decoys of a map are two draws of ITS distribution;
    # if ours is the same distribution, (i) the reference's pair is as far apart as two of our draws typically are, and (ii) a draw
    # of ours is as far from ONE reference draw as from another draw of ours.  Measured on 1024 decoys (non-mirror ones), default
    # protocol / --no-fastrelax: NMR two of ours 0.836 / 0.880 A, the reference's two 0.861 A (percentile 54 / 47 of ours), ours to one
    # reference draw 0.866 / 0.895; X-ray 0.684 / 0.725, 0.615 (39 / 32), 0.690 / 0.725.
    ok = [r["xyz"][i, :, 1].astype(np.float64) for i in np.nonzero(best < 3.0)[0][:120]]
    n_ok = len(ok)
    D = np.zeros((n_ok, n_ok))
    for i in range(n_ok):
        for j in range(i + 1, n_ok):
            D[i, j] = D[j, i] = kabsch_rmsd(ok[i], ok[j])
    pw = D[np.triu_indices(n_ok, 1)]
    ref_pair = kabsch_rmsd(dec[refs[0]][:, 1], dec[refs[1]][:, 1])
    to_ref = np.array([[kabsch_rmsd(c, dec[k][:, 1]) for k in refs] for c in ok])
    pct = 100.0 * (pw < ref_pair).mean()
    # (iii) every draw has its own median distance to the other draws (central ones small, peripheral ones large): each reference
    # draw, measured the same way against our draws, must be one of them (tests/test_gpu_iteration_parity.py does the same for the
    # four iteration-phase decoys: percentiles 94 / 62 / 91 / 66)
    own = np.array([np.median(np.delete(D[i], i)) for i in range(n_ok)])
    pct_ref = [100.0 * (own < np.median(to_ref[:, q])).mean() for q in range(2)]
    print("   two draws of ours: median %.3f A; the reference's two: %.3f A (percentile %.0f of ours); ours to one reference draw: median %.3f A; "
          "a draw's median distance to the others %.2f-%.2f (5-95 %%), %s %.3f = percentile %.0f, %s %.3f = percentile %.0f"
          % (np.median(pw), ref_pair, pct, np.median(to_ref), np.percentile(own, 5), np.percentile(own, 95),
             refs[0], np.median(to_ref[:, 0]), pct_ref[0], refs[1], np.median(to_ref[:, 1]), pct_ref[1]))
    assert 15.0 <= pct <= 85.0, (pct, ref_pair, np.median(pw))
    assert abs(np.median(to_ref) - np.median(pw)) <= 0.12, (np.median(to_ref), np.median(pw))
    assert 1.0 <= min(pct_ref) and max(pct_ref) <= 99.0, pct_ref     # two-sided
    PCT_INITIAL[(tag, relax)] = pct_ref
    # Geometry spread.  --no-fastrelax: where the reference's decoys are (CA-C sd 0.011 A, N-CA-C sd 2.4 deg: the bonded term's
    # calibration, trx2_model.h).  With the relax stage the LAST run is a Cartesian minimisation WITHOUT restraints (folding.py:257-263)
    # under ref2015_cart's cart_bonded weight 0.5: nothing strains the backbone any more -- the reference's full-atom terms, which
    # do, do not exist here -- and the geometry returns to ideal (measured sd 0.0003 A / 0.2 deg): documented deviation, DESIGN.md.
    assert bond_sd < 0.02 and ang_sd < 4.5 and (relax or ang_sd > 1.5), (bond_sd, ang_sd)
    # twisted peptides: none with the relax stage (measured 0 of 1024 per map; the reference's eight decoys hold one cis peptide and
    # none twisted); without it 4-11 % (deviation 2): measured + margin each
    # round 5 (fitted omega tether, 3 x ref2015's stiffness about its own centre): 0.1 / 0.2 % of 4096 with the relax stage, 7.5 % (NMR) /
    # 28.9 % (X-ray) without (profiles/r05_model_final_n4096.txt; rounds 1-4's tether at 180 degrees: 4 % / 12 %); sd of a fraction of 29 % at n = 256: 2.8 %
    assert twisted <= (0.02 if relax else (0.36 if tag == "Xray" else 0.13)) * B, twisted


def test_the_four_initial_reference_decoys_are_jointly_typical_draws():
    """Joint statement over the four initial-phase reference decoys (two per map), default protocol: percentiles uniform on 0..100 if they
    are draws of this build's distributions; the mean of four has sd 14.4 -- asserted within 50 +- 25 (round 4 measured 77 / 23 and 29 / 86:
    mean 54)."""
    got = [p for (tag, relax), pr in PCT_INITIAL.items() if relax for p in pr]
    assert len(got) == 4, PCT_INITIAL
    print("\npercentiles of the four initial reference decoys among this build's draws (default protocol):", np.round(got), "mean %.1f" % np.mean(got))
    assert 25.0 <= np.mean(got) <= 75.0, got


def test_cartesian_run_on_a_chain_longer_than_256(ctx):
    """256 < L <= 512: the Cartesian role runs with 512 threads (one residue each) next to the 256-thread torsion role in the
    same launch.  Same short-horizon agreement with the oracle as at L=90, then the full default protocol on the same map."""
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    L, B = 300, 4
    m = S.make_map(L, seed=L, n_moves=150)
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    rng = np.random.default_rng(9)
    t0 = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.08 for _ in range(B)]).astype(np.float32)
    rows = []
    for n in (12, 40):
        r = ctx.fold_batch(B, cart_only(L), tors0=t0, max_evals=n)
        assert np.all(np.isfinite(r["xyz"]))
        for d in range(B):
            to, xo, st = O.fold(Tb, t0[d].astype(np.float64), cart_only(L), max_evals=n)
            rows.append((n, d, r["f"][d], st["f_final"], int(r["n_iters"][d]), st["n_iters"], kabsch_rmsd(r["xyz"][d].reshape(-1, 3), xo.reshape(-1, 3))))
    print("\nL=300  evals decoy      f_device      f_oracle   iters dev/orc   all-atom RMSD dev-vs-orc")
    for q in rows:
        print("       %5d %4d  %12.2f  %12.2f   %3d / %3d        %.4f" % q)
    short = [q for q in rows if q[0] == 12]
    rel = np.array([abs(q[2] - q[3]) / abs(q[3]) for q in short])
    assert np.median(rel) <= 5e-3 and rel.max() <= 5e-2, rel
    assert all(q[6] < 0.2 for q in short), short
    runs = P.build_runs(L, 2)
    assert any(q["cartesian"] for q in runs)
    r = ctx.fold_batch(B, runs, tors0=t0)
    assert np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"]))
    ca = S.nerf_backbone(m["tors"])[1]
    rm = [kabsch_rmsd(r["xyz"][i, :, 1], ca) for i in range(B)]
    geo = [O.extract_internal(r["xyz"][i].astype(np.float64))[1] for i in range(B)]
    print("L=300 full protocol from near-native starts: RMSD to target %s, CA-C sd %.3f, N-CA-C sd %.1f" % (np.round(rm, 2), np.mean([g[:, 1].std() for g in geo]), np.mean([np.degrees(g[:, 3]).std() for g in geo])))
    assert np.mean([g[:, 1].std() for g in geo]) < 0.03 and np.median(rm) < 3.0


@pytest.mark.parametrize("L", [150, 230])
def test_cartesian_history_staged_in_lds_tracks_oracle(ctx, L):
    """128 < L <= 256: the role's L-BFGS history is staged in LDS as far as the launch has room (at L=150 six pairs when the context is alone,
    three when another stream's pair kernel needs room; at L=230 four / two, the rest read from global memory inside the
    recursion) and the pair stored a moment ago is used from registers.  40 evaluations fill the 8-pair history and turn it over: every source is used.
    Same short-horizon agreement with the oracle as at L=90 (all pairs staged) and L=300 (none)."""
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    B = 4
    m = S.make_map(L, seed=L, n_moves=150)
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    # pairs staged in LDS: everything beside the kernel's static LDS when this context is alone in the process, two pair-kernel
    # workgroups' worth less (the other stream's) when another context is alive
    assert int(ctx.info(5)) in {150: (6, 3), 230: (4, 2)}[L]
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    rng = np.random.default_rng(L)
    t0 = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.08 for _ in range(B)]).astype(np.float32)
    rows = []
    for n in (12, 40):
        r = ctx.fold_batch(B, cart_only(L), tors0=t0, max_evals=n)
        assert np.all(np.isfinite(r["xyz"]))
        for d in range(B):
            to, xo, st = O.fold(Tb, t0[d].astype(np.float64), cart_only(L), max_evals=n)
            rows.append((n, d, r["f"][d], st["f_final"], int(r["n_iters"][d]), st["n_iters"], kabsch_rmsd(r["xyz"][d].reshape(-1, 3), xo.reshape(-1, 3))))
    print(f"\nL={L}  evals decoy      f_device      f_oracle   iters dev/orc   all-atom RMSD dev-vs-orc")
    for q in rows:
        print("       %5d %4d  %12.2f  %12.2f   %3d / %3d        %.4f" % q)
    short = [q for q in rows if q[0] == 12]
    rel = np.array([abs(q[2] - q[3]) / abs(q[3]) for q in short])
    assert np.median(rel) <= 5e-3 and rel.max() <= 5e-2, rel
    assert all(q[6] < 0.2 for q in short), short
    # By 40 evaluations the float32 and float64 trajectories have separated (every build variant does, each in its own decoys:
    # tools/cart_det2.py, 16 decoys -- 2 to 11 rejected trials in total where the oracle has none): energies stay within
    # percents, the structures within half an angstrom for most, and the device accepts nearly as many steps as the oracle.
    late = [q for q in rows if q[0] == 40]
    assert max(abs(q[2] - q[3]) / abs(q[3]) for q in late) <= 5e-2 and np.median([q[6] for q in late]) < 0.5 and max(q[6] for q in late) < 1.0, late
    assert sum(q[4] for q in late) >= 0.85 * sum(q[5] for q in late), late
