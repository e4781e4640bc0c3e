"""CPU: pin the oracle (oracle/trx2_oracle.c) to vectors captured from the reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

BITS = {"dist": 1, "omega": 2, "theta": 4, "phi": 8}


@pytest.mark.parametrize("tag", ["NMR", "Xray"])
def test_tables_match_reference_gen_rst(golden_dir, tag):
    """gen_rst (utils_ros.py:6-146): same restraint set, same probabilities, same knots; table values exact for
    dist (float64 path) and within one unit of the last printed decimal for the float32 angle channels, where
    numpy's float32 log differs from libm's in the last bit for a small fraction of entries."""
    npz = np.load(os.path.join(golden_dir, f"seq_{tag}.npz"))
    gold = np.load(os.path.join(golden_dir, f"gen_rst_{tag}.npz"))
    T = O.Tables(npz["dist"], npz["omega"], npz["theta"], npz["phi"])
    kn, gen = T.knots(), T.mask(False)
    for ch, bit in BITS.items():
        a, b, p, x, yi, sc = [gold[f"{ch}_{k}"] for k in ("a", "b", "p", "x", "yi", "scale")]
        ga, gb = np.nonzero(gen & bit)
        assert np.array_equal(ga, a) and np.array_equal(gb, b), f"{ch}: restraint set differs"
        assert np.array_equal(kn[ch], x), f"{ch}: knots differ"
        assert np.array_equal(T.prob(ch)[a, b].astype(np.float64), p), f"{ch}: summed probabilities differ"
        diff = np.abs(np.rint(T.y(ch)[a, b] * sc).astype(np.int64) - yi)
        if ch == "dist":
            assert diff.max() == 0
        else:
            assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (ch, diff.max(), (diff > 0).mean())


def test_selection_counts_match_survey(golden_dir):
    """add_rst thresholds (utils_ros.py:719-723) on the NMR map: 3226 / 2562 / 5142 / 2541."""
    npz = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    sel = O.Tables(npz["dist"], npz["omega"], npz["theta"], npz["phi"]).mask(True)
    assert [int((sel & b > 0).sum()) for b in (1, 2, 4, 8)] == [3226, 2562, 5142, 2541]


@pytest.mark.parametrize("tag", ["NMR", "Xray"])
def test_relax_reselection_matches_add_rst_nogly(golden_dir, seq, tag):
    """The full-atom stage re-selects the restraints twice (folding.py:230-231,236-237): add_rst(pose, rst, 1, nres, params, True)
    at PCUT 0.15, then 0.30 -- of the restraints gen_rst generated, p >= pcut (dist), pcut + 0.5 (omega, theta), pcut + 0.6
    (phi), both residues not glycine (utils_ros.py:713-717).  The reference's generated lists (a, b, p) are the committed golden
    vectors; the predicate is applied to them here and must give the oracle's selections (runs with pair_filter 2 / 3), pair
    for pair; the example has 7 glycines, so the filter bites."""
    npz = np.load(os.path.join(golden_dir, f"seq_{tag}.npz"))
    gold = np.load(os.path.join(golden_dir, f"gen_rst_{tag}.npz"))
    T = O.Tables(npz["dist"], npz["omega"], npz["theta"], npz["phi"], seq=seq)
    gly = np.array([c == "G" for c in seq])
    assert gly.sum() == 7
    add = {"dist": 0.0, "omega": 0.5, "theta": 0.5, "phi": 0.6}
    for k, pcut in enumerate((0.15, 0.30)):
        sel = T.relax_selection(k)
        n_sel = []
        for ch, bit in BITS.items():
            a, b, p = gold[f"{ch}_a"], gold[f"{ch}_b"], gold[f"{ch}_p"]
            keep = (np.abs(a - b) >= 1) & (np.abs(a - b) < len(seq)) & ~gly[a] & ~gly[b] & (p >= pcut + add[ch])
            want = np.zeros((len(seq), len(seq)), bool)
            want[a[keep], b[keep]] = True
            assert np.array_equal((sel & bit) > 0, want), (tag, pcut, ch)
            n_sel.append(int(keep.sum()))
        base = T.mask(True)
        assert np.all((sel & ~base) == 0)          # a subset of the PCUT 0.05 selection
        print(tag, pcut, "selected dist / omega / theta / phi:", n_sel, "of", [int((base & b_ > 0).sum()) for b_ in (1, 2, 4, 8)])
        assert 0 < n_sel[0] < int((base & 1 > 0).sum())


def test_no_orient_builds_dist_only(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "gen_rst_noorient.json")))
    npz = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    T = O.Tables(npz["dist"])
    assert not T.use_orient and g["NMR"]["channels"] == ["dist"]
    assert int((T.mask(False) & 1 > 0).sum()) == g["NMR"]["n_dist"]


def test_geometry_matches_reference_get_neighbors(golden_dir, seq):
    """dihedral / angle conventions vs get_neighbors (utils_trX2dy/utils.py:125-182) on decoy conf_2_1."""
    dec = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    fb = np.load(os.path.join(golden_dir, "feedback_NMR.npz"))
    X = dec["conf_2_1"].astype(np.float64)
    N, CA, C = X[:, 0], X[:, 1], X[:, 2]
    b, c = CA - N, C - CA
    vCB = -0.58273431 * np.cross(b, c) + 0.56802827 * b - 0.54067466 * c + CA
    CB = np.where(np.array([s == "G" for s in seq])[:, None], vCB, X[:, 4])
    worst = np.zeros(4)
    for i in range(0, 90, 5):
        for j in range(90):
            if i == j or fb["dist6d"][i, j] == 0:
                continue
            got = (np.linalg.norm(CB[i] - CB[j]), O.dihedral(CA[i], CB[i], CB[j], CA[j]),
                   O.dihedral(N[i], CA[i], CB[i], CB[j]), O.angle(CA[i], CB[i], CB[j]))
            ref = (fb["dist6d"][i, j], fb["omega6d"][i, j], fb["theta6d"][i, j], fb["phi6d"][i, j])
            worst = np.maximum(worst, np.abs(np.array(got) - np.array(ref)))
    assert worst.max() < 2e-5, worst  # the reference computed these in float32


def test_random_start_table(golden_dir):
    """random_dihedral thresholds (utils_ros.py:674-696): replay the reference's draws through the same cut table."""
    import random
    rd = json.load(open(os.path.join(golden_dir, "random_dihedral.json")))
    cum = [0.135, 0.29, 0.363, 0.485, 0.982, 2.0]
    tab = [(-140, 153), (-72, 145), (-122, 117), (-82, -14), (-61, -41), (57, 39)]
    for s, draws in rd.items():
        random.seed(int(s))
        for want in draws:
            r = random.random()
            k = next(i for i, c in enumerate(cum) if r <= c)
            assert list(tab[k]) == want
    t = O.random_torsions(90, 7, 3)
    assert np.allclose(t[-1], np.radians([180, 180, 180])) and np.allclose(t[:, 2], np.pi)
    assert {tuple(np.round(np.degrees(v[:2])).astype(int)) for v in t[:-1]} <= set(tab)


def test_gradients_by_finite_differences(golden_dir):
    rng = np.random.default_rng(0)
    P = rng.normal(size=(4, 3)) * 2
    for fn, n in ((O.dihedral, 4), (O.angle, 3)):
        _, g = fn(*P[:n], grad=True)
        fd = np.zeros((n, 3))
        for k in range(n):
            for c in range(3):
                Pp, Pm = P[:n].copy(), P[:n].copy()
                Pp[k, c] += 1e-6; Pm[k, c] -= 1e-6
                fd[k, c] = (fn(*Pp) - fn(*Pm)) / 2e-6
        assert np.abs(fd - g).max() < 1e-7
    npz = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    T = O.Tables(npz["dist"], npz["omega"], npz["theta"], npz["phi"])
    t0 = O.random_torsions(90, 1, 0) + rng.normal(size=(90, 3)) * 0.05
    w = np.array([5, 4, 4, 1, 1, 0.5, 0, 0.0])
    _, _, g, _ = O.evaluate(T, t0, w)
    for i, k in [(5, 0), (5, 1), (5, 2), (40, 0), (40, 1), (40, 2), (88, 1), (0, 1), (89, 0)]:
        tp, tm = t0.copy(), t0.copy()
        tp[i, k] += 1e-6; tm[i, k] -= 1e-6
        fd = (O.evaluate(T, tp, w, grad=False)[0] - O.evaluate(T, tm, w, grad=False)[0]) / 2e-6
        assert abs(fd - g[i, k]) <= 1e-5 * (1 + abs(fd)), (i, k, fd, g[i, k])


def test_nerf_reproduces_torsions_and_ideal_geometry():
    rng = np.random.default_rng(3)
    L = 40
    t = rng.uniform(-np.pi, np.pi, size=(L, 3))
    xyz = O.nerf(t)
    w = lambda x: (x + np.pi) % (2 * np.pi) - np.pi
    for i in range(L - 1):
        assert abs(w(O.dihedral(xyz[i, 0], xyz[i, 1], xyz[i, 2], xyz[i + 1, 0]) - t[i, 1])) < 1e-9
        assert abs(w(O.dihedral(xyz[i, 1], xyz[i, 2], xyz[i + 1, 0], xyz[i + 1, 1]) - t[i, 2])) < 1e-9
        assert abs(w(O.dihedral(xyz[i, 2], xyz[i + 1, 0], xyz[i + 1, 1], xyz[i + 1, 2]) - t[i + 1, 0])) < 1e-9
    assert np.allclose(np.linalg.norm(xyz[:, 1] - xyz[:, 0], axis=-1), 1.458)
    assert np.allclose(np.linalg.norm(xyz[1:, 0] - xyz[:-1, 2], axis=-1), 1.334)


def _decoy_with_virtual_gly_cb(golden_dir, name):
    X = np.load(os.path.join(golden_dir, "ref_decoys.npz"))[name].astype(np.float64)
    b, c = X[:, 1] - X[:, 0], X[:, 2] - X[:, 1]
    X[:, 4] = np.where(np.isnan(X[:, 4]), -0.58273431 * np.cross(b, c) + 0.56802827 * b - 0.54067466 * c + X[:, 1], X[:, 4])
    return X


@pytest.mark.parametrize("name", ["conf_2_1", "conf_1_1"])
def test_internal_coordinates_round_trip_on_reference_decoys(golden_dir, name):
    """coordinates -> (torsions, per-residue bond geometry) -> coordinates, on real non-ideal structures (conf_1_1 carries
    a twisted peptide): every interatomic distance is reproduced.  This is what lets torsion-space runs continue from a
    Cartesian-minimised structure without snapping it back to ideal geometry."""
    X = _decoy_with_virtual_gly_cb(golden_dir, name)
    t, g = O.extract_internal(X)
    Y = O.nerf_geom(t, g)
    D = lambda Z: np.linalg.norm(Z.reshape(-1, 1, 3) - Z.reshape(1, -1, 3), axis=-1)
    assert np.abs(D(X) - D(Y)).max() < 1e-9
    assert 1.5 < np.degrees(g[:, 3]).std() < 4.0  # N-CA-C spread of a real structure, not the ideal constant


def test_cartesian_evaluation_gradient_and_consistency(golden_dir):
    """sf_cart in Cartesian space (folding.py:83-84,100-102): analytic gradient vs finite differences for the whole score and
    for the new pieces alone; at ideal geometry it equals the torsion-space evaluation and the bonded term vanishes."""
    m = np.load(os.path.join(golden_dir, "seq_NMR.npz"))
    T = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    rng = np.random.default_rng(2)
    xyz = O.nerf(O.random_torsions(90, 3, 0) + rng.normal(size=(90, 3)) * 0.1) + rng.normal(size=(90, 5, 3)) * 0.05
    comps = [(0, 1, 1), (5, 3, 0), (40, 0, 2), (40, 2, 1), (40, 4, 0), (89, 0, 1), (89, 3, 2)]
    for w in ([5, 4, 4, 0.5, 1, 0.5, 0.1, 0], [0, 0, 0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0, 0]):
        w = np.array(w, float)
        _, _, gx = O.eval_cart(T, xyz, w)
        for (i, a, c) in comps:
            xp, xm = xyz.copy(), xyz.copy()
            xp[i, a, c] += 1e-6; xm[i, a, c] -= 1e-6
            fd = (O.eval_cart(T, xp, w, grad=False)[0] - O.eval_cart(T, xm, w, grad=False)[0]) / 2e-6
            assert abs(fd - gx[i, a, c]) <= 2e-5 * (1 + abs(fd)), (w, i, a, c, fd, gx[i, a, c])
    w = np.array([5, 4, 4, 1, 1, 0.5, 0.3, 0.0])
    ft, _, _, x1 = O.evaluate(T, O.random_torsions(90, 3, 1), w)
    fc, ec, _ = O.eval_cart(T, x1, w)
    assert abs(ft - fc) < 1e-6 * abs(ft) and ec[7] < 1e-12


def test_cartesian_stage_keeps_reference_like_geometry(golden_dir):
    """one decoy through the protocol WITH the Cartesian run: bond and angle spread land where the reference's decoys are
    (trx2_model.h, calibration of TRX2_CART_K*), peptides stay planar, and the fold is close to the reference decoys."""
    import importlib.util
    from oracle.kabsch import kabsch_rmsd
    spec = importlib.util.spec_from_file_location("protocol", os.path.join(os.path.dirname(golden_dir), "..", "trrosettax2-dynamics_amd", "protocol.py"))
    P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
    m = np.load(os.path.join(golden_dir, "seq_Xray.npz"))
    T = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    tors, xyz, st = O.fold(T, O.random_torsions(90, 4242, 1), P.build_runs(90, 2, cartesian_stage=True))
    _, g = O.extract_internal(xyz)
    assert st["status"] == 0
    assert g[:, 1].std() < 0.02 and 1.0 < np.degrees(g[:, 3]).std() < 4.5, (g[:, 1].std(), np.degrees(g[:, 3]).std())
    dec = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    assert min(kabsch_rmsd(xyz[:, 1], dec[k][:, 1]) for k in ("conf_1_1", "conf_1_2")) < 1.5


def test_oracle_outcome_fixture_belongs_to_this_model(golden_dir):
    """tests/golden/oracle_outcomes.npz (the oracle's side of tests/test_gpu_outcome_vs_oracle.py, 10 CPU-minutes to make) carries a digest of
    include/trx2_model.h, oracle/trx2_oracle.c and protocol.py: a change to any of them needs `python tests/golden/make_oracle_outcomes.py`."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_oracle_outcomes", os.path.join(root, "tests", "golden", "make_oracle_outcomes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fx = np.load(os.path.join(golden_dir, "oracle_outcomes.npz"))
    assert str(fx["digest"]) == mod.model_digest(), "stale fixture: run `python tests/golden/make_oracle_outcomes.py`"
    for key in ("NMR_initial", "Xray_initial", "NMR_stage1", "NMR_stage2", "Xray_stage1", "Xray_stage2", "NMR_initial_nofastrelax", "Xray_initial_nofastrelax"):
        assert fx[key + "_f"].shape == (int(fx["n"]),) and np.all(np.isfinite(fx[key + "_rmsd"])), key
