"""GPU: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (device arithmetic is float32, the oracle float64):
  tables   : masks / summed probabilities / knots bit-identical; y identical after the float32 cast except where the
             device's f64 log differs from libm's in the last ulp AND that flips the 3rd/5th printed decimal
             (<= 1 unit, < 0.1 % of entries); y'' to 1e-5 relative.
  NeRF     : 1e-3 A on coordinates of magnitude <= ~150 A (float32 eps 6e-8 x 90 composed frames).
  energies : 1e-4 relative + 0.05 absolute per term; gradient: 5e-3 of the largest component.
  fold     : outcome level -- statuses, energy depth and C-alpha RMSD to the reference's PyRosetta decoys.
"""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd

T = importlib.import_module("trrosettax2-dynamics_amd")
SF = np.array(T.protocol.SF, np.float64)
TERMS = [0, 1, 2, 3, 4, 5, 6, 8]   # every term of a torsion-space evaluation (7 = cart_bonded: Cartesian runs only)


@pytest.fixture(scope="module")
def maps(golden_dir):
    return {t: np.load(os.path.join(golden_dir, f"seq_{t}.npz")) for t in ("NMR", "Xray")}


@pytest.fixture(scope="module")
def ctx():
    c = T.Context(0)
    yield c
    c.close()


def start_torsions(B, L, seed, noise=0.05):
    rng = np.random.default_rng(seed)
    return np.stack([O.random_torsions(L, seed, d) + rng.normal(size=(L, 3)) * noise for d in range(B)])


@pytest.mark.parametrize("tag", ["NMR", "Xray"])
def test_tables_match_oracle(ctx, maps, seq, tag):
    m = maps[tag]
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    gen, sel, kn = Tb.mask(False), Tb.mask(True), Tb.knots()
    for ch, bit in (("dist", 1), ("omega", 2), ("theta", 4), ("phi", 8)):
        d = ctx.get_tables(ch)
        assert np.array_equal(d["gen"], gen) and np.array_equal(d["sel"], sel)
        assert np.array_equal(d["prob"], Tb.prob(ch)), ch
        assert np.array_equal(d["knots"], kn[ch].astype(np.float32)), ch
        rows = (gen & bit) > 0
        scale = 1e5 if ch == "omega" else 1e3
        dy = np.abs(np.rint(d["y"][rows].astype(np.float64) * scale) - np.rint(Tb.y(ch)[rows] * scale))
        assert dy.max() <= 1 and (dy > 0).mean() < 1e-3, (ch, dy.max(), (dy > 0).mean())
        exact = dy.max(axis=-1) == 0  # compare y'' only on rows whose y agree exactly
        assert np.allclose(d["y2"][rows][exact], Tb.y2(ch)[rows][exact], rtol=1e-5, atol=1e-5), ch


@pytest.mark.parametrize("B", [1, 2, 5, 64, 70])
def test_eval_matches_oracle(ctx, maps, seq, B):
    m = maps["NMR"]
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    tors = start_torsions(B, 90, 11 + B)
    f, e, g, xyz = ctx.eval_batch(tors, SF)
    for d in sorted(set([0, B // 2, B - 1])):
        fo, eo, go, xo = O.evaluate(Tb, tors[d].astype(np.float32).astype(np.float64), SF)
        assert np.abs(xyz[d] - xo).max() < 1e-3, ("xyz", d, np.abs(xyz[d] - xo).max())
        assert np.all(np.abs(e[d][TERMS] - eo[TERMS]) <= 1e-4 * np.abs(eo[TERMS]) + 0.05), ("terms", d, e[d], eo)
        assert abs(f[d] - fo) <= 1e-4 * abs(fo) + 0.5, ("total", d, f[d], fo)
        assert np.abs(g[d] - go).max() <= 5e-3 * np.abs(go).max(), ("grad", d, np.abs(g[d] - go).max(), np.abs(go).max())


def test_eval_no_orient_and_separation_window(ctx, maps, seq):
    """--no-orient (dist only, arguments.py:16) and the add_rst separation window (utils_ros.py:719)."""
    m = maps["Xray"]
    ctx.set_map(m["dist"], seq=seq)
    Tb = O.Tables(m["dist"], seq=seq)
    tors = start_torsions(3, 90, 5)
    for lo, hi in ((1, 90), (3, 24), (12, 90)):
        f, e, g, _ = ctx.eval_batch(tors, SF, lo, hi)
        fo, eo, go, _ = O.evaluate(Tb, tors[1].astype(np.float32).astype(np.float64), SF, lo, hi)
        assert e[1, 1] == 0 and e[1, 2] == 0 and e[1, 3] == 0
        assert abs(e[1, 0] - eo[0]) <= 1e-4 * abs(eo[0]) + 0.05, (lo, hi, e[1, 0], eo[0])
        assert np.abs(g[1] - go).max() <= 5e-3 * np.abs(go).max()


@pytest.mark.parametrize("term,w", [("rama", [0, 0, 0, 0, 1, 0, 0, 0]), ("omega", [0, 0, 0, 0, 0, 1, 0, 0])])
def test_backbone_terms_alone_match_oracle(ctx, maps, seq, term, w):
    """The fitted rama and omega terms (include/trx2_model.h TRX2_RAMA_FIT_* / TRX2_OMEGA_FIT) with every other weight at zero, so that
    their gradient is not hidden under the restraints' (whose largest component sets the tolerance of the whole-function checks).  Torsions
    drawn over the WHOLE (phi, psi) plane and omega up to 40 degrees off planar; the example sequence has all four residue classes
    (general, 7 glycines, a proline, a residue before it).  Torsion role here; the Cartesian role's chain rule is covered by
    tests/test_gpu_cartesian.py's tracking against the oracle."""
    m = maps["NMR"]
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    rng = np.random.default_rng(77)
    B = 8
    tors = np.stack([np.stack([rng.uniform(-np.pi, np.pi, 90), rng.uniform(-np.pi, np.pi, 90), np.pi + rng.normal(size=90) * 0.25], 1) for _ in range(B)]).astype(np.float32)
    w = np.array(w, np.float64)
    f, e, g, _ = ctx.eval_batch(tors, w)
    k = 5 if term == "rama" else 6
    for d in range(B):
        fo, eo, go, _ = O.evaluate(Tb, tors[d].astype(np.float64), w)
        assert abs(e[d][k] - eo[k]) <= 2e-5 * abs(eo[k]) + 2e-3, (term, d, e[d][k], eo[k])
        assert abs(f[d] - fo) <= 2e-5 * abs(fo) + 2e-3, (term, d, f[d], fo)
        assert np.abs(g[d] - go).max() <= 2e-4 * np.abs(go).max() + 1e-4, (term, d, np.abs(g[d] - go).max(), np.abs(go).max())


def test_eval_is_bitwise_reproducible(ctx, maps, seq):
    """no atomics anywhere on the path: the same inputs give the same bits."""
    m = maps["NMR"]
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    tors = start_torsions(64, 90, 3)
    a, b = ctx.eval_batch(tors, SF), ctx.eval_batch(tors, SF)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def test_fold_reaches_reference_decoys(ctx, maps, seq, golden_dir):
    """TORSION-SPACE protocol (the Cartesian run replaced by its torsion-space stand-in; full protocol: test_gpu_cartesian.py).
    Outcome parity: fold the committed NMR map and compare with the reference's PyRosetta decoys
    conf_2_1 / conf_2_2 (NMR/initial0,1).  Their own mutual RMSD is 0.86 A (SURVEY.md 8c's criterion: median <= 0.5 + 0.86 A).
    Asserted is measured + margin instead: 0.92 A median over 1024 decoys of this torsion-only protocol
    (quartiles 0.78-1.09, profiles/r02_outcome_parity_n1024.txt); with 128 decoys the median's sampling error is ~0.04 A:
    <= 1.05 A."""
    m = maps["NMR"]
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    dec = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    B = 128
    r = ctx.fold_batch(B, T.protocol.build_runs(90, 2, cartesian_stage=False), seed=2024)  # torsion space only: the exact
    assert np.all(r["status"] == 0), r["status"]                                             # ideal-geometry invariants below
    assert np.all(np.isfinite(r["xyz"])) and np.all(np.isfinite(r["f"]))
    x = r["xyz"]
    ca = x[:, :, 1]
    cc = np.linalg.norm(ca[:, 1:] - ca[:, :-1], axis=-1)
    dw = np.degrees(np.abs((r["tors"][:, :-1, 2] % (2 * np.pi)) - np.pi))
    best = np.array([min(kabsch_rmsd(ca[i], dec[k][:, 1]) for k in ("conf_2_1", "conf_2_2")) for i in range(B)])
    print("\nfold: evals", r["n_evals"].min(), r["n_evals"].max(), "iters", r["n_iters"].min(), r["n_iters"].max(),
          "seconds", round(r["seconds"], 3), "launches", r["launches"], "\n rmsd", np.round(np.sort(best), 2),
          "\n CA-CA min", np.round(cc.min(1), 2), "\n |dw|max", np.round(dw.max(1), 0),
          "\n dist", np.round(r["e_terms"][:, 0], 0), "\n theta", np.round(r["e_terms"][:, 2], 0))
    # torsion-space moves keep the ideal bond lengths exactly (trx2_model.h), whatever omega does
    for (i, j, shift, want) in ((0, 1, 0, 1.458), (1, 2, 0, 1.524), (2, 0, 1, 1.334), (2, 3, 0, 1.232)):
        p, q = (x[:, :-1, i], x[:, 1:, j]) if shift else (x[:, :, i], x[:, :, j])
        assert np.abs(np.linalg.norm(p - q, axis=-1) - want).max() < 2e-3, (i, j, want)
    # the returned coordinates are exactly the backbone of the returned torsions (device NeRF vs oracle NeRF)
    for i in (0, 1, B // 2, B - 1):
        xo = O.nerf(r["tors"][i].astype(np.float64))
        assert np.abs(x[i] - xo).max() < 1e-3, (i, np.abs(x[i] - xo).max())
    # CA(i)-CA(i+1) is a function of omega alone; its range is [cis, trans], both COMPUTED from the model's ideal
    # geometry (2.7789 / 3.8093 A) rather than typed in.  It is NOT "3.8 everywhere": the reference's own conf_1_1 has
    # one at 3.19 A, and the shared energy model lets a few peptides twist (known deviations).
    def ca_ca(omega_deg):
        t = np.zeros((2, 3)); t[0, 2] = np.radians(omega_deg)
        xx = O.nerf(t)
        return np.linalg.norm(xx[1, 1] - xx[0, 1])
    cis, trans = ca_ca(0.0), ca_ca(180.0)
    assert cis - 2e-3 <= cc.min() and cc.max() <= trans + 2e-3, (cc.min(), cc.max(), cis, trans)
    # The median hides gross failures, so count them: with all channels on, the oracle itself lands in the MIRROR-image
    # topology (RMSD ~12 A, mirror ~3 A) for roughly 1 start in 6-20 -- report it, and bound it loosely.
    n_gross = int((best > 3.0).sum())
    print(" decoys with RMSD > 3 A (mirror-trapped / misfolded):", n_gross, "of", B)
    assert n_gross <= 0.09 * B, np.sort(best)          # measured 3.6 % of 1024 (sd of the count at B=128: 2.1 decoys)
    assert np.median(best) <= 1.05, np.sort(best)[::8]
    # depth of optimisation: restraint energies of the reference decoys under the same tables are
    # dist -19679/-19689, theta -28121/-28288
    assert np.median(r["e_terms"][:, 0]) < -19000 and np.median(r["e_terms"][:, 2]) < -27000, r["e_terms"][:, :4].mean(0)


@pytest.mark.parametrize("L,B", [(60, 3), (60, 32), (300, 4), (520, 2)])
def test_eval_other_widths_and_long_chains(ctx, L, B):
    """decoy-group widths 4 and 32 (B=3, 32) and the chain kernel's 2- and 4-residues-per-thread paths (L=300, 520),
    on synthetic maps (the reference ships L=90 only)."""
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    m = S.make_map(L, seed=L, n_moves=150)
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    rng = np.random.default_rng(L + B)
    tors = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.08 for _ in range(B)])  # near the compact target
    f, e, g, xyz = ctx.eval_batch(tors, SF)
    for d in sorted({0, B - 1}):
        fo, eo, go, xo = O.evaluate(Tb, tors[d].astype(np.float32).astype(np.float64), SF)
        assert np.abs(xyz[d] - xo).max() < 3e-3, ("xyz", L, d, np.abs(xyz[d] - xo).max())
        assert np.all(np.abs(e[d][TERMS] - eo[TERMS]) <= 2e-4 * np.abs(eo[TERMS]) + 0.1), ("terms", L, d, e[d], eo)
        assert np.abs(g[d] - go).max() <= 1e-2 * np.abs(go).max(), ("grad", L, d, np.abs(g[d] - go).max(), np.abs(go).max())


@pytest.mark.parametrize("L,B,orient", [(4, 1, True), (5, 2, True), (17, 5, True), (64, 64, False), (65, 33, True), (129, 9, True),
                                         (200, 70, False)])
def test_eval_edge_shapes(ctx, L, B, orient):
    """The smallest chain the ABI accepts (4 residues), chain lengths around the wave size, a partial second decoy group
    (B = 70: 64 + 6 live lanes), distance-only maps with full groups (the pair kernel's shared repulsion walk, with dead lanes
    in the second group at B = 70), and a 128-thread / 256-thread step boundary (L = 129).  Energy terms, gradient and
    coordinates against the oracle."""
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    m = S.make_map(L, seed=1000 + L, n_moves=60)
    ang = [m["omega"], m["theta"], m["phi"]] if orient else []
    ctx.set_map(m["dist"], *ang, seq=m["seq"])
    Tb = O.Tables(m["dist"], *(ang if orient else [None, None, None]))
    rng = np.random.default_rng(L * 100 + B)
    tors = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.1 for _ in range(B)])
    f, e, g, xyz = ctx.eval_batch(tors, SF)
    assert np.all(np.isfinite(f)) and np.all(np.isfinite(g))
    for d in sorted({0, B // 2, B - 1}):
        fo, eo, go, xo = O.evaluate(Tb, tors[d].astype(np.float32).astype(np.float64), SF)
        assert np.abs(xyz[d] - xo).max() < 2e-3, ("xyz", L, B, d)
        assert np.all(np.abs(e[d][TERMS] - eo[TERMS]) <= 2e-4 * np.abs(eo[TERMS]) + 0.1), ("terms", L, B, d, e[d], eo)
        assert np.abs(g[d] - go).max() <= 1e-2 * max(np.abs(go).max(), 1.0), ("grad", L, B, d, np.abs(g[d] - go).max(), np.abs(go).max())
    if L <= 64:   # a short fold must run through every stage and every role at these sizes too
        r = ctx.fold_batch(B, T.protocol.build_runs(L, 2), seed=3, max_evals=300)
        assert np.all(np.isfinite(r["xyz"])) and np.all(np.isfinite(r["f"])) and np.all((r["status"] == 0) | (r["n_evals"] == 300))
        again = ctx.fold_batch(B, T.protocol.build_runs(L, 2), seed=3, max_evals=300)
        assert np.array_equal(r["xyz"], again["xyz"])                            # and reproducibly


def test_maximum_chain_length(ctx):
    """L = 1024, the largest chain the ABI accepts (include/trx2fold.h, trx2_set_map; the pair kernel packs a row index into ten bits): tables,
    energy terms, gradient and coordinates against the oracle with all four channels, then a short torsion-space minimisation that must accept
    steps and stay finite; one residue more is refused with a message, not folded."""
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    L, B = 1024, 2
    m = S.make_map(L, seed=2024, n_moves=60)
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    rng = np.random.default_rng(L)
    tors = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.05 for _ in range(B)])
    f, e, g, xyz = ctx.eval_batch(tors, SF)
    for d in range(B):
        fo, eo, go, xo = O.evaluate(Tb, tors[d].astype(np.float32).astype(np.float64), SF)
        assert np.abs(xyz[d] - xo).max() < 8e-3, ("xyz", d, np.abs(xyz[d] - xo).max())      # float32 NeRF over 1024 chained frames (3e-3 at L = 520)
        assert np.all(np.abs(e[d][TERMS] - eo[TERMS]) <= 3e-4 * np.abs(eo[TERMS]) + 0.2), ("terms", d, e[d], eo)
        assert np.abs(g[d] - go).max() <= 1e-2 * np.abs(go).max(), ("grad", d, np.abs(g[d] - go).max(), np.abs(go).max())
    runs = T.protocol.build_runs(L, 2)
    assert not any(q["cartesian"] for q in runs)
    t0 = tors.astype(np.float32)
    r = ctx.fold_batch(B, runs[5:], tors0=t0, max_evals=25)
    assert np.all(np.isfinite(r["xyz"])) and np.all(np.isfinite(r["f"])) and np.all(r["n_evals"] == 25) and np.all(r["n_iters"] >= 10), (r["n_evals"], r["n_iters"])
    too_long = np.zeros((L + 1, L + 1, m["dist"].shape[2]), np.float32)
    with pytest.raises(Exception, match="1024"):
        ctx.set_map(too_long, seq="A" * (L + 1))
    ctx.set_map(m["dist"], seq=m["seq"])          # the context is still usable after the refusal
    assert np.all(np.isfinite(ctx.eval_batch(tors[:1], SF)[0]))


@pytest.mark.parametrize("orient", [True, False], ids=["all-channels", "dist-only"])
def test_a_map_that_selects_no_restraint(ctx, orient):
    """A distogram that puts every pair beyond the last bin (P(no contact) = 1): gen_rst selects nothing (utils_ros.py:54-144), every row of the
    restraint lists is EMPTY, and what is left is the backbone model alone.  The oracle agrees on the counts (0) and on every term; a fold runs
    through the whole default protocol on clash, rama, omega and hydrogen-bond terms only and is reproducible."""
    L, B = 70, 5
    dist = np.zeros((L, L, 37), np.float32); dist[..., 0] = 1.0
    ang = []
    if orient:
        om = np.zeros((L, L, 25), np.float32); om[..., 0] = 1.0
        ph = np.zeros((L, L, 13), np.float32); ph[..., 0] = 1.0
        ang = [om, om.copy(), ph]
    seq = ("ACDEFGHIKLMNPQRSTVWY" * 4)[:L]
    ctx.set_map(dist, *ang, seq=seq)
    Tb = O.Tables(dist, *(ang if orient else [None, None, None]), seq=seq)
    rng = np.random.default_rng(11)
    tors = np.stack([O.random_torsions(L, 5, d) + rng.normal(size=(L, 3)) * 0.05 for d in range(B)])
    f, e, g, xyz = ctx.eval_batch(tors, SF)
    for d in (0, B - 1):
        fo, eo, go, xo = O.evaluate(Tb, tors[d].astype(np.float32).astype(np.float64), SF)
        assert abs(eo[0]) + abs(eo[1]) + abs(eo[2]) + abs(eo[3]) == 0.0 and np.all(e[d][:4] == 0.0), (eo[:4], e[d][:4])      # no restraint term at all
        assert np.abs(xyz[d] - xo).max() < 2e-3
        bb = [4, 5, 6, 8]      # clash, rama, omega, hydrogen bonds
        assert np.all(np.abs(e[d][bb] - eo[bb]) <= 2e-4 * np.abs(eo[bb]) + 0.05), (e[d], eo)
        assert np.abs(g[d] - go).max() <= 1e-2 * max(np.abs(go).max(), 1.0)
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    r = ctx.fold_batch(B, runs, seed=9)
    assert np.all(r["status"] == 0) and np.all(np.isfinite(r["xyz"])) and np.all(np.isfinite(r["f"]))
    again = ctx.fold_batch(B, runs, seed=9)
    assert np.array_equal(r["xyz"], again["xyz"]) and np.array_equal(r["n_evals"], again["n_evals"])


def test_minimiser_on_a_chain_longer_than_512(ctx):
    """L = 520: four residues per thread in the step kernel, history read from global memory (the LDS-staged history is for
    L <= 256), more than 64 visits per wave in the pair kernel (three blocks of its contact-bit walk).  Torsion-space
    protocol (the Cartesian kernel stops at 512 residues); same start and evaluation budget as the oracle.
    tests/diag/scratch_r01_r04/horizon_check.py on this start: agreement to 1e-7 .. 1e-6 for the first evaluations, identical iteration counts up
    to 8, then the float32 and float64 trajectories separate (1e-5 at 3 evaluations for one decoy, 3e-4 at 4, percent level
    from 8 on) -- so the comparison is pinned at 4 evaluations and the longer run is a sanity check."""
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    L, B = 520, 3
    m = S.make_map(L, seed=L, n_moves=150)
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"])
    runs = T.protocol.build_runs(L, 2)
    assert not any(q["cartesian"] for q in runs)
    rng = np.random.default_rng(7)
    t0 = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.15 for _ in range(B)]).astype(np.float32)
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=4)
    rel, same = [], 0
    for d in range(B):
        to, xo, st = O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=4)
        rel.append(abs(r["f"][d] - st["f_final"]) / abs(st["f_final"]))
        same += int(r["n_iters"][d] == st["n_iters"])
    print("\nL=520, 4 evaluations: relative energy difference device vs oracle %s, identical iteration counts %d of %d" % (np.round(rel, 7), same, B))
    # (round 5: with the fitted rama / omega terms one decoy of the three flips a line-search decision inside these 4 evaluations -- 6.9e-2 --
    # while the other two stay at 8e-7 / 9e-7; the pin is on the median and on all decoys but one, as in the 20-evaluation checks)
    assert np.sort(rel)[B - 2] <= 2e-3 and np.median(rel) <= 2e-5 and max(rel) <= 0.15 and same == B, (rel, same)
    r = ctx.fold_batch(B, runs, tors0=t0, max_evals=30)
    # (a sanity check, not a pin: accepted iterations within 30 evaluations from this perturbed start -- 12 / 24 / 17 with round 6's rama constants)
    assert np.all(np.isfinite(r["xyz"])) and np.all(r["n_evals"] == 30) and np.all(r["n_iters"] >= 10)


def test_minimiser_tracks_oracle_over_short_horizons(ctx, maps, seq):
    """Same start, same protocol, same evaluation budget as the oracle.  Outcome tests cannot see a minimiser that
    converges wastefully; this can.  Statistic: accepted iterations summed over 12 decoys, device / oracle.
    Calibration (profiles/README.md, tests/diag/scratch_r01_r04/traj_stats.py): a healthy minimiser gives 0.99-1.00 at 20 evaluations (11-12 of
    12 per-decoy counts identical), 0.89-0.92 at 80 and 0.87-0.98 at 160 over three seeds -- the float32 device is 5-10 %
    less efficient per evaluation than the float64 oracle once the trajectories have separated.  An experimental
    Gram-matrix L-BFGS with float-accumulated dot products lost its directions to cancellation and scored 0.69 at 80."""
    m = maps["NMR"]
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    runs = T.protocol.build_runs(90, 2)
    B = 12
    t0 = np.stack([O.random_torsions(90, 99, d) for d in range(B)]).astype(np.float32)
    ratio, same = {}, {}
    for n in (20, 80, 160):
        r = ctx.fold_batch(B, runs, tors0=t0, max_evals=n)
        orc = [O.fold(Tb, t0[d].astype(np.float64), runs, max_evals=n)[2] for d in range(B)]
        oi = np.array([o["n_iters"] for o in orc])
        ratio[n], same[n] = r["n_iters"].sum() / oi.sum(), int((oi == r["n_iters"]).sum())
        if n == 20:  # before most line-search decisions have flipped.  Calibration (tests/diag/scratch_r01_r04/traj20.py, 3 seeds x 12 decoys):
            # relative energy difference median 4e-5 .. 3e-4, 11-12 of 12 below 5e-3; identical iteration counts 10-12 of 12.
            # A single flipped line-search decision in this steep phase moves one decoy by up to 10 % (seen once in 36), so the
            # bound is on the median and on all decoys but one.
            rel = np.array([abs(r["f"][d] - orc[d]["f_final"]) / abs(orc[d]["f_final"]) for d in range(B)])
            assert np.median(rel) <= 1e-3 and np.sort(rel)[B - 2] <= 2e-2, np.sort(rel)
    print("\naccepted iterations, device / oracle:", {k: round(float(v), 2) for k, v in ratio.items()}, "identical counts:", same)
    assert ratio[20] >= 0.95 and same[20] >= B - 3, (ratio, same)   # measured 0.98-1.01 and 10-11
    assert ratio[80] >= 0.80, ratio                                  # measured 0.85-0.92; the broken variant scored 0.69


# The whole-protocol comparison of device and oracle (rounds 3-4: 64 decoys of the 14-run protocol, live oracle) is now
# tests/test_gpu_outcome_vs_oracle.py: the protocol that ships (35 runs) and --no-fastrelax, 256 decoys per map, all six example maps.
