"""GPU: the feedback step on the device (K7) against the host mirror feedback.py, which tests/test_host_boundary.py pins
bit for bit to the reference (SHA-256 of its full outputs).  Inputs: the reference's own decoys and distograms."""
import importlib
import os

import numpy as np
import pytest
import torch  # noqa: F401  -- before libtrx2fold.so (see test_gpu_boundary.py)

pytestmark = pytest.mark.gpu

T = importlib.import_module("trrosettax2-dynamics_amd")
FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
P = importlib.import_module("trrosettax2-dynamics_amd.pdbio")


def pdb_rounded(xyz, seq, tmp_path, name):
    """what the pipeline hands to the feedback: coordinates after the PDB round trip (%8.3f)"""
    path = str(tmp_path / name)
    P.write_pdb(path, seq, np.nan_to_num(np.asarray(xyz, np.float32)))
    return path, P.read_backbone(path)[0]


@pytest.mark.parametrize("tag,decoys", [("NMR", ("conf_2_1", "conf_2_2")), ("Xray", ("conf_1_1", "conf_1_2"))])
def test_device_feedback_equals_host_feedback(golden_dir, seq, tmp_path, tag, decoys):
    m = dict(np.load(os.path.join(golden_dir, f"seq_{tag}.npz")))
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    ctx = T.Context(0)
    try:
        cur_h = {k: m[k] for k in ("dist", "theta", "omega", "phi")}
        cur_d = dict(cur_h)
        for it, name in enumerate(decoys):                       # two chained iterations: the second one consumes `tmp`
            path, xyz = pdb_rounded(ref[name], seq, tmp_path, f"{name}.pdb")
            hb = FB.get_distribution_from_pdb(path)              # jd, jt, jo, jp
            db = ctx.feedback_bins(xyz, seq)
            diff = [int((np.asarray(a) != b).sum()) for a, b in zip(hb, db)]
            print(f"\n{tag} {name}: pairs whose bin differs from the host's (dist, theta, omega, phi): {diff} of {90 * 89}")
            # an angle within one ulp of a bin edge may fall on the other side (atan2f of glibc vs the device library)
            assert diff[0] == 0 and max(diff) <= 2, diff
            assert db[0].dtype == np.int8 and int(db[0].max()) <= 36 and int(db[2].max()) <= 24 and int(db[3].max()) <= 12
            lab_h = FB.feedback_labels(cur_h, path, 1.0, True)
            lab_d = ctx.feedback_labels(cur_d, xyz, seq, 1.0, True)
            assert sorted(lab_h) == sorted(lab_d) == ["dist", "omega", "phi", "theta", "tmp"]
            for k in lab_h:
                same = np.array_equal(lab_h[k], lab_d[k])
                rows = int((lab_h[k] != lab_d[k]).any(-1).sum())
                print(f"   {k:5s}: bitwise equal {same}; pair rows that differ {rows}; max abs diff {np.abs(lab_h[k] - lab_d[k]).max():.2e}")
                if max(diff) == 0:
                    assert same, k                               # same bins -> same arithmetic -> same bits
                else:
                    assert rows <= 2 * max(diff)
            cur_h, cur_d = lab_h, lab_d
        # the individual variants of the reference's function
        jd = np.asarray(hb[0])
        for norm, smooth in ((True, False), (False, False)):
            a = FB.process_distribution_with_pred_distribution(m["dist"], jd, norm=norm, smooth=smooth)
            assert np.array_equal(a, ctx.feedback_process(m["dist"], jd, norm=norm, smooth=smooth)), (norm, smooth)
    finally:
        ctx.close()


def test_device_feedback_glycine_missing_cb_and_errors(golden_dir, seq):
    """virtual C-beta for glycine and for residues whose CB record is absent; argument checks fail loudly"""
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    xyz = np.nan_to_num(ref["conf_1_1"].astype(np.float32)).round(3)
    s2 = "G" + seq[1:10] + "G" + seq[11:]
    xyz[20, 4] = np.nan                                          # a residue that lost its CB
    ctx = T.Context(0)
    try:
        host = FB.bin_geometry(*FB.get_neighbors(xyz, s2))       # jd, jo, jt, jp
        jd, jt, jo, jp = ctx.feedback_bins(xyz, s2)
        assert np.array_equal(host[0], jd)
        assert max(int((host[1] != jo).sum()), int((host[2] != jt).sum()), int((host[3] != jp).sum())) <= 2
        with pytest.raises(ValueError):
            ctx.feedback_bins(xyz, seq[:-1])
        with pytest.raises(ValueError):
            ctx.feedback_process(np.zeros((90, 90, 37), np.float32), np.zeros((90, 89), np.int8))
        with pytest.raises(RuntimeError):
            ctx.feedback_process(np.zeros((90, 90, 4), np.float32), np.zeros((90, 90), np.int8))   # fewer than 8 bins
        with pytest.raises(ValueError):
            T.Context.gaussian_weights(2.0)
    finally:
        ctx.close()


def test_resident_feedback_chain_equals_the_host_chain(golden_dir, tmp_path):
    """pipeline.generate_npz_and_pdb three ways -- distograms resident on the device (default), device kernels on host
    arrays, numpy -- must produce the same decoys, the same number of iterations and the same intermediate arrays."""
    PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    npz, fa = os.path.join(golden_dir, "seq_NMR.npz"), os.path.join(golden_dir, "seq.fasta")
    runs = {}
    for tag, mode in (("resident", True), ("arrays", "arrays"), ("numpy", False)):
        tmpd, pdbd = str(tmp_path / tag / "tmp"), str(tmp_path / tag / "pdb")
        last = PL.generate_npz_and_pdb("s", tmpd, pdbd, npz, fa, N=4, Nmax=4, seed=21, device_feedback=mode, write_tmp_npz=True)
        runs[tag] = (last, {f: open(os.path.join(pdbd, f)).read() for f in sorted(os.listdir(pdbd))},
                     {f: dict(np.load(os.path.join(tmpd, f))) for f in sorted(os.listdir(tmpd))})
    ref = runs["numpy"]
    assert ref[0] >= 2 and len(ref[2]) >= 2
    for tag in ("resident", "arrays"):
        last, pdbs, arrs = runs[tag]
        assert last == ref[0] and sorted(pdbs) == sorted(ref[1]) and sorted(arrs) == sorted(ref[2]), tag
        assert all(pdbs[f] == ref[1][f] for f in pdbs), tag                       # every decoy, byte for byte
        for f in arrs:
            assert sorted(arrs[f]) == sorted(ref[2][f]) == ["dist", "omega", "phi", "theta", "tmp"]
            assert all(np.array_equal(arrs[f][k], ref[2][f][k]) for k in arrs[f]), (tag, f)
    # dist-only iteration (--no-angle): only dist and tmp move
    a = PL.generate_npz_and_pdb("t", str(tmp_path / "a" / "tmp"), str(tmp_path / "a" / "pdb"), npz, fa, N=2, Nmax=3, seed=5, angle=False,
                                tta_opt="-m 2 --no-orient -r no-idp", write_tmp_npz=True)
    b = PL.generate_npz_and_pdb("t", str(tmp_path / "b" / "tmp"), str(tmp_path / "b" / "pdb"), npz, fa, N=2, Nmax=3, seed=5, angle=False,
                                tta_opt="-m 2 --no-orient -r no-idp", write_tmp_npz=True, device_feedback=False)
    assert a == b
    for f in sorted(os.listdir(tmp_path / "b" / "tmp")):
        za, zb = np.load(tmp_path / "a" / "tmp" / f), np.load(tmp_path / "b" / "tmp" / f)
        assert sorted(za.files) == sorted(zb.files) == ["dist", "tmp"] and all(np.array_equal(za[k], zb[k]) for k in za.files)
    assert open(tmp_path / "a" / "pdb" / f"t{a}.pdb").read() == open(tmp_path / "b" / "pdb" / f"t{b}.pdb").read()


def test_device_glocon_matrix_equals_host(golden_dir, seq, tmp_path):
    """trx2_glocon_matrix sums in numpy's pairwise order: bitwise the host matrix (which is pinned to the reference's)"""
    CL = importlib.import_module("trrosettax2-dynamics_amd.cluster")
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    d = tmp_path / "pdb"
    d.mkdir()
    for name in ("conf_1_1", "conf_1_2", "conf_1_3", "conf_1_4", "conf_2_1", "conf_2_2", "conf_2_3", "conf_2_4"):
        P.write_pdb(str(d / f"{name}.pdb"), seq, np.nan_to_num(ref[name].astype(np.float32)))
    host, files = CL.get_glocon_matrix(str(d))
    dev, files2 = CL.get_glocon_matrix(str(d), device=0)
    assert files == files2 and np.array_equal(host, dev), np.abs(host - dev).max()
    assert np.allclose(np.diag(dev), 0) and np.array_equal(dev, dev.T)
    out = CL.save_cluster_result(str(d), n_clusters=2, n_files=2, mode="glocon", device=0)
    assert sorted(len(v) for v in out.values()) == [4, 4]


def test_device_superposition_matrices_equal_host(golden_dir, seq, tmp_path):
    """trx2_superpose_matrix against evaluate.rmsd_common / evaluate.tm_score (the numpy statements of the TM-score program's
    numbers, pinned by the reference's committed summary.txt): all pairs of the eight example decoys, a rectangular batch against
    the natives with another normalisation length, and a long chain (four residues per lane)."""
    EV = importlib.import_module("trrosettax2-dynamics_amd.evaluate")
    CL = importlib.import_module("trrosettax2-dynamics_amd.cluster")
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    names = ("conf_1_1", "conf_1_2", "conf_1_3", "conf_1_4", "conf_2_1", "conf_2_2", "conf_2_3", "conf_2_4")
    ca = np.stack([ref[n][:, 1] for n in names]).astype(np.float32)
    ctx = T.Context(0)
    try:
        rm, tm = ctx.superpose_matrix(ca)
        for i in range(8):
            for j in range(i):
                x, y = ca[i].astype(np.float64), ca[j].astype(np.float64)
                assert abs(rm[i, j] - EV.rmsd_common(x, y)) < 1e-9 and abs(tm[i, j] - EV.tm_score(x, y)) < 1e-9, (i, j)
        assert np.array_equal(rm, rm.T) and np.array_equal(tm, tm.T) and np.allclose(np.diag(rm), 0, atol=1e-6) and np.allclose(np.diag(tm), 1.0)
        # natives x models on their common residues, normalised by the model's length (evaluate.compare)
        nat = np.nan_to_num(ref["apo"][:, 1])
        okr = np.ones(len(nat), bool)
        xa = nat[okr][None].astype(np.float32)
        xb = ca[:, okr]
        rm2, tm2 = ctx.superpose_matrix(xa, xb, l_norm=90)
        for j in range(8):
            assert abs(tm2[0, j] - EV.tm_score(xa[0].astype(np.float64), xb[j].astype(np.float64), l_norm=90)) < 1e-9
            assert abs(rm2[0, j] - EV.rmsd_common(xa[0].astype(np.float64), xb[j].astype(np.float64))) < 1e-9
        # a long chain: 200 residues = four per lane
        S = importlib.import_module("trrosettax2-dynamics_amd.synth")
        rng = np.random.default_rng(3)
        base = S.nerf_backbone(S.make_map(200, seed=200, n_moves=50)["tors"])[1]
        pts = np.stack([base + rng.normal(size=base.shape) * s for s in (0.3, 1.0, 3.0)]).astype(np.float32)
        rm3, tm3 = ctx.superpose_matrix(pts)
        assert abs(tm3[0, 2] - EV.tm_score(pts[0].astype(np.float64), pts[2].astype(np.float64))) < 1e-9
        assert abs(rm3[1, 2] - EV.rmsd_common(pts[1].astype(np.float64), pts[2].astype(np.float64))) < 1e-9
        # reliability scores of a batch (the reference ranks its initial decoys by them) against the per-file host function
        xyz = np.stack([P.as_read_from_pdb(seq, np.nan_to_num(ref[n].astype(np.float32)))[0] for n in names])
        sc = ctx.reliability_scores(xyz)
        for k, n in enumerate(names):
            path, _ = pdb_rounded(ref[n], seq, tmp_path, f"{n}.pdb")
            assert sc[k] == FB.calculate_reliability_score(path), (n, sc[k])
        assert sc[4] == pytest.approx(81 / 88) and sc[0] == pytest.approx(83 / 88)
    finally:
        ctx.close()
    # cluster.py -m tmscore / -m rmsd on the device, and evaluate.py --device reproducing the reference's summary.txt
    d = tmp_path / "pdb"; d.mkdir()
    nat_d = tmp_path / "nat"; nat_d.mkdir()
    for n in names:
        P.write_pdb(str(d / f"{n}.pdb"), seq, np.nan_to_num(ref[n].astype(np.float32)))
    for n in ("apo", "holo"):
        P.write_pdb(str(nat_d / f"{n}.pdb"), seq, np.nan_to_num(ref[n].astype(np.float32)))
    tm_d, rm_d, files = CL.get_tmscore_and_rmsd_matrix(str(d), device=0)
    tm_h, rm_h, _ = CL.get_tmscore_and_rmsd_matrix(str(d))
    assert np.abs(tm_d - tm_h).max() < 1e-9 and np.abs(rm_d - rm_h).max() < 1e-9
    out = CL.save_cluster_result(str(d), n_clusters=2, n_files=2, mode="tmscore", device=0)
    assert sorted(len(v) for v in out.values()) == [4, 4]
    host = EV.run_score(str(nat_d), str(d))
    dev = EV.run_score(str(nat_d), str(d), device=0, save_summary=True, save_dir=str(tmp_path / "sum"))
    assert host == dev
    # --align (TM-score's -seq): correspondence by sequence alignment, superposition of the aligned pairs on the device
    assert EV.run_score(str(nat_d), str(d), align=True, device=0) == host


def test_superposition_of_more_than_256_structures_and_nonfinite_input(golden_dir):
    """(pair) used to sit on gridDim.y, so 257 structures against themselves exceeded its 65 535 blocks; now the
    upper triangle is enumerated on gridDim.x.  300 noisy copies of a reference decoy (L = 40 residues of it: 45 150 pairs):
    sampled entries against the host functions, symmetry, diagonal; and non-finite coordinates are refused at the C ABI
    (the TM-score search's cutoff-widening loop never ended on a NaN)."""
    EV = importlib.import_module("trrosettax2-dynamics_amd.evaluate")
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    base = ref["conf_1_1"][:40, 1].astype(np.float64)
    rng = np.random.default_rng(11)
    n = 300
    pts = (base[None] + rng.normal(size=(n, 40, 3)) * rng.uniform(0.1, 2.0, size=(n, 1, 1))).astype(np.float32)
    ctx = T.Context(0)
    try:
        rm, tm = ctx.superpose_matrix(pts)
        assert rm.shape == (n, n) and np.array_equal(rm, rm.T) and np.array_equal(tm, tm.T)
        assert np.allclose(np.diag(rm), 0, atol=1e-6) and np.allclose(np.diag(tm), 1.0)
        for i, j in [(0, 1), (0, 299), (255, 256), (256, 257), (299, 298), (128, 290)] + [tuple(rng.integers(0, n, 2)) for _ in range(12)]:
            if i == j:   # the diagonal is a square root of rounding noise (1e-7 against 1e-15): checked above with its own tolerance
                continue
            x, y = pts[i].astype(np.float64), pts[j].astype(np.float64)
            assert abs(rm[i, j] - EV.rmsd_common(x, y)) < 1e-9 and abs(tm[i, j] - EV.tm_score(x, y)) < 1e-9, (i, j)
        # rectangular, more than 65 535 pairs: 300 x 260
        rm2, tm2 = ctx.superpose_matrix(pts, pts[:260])
        assert np.abs(rm2 - rm[:, :260]).max() < 1e-12 and np.abs(tm2 - tm[:, :260]).max() < 1e-12
        bad = pts[:3].copy()
        bad[1, 5, 2] = np.nan
        L = ctx._l
        import ctypes as C
        out = np.zeros((3, 3))
        rc = L.trx2_superpose_matrix(ctx._h, 3, 3, 40, bad.ctypes.data_as(C.c_void_p), None, C.c_double(0), out.ctypes.data_as(C.c_void_p),
                                     out.ctypes.data_as(C.c_void_p))
        assert rc != 0 and b"non-finite" in L.trx2_last_error(ctx._h)
    finally:
        ctx.close()
