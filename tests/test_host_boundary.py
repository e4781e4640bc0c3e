"""CPU: host logic around the fold -- feedback (pinned bit-for-bit to the reference), PDB I/O, option parsing, output layout."""
import hashlib
import importlib
import os

import numpy as np
import pytest

F = importlib.import_module("trrosettax2-dynamics_amd.feedback")
P = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
FO = importlib.import_module("trrosettax2-dynamics_amd.fold")
PL = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def decoy_pdb(golden_dir, tmp_path, seq, name):
    xyz = np.load(os.path.join(golden_dir, "ref_decoys.npz"))[name].copy()
    xyz[np.isnan(xyz[:, 4, 0]), 4] = 0.0  # Gly: write_pdb omits CB anyway
    path = str(tmp_path / f"{name}.pdb")
    P.write_pdb(path, seq, xyz)
    return path


@pytest.mark.parametrize("tag,name", [("NMR", "conf_2_1"), ("Xray", "conf_1_1")])
def test_feedback_is_bit_identical_to_reference(golden_dir, tmp_path, seq, tag, name):
    """get_neighbors / pros / process_distribution_with_pred_distribution (utils.py:125-475), incl. quirks R1-R4, R11:
    6-D geometry equal, bins equal, SHA-256 of every full output array equal to the reference's."""
    g = np.load(os.path.join(golden_dir, f"feedback_{tag}.npz"))
    pdb = decoy_pdb(golden_dir, tmp_path, seq, name)
    xyz, s = P.read_backbone(pdb)
    assert s == seq and np.isnan(xyz[[i for i, a in enumerate(seq) if a == "G"], 4]).all()
    for a, k in zip(F.get_neighbors(xyz, seq), ("dist6d", "omega6d", "theta6d", "phi6d")):
        assert np.array_equal(a, g[k]), k
    jd, jt, jo, jp = F.get_distribution_from_pdb(pdb)
    for k, v in (("bin_dist", jd), ("bin_omega", jo), ("bin_theta", jt), ("bin_phi", jp)):
        assert np.array_equal(v, g[k]), k
    npz = os.path.join(golden_dir, f"seq_{tag}.npz")
    outs = dict(zip(("dist", "omega", "theta", "phi"), F.get_npz_from_pred_pdb(npz, pdb)))
    outs["tmp"] = F.get_npz_from_pred_pdb(npz, pdb, tmp=True)
    for ch, arr in outs.items():
        assert str(arr.dtype) == str(g[f"{ch}_dtype"]) and sha(arr) == str(g[f"{ch}_sha256"]), ch
    assert np.array_equal(F.get_npz_from_pred_pdb(npz, pdb, angle=False), outs["dist"])


def test_phi_bins_come_from_theta(golden_dir, tmp_path, seq):
    """quirk R1 (utils.py:226): the phi one-hot is theta binned on phi's edges, so negative theta gives bin 0"""
    xyz, _ = P.read_backbone(decoy_pdb(golden_dir, tmp_path, seq, "conf_2_1"))
    d6, o6, t6, p6 = F.get_neighbors(xyz, seq)
    jd, jo, jt, jp = F.bin_geometry(d6, o6, t6, p6)
    c = jd > 0
    assert np.all(jp[c & (t6 <= 0)] == 0) and np.all(jp[c & (t6 > 0)] >= 1)
    assert np.all(jo[~c] == 0) and np.all(jt[~c] == 0) and np.all(jp[~c] == 0)  # R2


def test_reliability_score_matches_survey(golden_dir, tmp_path, seq):
    """NMR initials 81/88 both, X-ray initials 83/88 both (ties -> initial0 wins)"""
    for name, want in (("conf_2_1", 81), ("conf_2_2", 81), ("conf_1_1", 83), ("conf_1_2", 83)):
        assert F.calculate_reliability_score(decoy_pdb(golden_dir, tmp_path, seq, name)) == pytest.approx(want / 88)


def test_pdb_writer_format_and_safety(tmp_path):
    seq = "AGW"
    xyz = np.arange(45, dtype=np.float64).reshape(3, 5, 3) * 1.001
    path = str(tmp_path / "o.pdb")
    P.write_pdb(path, seq, xyz)
    lines = [l for l in open(path).read().splitlines() if l.startswith("ATOM")]
    assert len(lines) == 5 + 4 + 5 and not any(" CB  GLY" in l for l in lines)
    assert lines[0][12:16] == " N  " and lines[0][17:20] == "ALA" and lines[0][21] == "A" and int(lines[0][22:26]) == 1
    assert all(len(l) >= 54 and float(l[30:38]) == pytest.approx(round(float(l[30:38]), 3)) for l in lines)
    back, s = P.read_backbone(path)
    assert s == seq and np.allclose(np.nan_to_num(back), np.nan_to_num(np.where(np.isnan(back), np.nan, xyz)), atol=5e-4)
    with pytest.raises(ValueError):
        P.write_pdb(str(tmp_path / "bad.pdb"), seq, np.full((3, 5, 3), np.nan))
    assert not os.path.exists(tmp_path / "bad.pdb") and not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]


def test_option_string_parsing():
    """the `options` string built at run_inference.py:295 and the flags of arguments.py:5-25"""
    a = FO.parse_options("-m 2 --orient -r no-idp")
    assert (a.mode, a.use_orient, a.rst, a.pcut, a.fastrelax) == (2, True, "no-idp", 0.05, True)
    a = FO.parse_options("-m 0 --no-orient -pd 0.15 --no-fastrelax")
    assert (a.mode, a.use_orient, a.pcut, a.fastrelax) == (0, False, 0.15, False)
    assert FO._unquote('"/a b/c.npz"') == "/a b/c.npz" and FO._unquote("x.npz") == "x.npz"
    a = FO.parse_options("-r idp -m 3 --orient")                     # the other builders and mode 3
    assert (a.rst, a.mode) == ("idp", 3)
    with pytest.raises(RuntimeError):                                 # utils_ros.py:149-150: gen_rst_af2 refuses --orient
        FO.parse_options("-r af2 --orient")
    assert FO.parse_options("-r af2 --no-orient").rst == "af2"
    with pytest.raises(ValueError):                                   # folding.py:66-67: -r gpcr loads args.KNOWN
        FO.parse_options("-r gpcr --orient")


def test_output_layout_matches_committed_example(tmp_path):
    """provenance of example/output/seq/pred_pdb: conf_1_1/2 = Xray initials, conf_1_3/4 = NMR
    iterations, conf_2_1/2 = NMR initials, conf_2_3/4 = Xray iterations; iteration files ordered by number."""
    d = str(tmp_path)
    files = {"NMR": ["initial0", "initial1", "seq1", "seq2"], "Xray": ["initial0", "initial1", "seq3", "seq4"]}
    for sub, names in files.items():
        os.makedirs(os.path.join(d, sub))
        for n in names:
            open(os.path.join(d, sub, n + ".pdb"), "w").write(f"{sub}/{n}")
    PL.flatten_and_rename(d, 2)
    got = {n: open(os.path.join(d, n)).read() for n in sorted(os.listdir(d))}
    assert got == {"conf_1_1.pdb": "Xray/initial0", "conf_1_2.pdb": "Xray/initial1", "conf_1_3.pdb": "NMR/seq1",
                   "conf_1_4.pdb": "NMR/seq2", "conf_2_1.pdb": "NMR/initial0", "conf_2_2.pdb": "NMR/initial1",
                   "conf_2_3.pdb": "Xray/seq3", "conf_2_4.pdb": "Xray/seq4"}
    d2 = str(tmp_path / "many")
    os.makedirs(os.path.join(d2, "NMR"))
    for k in range(1, 13):
        open(os.path.join(d2, "NMR", f"t{k}.pdb"), "w").write(str(k))
    PL.flatten_and_rename(d2, 12)
    assert [open(os.path.join(d2, f"conf_1_{k}.pdb")).read() for k in range(1, 13)] == [str(k) for k in range(1, 13)]


def test_synthetic_map_generator_is_deterministic_and_realisable():
    S = importlib.import_module("trrosettax2-dynamics_amd.synth")
    from oracle import oracle as O
    a, b = S._make_map(40, 40, 100), S._make_map(40, 40, 100)
    assert all(np.array_equal(a[k], b[k]) for k in ("dist", "omega", "theta", "phi", "tors"))
    for k, nb in (("dist", 37), ("omega", 25), ("theta", 25), ("phi", 13)):
        assert a[k].shape == (40, 40, nb) and a[k].dtype == np.float32 and np.allclose(a[k].sum(-1), 1, atol=1e-5)
    assert np.allclose(a["dist"], a["dist"].transpose(1, 0, 2)) and np.allclose(a["omega"], a["omega"].transpose(1, 0, 2))
    N, CA, C, CB = S.nerf_backbone(a["tors"])
    xo = O.nerf(a["tors"])
    assert max(np.abs(N - xo[:, 0]).max(), np.abs(CA - xo[:, 1]).max(), np.abs(CB - xo[:, 4]).max()) < 1e-9


@pytest.mark.parametrize("angle", [True, False])
def test_single_parse_feedback_equals_the_two_reference_calls(golden_dir, tmp_path, seq, angle):
    """feedback_labels (one PDB parse, arrays in memory) == get_npz_from_pred_pdb(...) + get_npz_from_pred_pdb(..., tmp=True),
    the two calls run_inference.py:75-77,116-118 makes; also on a second iteration, where `tmp` comes from the first."""
    npz = os.path.join(golden_dir, "seq_NMR.npz")
    pdb1, pdb2 = decoy_pdb(golden_dir, tmp_path, seq, "conf_2_1"), decoy_pdb(golden_dir, tmp_path, seq, "conf_2_2")
    z = np.load(npz)
    lab = F.feedback_labels({k: z[k] for k in (("dist", "theta", "omega", "phi") if angle else ("dist",))}, pdb1, 1.0, angle)
    ref = F.get_npz_from_pred_pdb(npz, pdb1, angle=angle)
    want = dict(zip(("dist", "omega", "theta", "phi"), ref)) if angle else {"dist": ref}
    want["tmp"] = F.get_npz_from_pred_pdb(npz, pdb1, tmp=True, angle=angle)
    assert sorted(lab) == sorted(want) and all(np.array_equal(lab[k], want[k]) for k in want)
    it2 = str(tmp_path / "it2.npz")
    np.savez(it2, **lab)
    lab2 = F.feedback_labels(lab, pdb2, 1.0, angle)
    ref2 = F.get_npz_from_pred_pdb(it2, pdb2, angle=angle)
    want2 = dict(zip(("dist", "omega", "theta", "phi"), ref2)) if angle else {"dist": ref2}
    want2["tmp"] = F.get_npz_from_pred_pdb(it2, pdb2, tmp=True, angle=angle)
    assert all(np.array_equal(lab2[k], want2[k]) for k in want2)


def test_pdb_round_trip_without_the_file(tmp_path, golden_dir):
    """pdbio.as_read_from_pdb(seq, xyz) must equal read_backbone(write_pdb(seq, xyz)) bit for bit: the device feedback
    consumes the decoy "as the reference sees it" without reading the file back.  Includes exact .0005 ties and glycine."""
    P = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
    rng = np.random.default_rng(3)
    seq = "GASTGVLKDEGNQRHFWYCMPI" * 4
    xyz = (rng.normal(size=(len(seq), 5, 3)) * 30).astype(np.float32)
    ties = np.array([0.0005, -0.0005, 1.0005, 2.0015, -3.0025, 12.3455, 0.0, -0.0, 123.4565, -99.9995, 7.1245, 0.00049999], np.float32)
    xyz.reshape(-1)[:len(ties)] = ties
    xyz.reshape(-1)[40:52] = (np.arange(12) * 0.001 + 0.0005).astype(np.float32)
    path = str(tmp_path / "x.pdb")
    P.write_pdb(path, seq, xyz)
    a, sa = P.read_backbone(path)
    b, sb = P.as_read_from_pdb(seq, xyz)
    assert sa == sb == seq
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(a).sum() == 3 * seq.count("G")
    assert np.array_equal(np.nan_to_num(a).view(np.uint32), np.nan_to_num(b).view(np.uint32))     # bits, including signed zeros
    with pytest.raises(ValueError):
        P.as_read_from_pdb("A@A", xyz[:3])


def test_glocon_matrix_is_bit_identical_to_reference(golden_dir, tmp_path, seq):
    """cluster.get_glocon_matrix (utils_trX2dy/utils.py:543-569) on the reference's eight example decoys against the matrix its
    own function returned (tests/golden/glocon.json: get_glocon_matrix called as is, only the Biopython reader substituted);
    plus the RMSD matrix against SURVEY.md section 4 and the clustering front end."""
    import json
    CL = importlib.import_module("trrosettax2-dynamics_amd.cluster")
    g = json.load(open(os.path.join(golden_dir, "glocon.json")))
    d = tmp_path / "pdb"
    d.mkdir()
    for f in g["files"]:
        decoy_pdb(golden_dir, d, seq, f[:-4])
    m, got_files = CL.get_glocon_matrix(str(d))
    assert got_files == g["files"]                                             # sorted order on both sides
    want = np.array([[float.fromhex(v) for v in row] for row in g["matrix"]])
    assert np.array_equal(m, want)
    r, _ = CL.get_rmsd_matrix(str(d))
    assert abs(r[0, 1] - 0.62) < 0.01 and abs(r[4, 5] - 0.86) < 0.01          # conf_1_1/1_2, conf_2_1/2_2
    out = CL.save_cluster_result(str(d), n_clusters=2, n_files=1, mode="glocon")
    assert sorted(len(v) for v in out.values()) == [4, 4] and len(os.listdir(d / "clusters_result")) == 2   # the two models separate
    # mode tmscore (utils.py:524-541): TM-score matrix of all pairs, zero diagonal; the two models of the example separate on it too.
    # (host path here: 28 pairs of the pure-numpy search; the device path is tested in test_gpu_feedback.py)
    tm, rm, files = CL.get_tmscore_and_rmsd_matrix(str(d))
    assert files == g["files"] and np.allclose(np.diag(tm), 0) and np.array_equal(tm, tm.T) and np.allclose(rm, r)
    assert tm[0, 1] > 0.9 and tm[4, 5] > 0.9 and tm[0, 4] < 0.9            # same-map initials agree, the two maps differ (2.1-2.5 A)


def test_evaluation_reproduces_the_reference_summary(golden_dir, tmp_path, seq):
    """evaluate.py / run_score (evaluate_utils.py:33-100) without the external TM-score binary: on the reference's committed
    example (eight decoys against apo and holo) the summary must carry the numbers of its committed summary.txt:
      holo best_RMSD: 3.931 ... best_TM_score: 0.6269   apo best_RMSD: 3.018 ... best_TM_score: 0.6661
      Mean RMSD: 3.47  Mean TM-score: 0.65  Min RMSD: 3.02  Max TM-score: 0.67"""
    EV = importlib.import_module("trrosettax2-dynamics_amd.evaluate")
    CLI = importlib.import_module("evaluate")
    nat, pred = tmp_path / "native", tmp_path / "pred"
    nat.mkdir(); pred.mkdir()
    for k in ("apo", "holo"):
        decoy_pdb(golden_dir, nat, seq, k)
    for k in ("conf_1_1", "conf_1_2", "conf_1_3", "conf_1_4", "conf_2_1", "conf_2_2", "conf_2_3", "conf_2_4"):
        decoy_pdb(golden_dir, pred, seq, k)
    assert CLI.main(["-n", str(nat), "-p", str(pred), "-o", str(tmp_path / "out" / "s.txt")]) == 0
    txt = open(tmp_path / "out" / "s.txt").read().splitlines()
    rows = {l.split()[0]: l.split() for l in txt[:2]}
    assert (rows["apo"][2], rows["apo"][6]) == ("3.018", "0.6661") and rows["apo"][4] == rows["apo"][8] == "conf_2_3"
    assert (rows["holo"][2], rows["holo"][6]) == ("3.931", "0.6269")
    assert txt[2:] == ["Mean RMSD: 3.47", "Mean TM-score: 0.65", "Min RMSD: 3.02", "Max TM-score: 0.67"]
    x = np.load(os.path.join(golden_dir, "ref_decoys.npz"))["conf_1_1"][:, 1].astype(np.float64)
    assert EV.tm_score(x, x) == pytest.approx(1.0) and EV.rmsd_common(x, x[::-1].copy()[::-1]) == pytest.approx(0.0, abs=1e-9)
    # --align on files whose residues already correspond one to one gives the same summary (identical sequences align trivially)
    assert EV.run_score(str(nat), str(pred), align=True) == EV.run_score(str(nat), str(pred))


def test_align_reproduces_the_tmscore_programs_seq_option(golden_dir, tmp_path):
    """evaluate.py --align = `TMscore native model -seq` (evaluate_utils.py:56-58).  Golden vectors captured from the reference's
    prebuilt bin/TMscore in the build container (tests/golden/make_golden_align.py): 48 pairs of chains cut from its example
    natives with internal deletions, truncated termini, point mutations and unrelated numbering.  The alignment must equal
    the program's printed one RESIDUE PAIR FOR RESIDUE PAIR; the two numbers the pipeline parses from its output then follow:
    the RMSD of the aligned residues to the 3 printed decimals, the TM-score within 5e-4 (the program's search heuristics are
    not in the tree: the module's docstring) -- measured worst 0 / 5e-5 on the 12 pairs scored."""
    import json
    EV = importlib.import_module("trrosettax2-dynamics_amd.evaluate")
    cases = json.load(open(os.path.join(golden_dir, "align_tmscore.json")))["cases"]
    assert len(cases) == 48
    worst_r = worst_t = 0.0
    for k, c in enumerate(cases):
        pairs = EV.nw_align(c["seq_a"], c["seq_b"])
        assert [list(p) for p in pairs] == c["pairs"] and len(pairs) == c["n_common"], k
        x = np.array(c["ca_a"])[[i for i, _ in pairs]]
        y = np.array(c["ca_b"])[[j for _, j in pairs]]
        worst_r = max(worst_r, abs(round(EV.rmsd_common(x, y), 3) - c["rmsd"]))
        if k % 4 == 0:   # the host TM-score search is ~1 s per pair
            worst_t = max(worst_t, abs(EV.tm_score(x, y, l_norm=len(c["seq_b"])) - c["tm"]))
    print("\nworst |RMSD - program| %.4f, worst |TM - program| %.5f" % (worst_r, worst_t))
    assert worst_r <= 1.001e-3 and worst_t <= 5e-4
    # through the files: two chains with unrelated numbering are matched only by the alignment
    c = cases[1]
    three = {v: k for k, v in EV._AA3.items()}
    for name, seq_, ca, first in (("a", c["seq_a"], c["ca_a"], 7), ("b", c["seq_b"], c["ca_b"], 31)):
        with open(tmp_path / f"{name}.pdb", "w") as f:
            for n, (aa, xyz) in enumerate(zip(seq_, ca)):
                f.write("ATOM  %5d  CA  %s A%4d    %8.3f%8.3f%8.3f  1.00  0.00           C\n" % (n + 1, three[aa], first + n, *xyz))
    r, t = EV.compare(str(tmp_path / "a.pdb"), str(tmp_path / "b.pdb"), align=True)
    assert abs(round(r, 3) - c["rmsd"]) <= 1.001e-3 and abs(t - c["tm"]) <= 5e-4


def test_one_failed_decoy_does_not_discard_the_batch(golden_dir, tmp_path, seq):
    """the decoys that folded are written, the failed one gets no file, and the call still raises
    (the reference passes silently, utils.py:498; SURVEY.md 8b: raise, never a partial PDB)."""
    xyz = np.load(os.path.join(golden_dir, "ref_decoys.npz"))["conf_2_1"].copy()
    xyz[np.isnan(xyz[:, 4, 0]), 4] = 0.0
    r = dict(xyz=np.stack([xyz, xyz, xyz]).astype(np.float32), status=np.array([0, 1, 0], np.int32), n_evals=np.array([10, 20, 30], np.int32))
    r["xyz"][1, 3, 1, 0] = np.nan
    names = ["a0.pdb", "a1.pdb", "a2.pdb"]
    with pytest.raises(FO.FoldError, match=r"decoys \[1\]") as ei:
        FO._write_decoys(r, seq, str(tmp_path), names, seed=7)
    assert ei.value.bad == [1] and ei.value.result is r and isinstance(ei.value, RuntimeError)
    assert sorted(os.listdir(tmp_path)) == ["a0.pdb", "a2.pdb"]
    got, s = P.read_backbone(str(tmp_path / "a2.pdb"))
    assert s == seq and np.allclose(got[:, 1], xyz[:, 1], atol=6e-4)
