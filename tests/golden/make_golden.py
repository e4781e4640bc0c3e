#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's numpy halves.

Runs only in the build container (needs /root/reference). The reference's Python never ships: only the
vectors (inputs + expected outputs) written here are committed.  `pyrosetta` and `Bio` are absent from the
image (ordinary ModuleNotFoundError), so empty stub modules are registered for the import statements at
folding/utils_ros/utils_ros.py:3 and utils_trX2dy/utils.py:12,18; none of the captured functions touches them.

Fixtures written:
  seq_NMR.npz / seq_Xray.npz / seq.fasta      data files of the reference example (inputs of the fold)
  ref_decoys.npz                              backbone atoms of the 8 committed PyRosetta decoys + apo/holo CA
  gen_rst_{NMR,Xray}.npz                      G1: gen_rst(USE_ORIENT=True) restraint lists + parsed tables
  gen_rst_noorient.json                       G2: counts with USE_ORIENT=False
  feedback_{NMR,Xray}.npz                     G3+G4: get_neighbors / pros / process_distribution outputs
  random_dihedral.json                        G5: random_dihedral() draws under random.seed(s)
  constants.json                              G6: params.json + *.wts (data)
  glocon.json                                 G7: get_glocon_matrix() on the 8 committed example decoys
"""
import hashlib
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("pyrosetta", __all__=[])
    bio = _stub("Bio")
    pdb = _stub("Bio.PDB", PPBuilder=None, PDBParser=None)
    bio.PDB = pdb
    sys.path.insert(0, os.path.join(REF, "folding"))
    sys.path.insert(0, REF)
    from utils_ros import utils_ros  # folding/utils_ros/utils_ros.py
    import utils_trX2dy.utils as U  # utils_trX2dy/utils.py
    return utils_ros, U


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def parse_pdb_backbone(path):
    """Own minimal PDB reader: N, CA, C, O, CB per residue (first model, chain order), NaN when absent."""
    names = ["N", "CA", "C", "O", "CB"]
    res = {}
    order = []
    seq = {}
    with open(path) as f:
        for line in f:
            if line.startswith("ENDMDL"):
                break
            if not line.startswith("ATOM"):
                continue
            an = line[12:16].strip()
            if line[16] not in (" ", "A"):
                continue
            key = (line[21], int(line[22:26]), line[26])
            if key not in res:
                res[key] = np.full((5, 3), np.nan)
                order.append(key)
                seq[key] = line[17:20]
            if an in names:
                res[key][names.index(an)] = [float(line[30:38]), float(line[38:46]), float(line[46:54])]
    xyz = np.stack([res[k] for k in order])
    return xyz, [seq[k] for k in order]


def capture_gen_rst(utils_ros, npz_path, seq, use_orient):
    """Run the reference gen_rst and parse back the text files it wrote (what Rosetta would read)."""
    params = json.load(open(os.path.join(REF, "folding/data/params.json")))
    params["seq"] = seq
    params["USE_ORIENT"] = use_orient
    npz = np.load(npz_path)
    tmp = tempfile.TemporaryDirectory(prefix="/dev/shm/")
    try:
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            rst = utils_ros.gen_rst(npz, tmp, params)
        out = {}
        for ch in rst:
            items = rst[ch]
            a = np.array([it[0] for it in items], dtype=np.int32)
            b = np.array([it[1] for it in items], dtype=np.int32)
            p = np.array([it[2] for it in items], dtype=np.float64)
            xs, ys = [], []
            for it in items:
                fn = it[3].split(" TAG ")[1].split()[0]
                with open(fn) as f:
                    lx = f.readline().split("\t")
                    ly = f.readline().split("\t")
                assert lx[0] == "x_axis" and ly[0] == "y_axis"
                xs.append([float(v) for v in lx[1:]])
                ys.append([float(v) for v in ly[1:]])
            out[ch] = dict(a=a, b=b, p=p, x=np.array(xs), y=np.array(ys),
                           line0=items[0][3].replace(tmp.name, "TMP") if items else "")
        return out
    finally:
        tmp.cleanup()


def main():
    utils_ros, U = import_reference()
    ex = os.path.join(REF, "example")
    for f in ("output/seq/pred_npz/seq_NMR.npz", "output/seq/pred_npz/seq_Xray.npz", "seq.fasta"):
        shutil.copyfile(os.path.join(ex, f), os.path.join(OUT, os.path.basename(f)))
        os.chmod(os.path.join(OUT, os.path.basename(f)), 0o644)
    seq = "".join(l.strip() for l in open(os.path.join(ex, "seq.fasta")) if not l.startswith(">"))
    L = len(seq)

    # ---- reference decoys (outputs of the PyRosetta path) + natives ------------------------------------
    dec = {}
    for m in (1, 2):
        for k in (1, 2, 3, 4):
            xyz, _ = parse_pdb_backbone(os.path.join(ex, f"output/seq/pred_pdb/conf_{m}_{k}.pdb"))
            assert xyz.shape == (L, 5, 3)
            dec[f"conf_{m}_{k}"] = xyz.astype(np.float32)
    for n in ("apo", "holo"):
        xyz, _ = parse_pdb_backbone(os.path.join(ex, f"{n}.pdb"))
        dec[n] = xyz.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ref_decoys.npz"), **dec)

    # ---- G1 / G2: gen_rst -----------------------------------------------------------------------------
    noorient = {}
    for tag in ("NMR", "Xray"):
        npzp = os.path.join(OUT, f"seq_{tag}.npz")
        g = capture_gen_rst(utils_ros, npzp, seq, True)
        save = {}
        for ch, d in g.items():
            if ch == "rep":
                continue
            scale = 1e5 if ch == "omega" else 1e3
            yi = np.rint(d["y"] * scale).astype(np.int32)
            assert np.abs(yi / scale - d["y"]).max() < 1e-9
            save[f"{ch}_a"], save[f"{ch}_b"], save[f"{ch}_p"] = d["a"], d["b"], d["p"]
            save[f"{ch}_x"] = d["x"][0]
            assert np.all(d["x"] == d["x"][0])
            save[f"{ch}_yi"] = yi
            save[f"{ch}_scale"] = np.float64(scale)
            print(tag, ch, len(d["a"]), d["line0"])
        np.savez_compressed(os.path.join(OUT, f"gen_rst_{tag}.npz"), **save)
        g2 = capture_gen_rst(utils_ros, npzp, seq, False)
        noorient[tag] = {"channels": sorted(g2.keys()), "n_dist": int(len(g2["dist"]["a"])),
                         "dist_y_sha256": sha(np.rint(g2["dist"]["y"] * 1e3).astype(np.int32))}
        assert noorient[tag]["dist_y_sha256"] == sha(save["dist_yi"])
    json.dump(noorient, open(os.path.join(OUT, "gen_rst_noorient.json"), "w"), indent=1)

    # ---- G3 / G4: feedback -----------------------------------------------------------------------------
    # provenance: conf_2_1 = NMR/initial0, conf_1_1 = Xray/initial0
    for tag, decoy in (("NMR", "conf_2_1"), ("Xray", "conf_1_1")):
        # What get_atom_positions_pdb hands over (utils.py:252-291): float64 arrays (np.nan * np.zeros) holding Biopython's
        # float32 coordinates -- so the reference's geometry arithmetic runs in float64.
        xyz = dec[decoy].astype(np.float64)
        xyzs = {"N": xyz[:, 0], "CA": xyz[:, 1], "C": xyz[:, 2], "CB": xyz[:, 4]}
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            key, d6, o6, t6, p6 = U.get_neighbors({k: v.copy() for k, v in xyzs.items()}, seq, 20)
        assert key is False
        pf = U.pros(d6[None], o6[None], t6[None], p6[None], angle=True)
        fd, ft, fo, fp = pf[0][0, 0], pf[1][0, 0], pf[2][0, 0], pf[3][0, 0]
        npz = np.load(os.path.join(OUT, f"seq_{tag}.npz"))
        outs = {}
        for ch, fact in (("dist", fd), ("omega", fo), ("theta", ft), ("phi", fp)):
            outs[ch] = U.process_distribution_with_pred_distribution(npz[ch], fact, norm=True, smooth=True, sigma=1.0)
        outs["tmp"] = U.process_distribution_with_pred_distribution(npz["dist"], fd, norm=False)
        rng = np.random.default_rng(7)
        ii = rng.integers(0, L, 512)
        jj = rng.integers(0, L, 512)
        save = dict(decoy=np.array(decoy), dist6d=d6, omega6d=o6, theta6d=t6, phi6d=p6,
                    bin_dist=fd.argmax(-1).astype(np.uint8), bin_omega=fo.argmax(-1).astype(np.uint8),
                    bin_theta=ft.argmax(-1).astype(np.uint8), bin_phi=fp.argmax(-1).astype(np.uint8),
                    sample_i=ii.astype(np.int32), sample_j=jj.astype(np.int32))
        for ch, arr in outs.items():
            save[f"{ch}_sample"] = arr[ii, jj]
            save[f"{ch}_sha256"] = np.array(sha(arr))
            save[f"{ch}_dtype"] = np.array(str(arr.dtype))
            save[f"{ch}_sum"] = np.float64(arr.astype(np.float64).sum())
        np.savez_compressed(os.path.join(OUT, f"feedback_{tag}.npz"), **save)
        print(tag, "feedback", {k: str(v.dtype) for k, v in outs.items()})

    # ---- G7: GloCon matrix of the reference's own 8 example decoys (utils.py:543-569) ----------------------------
    # The reference's function is called as is; only its PDB reader (Biopython, absent) is replaced by one that returns
    # what get_atom_positions_pdb returns: float64 arrays holding float32 coordinates, the residue ids, the sequence.
    ex_pdb = os.path.join(REF, "example", "output", "seq", "pred_pdb")
    res1 = {"ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G", "HIS": "H", "ILE": "I",
            "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S", "THR": "T", "TRP": "W", "TYR": "Y", "VAL": "V"}

    def reader(pdb_file, model=0, retain_all_res=True):
        xyz, names = parse_pdb_backbone(pdb_file)
        xyz = xyz.astype(np.float32).astype(np.float64)
        return {"N": xyz[:, 0], "CA": xyz[:, 1], "C": xyz[:, 2], "CB": xyz[:, 4]}, np.arange(len(xyz)), "".join(res1[n] for n in names)

    U.get_atom_positions_pdb = reader
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        gm, gfiles = U.get_glocon_matrix(ex_pdb)
    order = np.argsort(gfiles)
    gm = gm[np.ix_(order, order)]
    json.dump({"files": [gfiles[i] for i in order], "matrix": [[float.hex(float(v)) for v in row] for row in gm]},
              open(os.path.join(OUT, "glocon.json"), "w"), indent=1)
    print("glocon", [gfiles[i] for i in order], np.round(gm[0], 4))

    # ---- G5: random_dihedral ---------------------------------------------------------------------------
    rd = {}
    for s in (0, 1, 2024):
        random.seed(s)
        rd[str(s)] = [list(utils_ros.random_dihedral()) for _ in range(200)]
    json.dump(rd, open(os.path.join(OUT, "random_dihedral.json"), "w"))

    # ---- G6: constants (data) --------------------------------------------------------------------------
    const = {"params": json.load(open(os.path.join(REF, "folding/data/params.json")))}
    for w in ("scorefxn", "scorefxn1", "scorefxn_vdw", "scorefxn_cart"):
        const[w] = {l.split()[0]: float(l.split()[1]) for l in open(os.path.join(REF, f"folding/data/{w}.wts")) if l.strip()}
    const["summary_txt"] = open(os.path.join(ex, "output/seq/summary.txt")).read()
    json.dump(const, open(os.path.join(OUT, "constants.json"), "w"), indent=1)
    print("done")


if __name__ == "__main__":
    main()
