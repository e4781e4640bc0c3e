#!/usr/bin/env python
"""Golden vectors for the other restraint builders of the reference, captured by IMPORTING them
(folding/utils_ros/utils_ros.py: gen_idp_rst :196-373, gen_rst_af2 :148-194, gen_gpcr_rst :484-654; pyrosetta stubbed as in
make_golden.py).  Runs only in the build container; only inputs + outputs are committed:

  rst_variants_inputs.npz   idr mask [L][L] (residues 30-44 "disordered"), an AF2-style 64-bin C-alpha distogram + its 63 bin
                            edges (built here from the reference decoy conf_2_1), seeded
  gen_rst_idp_NMR.npz       gen_idp_rst(seq_NMR.npz + idr): the rows of the idr pairs (all other rows are asserted equal to
                            gen_rst_NMR.npz) + SHA-256 of the full integer tables
  gen_rst_af2.npz           gen_rst_af2(af2 distogram)
  gen_rst_gpcr_NMR.npz      gen_gpcr_rst(seq_NMR.npz + idr, known = 6-D geometry of the 8 committed decoys)
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def capture(fn, *args):
    tmp = tempfile.TemporaryDirectory(prefix="/dev/shm/")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            rst = fn(*args[:-1], tmp, args[-1])
        out = {}
        for ch, items in rst.items():
            if ch == "rep" or not items:
                continue
            xs, ys = [], []
            for it in items:
                fnm = it[3].split(" TAG ")[1].split()[0]
                with open(fnm) as f:
                    lx, ly = f.readline().split("\t"), f.readline().split("\t")
                xs.append([float(v) for v in lx[1:]]); ys.append([float(v) for v in ly[1:]])
            scale = 1e5 if ch == "omega" else 1e3
            x, y = np.array(xs), np.array(ys)
            assert np.all(x == x[0])
            yi = np.rint(y * scale).astype(np.int64)
            assert np.abs(yi / scale - y).max() < 1e-9
            out[f"{ch}_a"] = np.array([it[0] for it in items], np.int32); out[f"{ch}_b"] = np.array([it[1] for it in items], np.int32)
            out[f"{ch}_p"] = np.array([it[2] for it in items], np.float64); out[f"{ch}_x"] = x[0]; out[f"{ch}_yi"] = yi.astype(np.int32)
            out[f"{ch}_scale"] = np.float64(scale); out[f"{ch}_line0"] = np.array(items[0][3].replace(tmp.name, "TMP"))
        return out
    finally:
        tmp.cleanup()


def only_idr_rows(out, idr):
    """gen_idp_rst / gen_gpcr_rst differ from gen_rst only on the pairs flagged in idr: keep those rows and the digest of the whole"""
    base = np.load(os.path.join(HERE, "gen_rst_NMR.npz"))
    keep = {}
    for ch in ("dist", "omega", "theta", "phi"):
        a, b = out[f"{ch}_a"], out[f"{ch}_b"]
        assert all(np.array_equal(out[f"{ch}_{k}"], base[f"{ch}_{k}"]) for k in ("a", "b", "p", "x"))
        rows = np.nonzero(idr[a, b])[0]
        assert np.array_equal(np.delete(out[f"{ch}_yi"], rows, axis=0), np.delete(base[f"{ch}_yi"], rows, axis=0))
        keep[f"{ch}_rows"] = rows.astype(np.int32); keep[f"{ch}_yi_rows"] = out[f"{ch}_yi"][rows]
        keep[f"{ch}_yi_sha256"] = np.array(MG.sha(out[f"{ch}_yi"])); keep[f"{ch}_line0"] = out[f"{ch}_line0"]
    return keep


def main():
    utils_ros, U = MG.import_reference()
    seq = "".join(l.strip() for l in open(os.path.join(HERE, "seq.fasta")) if not l.startswith(">"))
    L = len(seq)
    params = json.load(open(os.path.join(MG.REF, "folding/data/params.json")))
    params["seq"] = seq
    params["USE_ORIENT"] = True
    npz = dict(np.load(os.path.join(HERE, "seq_NMR.npz")))
    dis = np.zeros(L, bool); dis[30:45] = True
    idr = (dis[:, None] | dis[None, :])
    npz["idr"] = idr
    dec = np.load(os.path.join(HERE, "ref_decoys.npz"))
    # AF2-style distogram: 64 bins between the 63 edges 2.3125 .. 21.6875, one-hot of conf_2_1's CA-CA distance, blurred and mixed
    ca = dec["conf_2_1"][:, 1].astype(np.float64)
    d = np.linalg.norm(ca[:, None] - ca[None], axis=-1)
    edges = np.linspace(2.3125, 21.6875, 63)
    k = (d[..., None] > edges).sum(-1)
    oh = np.eye(64)[k]
    kk = np.arange(64)
    G = np.exp(-0.5 * ((kk[:, None] - kk[None]) / 2.0) ** 2); G /= G.sum(1, keepdims=True)
    af = 0.97 * (oh @ G.T) + 0.03 / 64
    af = (af / af.sum(-1, keepdims=True)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "rst_variants_inputs.npz"), idr=idr, af2_dist=af, af2_bins=edges)

    out = capture(utils_ros.gen_idp_rst, npz, params)
    np.savez_compressed(os.path.join(HERE, "gen_rst_idp_NMR.npz"), **only_idr_rows(out, idr))
    print("idp", {k: v.shape for k, v in out.items() if k.endswith("_yi")}, out["dist_line0"], out["phi_line0"])

    p2 = dict(params); p2["USE_ORIENT"] = False
    out = capture(utils_ros.gen_rst_af2, {"dist": af, "bins": edges}, p2)
    np.savez_compressed(os.path.join(HERE, "gen_rst_af2.npz"), **out)
    print("af2", {k: v.shape for k, v in out.items() if k.endswith("_yi")}, out["dist_line0"], out["dist_x"][:6])

    # known structures for gpcr: the reference's get_neighbors on the 8 committed decoys (what a caller of -KNOWN would provide)
    known = {"dist": [], "omega": [], "theta_asym": [], "phi_asym": []}
    for name in ("conf_1_1", "conf_1_2", "conf_1_3", "conf_1_4", "conf_2_1", "conf_2_2", "conf_2_3", "conf_2_4"):
        xyz = dec[name].astype(np.float64)
        with contextlib.redirect_stdout(io.StringIO()):
            _, d6, o6, t6, p6 = U.get_neighbors({"N": xyz[:, 0].copy(), "CA": xyz[:, 1].copy(), "C": xyz[:, 2].copy(), "CB": xyz[:, 4].copy()}, seq, 20)
        known["dist"].append(d6); known["omega"].append(o6); known["theta_asym"].append(t6); known["phi_asym"].append(p6)
    known = {k: np.array(v) for k, v in known.items()}
    out = capture(utils_ros.gen_gpcr_rst, npz, known, params)
    np.savez_compressed(os.path.join(HERE, "gen_rst_gpcr_NMR.npz"), **only_idr_rows(out, idr))
    print("gpcr", {k: v.shape for k, v in out.items() if k.endswith("_yi")})


if __name__ == "__main__":
    main()
