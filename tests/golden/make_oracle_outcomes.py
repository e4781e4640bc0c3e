"""Fixture: what the CPU oracle (oracle/trx2_oracle.c, float64) makes of the DEFAULT protocol (-m 2 --fastrelax, 35 runs) on the six maps the
reference's eight example decoys were folded from -- the two committed distograms and the four fed-back maps of the iteration phase
(npz1 = feedback(seq_{tag}.npz, reference initial0), npz2 = feedback(npz1, reference seq{k}); host mirror of the reference's feedback,
pinned bit for bit) -- 256 decoys each from seeded random starts.  tests/test_gpu_outcome_vs_oracle.py folds the SAME starts on the GPU and
compares the distributions.  ~10 minutes on 8 cores, which is why it is a fixture and not part of the GPU suite.

The file carries a digest of the sources that define the model (include/trx2_model.h, oracle/trx2_oracle.c, protocol.py): the test refuses a
fixture made with another model.  Needs nothing from /root/reference (every input is a committed fixture).
usage: python tests/golden/make_oracle_outcomes.py [decoys per map = 256]"""
import hashlib
import importlib
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle.kabsch import kabsch_rmsd  # noqa: E402

T = importlib.import_module("trrosettax2-dynamics_amd")
FB = importlib.import_module("trrosettax2-dynamics_amd.feedback")
PD = importlib.import_module("trrosettax2-dynamics_amd.pdbio")

CHAINS = {"NMR": ("conf_2_1", "conf_2_2", "conf_1_3", "conf_1_4"), "Xray": ("conf_1_1", "conf_1_2", "conf_2_3", "conf_2_4")}
SEED = 5150


def model_digest():
    """Digest of what DEFINES the model: the C sources without comments and white space, protocol.py as its syntax tree -- a comment or
    layout edit does not invalidate a 10-minute fixture, a changed constant or statement does."""
    import ast
    import re
    h = hashlib.sha256()
    for rel in ("include/trx2_model.h", "oracle/trx2_oracle.c"):
        src = open(os.path.join(ROOT, rel)).read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        h.update(re.sub(r"\s+", " ", src).encode())
    h.update(ast.dump(ast.parse(open(os.path.join(ROOT, "trrosettax2-dynamics_amd/protocol.py")).read())).encode())
    return h.hexdigest()


def maps_and_targets(golden=HERE):
    """-> list of (key, arrays dict(dist, omega, theta, phi), names of the reference decoys to measure against)"""
    ref = np.load(os.path.join(golden, "ref_decoys.npz"))
    seq = "".join(l.strip() for l in open(os.path.join(golden, "seq.fasta")) if not l.startswith(">"))
    out = []
    for tag, (i0, i1, s1, s2) in CHAINS.items():
        m = dict(np.load(os.path.join(golden, f"seq_{tag}.npz")))
        host = {k: m[k] for k in ("dist", "theta", "omega", "phi")}
        out.append((f"{tag}/initial", dict(host), (i0, i1)))
        with tempfile.TemporaryDirectory() as tmp:
            for stage, (fed, target) in enumerate(((i0, s1), (s1, s2)), start=1):
                path = os.path.join(tmp, f"{fed}.pdb")
                PD.write_pdb(path, seq, np.nan_to_num(ref[fed]))
                host = FB.feedback_labels(host, path, 1.0, True)
                out.append((f"{tag}/stage{stage}", {k: host[k] for k in ("dist", "theta", "omega", "phi")}, (target, i0, i1)))
    return seq, ref, out


def main(n=256):
    seq, ref, cases = maps_and_targets()
    protocols = {"": T.protocol.build_runs(90, 2, fastrelax=True), "_nofastrelax": T.protocol.build_runs(90, 2)}
    cases = [(k, a, nm, "") for k, a, nm in cases] + [(k, a, nm, "_nofastrelax") for k, a, nm in cases if k.endswith("initial")]
    t0 = np.stack([O.random_torsions(90, SEED, d) for d in range(n)])
    out = os.path.join(HERE, "oracle_outcomes.npz")
    rec = {}
    if os.path.exists(out):       # keep what an interrupted or earlier run of the SAME model has made
        old = np.load(out)
        if str(old["digest"]) == model_digest() and int(old["n"]) == n:
            rec = {k: old[k] for k in old.files}
    rec.update({"digest": np.array(model_digest()), "seed": np.array(SEED), "n": np.array(n)})
    for key, arrs, names, proto in cases:
        k = key.replace("/", "_") + proto
        if k + "_f" in rec:
            continue
        Tb = O.Tables(arrs["dist"], arrs["omega"], arrs["theta"], arrs["phi"], seq=seq)
        t = time.time()
        _, xo, st, used = O.fold_batch(Tb, t0, protocols[proto])
        assert all(s["status"] == 0 for s in st), key
        ca = np.asarray(xo)[:, :, 1]
        rec[k + "_f"] = np.array([s["f_final"] for s in st])
        rec[k + "_evals"] = np.array([s["n_evals"] for s in st], np.int32)
        rec[k + "_iters"] = np.array([s["n_iters"] for s in st], np.int32)
        rec[k + "_rmsd"] = np.array([[kabsch_rmsd(ca[d], ref[nm][:, 1]) for nm in names] for d in range(n)], np.float32)
        rec[k + "_refs"] = np.array(",".join(names))
        rm = rec[k + "_rmsd"][:, 0] if "stage" in key else rec[k + "_rmsd"].min(1)
        print(f"{k}: {n} decoys in {time.time() - t:.0f} s on {used} threads; RMSD median {np.median(rm):.3f}, evaluations median {np.median(rec[k + '_evals']):.0f}", flush=True)
        np.savez_compressed(out, **rec)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
