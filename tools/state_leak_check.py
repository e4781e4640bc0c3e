"""Does anything survive in a context from one fold to the next?  The same job (X-ray example map, full protocol, 64 decoys,
one seed) on a fresh context and on contexts that first ran other jobs (smaller batch stopped at max_evals, other map,
larger batch).  Every result must be bitwise identical.  usage: state_leak_check.py <repo>"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden")
mx = np.load(os.path.join(g, "seq_Xray.npz")); mn = np.load(os.path.join(g, "seq_NMR.npz"))
runs = T.protocol.build_runs(90, 2)
cart = [q for q in runs if q["cartesian"]]
def job(c, seed=4242, B=64):
    c.set_map(mx["dist"], mx["omega"], mx["theta"], mx["phi"])
    return c.fold_batch(B, runs, seed=seed)
def stats(r):
    return "evals median %d  f median %.1f" % (np.median(r["n_evals"]), np.median(r["f"]))
a = T.Context(0); ra = job(a); print("fresh context:", stats(ra))
pre = {
    "small Cartesian-only batch stopped at max_evals": lambda c: (c.set_map(mn["dist"], mn["omega"], mn["theta"], mn["phi"]), c.fold_batch(6, cart, seed=3, max_evals=40)),
    "torsion-only batch of 24 on the other map": lambda c: (c.set_map(mn["dist"], mn["omega"], mn["theta"], mn["phi"]), c.fold_batch(24, T.protocol.build_runs(90, 2, cartesian_stage=False), seed=9)),
    "larger batch (128) of the same job": lambda c: job(c, seed=77, B=128),
    "the same job": lambda c: job(c),
}
bad = 0
for name, f in pre.items():
    c = T.Context(0); f(c); r = job(c)
    same = all(np.array_equal(ra[k], r[k]) for k in ("xyz", "tors", "f", "n_evals", "status"))
    ndiff = int((np.abs(ra["xyz"] - r["xyz"]).reshape(64, -1).max(1) > 0).sum())
    print(f"after {name:52s}: identical {same}  decoys differing {ndiff}  {stats(r)}")
    bad += not same
    c.close()
a.close()
sys.exit(1 if bad else 0)
