"""Is the flipped peptide between G33 and G34 on the NMR map (profiles/README.md, round 5) a SEARCH result or the MODEL's preference?  Folds n decoys (default
protocol), takes those with the flip (phi 34 > 0), sets psi 33 / phi 34 to the reference's values (26 / -100 degrees), runs the relax stage again from there and
from the unchanged torsions (control), and compares the two under the last restrained run's weights.  usage: flip_probe.py <repo> [n = 512]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = os.path.join(sys.argv[1], "tests", "golden")
seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
m = np.load(os.path.join(g, "seq_NMR.npz"))
runs = T.protocol.build_runs(90, 2, fastrelax=True)
ctx = T.Context(0, lanes=2)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
r = ctx.fold_batch(n, runs, seed=77)
tors = r["tors"].astype(np.float64)
flipped = np.degrees(tors[:, 33, 0]) > 0
print(f"{n} decoys: phi(34) > 0 in {100 * flipped.mean():.0f} %; psi(33) circular mean of those {np.degrees(np.angle(np.exp(1j * tors[flipped, 32, 1]).mean())):.0f}")
sel = np.where(flipped)[0][:256]
ctrl = tors[sel].copy()
fixed = tors[sel].copy()
fixed[:, 32, 1] = np.radians(26.0); fixed[:, 33, 0] = np.radians(-100.0)
relax = runs[14:-1]                                   # the relax stage without the closing unrestrained run
w = np.array(relax[-1]["w"], np.float64)
out = {}
for name, t0 in (("control", ctrl), ("psi33 / phi34 set to the reference's", fixed)):
    rr = ctx.fold_batch(len(sel), relax, tors0=t0.astype(np.float32))
    tf = rr["tors"].astype(np.float64)
    f, e, _, _ = ctx.eval_batch(tf, w)
    out[name] = (f, e, tf)
    print(f"{name}: after the relax stage phi(34) < 0 in {100 * np.mean(np.degrees(tf[:, 33, 0]) < 0):.0f} %, psi(33) in (-60, 90) in {100 * np.mean((np.degrees(tf[:, 32, 1]) > -60) & (np.degrees(tf[:, 32, 1]) < 90)):.0f} %; "
          f"total under the last restrained run's weights: median {np.median(f):.1f}")
fc, ec, _ = out["control"]; ff, ef, tf = out["psi33 / phi34 set to the reference's"]
stay = np.degrees(tf[:, 33, 0]) < 0
d = ff - fc
print(f"difference (set - control), decoys that kept the reference's orientation ({int(stay.sum())}): median {np.median(d[stay]):.1f}, quartiles {np.percentile(d[stay], 25):.1f} .. {np.percentile(d[stay], 75):.1f}; lower in {100 * np.mean(d[stay] < 0):.0f} %")
names = ["dist", "omega", "theta", "phi", "vdw", "rama", "omega_bb", "bonded", "hbond"]
print("   by term (median difference, unweighted): " + ", ".join(f"{names[k]} {np.median((ef - ec)[stay, k]):.2f}" for k in range(min(9, ef.shape[1]))))
ctx.close()
