"""Which minimiser runs need the tight convergence tolerance?  Outcome (C-alpha RMSD to the reference's PyRosetta decoys of the same
map, as tools/outcome_sample.py) and evaluations per decoy for per-run tolerance schedules of the mode-2 protocol:
runs 0-4 declash (sf_vdw), 5-7 the repeated sf run, 8 the Cartesian run, 9-13 declash (sf1).
usage: tol_sweep.py <repo> [n_batches of 64 = 8] [first seed = 1000] [part: cum | sweep | all]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden"); dec = np.load(os.path.join(g, "ref_decoys.npz"))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
part = sys.argv[4] if len(sys.argv) > 4 else "all"


def rmsd(P, Q):
    P = P - P.mean(0); Q = Q - Q.mean(0)
    U, S_, Vt = np.linalg.svd(P.T @ Q)
    d = np.sign(np.linalg.det(U @ Vt))
    return float(np.sqrt(max(0.0, ((P ** 2).sum() + (Q ** 2).sum() - 2 * (S_[0] + S_[1] + d * S_[2])) / len(P))))


seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
ctx = T.Context(0)
MAPS = (("NMR", ("conf_2_1", "conf_2_2")), ("Xray", ("conf_1_1", "conf_1_2")))


def sample(tag, refs, runs, nb):
    m = np.load(os.path.join(g, f"seq_{tag}.npz")); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    rm, tw, ev, fe = [], [], [], []
    for b in range(nb):
        r = ctx.fold_batch(64, runs, seed=seed0 + b)
        ev += list(r["n_evals"]); fe += list(r["f"]) if "f" in r else []
        for i in range(64):
            ca = r["xyz"][i, :, 1].astype(np.float64)
            rm.append(min(rmsd(ca, dec[k][:, 1]) for k in refs))
            dw = np.degrees(np.abs((r["tors"][i, :-1, 2] % (2 * np.pi)) - np.pi)); tw.append(dw.max() > 60)
    return np.array(rm), np.array(tw), np.array(ev, float)


def schedule(tols):
    runs = T.protocol.build_runs(90, 2)
    assert len(runs) == len(tols)
    for r, t in zip(runs, tols): r["tol"] = t
    return runs


if part in ("cum", "all"):
    # evaluations spent per run: mean evaluations of the protocol cut after k runs
    runs = T.protocol.build_runs(90, 2)
    for tag, refs in MAPS:
        prev = 0.0; out = []
        for k in range(1, len(runs) + 1):
            _, _, ev = sample(tag, refs, runs[:k], 2)
            out.append(ev.mean() - prev); prev = ev.mean()
        print(f"{tag}: mean evaluations per run: " + " ".join(f"{x:.0f}" for x in out) + f"  total {prev:.0f}", flush=True)

if part in ("sweep", "all"):
    t6, t5, t4 = 1e-6, 1e-5, 1e-4
    S = {
        "all 1e-6 (default)": [t6] * 14,
        "all 3e-6": [3e-6] * 14,
        "all 1e-5": [t5] * 14,
        "all 1e-4 (reference's value)": [t4] * 14,
        "first declash 1e-4": [t4] * 5 + [t6] * 9,
        "first declash 1e-4, sf x2 1e-5": [t4] * 5 + [t5, t5, t6] + [t6] + [t6] * 5,
        "first declash 1e-4, sf x2 1e-4": [t4] * 5 + [t4, t4, t6] + [t6] + [t6] * 5,
        "first declash 1e-4, sf x3 1e-4, cart 1e-6": [t4] * 5 + [t4, t4, t4] + [t6] + [t6] * 5,
        "all 1e-4 but the last declash": [t4] * 9 + [t6] * 5,
        "all 1e-4 but cart + last declash 1e-5": [t4] * 8 + [t5] + [t5] * 5,
        "first declash 1e-4, sf x2 1e-4, last declash 1e-5 x4 + 1e-6": [t4] * 5 + [t4, t4, t6] + [t6] + [t5] * 4 + [t6],
    }
    for name, tols in S.items():
        line = f"{name:58s}"
        for tag, refs in MAPS:
            rm, tw, ev = sample(tag, refs, schedule(tols), nb)
            line += f" | {tag} n={len(rm)} med {np.median(rm):.3f} <=0.5 {100*(rm<=0.5).mean():.0f}% <=1 {100*(rm<=1).mean():.0f}% >3 {100*(rm>3).mean():.1f}% tw {100*tw.mean():.0f}% ev {ev.mean():.0f}"
        print(line, flush=True)
ctx.close()
