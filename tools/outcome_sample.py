"""Large-sample outcome parity on the reference's example maps: C-alpha RMSD of folded decoys to the reference's PyRosetta decoys
(the closer of the two initial decoys of the same map), full protocol.  TRX2FOLD_LIB selects the build (model-constant A/B).
usage: outcome_sample.py <repo> [n_batches of 64 = 16] [first seed = 1000] [options, e.g. "--fastrelax"]"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
g = os.path.join(sys.argv[1], "tests", "golden"); dec = np.load(os.path.join(g, "ref_decoys.npz"))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
relax = len(sys.argv) > 4 and "fastrelax" in sys.argv[4]


def rmsd(P, Q):
    P = P - P.mean(0); Q = Q - Q.mean(0)
    U, S_, Vt = np.linalg.svd(P.T @ Q)
    d = np.sign(np.linalg.det(U @ Vt))
    return float(np.sqrt(max(0.0, ((P ** 2).sum() + (Q ** 2).sum() - 2 * (S_[0] + S_[1] + d * S_[2])) / len(P))))


seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
ctx = T.Context(0)
for tag, refs in (("NMR", ("conf_2_1", "conf_2_2")), ("Xray", ("conf_1_1", "conf_1_2"))):
    m = np.load(os.path.join(g, f"seq_{tag}.npz")); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    runs = T.protocol.build_runs(90, 2, fastrelax=True) if relax else T.protocol.build_runs(90, 2)
    if relax and os.environ.get("OUTCOME_CLOSING_ITERS"):      # experiment: the closing unrestrained minimisation shortened (0 = dropped)
        n_it = int(os.environ["OUTCOME_CLOSING_ITERS"])
        runs = runs[:-1] if n_it == 0 else runs[:-1] + [dict(runs[-1], max_iter=n_it)]
    if relax and os.environ.get("OUTCOME_CLOSING_RAMA"):       # experiment: the torsion-potential weight of the closing run scaled
        w = list(runs[-1]["w"]); w[4] *= float(os.environ["OUTCOME_CLOSING_RAMA"])
        runs = runs[:-1] + [dict(runs[-1], w=w)]
    rm, mir, tw, ev, sec, cas, xyzs = [], [], [], [], 0.0, [], []
    for b in range(nb):
        r = ctx.fold_batch(64, runs, seed=seed0 + b)
        assert np.all(r["status"] == 0)
        sec += r["seconds"]; ev += list(r["n_evals"])
        for i in range(64):
            ca = r["xyz"][i, :, 1].astype(np.float64)
            cas.append(ca); xyzs.append(r["xyz"][i])
            rm.append(min(rmsd(ca, dec[k][:, 1]) for k in refs)); mir.append(min(rmsd(ca * [1, 1, -1], dec[k][:, 1]) for k in refs))
            dw = np.degrees(np.abs((r["tors"][i, :-1, 2] % (2 * np.pi)) - np.pi)); tw.append(dw.max() > 60)
    rm, mir = np.array(rm), np.array(mir); n = len(rm); gross = rm > 3
    print(f"{os.path.basename(os.environ.get('TRX2FOLD_LIB', 'default')):28s} {tag:4s} n={n}: RMSD median {np.median(rm):.3f}  quartiles {np.percentile(rm,25):.2f}-{np.percentile(rm,75):.2f}  "
          f"<=0.5A {100*(rm<=0.5).mean():.0f}%  <=1A {100*(rm<=1).mean():.0f}%  >3A {100*gross.mean():.1f}% (mirror {100*(gross&(mir<rm)).mean():.1f}%)  "
          f"twisted>60 {100*np.mean(tw):.0f}%  evals median {np.median(ev):.0f} mean {np.mean(ev):.0f}  {n/sec:.0f} decoys/s")
    # Two-sample reading: the reference's two decoys of this map are two draws of ITS distribution; how far apart are two draws of
    # OURS?  (non-mirror decoys only -- the reference's pair holds none; 300 decoys -> 44 850 pairs)
    ok = [c for c, g_ in zip(cas, gross) if not g_][:300]
    pw = np.array([rmsd(ok[i], ok[j]) for i in range(len(ok)) for j in range(i + 1, len(ok))])
    ref_pair = rmsd(dec[refs[0]][:, 1], dec[refs[1]][:, 1])
    to_ref = np.array([[rmsd(c, dec[k][:, 1]) for k in refs] for c in ok])
    print(f"{'':28s} {tag:4s} two draws of this build: median {np.median(pw):.3f} A (5-95 %: {np.percentile(pw, 5):.2f}-{np.percentile(pw, 95):.2f}); "
          f"the reference's two draws: {ref_pair:.3f} A = percentile {100 * (pw < ref_pair).mean():.0f} of ours; "
          f"a draw of ours to ONE reference draw: median {np.median(to_ref):.3f} A ({refs[0]} {np.median(to_ref[:, 0]):.3f}, {refs[1]} {np.median(to_ref[:, 1]):.3f})")
    # Torsion-level agreement (global superposition says nothing about the local angles): mean absolute circular difference of
    # phi / psi over the residues, between a draw of ours and the reference's draws, between two of ours, between the reference's two
    def phipsi(xyz):           # xyz[L, 5, 3] (N, CA, C, O, CB) -> [L - 2, 2] for residues 1 .. L - 2
        N_, CA_, C_ = xyz[:, 0].astype(np.float64), xyz[:, 1].astype(np.float64), xyz[:, 2].astype(np.float64)
        def dih(a, b, c, d_):
            b0, b1, b2 = a - b, c - b, d_ - c
            b1 = b1 / np.linalg.norm(b1, axis=-1, keepdims=True)
            v = b0 - (b0 * b1).sum(-1, keepdims=True) * b1; w = b2 - (b2 * b1).sum(-1, keepdims=True) * b1
            return np.arctan2((np.cross(b1, v) * w).sum(-1), (v * w).sum(-1))
        return np.stack([dih(C_[:-2], N_[1:-1], CA_[1:-1], C_[1:-1]), dih(N_[1:-1], CA_[1:-1], C_[1:-1], N_[2:])], 1)
    cdiff = lambda a, b: np.degrees(np.abs((a - b + np.pi) % (2 * np.pi) - np.pi)).mean()
    pp_ref = [phipsi(dec[k]) for k in refs]
    pp = [phipsi(x) for x, g_ in zip(xyzs, gross) if not g_][:300]
    to = np.array([min(cdiff(q, pr) for pr in pp_ref) for q in pp])
    rng = np.random.default_rng(0); ij = rng.integers(0, len(pp), size=(4000, 2)); ij = ij[ij[:, 0] != ij[:, 1]]
    own2 = np.array([cdiff(pp[i], pp[j]) for i, j in ij])
    print(f"{'':28s} {tag:4s} phi/psi mean |difference| over residues: ours to the closer reference draw median {np.median(to):.1f} deg, two of ours {np.median(own2):.1f} deg, "
          f"the reference's two {cdiff(pp_ref[0], pp_ref[1]):.1f} deg")
ctx.close()
