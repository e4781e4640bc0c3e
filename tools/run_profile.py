"""Round 5: where the evaluations of the default protocol (35 runs) go.  The protocol cut after k runs, k = 1..35: mean evaluations and accepted
iterations per run (differences of the means), energy under the full score after each cut is not comparable (weights change), so only counts.
usage: run_profile.py <repo> [decoys = 256] [L = 90 | 150 (synthetic, dist-only) | 151 (synthetic L=150, all channels)]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Lc = int(sys.argv[3]) if len(sys.argv) > 3 else 90
g = os.path.join(sys.argv[1], "tests", "golden")
ctx = T.Context(0, lanes=2)
if Lc == 90:
    seq = "".join(l.strip() for l in open(os.path.join(g, "seq.fasta")) if not l.startswith(">"))
    cases = []
    for tag in ("NMR", "Xray"):
        m = np.load(os.path.join(g, f"seq_{tag}.npz")); cases.append((tag, 90, (m["dist"], m["omega"], m["theta"], m["phi"]), seq))
else:
    L = 150; m = S.make_map(L)
    cases = [("synthetic L=150 " + ("dist-only" if Lc == 150 else "all channels"), L, (m["dist"],) if Lc == 150 else (m["dist"], m["omega"], m["theta"], m["phi"]), m["seq"])]
for tag, L, arrs, seq in cases:
    ctx.set_map(*arrs, seq=seq)
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    pe = pi = 0.0
    print(f"{tag}: run | weights (pair dih ang vdw rama omega bonded hb) | cart filter tol max_iter | mean evaluations, accepted iterations of the run | evals/iter")
    for k in range(1, len(runs) + 1):
        r = ctx.fold_batch(n, runs[:k], seed=1000)
        e, i = r["n_evals"].mean(), r["n_iters"].mean()
        q = runs[k - 1]
        print(f"  {k-1:2d} | {' '.join('%4.2f' % w for w in q['w'])} | {q['cartesian']} {q['pair_filter']} {q['tol']:.0e} {q['max_iter']:4d} pre={q['precheck']} | {e-pe:7.1f} {i-pi:7.1f} | {(e-pe)/max(i-pi,1e-9):5.2f}", flush=True)
        pe, pi = e, i
    print(f"  total {pe:.0f} evaluations, {pi:.0f} iterations, {pe/pi:.3f} evaluations per iteration")
ctx.close()
