"""A/B timing of one build: the bench's queue (K x 64 decoys, L=150, distances only) on 2 lanes x 32 slots, 1 x 64, 2 x 64; best and
median of R repeats.  Run it once per library in the SAME gpurun call (TRX2FOLD_LIB=...) -- boxes differ by a few per cent.
usage: ab_fold.py <repo> [K=4] [R=5]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L, B = 150, 64
m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
ctx = T.Context(0, lanes=2); ctx.set_map(m["dist"], seq=m["seq"])
out = []
for name, lanes, pool in (("2x32", 2, 32), ("1x64", 1, 64), ("2x64", 2, 64)):
    ctx.set_lanes(lanes); ctx.set_pool(pool)
    ctx.fold_batch(B, runs, seed=150, decoy0=900 * B)
    v = []
    for i in range(R):
        t0 = time.perf_counter(); r = ctx.fold_batch(K * B, runs, seed=150, decoy0=0); v.append(K * B / (time.perf_counter() - t0))
        assert np.all(r["status"] == 0)
    out.append(f"{name}: best {max(v):6.1f} median {np.median(v):6.1f} ({r['launches']} launches, {r['n_evals'].mean():.0f} evals/decoy, {r['n_iters'].mean():.0f} iters)")
print(" | ".join(out), "decoys/s")
ctx.close()
