"""A/B of one build at the bench's headline shape: 320 decoys (L=150, distances only) on 2 lanes x 160 slots, and 256 on 2 x 32; best of R.
Run once per library in the SAME gpurun call (TRX2FOLD_LIB=...).  usage: ab_wide.py <repo> [R=3]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 150; m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
ctx = T.Context(0, lanes=2); ctx.set_map(m["dist"], seq=m["seq"])
out = []
for N, slots in ((320, 160), (256, 32), (1024, 192)):
    ctx.set_pool(slots)
    ctx.fold_batch(min(N, 2 * slots), runs, seed=150, decoy0=900 * 64)
    v = []
    for i in range(R):
        t0 = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=150, decoy0=0); v.append(N / (time.perf_counter() - t0))
        assert np.all(r["status"] == 0)
    out.append(f"{N} on 2x{slots}: {max(v):6.1f} ({r['launches']} launch pairs)")
print(" | ".join(out), "decoys/s")
ctx.close()
