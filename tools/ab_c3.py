"""A/B at config 3's shape: 128 decoys (L=150, all four channels) on one lane of 128 slots, and 320 on 2 x 160.  usage: ab_c3.py <repo> [R=3]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 150; m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
out = []
for lanes, N, slots in ((1, 128, 128), (2, 320, 160)):
    ctx = T.Context(0, lanes=lanes, pool=slots); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    ctx.fold_batch(N, runs, seed=150, decoy0=900 * 64)
    v = []
    for i in range(R):
        t0 = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=150, decoy0=0); v.append(N / (time.perf_counter() - t0))
    out.append(f"{N} on {lanes}x{slots}: {max(v):6.1f}")
    ctx.close()
print(" | ".join(out), "decoys/s")
