import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = 150; m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
for lanes, slots in ((1, 320), (1, 192), (2, 160)):
    ctx = T.Context(0, lanes=lanes, pool=slots); ctx.set_map(m["dist"], seq=m["seq"])
    ctx.fold_batch(320, runs, seed=150, decoy0=900 * 64)
    v = []
    for rep in range(2):
        t0 = time.perf_counter(); r = ctx.fold_batch(320, runs, seed=150, decoy0=0); v.append(320 / (time.perf_counter() - t0))
    print(f"{lanes} lane(s) x {slots}: {max(v):.1f} decoys/s, {r['launches']} launch pairs")
    ctx.close()
