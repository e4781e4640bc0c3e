"""How many decoy slots should a job of N decoys use?  N decoys (L=150, distances only, full protocol) on 2 lanes x S slots.
usage: pool_sweep.py <repo> [N=512] [orient]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
orient = len(sys.argv) > 3
L = 150
m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
ctx = T.Context(0, lanes=2)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
for slots in (32, 48, 64, 96, 128, 192, 256):
    ctx.set_pool(slots)
    ctx.fold_batch(2 * slots, runs, seed=150, decoy0=900 * 64)
    v = []
    for rep in range(2):
        t0 = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=150, decoy0=0); v.append(N / (time.perf_counter() - t0))
        assert np.all(r["status"] == 0)
    print(f"   2 lanes x {slots:3d} slots: {max(v):6.1f} decoys/s  ({r['launches']} launch pairs per lane, slot efficiency {r['slot_efficiency']:.2f})", flush=True)
ctx.close()
