"""One queue of N decoys on two lanes x S slots (the slot pool refills on the device): decoys/s against S.
usage: pool_sweep.py <repo> <config 2|3> <N> [S ...]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S_ = importlib.import_module("trrosettax2-dynamics_amd.synth")
cfg, N = int(sys.argv[2]), int(sys.argv[3])
slots = [int(x) for x in sys.argv[4:]] or [128, 192, 256, 320]
L, orient = int(os.environ.get("POOL_L", "150")), cfg == 3   # POOL_L: another chain length
m = S_.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
for s in slots:
    ctx = T.Context(0, lanes=2, pool=s)
    ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
    ctx.fold_batch(2 * s, runs, seed=5)
    t0 = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=7); el = time.perf_counter() - t0
    print(json.dumps(dict(config=cfg, N=N, slots_per_lane=s, decoys_per_sec=round(N / el, 1), slot_efficiency=round(float(r["slot_efficiency"]), 3), ok=bool(np.all(r["status"] == 0)))), flush=True)
    ctx.close()
