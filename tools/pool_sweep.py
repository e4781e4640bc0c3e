"""Throughput of one queue of decoys against the number of slots (trx2_ctx_set_pool) and lanes.  usage: pool_sweep.py <repo> <config> <n decoys>"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, orient = {2: (150, 64, False), 3: (150, 64, True), 4: (400, 32, True)}[int(sys.argv[2])]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5 * B
m = S.make_map(L); runs = T.protocol.build_runs(L, 2)
for lanes in (1, 2):
    for pool in (B // 2, B, 3 * B // 2, 2 * B, 4 * B):
        ctx = T.Context(0, lanes=lanes, pool=pool)
        ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
        ctx.fold_batch(min(N, 2 * pool), runs, seed=1, decoy0=10000)
        t = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=150); el = time.perf_counter() - t
        print(f"config {sys.argv[2]} N={N} lanes={lanes} slots per lane={pool:4d}: {N/el:7.1f} decoys/s  slot efficiency {r['slot_efficiency']:.3f}  launches {r['launches']}  ok {bool(np.all(r['status']==0))}", flush=True)
        ctx.close()
