"""Do two concurrently folding contexts slow down depending on how many contexts (streams) the process created before them?
(HIP maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues in creation order: two streams on one queue serialize.)
usage: queue_collision.py <repo> [L=150] [B=64]"""
import importlib, json, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
runs = T.protocol.build_runs(L, 2)
maps = [S.make_map(L, seed=L + c) for c in range(2)]
held = []
for k in range(7):
    ctxs = [T.Context(0) for _ in range(2)]
    for c, m in zip(ctxs, maps):
        c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    def one(i):
        return ctxs[i].fold_batch(B, runs, seed=150 + i, decoy0=0)
    with ThreadPoolExecutor(max_workers=2) as ex:
        list(ex.map(one, range(2)))
        t0 = time.perf_counter(); rs = list(ex.map(one, range(2))); el = time.perf_counter() - t0
    print(json.dumps(dict(contexts_created_before=len(held), seconds=round(el, 4), decoys_per_s=round(2 * B / el, 1), launches=[r["launches"] for r in rs])))
    for c in ctxs:
        c.close()
    held.append(T.Context(0))   # one more stream stays alive
