"""Throughput of K separate trx2_fold_batch calls of a bench config (what bench.py's `value` times), for A/B runs of library
defaults.  usage: percall.py <repo> <config 2|3|4> <lanes 1|2> <K> [compaction mode]"""
import importlib, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
CFG = {2: (150, 64, False, 1), 3: (150, 64, True, 2), 4: (400, 32, True, 1)}
cfg, lanes, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 1
L, B, orient, nch = CFG[cfg]
runs = T.protocol.build_runs(L, 2, cartesian_stage=False if os.environ.get("PERCALL_NOCART") else None, fastrelax=not os.environ.get("PERCALL_NORELAX"))   # PERCALL_NOCART=1: the Cartesian run in torsion space (A/B of the step launch)
ctxs = []
for c in range(nch):
    m = S.make_map(L, seed=L + c); x = T.Context(0, lanes=lanes)
    x.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"]); x.set_tail_compaction(mode); ctxs.append(x)
def step(i):
    one = lambda c: ctxs[c].fold_batch(B, runs, seed=150 + c, decoy0=i * B)
    if nch == 1:
        return [one(0)]
    with ThreadPoolExecutor(max_workers=nch) as ex:
        return list(ex.map(one, range(nch)))
step(900)
t0 = time.perf_counter(); rs = [r for i in range(K) for r in step(i)]; el = time.perf_counter() - t0
ev = np.concatenate([r["n_evals"] for r in rs])
print(json.dumps(dict(config=cfg, lanes=lanes, compaction=mode, hwq=os.environ.get("GPU_MAX_HW_QUEUES"), calls=K, decoys_per_sec=round(K * B * nch / el, 1),
                      ms_per_call=round(1e3 * el / K, 1), evals_median=float(np.median(ev)), evals_max=int(ev.max()), launches_per_call=sum(r["launches"] for r in rs) / K / nch,
                      ok=bool(all(np.all(r["status"] == 0) for r in rs)))))
for x in ctxs:
    x.close()
