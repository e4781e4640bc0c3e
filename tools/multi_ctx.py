"""One call's decoys split over n contexts (streams) folding concurrently from n host threads: what more than two lanes would give.
usage: multi_ctx.py <repo> <config 2|4> <n contexts> <K calls>"""
import importlib, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
CFG = {2: (150, 64, False), 3: (150, 64, True), 4: (400, 32, True)}
cfg, n, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
L, B, orient = CFG[cfg]
B = int(os.environ.get("MULTI_B", B))   # MULTI_B: decoys per call instead of the config's
runs = T.protocol.build_runs(L, 2)
m = S.make_map(L, seed=L)
ctxs = []
for c in range(n):
    x = T.Context(0, lanes=1); x.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"]); ctxs.append(x)
per = B // n
ex = ThreadPoolExecutor(max_workers=n)
def step(i):
    return list(ex.map(lambda c: ctxs[c].fold_batch(per, runs, seed=150, decoy0=i * B + c * per), range(n)))
step(900)
t0 = time.perf_counter(); rs = [r for i in range(K) for r in step(i)]; el = time.perf_counter() - t0
ev = np.concatenate([r["n_evals"] for r in rs])
print(json.dumps(dict(config=cfg, contexts=n, decoys_per_context=per, hwq=os.environ.get("GPU_MAX_HW_QUEUES"), calls=K, decoys_per_sec=round(K * per * n / el, 1),
                      ms_per_call=round(1e3 * el / K, 1), evals_max=int(ev.max()), ok=bool(all(np.all(r["status"] == 0) for r in rs)))))
for x in ctxs:
    x.close()
