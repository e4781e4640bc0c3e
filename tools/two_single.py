"""Single-decoy folds on T concurrent host threads / contexts (the NMR and X-ray chains of run_inference's iteration phase):
does a chain slow the other down?  usage: two_single.py <repo> <L> <threads> <folds per thread>   (env TRX2_GRAPH=1: graph replay)"""
import importlib, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, nt, n = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
runs = T.protocol.build_runs(L, 2)
ctxs = []
for c in range(nt):
    m = S.make_map(L, seed=L + c); x = T.Context(0)
    x.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"]); ctxs.append(x)
def chain(c):
    ctxs[c].fold_batch(1, runs, seed=5 + c)
    t0 = time.perf_counter(); ev = 0
    for k in range(n):
        r = ctxs[c].fold_batch(1, runs, seed=7 + c, decoy0=k); ev += int(r["n_evals"][0])
    return time.perf_counter() - t0, ev
t0 = time.perf_counter()
with ThreadPoolExecutor(max_workers=nt) as ex:
    res = list(ex.map(chain, range(nt)))
wall = time.perf_counter() - t0
print(json.dumps(dict(L=L, threads=nt, graph=os.environ.get("TRX2_GRAPH"), hwq=os.environ.get("GPU_MAX_HW_QUEUES"), folds_per_thread=n,
                      ms_per_fold=[round(1e3 * t / n, 2) for t, _ in res], us_per_eval=[round(1e6 * t / e, 2) for t, e in res], wall_s=round(wall, 3))))
for x in ctxs:
    x.close()
