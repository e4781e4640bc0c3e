"""Why do bench.py's sub-records run 1.65 x slower than the same calls in a fresh process?  Times two concurrent 64-decoy chains
(config 3) at several points of a process that does what bench.py does in between.  usage: bench_slowdown_probe.py <repo>"""
import contextlib, importlib, io, json, os, sys, threading, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, sys.argv[1])
import bench
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
pipe = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
L, B = 150, 64
runs = T.protocol.build_runs(L, 2)
maps = [S.make_map(L, seed=L + c) for c in range(2)]


def probe(tag):
    ctxs = [T.Context(0) for _ in range(2)]
    for c, m in zip(ctxs, maps):
        c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    one = lambda i: ctxs[i].fold_batch(B, runs, seed=150 + i, decoy0=0)
    with ThreadPoolExecutor(max_workers=2) as ex:
        list(ex.map(one, range(2)))
        t0 = time.perf_counter(); list(ex.map(one, range(2))); el = time.perf_counter() - t0
    for c in ctxs:
        c.close()
    print(json.dumps(dict(at=tag, seconds=round(el, 4), decoys_per_s=round(2 * B / el, 1), threads=threading.active_count(), load=os.getloadavg()[0])), flush=True)


probe("fresh process")
from oracle import oracle as O
m = S.make_map(L, seed=L)
Tb = O.Tables(m["dist"])
O.fold_batch(Tb, np.stack([O.random_torsions(L, 1, d) for d in range(16)]), runs, nthreads=16, max_evals=400)
probe("after an OpenMP region of the CPU oracle (16 threads)")
time.sleep(3)
probe("3 s later")
with contextlib.redirect_stdout(io.StringIO()):
    bench.e2e_leg(pipe, S, L, 10)
probe("after the e2e leg (run_single, two chain threads, contexts cached per thread)")
