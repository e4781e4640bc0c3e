"""N decoys (L=150, distances only) over NL concurrent contexts (one stream, one host thread each) of S slots: does a third or fourth
stream pay with the current kernels and the tail compaction?  usage: lanes_sweep2.py <repo> [N=320]   (try GPU_MAX_HW_QUEUES=8 too)"""
import importlib, sys, threading, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 320
L = 150; m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
def mk(slots):
    c = T.Context(0, pool=slots); c.set_map(m["dist"], seq=m["seq"]); return c
for NL, slots in ((1, 320), (2, 160), (3, 107), (4, 80), (4, 64), (6, 54), (8, 40)):
    cs = [mk(slots) for _ in range(NL)]
    parts = np.array_split(np.arange(N), NL)
    def work(i, d0):
        cs[i].fold_batch(len(parts[i]), runs, seed=150, decoy0=d0 + int(parts[i][0]))
    def job(d0):
        th = [threading.Thread(target=work, args=(i, d0)) for i in range(NL)]
        [t.start() for t in th]; [t.join() for t in th]
    job(900 * 64)
    v = []
    for rep in range(3):
        t0 = time.perf_counter(); job(0); v.append(N / (time.perf_counter() - t0))
    print(f"   {NL} streams x {slots:3d} slots: best {max(v):6.1f} median {np.median(v):6.1f} decoys/s", flush=True)
    for x in cs: x.close()
