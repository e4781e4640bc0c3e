"""The bench's queue (K x 64 decoys, L=150) over NL concurrent contexts (one stream each) of S slots each: how many streams, how wide?
usage: lanes_sweep.py <repo> [orient]   (GPU_MAX_HW_QUEUES in the environment changes what more than 4 streams do)"""
import importlib, sys, threading, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
orient = len(sys.argv) > 2
L, B, K = 150, 64, 4
m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
def mk(slots):
    c = T.Context(0, pool=slots)
    c.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
    return c
for NL, slots in ((1, 64), (2, 32), (3, 21), (4, 16), (8, 8), (2, 64), (3, 43), (4, 32), (8, 16)):
    cs = [mk(slots) for _ in range(NL)]
    parts = np.array_split(np.arange(K * B), NL)
    ok = [True] * NL
    def work(i, d0):
        r = cs[i].fold_batch(len(parts[i]), runs, seed=150, decoy0=d0 + int(parts[i][0])); ok[i] = bool(np.all(r["status"] == 0))
    def job(d0):
        th = [threading.Thread(target=work, args=(i, d0)) for i in range(NL)]
        [t.start() for t in th]; [t.join() for t in th]
    job(900 * B)
    v = []
    for rep in range(3):
        t0 = time.perf_counter(); job(0); v.append(K * B / (time.perf_counter() - t0))
    print(f"   {NL} streams x {slots:2d} slots ({NL*slots:3d} in flight): best {max(v):6.1f} median {np.median(v):6.1f} decoys/s  ok {all(ok)}", flush=True)
    for x in cs: x.close()
