"""Does splitting ONE job of 64 decoys over K concurrent contexts (streams) beat one batch of 64?
One stream's serial step kernel can then overlap another's pair kernel.  usage: split_batch.py <repo> [orient]"""
import importlib, sys, threading, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
orient = len(sys.argv) > 2
L = 150; m = S.make_map(L); runs = T.protocol.build_runs(L, 2); TOTAL = 64
def mk():
    c = T.Context(0)
    if orient: c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    else: c.set_map(m["dist"], seq=m["seq"])
    return c
print("orient" if orient else "dist-only", "L=150, 64 decoys split over K contexts, 4 jobs each:")
ref = None
for K in (1, 2, 3, 4, 6, 8):
    cs = [mk() for _ in range(K)]
    parts = np.array_split(np.arange(TOTAL), K)
    out = [None] * K
    def work(i, seed):
        out[i] = cs[i].fold_batch(len(parts[i]), runs, seed=seed, decoy0=int(parts[i][0]))
    def job(seed):
        th = [threading.Thread(target=work, args=(i, seed)) for i in range(K)]
        [t.start() for t in th]; [t.join() for t in th]
        return np.concatenate([o["xyz"] for o in out])
    job(1)
    t0 = time.perf_counter(); xs = [job(7 + k) for k in range(4)]; dt = time.perf_counter() - t0
    if ref is None: ref = xs
    same = all(np.array_equal(a, b) for a, b in zip(ref, xs))
    print(f"   K={K}: {4*TOTAL/dt:6.0f} decoys/s   ({dt/4*1e3:.0f} ms per job)  identical to K=1: {same}", flush=True)
    for x in cs: x.close()
