"""How to fill one MI355X with fold work: N concurrent contexts of 64 decoys vs one context with more decoys.
L=150, all channels, full protocol.  usage: fill_gpu.py <repo>"""
import importlib, sys, threading, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = 150; m = S.make_map(L); runs = T.protocol.build_runs(L, 2)
def mk():
    c = T.Context(0); c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"]); return c
print("one context, B decoys per batch (2 batches each):")
c = mk()
for B in (32, 64, 128, 256):
    c.fold_batch(B, runs, seed=1)  # warm (allocations)
    t0 = time.perf_counter(); r = [c.fold_batch(B, runs, seed=2 + k) for k in range(2)]; dt = time.perf_counter() - t0
    print(f"   B={B:3d}: {2*B/dt:6.0f} decoys/s   ({dt/2*1e3:.0f} ms per batch, converged {all(np.all(x['status']==0) for x in r)})")
c.close()
print("N concurrent contexts (threads), 64 decoys per batch, 3 batches each:")
for N in (1, 2, 3, 4):
    cs = [mk() for _ in range(N)]
    for x in cs: x.fold_batch(64, runs, seed=1)
    def work(i): [cs[i].fold_batch(64, runs, seed=5 + i * 10 + k) for k in range(3)]
    th = [threading.Thread(target=work, args=(i,)) for i in range(N)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print(f"   N={N}: {N*3*64/dt:6.0f} decoys/s")
    for x in cs: x.close()
