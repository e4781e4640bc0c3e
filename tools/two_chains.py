"""Two independent chains (two contexts = two streams) folded concurrently from two host threads, as the NMR and X-ray
chains of run_inference.py:310-318 are independent.  Checks results against solo runs and reports aggregate throughput.
usage: two_chains.py <repo> [steps]"""
import importlib, sys, threading, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B = 150, 64; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
maps = [S.make_map(L, seed=L), S.make_map(L, seed=L + 1)]   # second map = seed L+1
runs = T.protocol.build_runs(L, 2)
ctxs = [T.Context(0), T.Context(0)]
for c, m in zip(ctxs, maps): c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
def work(i, out): out[i] = [ctxs[i].fold_batch(B, runs, seed=10 + i, decoy0=k * B) for k in range(steps)]
# solo
solo = [None, None]; t0 = time.perf_counter(); work(0, solo); t1 = time.perf_counter(); work(1, solo); t2 = time.perf_counter()
# concurrent
both = [None, None]; th = [threading.Thread(target=work, args=(i, both)) for i in range(2)]
t3 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t4 = time.perf_counter()
same = all(np.array_equal(a["xyz"], b["xyz"]) and np.array_equal(a["n_evals"], b["n_evals"]) for i in range(2) for a, b in zip(solo[i], both[i]))
ok = all(np.all(r["status"] == 0) for i in range(2) for r in both[i])
n = steps * B
print(f"solo: chain A {n/(t1-t0):.0f} decoys/s, chain B {n/(t2-t1):.0f} decoys/s, one after the other {2*n/(t2-t0):.0f} decoys/s")
print(f"concurrent (2 contexts, 2 threads): {2*n/(t4-t3):.0f} decoys/s  -> x{(t2-t0)/(t4-t3):.2f};  results bitwise identical to the solo runs: {same};  all converged: {ok}")
for c in ctxs: c.close()
