"""Round 5: where do the STRAGGLERS of a 64-decoy call spend their evaluations?  (The call lasts as long as its slowest decoy.)  Config 2 (L=150,
distances only): cumulative evaluations per decoy after k runs of the default protocol, for the slowest and the median decoys.
usage: straggler_profile.py <repo>"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = 150; m = S.make_map(L)
ctx = T.Context(0, lanes=2); ctx.set_map(m["dist"], seq=m["seq"])
runs = T.protocol.build_runs(L, 2, fastrelax=True)
cuts = [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 18, 22, 26, 30, 34, 35]
ev = {}
for k in cuts:
    r = ctx.fold_batch(64, runs[:k], seed=150, decoy0=0)
    ev[k] = r["n_evals"].astype(int); it = r["n_iters"].astype(int)
tot = ev[35]; order = np.argsort(-tot)
print("evaluations of the whole protocol: min %d median %d max %d" % (tot.min(), np.median(tot), tot.max()))
print("decoy | total | per segment: " + " ".join(f"r<{k}" for k in cuts))
for d in list(order[:6]) + list(order[30:33]) + list(order[-2:]):
    prev = 0; seg = []
    for k in cuts:
        seg.append(ev[k][d] - prev); prev = ev[k][d]
    print(f"{d:5d} | {tot[d]:5d} | " + " ".join(f"{x:5d}" for x in seg))
ctx.close()
