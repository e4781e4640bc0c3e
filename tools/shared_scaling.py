"""Shared launches against the number of folds in flight: N contexts (each with ITS OWN copy of the tables of one synthetic map, as
the chains of a batch job have), one single-decoy fold each at the same time, a fixed evaluation budget.
usage: [SCALING_WAVES=1|4] shared_scaling.py <repo> <L> <evals> <N> [N ...]   (SCALING_WAVES: Context.set_single_decoy_waves, default 4)     prints one JSON line per N: wall, launch pairs, folds per launch,
microseconds per launch pair and per fold-evaluation.  Under rocprofv3 --kernel-trace --stats the kernel averages belong to the LAST N."""
import importlib, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth"); LB = importlib.import_module("trrosettax2-dynamics_amd._lib")
L, evals = int(sys.argv[2]), int(sys.argv[3])
m = S.make_map(L, seed=L)
runs = T.protocol.build_runs(L, 2, fastrelax=True)
rng = np.random.default_rng(1)
for N in [int(x) for x in sys.argv[4:]]:
    ctxs = [T.Context(0) for _ in range(N)]
    for c in ctxs:
        c.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
        c.set_single_decoy_waves(int(os.environ.get("SCALING_WAVES", "4")))
    t0s = [(m["tors"] + rng.normal(size=m["tors"].shape) * 0.3).astype(np.float32)[None] for _ in range(N)]   # near the structure: the state a fold spends its time in
    for c, t in zip(ctxs, t0s):
        c.fold_batch(1, runs[5:], tors0=t, max_evals=20)                         # buffers, engines started
    s0 = LB.shared_launch_stats(0)
    out = [None] * N
    def work(i):
        out[i] = ctxs[i].fold_batch(1, runs[5:], tors0=t0s[i], max_evals=evals)
    th = [threading.Thread(target=work, args=(i,)) for i in range(N)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; el = time.perf_counter() - t0
    s1 = LB.shared_launch_stats(0)
    chunks = s1["chunks"] - s0["chunks"]; jobs = s1["chunks"] * s1["folds_per_launch"] - s0["chunks"] * s0["folds_per_launch"]
    ev = sum(int(r["n_evals"][0]) for r in out)
    print(json.dumps(dict(L=L, waves_per_row=int(os.environ.get("SCALING_WAVES", "4")), folds=N, evals_each=evals, wall_s=round(el, 4), launch_pairs=int(chunks * 16), folds_per_launch=round(jobs / max(chunks, 1), 2),
                          us_per_fold_eval=round(1e6 * el / ev, 3), fold_evals_per_s=round(ev / el))), flush=True)
    for c in ctxs:
        c.close()
