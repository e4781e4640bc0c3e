"""One chain of run_inference's loop (run_inference.py:97-139) with per-iteration timers: where does an iteration's time go?
usage: e2e_chain_profile.py <repo> <L> <init_num> <iterations>"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
P = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
L, N, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = S.make_map(L, seed=L); seq = m["seq"]
ctx = T.Context(0)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
runs = T.protocol.build_runs(L, 2)
t0 = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=7); t_init = time.perf_counter() - t0
xyz = r["xyz"][0]
rows = []
for it in range(iters):
    t0 = time.perf_counter()
    x, s_pdb = P.as_read_from_pdb(seq, xyz)
    t1 = time.perf_counter()
    d = ctx.feedback_step(x, s_pdb, 1.0, True)
    t2 = time.perf_counter()
    r = ctx.fold_batch(1, runs, seed=100 + it)
    t3 = time.perf_counter()
    xyz = r["xyz"][0]
    rows.append(dict(it=it, as_read_ms=1e3 * (t1 - t0), feedback_ms=1e3 * (t2 - t1), fold_ms=1e3 * (t3 - t2), fold_c_ms=1e3 * r["seconds"], evals=int(r["n_evals"][0]),
                     launches=int(r["launches"]), delta=d))
print(json.dumps(dict(L=L, init_num=N, init_s=t_init, rows=rows[:3] + rows[-2:], evals_by_block_of_50=[float(np.mean([q["evals"] for q in rows[k:k + 50]])) for k in range(0, len(rows), 50)],
                      delta_by_block_of_50=[float(np.mean([q["delta"] for q in rows[k:k + 50]])) for k in range(0, len(rows), 50)],
                      mean=dict(as_read_ms=np.mean([q["as_read_ms"] for q in rows]), feedback_ms=np.mean([q["feedback_ms"] for q in rows]), fold_ms=np.mean([q["fold_ms"] for q in rows]),
                                fold_c_ms=np.mean([q["fold_c_ms"] for q in rows]), evals=np.mean([q["evals"] for q in rows]), us_per_eval=1e3 * np.mean([q["fold_c_ms"] for q in rows]) / np.mean([q["evals"] for q in rows])))))
ctx.close()
