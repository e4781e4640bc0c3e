"""Average step / pair launch of a fold, sampled live with events (Context.set_profiling): usage step_time.py <repo> <L> <B> <max_evals> [lanes=1]"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, ne = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lanes = int(sys.argv[5]) if len(sys.argv) > 5 else 1
m = S.make_map(L, seed=L); ctx = T.Context(0, lanes=lanes)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
runs = T.protocol.build_runs(L, 2, fastrelax=True)
ctx.fold_batch(B, runs, seed=3, max_evals=200)
ctx.set_profiling(4)
r = ctx.fold_batch(B, runs, seed=3, max_evals=ne)
p, s_, n = ctx.last_fold_kernel_times()
print(f"L={L} B={B} lanes={lanes} evals {int(np.median(r['n_evals']))}: pair {p*1e3:.2f} us, step {s_*1e3:.2f} us over {n} samples; fold {r['seconds']*1e3:.0f} ms")
ctx.close()
