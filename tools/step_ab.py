"""Round 5: live launch durations of the pair and step kernels (HIP events on the fold's own stream, every 7th evaluation) for the three
shapes VERDICT r4 item 3 names: one decoy (L = 150, all channels), 32 decoys per launch (config 2, distances only), 16 per launch at L = 400.
TRX2FOLD_LIB selects the build.  usage: step_ab.py <repo> [repeats = 3]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tag = os.path.basename(os.environ.get("TRX2FOLD_LIB", "libtrx2fold.so"))
m150, m400 = S.make_map(150), S.make_map(400)
out = []
for name, m, orient, B in (("1 decoy L=150 all channels", m150, True, 1), ("32 decoys L=150 dist-only", m150, False, 32), ("16 decoys L=400 all channels", m400, True, 16)):
    ctx = T.Context(0)
    L = len(m["tors"])
    ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    ctx.fold_batch(B, runs, seed=1)
    ctx.set_profiling(7)
    ps, ss, ev, sec = [], [], [], []
    for k in range(rep):
        r = ctx.fold_batch(B, runs, seed=2 + k)
        p, s, n = ctx.last_fold_kernel_times()
        ps.append(p * 1e3); ss.append(s * 1e3); ev.append(r["n_evals"].max()); sec.append(r["seconds"])
    ctx.set_profiling(0)
    r = ctx.fold_batch(B, runs, seed=2)
    out.append(f"{name:30s} pair {np.mean(ps):6.2f} us  step {np.mean(ss):6.2f} us  (spread {np.ptp(ss):.2f}) | unprofiled fold: {r['seconds']*1e3:.1f} ms for {r['n_evals'].max()} evaluations = {r['seconds']*1e6/r['n_evals'].max():.2f} us per evaluation")
    ctx.close()
# the pooled shape (bench.py's pooled_queue leg): 1280 decoys on 2 lanes x 640 slots, the low-register step instantiation
ctx = T.Context(0, lanes=2); ctx.set_map(m150["dist"], seq=m150["seq"]); ctx.set_pool(640)
runs = T.protocol.build_runs(150, 2, fastrelax=True)
ctx.fold_batch(1280, runs, seed=149, decoy0=10 ** 6, max_evals=1)
import time
t0 = time.perf_counter(); r = ctx.fold_batch(1280, runs, seed=150); el = time.perf_counter() - t0
out.append(f"{'pooled 1280 on 2 x 640 slots':30s} {1280 / el:7.1f} decoys/s ({el:.3f} s)")
ctx.close()
print(f"== {tag}")
print("\n".join(out), flush=True)
