"""Seconds of one fold call against (chain length, decoys): the samples behind sched.CostModel's constants.
usage: fit_cost_model.py <repo> [out.json]     (one GPU; ~1 minute)
Every call: one context, one lane, every decoy in flight, the default protocol (-m 2 --fastrelax), all channels, synthetic map seed L;
the median of three calls with different decoys.  Prints the non-negative least-squares fit and its relative errors."""
import importlib, json, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth"); SC = importlib.import_module("trrosettax2-dynamics_amd.sched")
samples, rows = [], []
for L in (90, 150, 220, 300, 400):
    m = S.make_map(L, seed=L)
    ctx = T.Context(0)
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
    for n in (1, 4, 16, 32, 64, 128):
        ctx.fold_batch(n, runs, seed=1, decoy0=9000, max_evals=5)          # buffers
        ts, ev = [], []
        for rep in range(3):
            t0 = time.perf_counter(); r = ctx.fold_batch(n, runs, seed=L, decoy0=rep * 1000); ts.append(time.perf_counter() - t0); ev.append(int(r["n_evals"].max()))
        t = float(np.median(ts))
        samples.append((L, n, t)); rows.append(dict(L=L, n=n, seconds=round(t, 4), evals_max=int(np.median(ev))))
        print(rows[-1], flush=True)
    ctx.close()
fit = SC.CostModel.fit(samples)
err = fit.rel_errors(samples)
out = dict(model="seconds(L, n) = c0 + c1 L + n (c2 L + c3 L^2)", c=[fit.c0, fit.c1, fit.c2, fit.c3], rel_error_max=float(np.max(np.abs(err))), rel_error_median=float(np.median(np.abs(err))),
           shipped=[SC.MODEL.c0, SC.MODEL.c1, SC.MODEL.c2, SC.MODEL.c3], shipped_rel_error_max=float(np.max(np.abs(SC.MODEL.rel_errors(samples)))), samples=rows)
print(json.dumps({k: v for k, v in out.items() if k != "samples"}))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
