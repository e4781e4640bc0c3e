"""Pair-kernel launch time of ONE library build at one bench shape (A/B runs: TRX2FOLD_LIB selects the build).
Folds a batch first so that the replays run on folded coordinates, lays its final torsions out again, then 200 replays.
usage: pair_ab.py <repo> <config 2|3|4> <decoys per launch> [fold: 1 = also time a whole fold of that batch]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
CFG = {2: (150, False), 3: (150, True), 4: (400, True)}
cfg, B = int(sys.argv[2]), int(sys.argv[3])
L, orient = CFG[cfg]
m = S.make_map(L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
runs = T.protocol.build_runs(L, 2)
w = np.array(T.protocol.SF, np.float32)
r = ctx.fold_batch(B, runs, seed=150)
t0 = time.perf_counter(); r = ctx.fold_batch(B, runs, seed=150); el = time.perf_counter() - t0
ctx.eval_batch(r["tors"], w)
ms, terms = ctx.time_pair_kernel(B, w, 1, L, n_rep=200)
# declash-stage weights (no restraints, repulsion only) and an unfolded start: the other regime of a fold
t_rand = np.stack([S.nerf_backbone(m["tors"])[0] * 0 + 0 for _ in range(0)]) if False else None
ctx.set_profiling(8); ctx.fold_batch(B, runs, seed=151); pm, sm, n = ctx.last_fold_kernel_times(); ctx.set_profiling(0)
print(json.dumps(dict(lib=os.path.basename(os.environ.get("TRX2FOLD_LIB", "default")), config=cfg, B=B, pair_us_final=round(ms * 1e3, 2), terms_per_decoy=terms / B,
                      workgroups=int(ctx.info(4)), pair_us_live=round(pm * 1e3, 2), step_us_live=round(sm * 1e3, 2), fold_s=round(el, 4),
                      decoys_per_s=round(B / el, 1), evals_median=float(np.median(r["n_evals"])), ok=bool(np.all(r["status"] == 0)))))
ctx.close()
