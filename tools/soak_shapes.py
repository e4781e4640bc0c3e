"""Soak of the fold entry point over random shapes: chain lengths in every kernel class (<= 128, <= 256, <= 512 residues, beyond),
batch sizes, lanes, slot pools, tail-compaction modes, evaluation budgets, with and without the angle channels.  Every fold must end
with status 0 (or 2 = budget spent when one was set), finite coordinates and energies, and repeat bit for bit.
usage: soak_shapes.py <repo> [n = 40] [seed = 1] [big: batches of 260-700 decoys for chains of up to 256 residues]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
big = len(sys.argv) > 4 and sys.argv[4] == "big"
bad = 0
t_all = time.perf_counter()
for it in range(n):
    cls = it % 5
    L = int([rng.integers(24, 65), rng.integers(65, 129), rng.integers(129, 257), rng.integers(257, 513), rng.integers(513, 560)][cls])
    orient = bool(rng.integers(0, 2))
    lanes = int(rng.integers(1, 3))
    B = int(rng.integers(1, 81)) if L <= 256 else int(rng.integers(1, 25))
    if big and L <= 256:
        B = int(rng.integers(260, 700))     # folds that start on enough slots for the low-register step instantiation
    pool = 0 if rng.random() < 0.5 else int(rng.integers(1, B + 1))
    mode = int(rng.integers(0, 3))
    budget = 0 if (rng.random() < 0.4 and L <= 300) else int(rng.integers(20, 400))
    m = S.make_map(L, seed=1000 + it)
    ctx = T.Context(0, lanes=lanes, pool=pool)
    try:
        ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
        ctx.set_tail_compaction(mode)
        runs = T.protocol.build_runs(L, int(rng.integers(0, 3)))
        t0 = time.perf_counter()
        a = ctx.fold_batch(B, runs, seed=7 + it, max_evals=budget)
        b = ctx.fold_batch(B, runs, seed=7 + it, max_evals=budget)
        dt = time.perf_counter() - t0
        ok_status = np.all((a["status"] == 0) | ((a["status"] == 2) if budget else False))
        ok = bool(ok_status and np.all(np.isfinite(a["xyz"])) and np.all(np.isfinite(a["f"])) and np.array_equal(a["xyz"], b["xyz"])
                  and np.array_equal(a["n_evals"], b["n_evals"]))
        bad += not ok
        print(f"{'ok  ' if ok else 'FAIL'} L={L:3d} B={B:2d} lanes={lanes} pool={pool:2d} compaction={mode} budget={budget:3d} orient={int(orient)} runs={len(runs):2d} "
              f"evals median {int(np.median(a['n_evals'])):5d} status {sorted(set(a['status'].tolist()))} {dt:5.2f} s", flush=True)
    finally:
        ctx.close()
print(f"{n} shapes, {bad} failures, {time.perf_counter() - t_all:.0f} s")
sys.exit(1 if bad else 0)
