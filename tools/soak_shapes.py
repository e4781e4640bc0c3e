"""Soak over batch shapes: random decoy counts, slot pools, lanes, tail-compaction modes, chain lengths and evaluation budgets on one
context per map, with a pair-kernel replay and an evaluation batch after every fold (the calls that reuse the fold's buffers).
Every fold must report every decoy.  usage: soak_shapes.py <repo> [seconds=120] [seed=1]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    L = int(rng.choice([40, 90, 128, 150, 200, 257, 300]))
    orient = bool(rng.integers(2))
    m = S.make_map(L, seed=L)
    lanes = int(rng.integers(1, 3))
    ctx = T.Context(0, lanes=lanes)
    ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
    runs = T.protocol.build_runs(L, 2, cartesian_stage=bool(rng.integers(4)))
    for rep in range(int(rng.integers(2, 6))):
        N = int(rng.choice([1, 2, 3, 7, 20, 33, 64, 65, 100, 130, 200, 320]))
        if L >= 257: N = min(N, 100)
        pool = int(rng.choice([0, 0, 5, 32, 48, 64, 96, 160, 192]))
        mode = int(rng.integers(3))
        me = int(rng.choice([0, 0, 30, 300]))
        ctx.set_pool(pool); ctx.set_tail_compaction(mode)
        r = ctx.fold_batch(N, runs, seed=int(rng.integers(1 << 30)), max_evals=me)
        assert r["xyz"].shape[0] == N and np.all(np.isfinite(r["xyz"])) and np.all(r["n_evals"] > 0), (L, N, pool, mode, me)
        assert np.all((r["status"] == 0) | (r["status"] == 2)), r["status"]
        B = min(N, pool) if pool else N
        if lanes == 1 or N < 32:
            ms, _ = ctx.time_pair_kernel(B, np.array(T.protocol.SF, np.float32), 1, L, n_rep=2)
            assert 0 < ms < 50
        k = int(rng.integers(1, 9))
        f, e, g, xyz = ctx.eval_batch(r["tors"][:k], np.array(T.protocol.SF, np.float32))
        assert np.all(np.isfinite(f)) and np.all(np.isfinite(g))
        n += 1
    ctx.close()
    print(f"{time.time() - t0:6.1f} s  {n} folds  last: L={L} orient={orient} lanes={lanes} N={N} pool={pool} compaction={mode} max_evals={me}", flush=True)
print("soak ok:", n, "folds")
