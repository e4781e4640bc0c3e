"""What does the Cartesian stage cost per launch pair at the headline shape?  320 decoys (L=150, distances only) on 2 lanes x 160 slots,
protocol with and without it (different evaluation counts: compare us per launch pair).  usage: cart_cost.py <repo>"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = 150; m = S.make_map(L, seed=L)
ctx = T.Context(0, lanes=2, pool=160); ctx.set_map(m["dist"], seq=m["seq"])
for cart in (True, False, True, False):
    runs = T.protocol.build_runs(L, 2, cartesian_stage=cart)
    ctx.fold_batch(320, runs, seed=150, decoy0=900 * 64)
    t0 = time.perf_counter(); r = ctx.fold_batch(320, runs, seed=150, decoy0=0); el = time.perf_counter() - t0
    print(f"cartesian stage {cart}: {320 / el:6.1f} decoys/s, {r['launches']} launch pairs per lane, {1e6 * el / r['launches']:.1f} us per launch pair, {r['n_evals'].mean():.0f} evals per decoy")
ctx.close()
