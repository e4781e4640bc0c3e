"""N decoys on ONE lane of S slots (run under rocprofv3 --kernel-trace --stats to see the kernels without a second lane beside them).
usage: one_lane.py <repo> N S [lanes=1]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
N, slots = int(sys.argv[2]), int(sys.argv[3]); lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L = 150; m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
ctx = T.Context(0, lanes=lanes, pool=slots); ctx.set_map(m["dist"], seq=m["seq"])
t0 = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=150, decoy0=0); el = time.perf_counter() - t0
print(f"{lanes} lane(s) x {slots}: {N / el:.1f} decoys/s, {r['launches']} launch pairs, {1e6 * el / r['launches']:.1f} us per launch pair")
ctx.close()
