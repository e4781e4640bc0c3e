"""Tail compaction off / on: N decoys (L=150) on 2 lanes x S slots.  usage: compact_ab.py <repo> N S [orient]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
N, slots = int(sys.argv[2]), int(sys.argv[3]); orient = len(sys.argv) > 4
L = 150; m = S.make_map(L, seed=L); runs = T.protocol.build_runs(L, 2)
ctx = T.Context(0, lanes=2, pool=slots)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
ctx.fold_batch(min(N, 2 * slots), runs, seed=150, decoy0=900 * 64)
for mode in (0, 1, 0, 1):
    ctx.set_tail_compaction(mode)
    t0 = time.perf_counter(); r = ctx.fold_batch(N, runs, seed=150, decoy0=0); el = time.perf_counter() - t0
    print(f"N={N} 2 x {slots} compaction {mode}: {N / el:6.1f} decoys/s, {r['launches']} launch pairs per lane, efficiency {r['slot_efficiency']:.2f}, ok {bool(np.all(r['status'] == 0))}")
ctx.close()
