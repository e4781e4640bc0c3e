import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 90
if L == 90:
    g = os.path.join(sys.argv[1], "tests", "golden"); m = np.load(os.path.join(g, "seq_NMR.npz")); seq = None
else:
    m = importlib.import_module("trrosettax2-dynamics_amd.synth").make_map(L); seq = m["seq"]
ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq); runs = T.protocol.build_runs(L, 2, fastrelax=True)
lib = T.load(); out = (C.c_ulonglong * 32)()
ctx.fold_batch(1, runs, seed=1); lib.trx2_debug_chain_stamps(out, 1)
for i in range(5): ctx.fold_batch(1, runs, seed=2 + i)
lib.trx2_debug_chain_stamps(out, 1)
v = np.array(out[:13], float); n, nd = out[30], out[31]
names = ["state load + role test", "slab sums, rama/omega", "suffix scan + torsion gradient", "energy reduction + X,G,D loads", "Armijo / (s,y) pair",
         "two-loop: first loop", "two-loop: gamma", "two-loop: second loop", "descent test / restart", "trial point + state stores",
         "NeRF: local frames", "NeRF: transform scan", "NeRF: atoms + stores"]
print(f"L={L} single decoy: {n} torsion steps, {nd} new-direction; total {v.sum()/n:.0f} ticks per step")
for k,(nm,x) in enumerate(zip(names,v)): print(f"   {nm:34s} {100*x/v.sum():5.1f} %  {x/(nd if k in (5,6,7) else n):8.0f}")
v = np.array(out[16:27], float); n2 = out[28]
cn = ["state load + role test", "coordinates -> LDS, slab sums", "backbone H, rama/omega", "bonded term, hand-over", "neighbours' parts, assembly",
      "energy reduction + X,G,D loads", "Armijo / (s,y) pair", "two-loop: first loop", "two-loop: gamma", "two-loop: second loop", "direction, trial, stores"]
if n2:
    print(f"Cartesian role: {n2} steps, total {v.sum()/n2:.0f} ticks per step")
    for nm, x in zip(cn, v): print(f"   {nm:34s} {100*x/v.sum():5.1f} %  {x/n2:8.0f}")
