"""Per-phase cycle shares of the torsion-space step (diagnostic -DTRX2_STAMP build, loaded through TRX2FOLD_LIB).
Thread 0 of decoy 0's workgroup stamps every STEP launch of one fold.  usage: stamp_chain.py <repo> <config 2|3|4> [decoys per launch]"""
import ctypes as C, importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, orient = {2: (150, 64, False), 3: (150, 64, True), 4: (400, 16, True)}[int(sys.argv[2])]
if len(sys.argv) > 3: B = int(sys.argv[3])   # decoys per launch
m = S.make_map(L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
runs = T.protocol.build_runs(L, 2, fastrelax=True)
lib = T.load(); out = (C.c_ulonglong * 32)()
ctx.fold_batch(B, runs, seed=1)
assert lib.trx2_debug_chain_stamps(out, 1) == 0
r = ctx.fold_batch(B, runs, seed=2)
assert lib.trx2_debug_chain_stamps(out, 1) == 0
v = np.array(out[:13], float); n, nd = out[30], out[31]
names = ["state load + role test", "slab sums, rama/omega", "suffix scan + torsion gradient", "energy reduction + X,G,D loads", "Armijo / (s,y) pair",
         "two-loop: first loop", "two-loop: gamma", "two-loop: second loop", "descent test / restart", "trial point + state stores",
         "NeRF: local frames", "NeRF: transform scan", "NeRF: atoms + stores"]
print(f"config {sys.argv[2]}: decoy 0, {n} torsion-space steps, {nd} with a new direction ({100*nd/max(n,1):.0f} %); fold took {r['seconds']*1e3:.0f} ms (stamped build)")
print(f"   cycles per step (100 MHz s_memtime ticks x shader clock ratio unknown: read SHARES): total {v.sum()/n:.0f}")
for k, (nm, x) in enumerate(zip(names, v)):
    per = x / (nd if k in (5, 6, 7) else n)
    print(f"   {nm:34s} {100*x/v.sum():5.1f} %   {per:8.0f} ticks per {'new-direction step' if k in (5,6,7) else 'step'}")
v = np.array(out[16:27], float); n, nd = out[28], out[29]
names = ["state load + role test", "coordinates -> LDS, slab sums", "backbone H, rama/omega", "bonded term, hand-over", "neighbours' parts, assembly",
         "energy reduction + X,G,D loads", "Armijo / (s,y) pair", "two-loop: first loop", "two-loop: gamma", "two-loop: second loop", "direction, trial, stores"]
if n:
    print(f"Cartesian role: decoy 0, {n} steps, {nd} with a new direction; ticks per step: total {v.sum()/n:.0f}")
    for k, (nm, xx) in enumerate(zip(names, v)):
        per = xx / (nd if k in (7, 8, 9) else n)
        print(f"   {nm:34s} {100*xx/v.sum():5.1f} %   {per:8.0f} ticks per {'new-direction step' if k in (7,8,9) else 'step'}")
print(f"staged Cartesian history entries that differ from their global copy (diagnostic counter 27): {out[27]}")
ctx.close()
