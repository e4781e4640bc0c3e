"""Per-phase cycle shares of one k_pair wave (diagnostic -DTRX2_STAMP build; load it with TRX2FOLD_LIB).  usage: stamp_pair.py <repo> <config> [decoys]"""
import ctypes as C, importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, orient = {2: (150, 64, False), 3: (150, 64, True), 4: (400, 16, True)}[int(sys.argv[2])]
if len(sys.argv) > 3: B = int(sys.argv[3])
m = S.make_map(L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
rng = np.random.default_rng(0); tors = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.03 for _ in range(B)])
w = np.array(T.protocol.SF, np.float32)
ctx.eval_batch(tors, w)
ms, _ = ctx.time_pair_kernel(B, w, 1, L, n_rep=20)
out = (C.c_ulonglong * 32)(); assert T.load().trx2_debug_stamps(out) == 0
v = np.array(out[:15], float); names = ["prologue", "entry+masks", "coords of b", "dist seek+fetch", "angles: seeks+fetches", "-", "-", "-", "values, gradients", "list done", "contact scan", "contact walk", "epilogue: column sums, stores", "epilogue: sub-lane sums, LDS image", "epilogue: barrier"]
print("config %s (stamped build: %.1f us per launch; shares matter, not the total)  wave total %.0f cycles" % (sys.argv[2], ms * 1e3, v.sum()))
for n, x in zip(names, v): print("   %-12s %8.0f cycles  %5.1f %%" % (n, x, 100 * x / v.sum()))
ctx.close()
