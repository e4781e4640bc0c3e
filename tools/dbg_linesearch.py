"""Diagnostic (-DTRX2_DBG build): decoy 0's line-search record per evaluation of an L=400 fold.
usage: dbg_linesearch.py <repo> [L=400] [n=70]"""
import ctypes as C, importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
m = S.make_map(L, seed=L); ctx = T.Context(0)
ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
r = ctx.fold_batch(8, T.protocol.build_runs(L, 2), seed=3, max_evals=n)
out = np.zeros((256, 12)); T.load().trx2_debug_linesearch(out.ctypes.data_as(C.c_void_p))
print("n_iters", r["n_iters"])
print("eval phase run          f_t              f        alpha           gdir  flags nls hl          fh0      xt0.x      gt0.x   (f_t-f)/alpha")
for k in range(1, n + 1):
    q = out[k]
    print(f"{k:4d} {int(q[0]):5d} {int(q[1]):3d} {q[2]:16.6f} {q[3]:16.6f} {q[4]:10.3e} {q[5]:14.4f} {int(q[6]):5d} {int(q[7]):3d} {int(q[8]):2d} {q[9]:16.6f} {q[10]:10.5f} {q[11]:10.4f} {((q[2]-q[3])/q[4] if q[4] else 0):14.4f}")
ctx.close()
