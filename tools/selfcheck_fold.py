"""Folds with a checking build of the library (TRX2FOLD_LIB=.../libtrx2fold_check.so): the step kernels compare every one-sum
energy total with the nine terms reduced one by one; prints the counters and the accepted iterations.
usage: selfcheck_fold.py <repo> <L> <B> <max_evals> [orient=1] [relax=0: 1 = the default protocol, with the relax stage]"""
import ctypes as C, importlib, json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, ne = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
orient = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
relax = len(sys.argv) > 6 and sys.argv[6] != "0"
m = S.make_map(L, seed=L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
lib = T.load()
has = hasattr(lib, "trx2_debug_selfcheck")
out = (C.c_ulonglong * 6)()
if has:
    lib.trx2_debug_selfcheck(out, 1)
r = ctx.fold_batch(B, T.protocol.build_runs(L, 2, fastrelax=relax), seed=3, max_evals=ne)
rec = dict(L=L, B=B, max_evals=ne, n_evals=r["n_evals"].tolist(), n_iters=r["n_iters"].tolist(), status=r["status"].tolist(), f=np.round(r["f"], 1).tolist())
if has:
    lib.trx2_debug_selfcheck(out, 0)
    rec["selfcheck"] = dict(torsion_checks=int(out[0]), torsion_mismatches=int(out[1]), cartesian_checks=int(out[2]), cartesian_mismatches=int(out[3]), run_starts=int(out[4]), run_starts_without_fh0=int(out[5]))
print(json.dumps(rec))
ctx.close()
