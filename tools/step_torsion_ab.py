"""A/B timing of the step kernel on the TORSION-SPACE part of the protocol (runs 0-7: declash + the three restrained minimisations; bond geometry stays
ideal, so a timing build that differs only in how the NeRF pass obtains its bond-angle sincos folds the SAME trajectory): microseconds per evaluation of
whole single-decoy folds and event-bracketed kernel durations.  TRX2FOLD_LIB selects the build.  usage: step_torsion_ab.py <repo> [repeats = 5]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tag = os.path.basename(os.environ.get("TRX2FOLD_LIB", "libtrx2fold.so"))
for L, B in ((150, 1), (150, 32), (400, 16)):
    m = S.make_map(L); ctx = T.Context(0)
    ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
    runs = T.protocol.build_runs(L, 2, fastrelax=True)[:8]
    ctx.fold_batch(B, runs, seed=1)
    us, ev = [], []
    for k in range(rep):
        r = ctx.fold_batch(B, runs, seed=2 + k)
        us.append(1e6 * r["seconds"] / r["n_evals"].max()); ev.append(int(r["n_evals"].max()))
    ctx.set_profiling(7)
    r = ctx.fold_batch(B, runs, seed=2)
    p, s, n = ctx.last_fold_kernel_times()
    ctx.set_profiling(0)
    print(f"{tag:24s} L={L} B={B:2d}: {np.mean(us):6.2f} us per evaluation (min {min(us):.2f}, max {max(us):.2f}; evaluations {ev}); events: pair {p * 1e3:.2f} us, step {s * 1e3:.2f} us; checksum {float(np.abs(r['xyz']).sum()):.3f}")
    ctx.close()
