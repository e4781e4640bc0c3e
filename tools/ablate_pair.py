"""Cost of k_pair's components by ablation through the public API (no rebuild): fold one batch, then time k_pair on the
final coordinates with (a) everything, (b) vdw weight 0, (c) empty separation window = no restraints, (d) both off
(loop skeleton + prologue/epilogue only).  usage: ablate_pair.py <repo> <config>"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B, orient = {2: (150, 64, False), 3: (150, 64, True), 4: (400, 32, True)}[int(sys.argv[2])]
m = S.make_map(L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
ctx.fold_batch(B, T.protocol.build_runs(L, 2), seed=150)
w = np.array(T.protocol.SF, np.float32); w0 = w.copy(); w0[3] = 0; w0[7] = 0; wv = w.copy(); wv[7] = 0; wh = w.copy(); wh[3] = 0
rows = [("everything", w, 1, L), ("contacts off (vdw, hb)", w0, 1, L), ("hb off", wv, 1, L), ("vdw off", wh, 1, L), ("restraints off (contacts only)", w, 0, 0),
        ("both off (skeleton)", w0, 0, 0)]
for rep in range(2):
    for name, ww, lo, hi in rows:
        ms, _ = ctx.time_pair_kernel(B, ww, lo, hi, n_rep=100)
        if rep: print(f"config {sys.argv[2]}  {name:28s} {ms*1e3:6.1f} us")
ctx.close()
