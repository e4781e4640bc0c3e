"""Driver for PMC passes over the pair kernel: fold one batch of a bench config, then replay k_pair N times on the final
coordinates (the same launches bench.py times for `roofline.achieved`).  usage: pmc_pair.py <repo> <config 2|3|4> [N]"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
# decoys per launch = the slots of one lane in a default bench.py run (bench.py: min(192, decoys of the run per lane)): config 2 folds
# 5 x 64 decoys on two lanes, configs 3 and 4 are sub-records of 2 steps (two chains of 128 on one lane each; 64 on two lanes)
SHAPES = {2: (150, 160, False), 3: (150, 128, True), 4: (400, 32, True)}
cfg = SHAPES[int(sys.argv[2])]
L, B, orient = cfg
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = S.make_map(L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
ctx.fold_batch(B, T.protocol.build_runs(L, 2), seed=150)
ms, terms = ctx.time_pair_kernel(B, np.array(T.protocol.SF, np.float32), 1, L, n_rep=n)
print(f"config {sys.argv[2]}: k_pair {ms*1e3:.1f} us, {terms/B:.0f} terms/decoy"); ctx.close()
