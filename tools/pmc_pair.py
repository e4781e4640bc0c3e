"""Driver for PMC passes over the pair kernel: fold one batch of a bench config, then replay k_pair N times on the final
coordinates (the same launches bench.py times for `roofline.achieved`).  usage: pmc_pair.py <repo> <config 2|3|4> [N]"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
# decoys per launch = the slots of one lane in bench.py: B/2 for the single-chain configs (two lanes), B for config 3 (two chains)
cfg = {2: (150, 32, False), 3: (150, 64, True), 4: (400, 16, True)}[int(sys.argv[2])]
L, B, orient = cfg
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = S.make_map(L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
ctx.fold_batch(B, T.protocol.build_runs(L, 2), seed=150)
ms, terms = ctx.time_pair_kernel(B, np.array(T.protocol.SF, np.float32), 1, L, n_rep=n)
print(f"config {sys.argv[2]}: k_pair {ms*1e3:.1f} us, {terms/B:.0f} terms/decoy"); ctx.close()
