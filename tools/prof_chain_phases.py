import sys, importlib, numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B = 150, 64
m = S.make_map(L); ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
rng = np.random.default_rng(0); tors = np.stack([m["tors"] + rng.normal(size=(L, 3)) * 0.1 for _ in range(B)])
w = np.array(T.protocol.SF, np.float32)
for _ in range(60): ctx.eval_batch(tors, w)
ctx.close(); print("done")
