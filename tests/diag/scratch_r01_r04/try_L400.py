"""BASELINE config 4: L=400, 32 decoys, all channels.  First end-to-end run at this size: does it converge, how long, how close
to the synthetic target?  usage: try_L400.py <repo>"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle.kabsch import kabsch_rmsd
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
L, B = 400, 32
t0 = time.time(); m = S.make_map(L); print(f"map L={L}: {time.time()-t0:.0f}s, contact fraction {m['contact_fraction']:.2f}", flush=True)
runs = T.protocol.build_runs(L, 2); print("cartesian run in protocol:", any(r["cartesian"] for r in runs), flush=True)
ctx = T.Context(0); ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=m["seq"])
r = ctx.fold_batch(B, runs, seed=400)
ca = S.nerf_backbone(m["tors"])[1]
rm = np.array([kabsch_rmsd(r["xyz"][i, :, 1], ca) for i in range(B)]); mir = np.array([kabsch_rmsd(r["xyz"][i, :, 1] * [1, 1, -1], ca) for i in range(B)])
print(f"fold: {r['seconds']:.2f} s for {B} decoys = {B/r['seconds']:.1f} decoys/s; status ok {bool(np.all(r['status']==0))}; evals {r['n_evals'].min()}..{r['n_evals'].max()} (median {np.median(r['n_evals']):.0f}); launches {r['launches']}")
print("RMSD to the synthetic target: median %.2f, sorted %s ; mirror-closer: %d" % (np.median(rm), np.round(np.sort(rm), 1), int((mir < rm).sum())))
w = np.array(T.protocol.SF, np.float32); ms, n = ctx.time_pair_kernel(B, w, 1, L, n_rep=30)
ab = B * (16.0 * n / B + 96.0 * L)
print(f"k_pair<32>: {ms*1e3:.0f} us per launch, {n/B:.0f} terms per decoy, algorithmic {ab/1e6:.0f} MB -> {ab/(ms*1e-3)/1e9:.0f} GB/s = {100*ab/(ms*1e-3)/1e9/8000:.1f} % of 8 TB/s")
ctx.close()
