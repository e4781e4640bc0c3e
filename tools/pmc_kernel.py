"""Driver for the PMC passes of tools/pmc_run.sh over one kernel at one bench launch shape.
usage: pmc_kernel.py <repo> <config 2|3|4|e> <decoys per launch> <pair|step> [N]
  config e = the metric's job (bench.py's `value`): ONE decoy on the L=150 all-channel map fed back once from a folded decoy (all pairs selected,
  segment cache on: k_pair_c / k_step<1,256,256>), the shape of every iteration fold of run_inference
  pair: fold one batch of that many decoys (one lane, one slot per decoy), lay its final torsions out again (eval_batch) and replay
        k_pair N times on them -- the launches bench.py times for `roofline.achieved`; the report averages the last N dispatches.
  step: fold the batch for 40 + N evaluations from the random start (every slot alive in every launch); the report averages the
        last N k_step dispatches."""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
CFG = {"2": (150, False), "3": (150, True), "4": (400, True), "e": (150, True)}
L, orient = CFG[sys.argv[2]]
B, what = int(sys.argv[3]), sys.argv[4]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 40
m = S.make_map(L); ctx = T.Context(0)
ctx.set_map(m["dist"], *([m["omega"], m["theta"], m["phi"]] if orient else []), seq=m["seq"])
runs = T.protocol.build_runs(L, 2)
if sys.argv[2] == "e":      # one feedback step from a folded decoy: the map an iteration fold sees
    P = importlib.import_module("trrosettax2-dynamics_amd.pdbio")
    r0 = ctx.fold_batch(1, T.protocol.build_runs(L, 2, fastrelax=True), seed=7)
    x, s_pdb = P.as_read_from_pdb(m["seq"], r0["xyz"][0])
    ctx.feedback_step(x, s_pdb, 1.0, True)
    runs = T.protocol.build_runs(L, 2, fastrelax=True)
if what == "pair":
    r = ctx.fold_batch(B, runs, seed=150)
    w = np.array(T.protocol.SF, np.float32)
    ctx.eval_batch(r["tors"], w)
    ms, terms = ctx.time_pair_kernel(B, w, 1, L, n_rep=n)
    print(f"config {sys.argv[2]} B={B}: k_pair {ms*1e3:.1f} us, {terms/B:.0f} terms/decoy")
else:
    # near the map's own structure: the restraint stages with stored pairs, the state a fold spends its time in
    rng = np.random.default_rng(1)
    t0 = np.stack([m["tors"] + rng.normal(size=m["tors"].shape) * 0.3 for _ in range(B)]).astype(np.float32)
    r = ctx.fold_batch(B, runs[5:], tors0=t0, max_evals=40 + n)
    print(f"config {sys.argv[2]} B={B}: k_step over {r['launches']} launch pairs, evaluations {r['n_evals'].min()}..{r['n_evals'].max()}")
ctx.close()
