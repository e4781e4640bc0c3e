"""Round 5 probe: would FOUR lanes of 16 decoys fold BASELINE config 2's call of 64 faster than the library's two lanes of 32?  Four independent
one-lane contexts on the same map (own streams: the library's pool has four), 16 decoys each from four host threads, against one two-lane
context folding 64; same decoy identities (seed, decoy0 + i).  usage: lanes4_probe.py <repo> [calls = 5] [config 2 | 3]"""
import importlib, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, sys.argv[1])
T = importlib.import_module("trrosettax2-dynamics_amd"); S = importlib.import_module("trrosettax2-dynamics_amd.synth")
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = 150; m = S.make_map(L); ang = [m["omega"], m["theta"], m["phi"]] if cfg == 3 else []
runs = T.protocol.build_runs(L, 2, fastrelax=True)
for split in (2, 4, 2, 4, 8):
    n = 64 // split
    if split == 2:
        ctxs = [T.Context(0, lanes=2)]
    else:
        ctxs = [T.Context(0) for _ in range(split)]
    for c in ctxs: c.set_map(m["dist"], *ang, seq=m["seq"])
    def call(i):
        if split == 2:
            return [ctxs[0].fold_batch(64, runs, seed=150, decoy0=64 * i)]
        with ThreadPoolExecutor(max_workers=split) as ex:
            return list(ex.map(lambda k: ctxs[k].fold_batch(n, runs, seed=150, decoy0=64 * i + n * k), range(split)))
    call(900)
    t0 = time.perf_counter(); rs = [call(i) for i in range(K)]; el = time.perf_counter() - t0
    ev = np.concatenate([r["n_evals"] for c in rs for r in c])
    print(f"{split} lane(s) x {n} decoys: {K * 64 / el:7.1f} decoys/s, {1e3 * el / K:6.1f} ms per call of 64; evaluations median {np.median(ev):.0f} max {ev.max()}", flush=True)
    for c in ctxs: c.close()
