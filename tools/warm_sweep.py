"""Round 5: which runs of the default protocol may take a warm first step (trx2_model.h TRX2_RUN_WARM) -- evaluations against outcome and
twisted peptides, 2 x n decoys per variant.  usage: warm_sweep.py <repo> [decoys = 2048]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.argv = [sys.argv[0], sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "2048", "1000", "none"]
exec(open(os.path.join(sys.argv[1], "tools", "tol_sweep_relax.py")).read().split("print(f\"# {n_dec}")[0])
ALL = set(range(35)); FIRST = {14, 18, 22, 26, 30}; REP = {1, 2, 3, 4, 6, 7, 10, 11, 12, 13}
V = {"none": set(), "all": ALL, "repeats + declash": REP, "all but ramp-first": ALL - FIRST, "all but closing": ALL - {34},
     "all but ramp-first and closing": ALL - FIRST - {34}, "all but run 14": ALL - {14}, "repeats + declash + tight ramp steps": REP | {17, 21, 25, 29, 33},
     "all but Cartesian ramp-firsts": ALL - {22, 26, 30}}
for name, on in V.items():
    runs = P.build_runs(90, 2, fastrelax=True)
    for i, r in enumerate(runs): r["warm"] = 1 if i in on else 0
    cell(name, runs)
ctx.close()
