"""GPU: the HIP path against the oracle on synthetic targets that are NOT helical bundles -- strand meanders and mixed
helix / strand chains (synth.make_map kind="meander" / "mixed"): extended residues in the beta basin of the torsion potential,
restraints that are mostly long-range, loose chains (radius of gyration 15-16 A at 100-120 residues against 12.6 for a bundle of
90).  The strands are not hydrogen-bonded sheets (synth.py says why): what these cases add is geometry, not a new energy term.

Per target: one evaluation of every decoy of a mixed batch (unfolded starts + starts near the target) against the oracle with the
tolerances of tests/test_gpu_configs.py, the minimiser-tracking check near the target, and a fold from random starts with all four
channels under the default protocol: bitwise reproducible, every decoy finite, and it folds (C-alpha RMSD to the map's own
structure; thresholds = measured + margin, written at the assert)."""
import importlib

import numpy as np
import pytest
import torch  # noqa: F401  -- before libtrx2fold.so (see test_gpu_boundary.py)

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd

from test_gpu_configs import SF, chans, check_eval_every_decoy, check_fold_properties, check_tracking, mixed_starts, near_starts, oracle_tables

T = importlib.import_module("trrosettax2-dynamics_amd")
S = importlib.import_module("trrosettax2-dynamics_amd.synth")


@pytest.mark.parametrize("kind,L", [("meander", 100), ("mixed", 120)])
def test_non_bundle_targets_evaluate_like_the_oracle_and_fold(kind, L):
    B = 32
    m = S.make_map(L, seed=L, kind=kind)
    ctx = T.Context(0)
    try:
        ctx.set_map(m["dist"], *chans(m, True), seq=m["seq"])
        Tb = oracle_tables(m, True)
        _, e_t, _, _ = O.evaluate(Tb, np.asarray(m["tors"], np.float64), SF, grad=False)
        f, e, g, xyz = ctx.eval_batch(np.asarray(m["tors"], np.float32)[None], SF)
        print(f"\n{kind} L={L}: energy terms at the target, oracle {np.round(e_t, 2)} device {np.round(e[0], 2)}")
        w = check_eval_every_decoy(ctx, Tb, mixed_starts(m, B, 11), SF, 2e-3)
        runs = T.protocol.build_runs(L, 2, fastrelax=True)
        # 20 evaluations of the restraint stage from near the target.  What is pinned hard is the minimiser's work: accepted iterations per
        # evaluation (measured ratio 0.993 / 1.002, 22 / 26 of 32 counts identical).  The energies at a FIXED evaluation count are taken on a
        # slope that drops by orders of magnitude within these evaluations (perturbed strands start from clashes): one flipped line-search
        # decision moves a decoy by its own magnitude -- measured median 0.13 / 0.04, worst eighth from 0.7 (round 5's rama / omega terms; the
        # terms themselves match the oracle to 2e-5 with the other weights at zero: tests/test_gpu_parity.py).
        trk = check_tracking(ctx, Tb, near_starts(m, B, 12), runs[5:], med_tol=0.25, tail_tol=1.5, ratio_min=0.97, same_frac=0.5)
        r, r2 = ctx.fold_batch(B, runs, seed=L), ctx.fold_batch(B, runs, seed=L)
        med, lo, rm, mir = check_fold_properties(r, r2, m, Tb, True, lambda q: ctx.eval_batch(q["tors"], SF)[1])
        print(f"{kind} L={L}: worst eval deviations {w}; tracking {trk}; evals median {int(np.median(r['n_evals']))}; "
              f"RMSD to target sorted {np.round(np.sort(rm), 2)}")
        # (the default protocol ends with the unrestrained closing minimisation: the restraint-energy depth of the last run is 0 by
        # construction, so the fold is judged on the structure)
        assert np.median(rm) < RMSD_MEDIAN_MAX[kind] and (rm < 2.0).sum() >= WITHIN_2A_MIN[kind] * B, np.sort(rm)
    finally:
        ctx.close()


# measured on MI355X (profiles/history/runs_r01_r04.sh.txt section r04_run19.sh): meander median 0.40 A, 32 of 32 within 2 A (worst 1.61); mixed 0.32 A, 32 of 32
# (worst 1.01); no mirror images (all four channels).  Limits = measured + margin.
RMSD_MEDIAN_MAX = {"meander": 0.7, "mixed": 0.6}
WITHIN_2A_MIN = {"meander": 0.9, "mixed": 0.9}
