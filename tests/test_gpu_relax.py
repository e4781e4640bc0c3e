"""GPU: the backbone-visible part of the reference's full-atom refinement (folding/folding.py:200-268; "a11-lite"): restraints
re-selected at PCUT 0.15 / 0.30 without glycine pairs (runs with pair_filter 2 / 3), score re-weighted to atom_pair 5 /
dihedral 1 / angle 1, the ramps of the two FastRelax scripts with their per-run tolerances.  The re-selections are pinned to
the reference's add_rst on the CPU side (tests/test_oracle_golden.py); here the device against the oracle."""
import importlib
import os

import numpy as np
import pytest
import torch  # noqa: F401  -- before libtrx2fold.so (see test_gpu_boundary.py)

pytestmark = pytest.mark.gpu

from oracle import oracle as O
from oracle.kabsch import kabsch_rmsd

T = importlib.import_module("trrosettax2-dynamics_amd")
P = T.protocol
TERMS = [0, 1, 2, 3, 4, 5, 6, 8]


@pytest.mark.parametrize("tag,pcut", [("NMR", 0.05), ("Xray", 0.05), ("NMR", 0.3)])
def test_relax_selections_on_the_device_equal_the_oracles(golden_dir, seq, tag, pcut):
    """one evaluation under each selection (all, relax round 1, relax round 2) with the relax stage's weights: every energy term
    of every decoy equals the oracle's (2e-4 relative + 0.1), and the three selections give three different restraint energies.
    -pd 0.3: the map's own selection is then NARROWER than the relax round-1 re-selection at 0.15, which
    add_rst(.., nogly=True) makes from all generated restraints (utils_ros.py:713-717) -- the row lists must keep those pairs."""
    m = np.load(os.path.join(golden_dir, f"seq_{tag}.npz"))
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq, pcut=pcut)
    ctx = T.Context(0)
    try:
        ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq, pcut=pcut)
        rng = np.random.default_rng(3)
        B = 12
        t0 = np.stack([O.random_torsions(90, 9, d) + rng.normal(size=(90, 3)) * 0.05 for d in range(B)]).astype(np.float32)
        seen = []
        for flt in (0, 2, 3):
            run = [dict(w=P.SF_FA, max_iter=1, sep_lo=1, sep_hi=90, pair_filter=flt)]
            r = ctx.fold_batch(B, run, tors0=t0, max_evals=1)
            assert np.all(r["status"] == 2)          # the one evaluation was the budget: the report is at the start torsions
            for d in range(B):
                _, _, st = O.fold(Tb, t0[d].astype(np.float64), run, max_evals=1)
                eo = st["e_final"]
                assert np.all(np.abs(r["e_terms"][d][TERMS] - eo[TERMS]) <= 2e-4 * np.abs(eo[TERMS]) + 0.1), (flt, d, r["e_terms"][d], eo)
                assert abs(r["f"][d] - st["f_final"]) <= 2e-4 * abs(st["f_final"]) + 1.0
            seen.append(r["e_terms"][:, :4].sum(0))
        assert np.all(np.abs(seen[0] - seen[1]) > 1.0) and np.all(np.abs(seen[1] - seen[2]) > 1.0), seen
    finally:
        ctx.close()


def test_fastrelax_protocol_runs_and_tracks_the_oracle(golden_dir, seq):
    """build_runs(.., fastrelax=True): 21 more runs (torsion ramps x 2, Cartesian ramps x 1 at PCUT 0.15; Cartesian ramps x 2 at
    0.30; the closing Cartesian minimisation without restraints).  The relax runs alone, from folded decoys, against the oracle
    over a short horizon (same start, same budget: accepted iterations within 10 %, energies within 1 %), then whole folds:
    every decoy converges, stays a chain (bond geometry), and lands near the plain protocol's decoy of the same seed."""
    m = np.load(os.path.join(golden_dir, "seq_Xray.npz"))
    Tb = O.Tables(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
    ctx = T.Context(0)
    try:
        ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
        plain, relax = P.build_runs(90, 2), P.build_runs(90, 2, fastrelax=True)
        assert len(relax) == len(plain) + 21 and relax[:len(plain)] == plain
        assert [q["pair_filter"] for q in relax[len(plain):]] == [2] * 12 + [3] * 9 and relax[-1]["w"][:3] == [0.0, 0.0, 0.0]
        B = 16
        a = ctx.fold_batch(B, plain, seed=77)
        # the relax runs alone from the folded decoys, 30 evaluations, device against oracle
        rr = relax[len(plain):]
        dev = ctx.fold_batch(B, rr, tors0=a["tors"], max_evals=30)
        _, _, st, _ = O.fold_batch(Tb, a["tors"].astype(np.float64), rr, max_evals=30)
        oi = np.array([s["n_iters"] for s in st]); of = np.array([s["f_final"] for s in st])
        rel = np.abs(dev["f"] - of) / np.abs(of)
        print(f"\\nrelax runs, 30 evaluations: iterations device {dev['n_iters'].sum()} oracle {oi.sum()}, rel energy median {np.median(rel):.1e} max {rel.max():.1e}")
        assert abs(int(dev["n_iters"].sum()) - int(oi.sum())) <= 0.1 * oi.sum() + 2 and np.median(rel) <= 1e-2
        b = ctx.fold_batch(B, relax, seed=77)
        assert np.all(b["status"] == 0) and np.all(np.isfinite(b["xyz"]))
        shift = np.array([kabsch_rmsd(a["xyz"][i, :, 1], b["xyz"][i, :, 1]) for i in range(B)])
        bonds = np.abs(np.linalg.norm(b["xyz"][:, 1:, 0] - b["xyz"][:, :-1, 2], axis=-1) - 1.334).max()
        print(f"fastrelax: evaluations {np.median(a['n_evals']):.0f} -> {np.median(b['n_evals']):.0f}, C-alpha shift median {np.median(shift):.2f} A, worst C-N bond deviation {bonds:.3f} A")
        assert np.median(shift) < 1.0 and bonds < 0.08
        assert np.median(b["n_evals"]) > np.median(a["n_evals"])
    finally:
        ctx.close()
