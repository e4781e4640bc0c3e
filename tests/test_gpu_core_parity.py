"""GPU: the north star's parity criterion -- "C-alpha RMSD to the reference PyRosetta decoys within 0.5 A" -- read where the reference itself is defined.

The global RMSD of a 90-residue decoy is dominated by a handful of residues on which the reference's OWN two decoys of a map disagree by 1 - 5 A
(both termini, the GGG loop 33-35, two more loops: tests/diag/per_residue_deviation.py prints the profile): the reference pair is 0.86 A (NMR map) and
0.62 A (X-ray map) apart globally, but 0.30 / 0.22 A on the 80 residues on which it agrees best.  The core is therefore defined BY THE REFERENCE --
the k residues with the smallest deviation between its two initial decoys of the map after their superposition -- never by this build's decoys,
and every decoy is superposed on the core alone.  Measured on 1024 decoys per map, default protocol (profiles/r05_core_parity.txt):

    core       NMR map: median, share within 0.5 A      X-ray map
    80 of 90   0.415 A, 91 %  (reference pair 0.295)    0.294 A, 93 %  (0.220)
    70 of 90   0.388 A, 95 %  (0.226)                   0.273 A, 92 %  (0.177)

Asserted on 512 decoys per map: median <= 0.45 / 0.33 A on the 80-residue core, at least 86 % of the decoys of the right topology within the north star's
0.5 A, and at least 80 % of ALL decoys (the mirror-image topology, 2 - 7 % of the starts, counted as a miss)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T = importlib.import_module("trrosettax2-dynamics_amd")


def fit(P, Q):
    """P superposed on Q (Kabsch, proper rotation)"""
    pc, qc = P.mean(0), Q.mean(0)
    U, S, Vt = np.linalg.svd((P - pc).T @ (Q - qc))
    R = Vt.T @ np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))]) @ U.T
    return (P - pc) @ R.T + qc


def rmsd_on(P, Q, idx):
    return float(np.sqrt(((fit(P[idx], Q[idx]) - Q[idx]) ** 2).sum(1).mean()))


@pytest.mark.parametrize("tag,refs,med_max,pair_core_max", [("NMR", ("conf_2_1", "conf_2_2"), 0.45, 0.31), ("Xray", ("conf_1_1", "conf_1_2"), 0.33, 0.24)])
def test_decoys_are_within_half_an_angstrom_of_the_reference_on_its_own_core(golden_dir, seq, tag, refs, med_max, pair_core_max):
    ref = np.load(os.path.join(golden_dir, "ref_decoys.npz"))
    R = [ref[k][:, 1].astype(np.float64) for k in refs]
    pair = np.sqrt(((fit(R[0], R[1]) - R[1]) ** 2).sum(1))            # per-residue deviation between the reference's own two decoys of the map
    core = np.sort(np.argsort(pair)[:80])
    left_out = sorted(int(i) + 1 for i in np.setdiff1d(np.arange(90), core))
    assert rmsd_on(R[0], R[1], core) <= pair_core_max and pair[np.setdiff1d(np.arange(90), core)].min() > 0.55, (rmsd_on(R[0], R[1], core), left_out)
    m = np.load(os.path.join(golden_dir, f"seq_{tag}.npz"))
    n = 512
    ctx = T.Context(0, lanes=2)
    try:
        ctx.set_map(m["dist"], m["omega"], m["theta"], m["phi"], seq=seq)
        r = ctx.fold_batch(n, T.protocol.build_runs(90, 2, fastrelax=True), seed=4242)
    finally:
        ctx.close()
    assert np.all(r["status"] == 0)
    ca = r["xyz"][:, :, 1].astype(np.float64)
    on_core = np.array([min(rmsd_on(ca[d], Q, core) for Q in R) for d in range(n)])
    glob = np.array([min(rmsd_on(ca[d], Q, np.arange(90)) for Q in R) for d in range(n)])
    print(f"\n{tag} map, {n} decoys, default protocol: C-alpha RMSD to the closer reference decoy, median {np.median(glob):.3f} A over all 90 residues "
          f"({100 * np.mean(glob <= 0.5):.0f} % within 0.5 A; the reference's own pair {rmsd_on(R[0], R[1], np.arange(90)):.3f});\n   on the 80 residues on which "
          f"the reference's two decoys agree best (left out: {left_out}): median {np.median(on_core):.3f} A, {100 * np.mean(on_core <= 0.5):.0f} % within 0.5 A "
          f"(the reference's own pair {rmsd_on(R[0], R[1], core):.3f})")
    assert np.median(on_core) <= med_max, np.median(on_core)
    same = on_core < 3.0                                                      # not the mirror-image topology (2 - 7 % of the starts)
    assert np.mean(on_core[same] <= 0.5) >= 0.86, np.mean(on_core[same] <= 0.5)        # the north star's tolerance (measured 91 - 95 %; sd of the share at n = 512: 1.3 %)
    assert np.mean(on_core <= 0.5) >= 0.80, np.mean(on_core <= 0.5)                    # ... and with the mirror-image decoys counted as misses (measured 87 - 89 %)
