"""GPU, run LAST (tests/conftest.py moves this file's items behind every other test, whatever the file names sort like): statements that are RED on
purpose while the finding they express is open.  VERDICT r4 item 1c asked for the percentile assert of the iteration-phase decoys to get teeth and
to stay red if it fails; it runs last so that `pytest -x` has every other result on record before it stops here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_the_four_iteration_decoys_are_jointly_typical_draws():
    """One reference decoy per fed-back map is one draw; four of them are four.  If they are draws of this build's distributions their
    percentiles (each decoy's median distance to our draws, ranked among our own draws' medians; tests/test_gpu_iteration_parity.py) are
    uniform on 0..100: the mean of four has expectation 50 and sd 14.4, so 50 +- 25 is a 92 % band (`<= 99` on each decoy alone, round 4's
    assert, would have passed at 98.9).
    OPEN FINDING, measured in rounds 4 and 5: 94 / 62 (NMR stages 1, 2), 91 / 66 (X-ray) in round 4; 93 / 61 / 89 / 73 with round 5's fitted
    rama / omega terms -- mean 78-79, outside the band.  Both STAGE-1 decoys (the first fold of a map fed back from the reference's initial0)
    sit near the 90th percentile: that fold of Rosetta's lies further from this build's draws than the draws lie from each other, on both
    maps; the stage-2 decoys and the four initial decoys (percentiles 32 / 82 / 81 / 38, asserted in tests/test_gpu_cartesian.py) do not.
    Device and oracle agree on these maps to a KS distance of 0.04 (tests/test_gpu_outcome_vs_oracle.py), so this is the ENERGY MODEL's
    response to the feedback step's perturbation (realised bins of low-confidence pairs halved), not the kernels': what PyRosetta's centroid
    and full-atom potentials do with those softened restraints is not in the reference tree ('parity unpinned').
    ROUND 6, MEASURED (tests/diag/iteration_drift.py, profiles/r06_iteration_drift.txt; 1024 draws per map, 80-residue core): the reference's stage-1
    decoy is displaced from the mean of its two initial decoys by 0.715 A (NMR) / 0.344 A (X-ray); the same statistic on draws of this build has a
    median of 0.424 / 0.200 A -- the reference's sits at percentile 98 / 86.  The DIRECTION is this build's (cosine with the shift of the ensemble
    mean +0.51 / +0.47, inside the null band 0.36..0.75 / 0.19..0.86) and along it the reference moved 1.5 / 1.4 times as far (null 5-95 %: 0.63..1.41);
    what is left over is spread: one reference draw of the fed-back map scatters ~1.7 x as far about the common response as a draw of this build
    (0.34 / 0.23 A about its ensemble mean).  Stage 2 is unremarkable on both maps (percentile 42 / 26).  So the finding is the UNDER-DISPERSION of this
    build's ensembles (deviation list), sharpest on the first fed-back map -- not a wrong response.  Variants scanned on the fed-back
    folds: restraint weights x 0.5 .. 2 leave the four percentiles at 81-90 / 67-81; the only variants that broaden the cloud enough are looser
    tolerances (1e-5: 78 / 59 / 88 / 70, mean 74; 1e-4, the reference's own number: 71 / 33 / 68 / 57, mean 57) and they pay with the distance to the
    reference's decoys (1.11 -> 1.25 A, 0.79 -> 0.88 A on the stage-1 maps) and with every initial-map figure (DESIGN deviation 1): not shipped."""
    try:
        import test_gpu_iteration_parity as IP
    except ImportError:
        pytest.skip("tests/test_gpu_iteration_parity.py was not collected in this run")
    if len(IP.TWO_SAMPLE) != 4:
        pytest.skip("the two parametrised tests of tests/test_gpu_iteration_parity.py fill TWO_SAMPLE; they did not run")
    pcts = {k: v[2] for k, v in sorted(IP.TWO_SAMPLE.items())}
    mean = float(np.mean(list(pcts.values())))
    print("\npercentiles of the four iteration-phase reference decoys among this build's draws:", {f"{k[0]}/stage{k[1]}": round(v) for k, v in pcts.items()}, "mean %.1f" % mean)
    assert 25.0 <= mean <= 75.0, (pcts, mean)
