"""GPU, collected LAST (the file name sorts behind every other test file): statements that are RED on purpose while the finding they express
is open.  VERDICT r4 item 1c asked for the percentile assert of the iteration-phase decoys to get teeth and to stay red if it fails; it runs
last so that `pytest -x` has every other result on record before it stops here."""
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
    and full-atom potentials do with those softened restraints is not in the reference tree ('parity unpinned')."""
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
