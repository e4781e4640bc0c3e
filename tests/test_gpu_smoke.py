"""GPU: the driver's smoke entry (__graft_entry__.smoke) is part of the suite, so a change that breaks it is seen before round end."""
import pytest
import torch  # noqa: F401  -- before libtrx2fold.so (see test_gpu_boundary.py)

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke():
    import __graft_entry__ as g
    g.smoke()
