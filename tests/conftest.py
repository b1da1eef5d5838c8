import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(autouse=True)
def _modes_as_requested():
    """The tests' grids are tiny, and 'auto' / 'fused' hand small problems to the stored-residual kernels: switch that
    rule off so that every backward mode a test asks for is the one that runs (the rule has a test of its own)."""
    from pde_amd import functional as PF
    old, PF.SMALL_FIELD_ELEMENTS = PF.SMALL_FIELD_ELEMENTS, 0
    yield
    PF.SMALL_FIELD_ELEMENTS = old
