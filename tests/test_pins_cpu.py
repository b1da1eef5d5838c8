"""The product's host logic (NumPy interpreter backends in place of the HIP library) against tests/golden/pins.npz."""
import pytest
import torch

from pde_amd import functional as PF
from pde_amd import reductions as R
from tests import pin_cases as PC
from tests import spec_interp
from tests.reduce_interp import InterpReductions


@pytest.fixture(autouse=True)
def _interp_backends():
    old_r = R.set_backend(InterpReductions())
    old_f = PF.set_backend(spec_interp.InterpBackend())
    yield
    R.set_backend(old_r)
    PF.set_backend(old_f)


@pytest.mark.parametrize("name", sorted(PC.CHECKS))
def test_pin_host_logic(name):
    PC.CHECKS[name](torch.device("cpu"))
