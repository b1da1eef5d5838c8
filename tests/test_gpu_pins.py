"""HIP kernels (through the C ABI) against tests/golden/pins.npz: the outputs of the reference's own verification
functions for the FD-c4 Gray-Scott residuals, the 3-D heat and Taylor-Green equations, forced incompressible NS, the
periodic-BC loss, the eval_nrmse loop and the boundary loss on 6-D tensors."""
import pytest
import torch

from pde_amd import functional as PF
from pde_amd import reductions as R
from tests import pin_cases as PC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(PC.CHECKS))
def test_pin(name):
    assert R.get_backend().name == "hip" and PF.get_backend().name == "hip"
    PC.CHECKS[name](torch.device("cuda:0"))
