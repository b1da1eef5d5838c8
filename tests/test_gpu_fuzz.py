"""Every symbolic spec's HIP kernels against the NumPy interpreter of the same spec on seeded random problems: ragged and
quad-width extents, clips of 1 to 17 residual frames, random lattice crops, weights, constants and per-sample parameters,
in every backward mode the PDE supports (tests/fuzz_parity.py runs longer campaigns of the same cases)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pde_amd import functional as PF
from pde_amd.stencil.specs import SPECS
from tests.spec_interp import InterpBackend
from tests.fuzz_parity import one_case


@pytest.mark.parametrize("name", list(SPECS))
def test_random_problems_match_the_interpreter(name):
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1000 + list(SPECS).index(name))
    hip, interp = PF.get_backend(), InterpBackend()
    for c in range(4):
        desc, bad = one_case(name, rng, dev, hip, interp, quad=(c % 2 == 1), whole=(c == 2))
        assert not bad, (desc, bad)


def test_random_glue_problems_match_the_oracle():
    """boundary variants, patch seams + no-slip walls, the statistics reductions and the channel gather on seeded random
    shapes, masks and strip widths, values and gradients, against the CPU oracle (tests/fuzz_glue.py)"""
    from tests import fuzz_glue as FG
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    for _ in range(12):
        for fn in (FG.boundary_case, FG.seam_case, FG.stats_case, FG.gather_case):
            desc, err = fn(rng, dev)
            assert np.isfinite(err) and err <= FG.TOL, (desc, err)
