"""GPU parity of the metric / boundary / glue kernels, through the C ABI: against the reference-generated golden
values, and against the CPU oracle on seeded inputs at sizes beyond the fixtures (ragged extents, many groups)."""
import numpy as np
import pytest
import torch

from oracle import glue_oracle as GO
from pde_amd import boundary as PB
from pde_amd import metrics as PM
from pde_amd import reductions as R
from tests import glue_cases as GC

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _hip_backend():
    assert R.get_backend().name == "hip"
    yield


@pytest.mark.parametrize("name", sorted(GC.BOUNDARY))
def test_boundary_golden(name):
    GC.check_boundary(name, DEV)


def test_boundary_scaled_upstream():
    GC.check_boundary_scaled_upstream(DEV)


def test_metrics_golden():
    GC.check_metrics(DEV)


def test_accumulator():
    GC.check_accumulator(DEV)


@pytest.mark.parametrize("tag", ["1d", "2d", "3d"])
def test_metric_func_golden(tag):
    GC.check_metric_func(DEV, tag)


@pytest.mark.parametrize("transpose", [False, True])
def test_gather(transpose):
    GC.check_gather(DEV, transpose)


@pytest.mark.parametrize("spatial", [(5, 7), (3, 4, 5), (9,)])
def test_gather_odd_extents(spatial):
    GC.check_gather(DEV, len(spatial) == 2, spatial)


@pytest.mark.parametrize("spatial,C,chans,transpose", [
    ((20, 13), 18, (0, 1, 15), False),      # two chunks of 256 points, the second with 4
    ((20, 13), 18, (0, 1, 3), True),
    ((33, 32), 18, (2, 2, 15), False),      # a channel listed twice collects both gradients
    ((24, 24), 6, (5, 0), False),
    ((16, 16), 44, (43, 0, 7), False),      # too many channels for the tiled kernels
    ((6, 5, 14), 18, (17, 0, 9), False),
])
def test_gather_tiled_and_fallback(spatial, C, chans, transpose):
    GC.check_gather(DEV, transpose, spatial, C, chans)


def test_compute_pde_loss():
    GC.check_compute_pde_loss(DEV)


@pytest.mark.parametrize("groups,rows,C", [(1, 1, 18), (3, 7, 18), (5, 1001, 18), (2, 4099, 1), (4, 333, 15), (2, 50, 84), (1, 3, 1000),
                                           (64, 257, 18), (1, 300000, 18)])
def test_pair_stats_vs_fp64(groups, rows, C):
    gen = torch.Generator().manual_seed(groups * 1000 + rows + C)
    p = torch.randn(groups, rows, C, generator=gen) + 0.3
    g = torch.randn(groups, rows, C, generator=gen) * 2 - 0.1
    st = R.get_backend().pair_stats(p.to(DEV), g.to(DEV), groups, rows, C, 1e-8).cpu().numpy()
    p64, g64 = p.double().numpy(), g.double().numpy()
    d = p64 - g64
    want = np.stack([(d * d).sum(1), (g64 * g64).sum(1), g64.sum(1), p64.sum(1), (d * d / (g64 * g64 + 1e-8)).sum(1), np.abs(d).max(1)], -1)
    # the sums are fp64 accumulations of fp32 terms: 1e-6 of the sum of magnitudes
    scale = np.stack([(d * d).sum(1), (g64 * g64).sum(1), np.abs(g64).sum(1), np.abs(p64).sum(1), (d * d / (g64 * g64 + 1e-8)).sum(1), np.abs(d).max(1)], -1)
    assert np.all(np.abs(st - want) <= 2e-6 * scale + 1e-30)
    assert np.array_equal(st[..., 5].astype(np.float32), np.abs(p.numpy() - g.numpy()).max(1))      # the max is exact in fp32


def test_pair_stats_empty():
    st = R.get_backend().pair_stats(torch.zeros(2, 0, 18, device=DEV), torch.zeros(2, 0, 18, device=DEV), 2, 0, 18)
    assert st.shape == (2, 18, R.NSTAT) and float(st.abs().sum()) == 0.0


def test_vrmse_large_against_oracle():
    gen = torch.Generator().manual_seed(9)
    p = torch.randn(4, 5, 64, 48, 18, generator=gen)
    g = torch.randn(4, 5, 64, 48, 18, generator=gen) + 0.5
    m = torch.zeros(18)
    m[[0, 1, 3, 15]] = 1
    v, r = PM.compute_vrmse(p.to(DEV), g.to(DEV), m.to(DEV))
    wv, wr = GO.vrmse(p.double(), g.double(), m)
    assert abs(float(v) - float(wv)) <= 1e-5 * float(wv) and abs(float(r) - float(wr)) <= 1e-5 * float(wr)


@pytest.mark.parametrize("shape,w", [((2, 3, 33, 17, 18), 3), ((1, 2, 8, 8, 18), 4), ((1, 1, 5, 40, 18), 16)])
def test_strips_ragged_against_oracle(shape, w):
    gen = torch.Generator().manual_seed(sum(shape) + w)
    p = torch.randn(shape, generator=gen)
    g = torch.randn(shape, generator=gen)
    m = torch.zeros(2, 18)
    m[:, [2, 5, 17]] = 1
    pd = p.clone().to(DEV).requires_grad_(True)
    val = PB.compute_boundary_loss(pd, g.to(DEV), m.to(DEV), w)
    val.backward()
    p64 = p.double().requires_grad_(True)
    want = GO.boundary_2d_strips(p64, g.double(), m, w)
    want.backward()
    assert abs(float(val.detach()) - float(want.detach())) <= 1e-5 * float(want.detach())
    assert GC._rel(pd.grad.cpu().numpy(), p64.grad.numpy()) <= 1e-5


def test_patch_seam_ragged_against_oracle():
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(2, 2, 37, 50, 18, generator=gen)      # extents that are not multiples of the patch size
    m = torch.zeros(18)
    m[[0, 3, 16]] = 1
    xd = x.clone().to(DEV).requires_grad_(True)
    val = PB.compute_patch_boundary_loss(xd, m.to(DEV), 8)
    val.backward()
    x64 = x.double().requires_grad_(True)
    want = GO.patch_seams(x64, m, 8)
    want.backward()
    assert abs(float(val.detach()) - float(want.detach())) <= 1e-5 * float(want.detach())
    assert GC._rel(xd.grad.cpu().numpy(), x64.grad.numpy()) <= 1e-5


def test_cpu_tensor_rejected():
    with pytest.raises(RuntimeError, match="HIP device"):
        PM.compute_vrmse(torch.zeros(1, 1, 4, 4, 18), torch.zeros(1, 1, 4, 4, 18), torch.ones(18))


def test_shared_grad():
    GC.check_shared_grad(DEV)
