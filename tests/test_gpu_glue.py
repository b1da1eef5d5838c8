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


def test_band_power():
    GC.check_band_power(DEV)


def test_denormalised():
    GC.check_denormalised(DEV)


def test_rollout():
    GC.check_rollout(DEV)


def test_scale_accumulator():
    GC.check_scale_accumulator(DEV)

def test_residuals_f64():
    GC.check_residuals_f64(DEV)


@pytest.mark.parametrize("channels", [18, 36])
def test_validation_scalars_single_sweep(channels):
    GC.check_validation_scalars(DEV, channels)



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


def test_full_size_metric_properties():
    """rollout-evaluation size [64, 512, 128, 18] (no CPU oracle run at this size): the statistics kernel against fp64
    torch reductions of the same tensors, scale invariance of nRMSE, sample permutation, and the gather/scatter pair as
    exact adjoints of each other"""
    import pde_amd
    from pde_amd import reductions as R
    gen = torch.Generator(device=DEV).manual_seed(41)
    M, H, W, C = 64, 512, 128, 18
    p = torch.randn(M, H, W, C, generator=gen, device=DEV)
    g = torch.randn(M, H, W, C, generator=gen, device=DEV) * 0.7 + 0.1
    st = R.get_backend().pair_stats(p, g, M, H * W, C)                     # [M, C, 6] fp64
    d = (p.double() - g.double())
    want_d2 = (d * d).sum(dim=(1, 2))
    want_g2 = (g.double() ** 2).sum(dim=(1, 2))
    assert float(((st[:, :, R.S_D2] - want_d2).abs() / want_d2).max()) <= 1e-7      # the kernel subtracts in fp32, like the reference
    assert float(((st[:, :, R.S_G2] - want_g2).abs() / want_g2).max()) <= 1e-7
    assert torch.equal(st[:, :, R.S_MAXD].float(), (p - g).abs().amax(dim=(1, 2)))
    names = [(0, "vx"), (1, "vy"), (15, "p")]
    a = pde_amd.nrmse_per_channel(p, g, names)
    b = pde_amd.nrmse_per_channel(3.0 * p, 3.0 * g, names)                 # scale invariance
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(2)).to(DEV)
    c = pde_amd.nrmse_per_channel(p[perm], g[perm], names)
    for _, n in names:
        assert float(((a[n] - b[n]).abs() / a[n]).max()) <= 1e-6
        assert torch.equal(a[n][perm], c[n])
    # <gather(x), y> == <x, scatter(y)> on the trainer layout [B, T, H, W, C]
    out = torch.randn(4, 5, 256, 256, C, generator=gen, device=DEV, requires_grad=True)
    inp = torch.randn(4, 2, 256, 256, C, generator=gen, device=DEV)
    fields = pde_amd.gather_fields(out, inp, (0, 1, 15))
    ys = [torch.randn(f.shape, generator=gen, device=DEV) for f in fields]
    sum((f * y).sum() for f, y in zip(fields, ys)).backward()
    for ci, (f, y) in zip((0, 1, 15), zip(fields, ys)):
        assert torch.equal(f[:, 1:], out.detach()[..., ci]) and torch.equal(f[:, 0], inp[:, 0, ..., ci])
        assert torch.equal(out.grad[..., ci], y[:, 1:])
    others = [c for c in range(C) if c not in (0, 1, 15)]
    assert float(out.grad[..., others].abs().max()) == 0.0


def test_pdebench_and_lp():
    GC.check_pdebench_and_lp(DEV)
