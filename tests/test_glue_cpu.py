"""CPU tests of the glue / metric layer: the oracle against the reference-generated golden values, and the
product's host logic against the same values through the NumPy interpreter backend."""
import numpy as np
import pytest
import torch

from tests import glue_cases as GC
from oracle import glue_oracle as GO
from pde_amd import glue as PG
from pde_amd import reductions as R
from tests.reduce_interp import InterpReductions
from tests import spec_interp
from pde_amd import functional as PF

CPU = torch.device("cpu")


@pytest.fixture(autouse=True)
def _interp_backends():
    old_r = R.set_backend(InterpReductions())
    old_f = PF.set_backend(spec_interp.InterpBackend())
    yield
    R.set_backend(old_r)
    PF.set_backend(old_f)


ORACLE = {
    "bc_strip_w1": lambda d, p: GO.boundary_2d_strips(p, torch.tensor(d["target"]), torch.tensor(d["mask"]), 1),
    "bc_strip_w2": lambda d, p: GO.boundary_2d_strips(p, torch.tensor(d["target"]), torch.tensor(d["mask"]), 2),
    "bc_edge": lambda d, p: GO.boundary_2d_edges(p, torch.tensor(d["target"]), torch.tensor(d["mask"])),
    "bc_1d": lambda d, p: GO.boundary_1d(p, torch.tensor(d["target1d"]), torch.tensor(d["mask"])),
    "bc_3d_faces": lambda d, p: GO.boundary_3d_faces(p, torch.tensor(d["target3d"]), torch.tensor(d["mask"])),
    "bc_3d_shells": lambda d, p: GO.boundary_3d_shells(p, torch.tensor(d["target3d"]), torch.tensor(d["mask"])),
    "noslip": lambda d, p: GO.noslip(p),
    "patch": lambda d, p: GO.patch_seams(p, torch.tensor(d["mask"])),
}


@pytest.mark.parametrize("name", sorted(ORACLE))
def test_oracle_boundary_matches_golden(name):
    d = GC.gold()
    p = torch.tensor(d[GC.BOUNDARY[name][0]]).requires_grad_(True)
    val = ORACLE[name](d, p)
    val.backward()
    assert abs(float(val.detach()) - float(d[name])) <= 1e-12 * abs(float(d[name]))
    assert np.allclose(p.grad.numpy(), d[name + "_grad"], rtol=1e-11, atol=1e-15)


def test_oracle_metrics_match_golden():
    d = GC.gold()
    p, g, m = torch.tensor(d["pred"]), torch.tensor(d["target"]), torch.tensor(d["mask"])
    v, r = GO.vrmse(p, g, m)
    assert abs(float(v) - float(d["vrmse"])) < 1e-12 and abs(float(r) - float(d["rmse"])) < 1e-12
    assert np.allclose(GO.nrmse_per_sample(p[:, 0, ..., 3], g[:, 0, ..., 3]).numpy(), d["nrmse_per_sample"], rtol=1e-12)
    pc = GO.nrmse_per_channel(p[:, 0], g[:, 0], [(0, "vx"), (1, "vy"), (15, "p")])
    assert np.allclose(np.stack([pc[k].numpy() for k in pc]), d["nrmse_per_channel"], rtol=1e-12)
    assert np.allclose(GO.rse(p[..., :3], g[..., :3]).numpy(), d["rse"], rtol=1e-12)


def test_oracle_pdebench_and_lp_match_golden():
    d = GC.gold()
    p, g = torch.tensor(d["pred"]), torch.tensor(d["target"])
    avg, per = GO.pdebench_nrmse(p[:, 0], g[:, 0])
    assert np.allclose(per, d["pdebench_nrmse"], rtol=1e-12) and abs(avg - float(d["pdebench_nrmse_avg"])) < 1e-12
    x, y = p[:, 0, ..., 2], g[:, 0, ..., 2]
    for pn in (1, 2, 3):
        for red in ("none", "mean", "sum"):
            assert np.allclose(GO.lp_loss(x, y, pn, red).numpy(), d[f"lp{pn}_{red}"], rtol=1e-12)
    assert np.allclose(GO.fft_lp_loss(x, y, 2, "none").numpy(), d["fftlp2_none"], rtol=1e-12)
    assert np.allclose(GO.fft_lp_loss(x, y, 2, "mean", 1, 5).numpy(), d["fftlp2_band_mean"], rtol=1e-12)
    assert np.allclose(GO.fft_lp_loss(x, y, 1, "sum").numpy(), d["fftlp1_sum"], rtol=1e-12)


def test_pdebench_and_lp_host_logic():
    GC.check_pdebench_and_lp(CPU)


@pytest.mark.parametrize("tag", ["1d", "2d", "3d"])
def test_oracle_metric_func_matches_golden(tag):
    d = np.load(GC.os.path.join(GC.GOLD, "metric_func.npz"))
    p, g = torch.tensor(d[f"{tag}_pred"]), torch.tensor(d[f"{tag}_target"])
    for mode, if_mean in (("mean", True), ("full", False)):
        res = GO.metric_func(p, g, if_mean=if_mean, iLow=2, iHigh=4)
        for i, r in enumerate(res):
            want = d[f"{tag}_{mode}_{i}"]
            assert tuple(r.shape) == tuple(want.shape)
            assert np.allclose(r.numpy(), want, rtol=2e-5, atol=1e-7, equal_nan=True), (tag, mode, i)


@pytest.mark.parametrize("name", sorted(GC.BOUNDARY))
def test_boundary_host_logic(name):
    GC.check_boundary(name, CPU)


def test_boundary_scaled_upstream():
    GC.check_boundary_scaled_upstream(CPU)


def test_metrics_host_logic():
    GC.check_metrics(CPU)


def test_accumulator_host_logic():
    GC.check_accumulator(CPU)


def test_band_power_host_logic():
    GC.check_band_power(CPU)


def test_denormalised_host_logic():
    GC.check_denormalised(CPU)


def test_rollout_host_logic():
    GC.check_rollout(CPU)


def test_scale_accumulator_host_logic():
    GC.check_scale_accumulator(CPU)

def test_residuals_f64():
    GC.check_residuals_f64(CPU)


def test_validation_scalars_single_sweep():
    GC.check_validation_scalars(CPU)



@pytest.mark.parametrize("tag", ["1d", "2d", "3d"])
def test_metric_func_host_logic(tag):
    GC.check_metric_func(CPU, tag)


@pytest.mark.parametrize("transpose", [False, True])
def test_gather_host_logic(transpose):
    GC.check_gather(CPU, transpose)


def test_gather_3d_and_1d_host_logic():
    GC.check_gather(CPU, False, (3, 4, 5))
    GC.check_gather(CPU, False, (9,))
    GC.check_gather(CPU, False, (7, 6), 18, (2, 2, 15))      # a channel listed twice


def test_compute_pde_loss_host_logic():
    GC.check_compute_pde_loss(CPU)


def test_registry_errors():
    with pytest.raises(ValueError, match="Unknown pde_type"):
        PG.create_pde_loss({"pde_type": "nope"}, CPU)
    with pytest.raises(ValueError, match="Unknown pde_type"):
        PG.compute_pde_loss_generic(torch.zeros(1, 1, 4, 4, 18), torch.zeros(1, 1, 4, 4, 18), None, {"pde_type": "nope"}, {})
    assert sorted(PG.PDE_LOSS_REGISTRY) == ["advdiff_2d", "burgers_2d_ch", "kp2_2d", "taylor_green_2d", "wave_2d"]


def test_faces_variant_index_error():
    from pde_amd import boundary as PB
    m = torch.zeros(18)
    m[[0, 15]] = 1
    x = torch.zeros(1, 1, 4, 8, 4, 18)
    with pytest.raises(IndexError):
        PB.compute_boundary_loss(x, x, m, variant="faces")


def test_shared_grad_host_logic():
    GC.check_shared_grad(CPU)
