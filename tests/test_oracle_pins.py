"""The CPU oracle against the outputs of the reference's own verification functions (tests/golden/pins.npz, produced by
tools/make_golden.py run_pins): what the reference tree has no loss class for, or only inside a driver loop."""
import json
import os

import numpy as np
import torch

from oracle import classes as OC
from oracle import glue_oracle as GO
from oracle import pde_oracle as O
from tests.helpers import GOLDEN, rel_l2

Z = np.load(os.path.join(GOLDEN, "pins.npz"))
T = lambda k: torch.tensor(Z[k])
J = lambda k: json.loads(bytes(Z[k]).decode())


def test_gray_scott_fd_residuals():
    """tools/test_gt_pde_gray_scott.py:27-79 (NumPy, fp64): the residual arrays themselves"""
    ra, rb = O.residuals_gray_scott_fd(T("gsfd_A"), T("gsfd_B"), **J("gsfd_params"))
    assert rel_l2(ra.numpy(), Z["gsfd_R_A"]) <= 1e-12
    assert rel_l2(rb.numpy(), Z["gsfd_R_B"]) <= 1e-12


def test_heat3d_mean_squared_residual():
    """finetune/generate_analytical_3d.py:160-178, one sample and one scalar alpha per call"""
    par = J("heat3d_params")
    u, alpha = T("heat3d_u"), T("heat3d_alpha")
    (r,) = O.residuals_heat3d(u, alpha, **par)
    mse = (r ** 2).mean(dim=(1, 2, 3, 4))
    assert np.allclose(mse.numpy(), Z["heat3d_mse"], rtol=1e-12, atol=0)
    # the class: mean over the batch of equally sized samples = mean of the per-sample values
    fn = OC.build("Heat3DPDELoss", nx=6, ny=5, nz=7, Lx=6 * par["dx"], Ly=5 * par["dy"], Lz=7 * par["dz"], dt=par["dt"])
    total, d = fn(u, alpha=alpha)
    assert abs(float(total) - Z["heat3d_mse"].mean()) <= 1e-12 * Z["heat3d_mse"].mean()
    assert sorted(d) == ["heat", "total"]


def test_taylor_green_3d():
    """finetune/generate_analytical_3d.py:97-157: divergence over every frame, u momentum over the interior frames"""
    par = J("tg3d_params")
    u, v, w, p, nu = (T(f"tg3d_{n}") for n in ("u", "v", "w", "p", "nu"))
    div_all = O.tg3d_face_divergence(u, v, w, dx=par["dx"], dy=par["dy"], dz=par["dz"])
    assert np.allclose((div_all ** 2).mean(dim=(1, 2, 3, 4)).numpy(), Z["tg3d_div_mse"], rtol=1e-12, atol=0)
    div, ru, rv, rw = O.residuals_tg3d(u, v, w, p, nu, **par)
    assert torch.equal(div, div_all[:, 1:-1])
    assert np.allclose((ru ** 2).mean(dim=(1, 2, 3, 4)).numpy(), Z["tg3d_u_momentum_mse"], rtol=1e-12, atol=0)
    # v and w momenta are the u momentum with the axes permuted: (x, y, z) -> (y, z, x) maps (u, v, w) -> (v, w, u)
    perm = lambda f: f.permute(0, 1, 3, 4, 2)
    _, ru2, rv2, rw2 = O.residuals_tg3d(perm(v), perm(w), perm(u), perm(p), nu, dx=par["dy"], dy=par["dz"], dz=par["dx"], dt=par["dt"])
    assert rel_l2(perm(rv).numpy(), ru2.numpy()) <= 1e-13 and rel_l2(perm(rw).numpy(), rv2.numpy()) <= 1e-13


def test_ns_incom_npinn_residuals():
    """tools/test_gt_pde_ns_incom.py:258-378 with zero pressure (the verification data has none)"""
    par = J("nsincom_params")
    vx, vy, d, fx, fy = (T(f"nsincom_{n}") for n in ("vx", "vy", "d", "fx", "fy"))
    kw = dict(dx=par["dx"], dy=par["dy"], dt=par["dt"], nu=par["nu"], skip=par["skip"])
    res = O.residuals_ns_incom(vx, vy, torch.zeros_like(vx), d, fx, fy, **kw)
    for r, name in zip(res, ("div", "R_u", "R_v", "R_d")):
        assert rel_l2(r.numpy(), Z[f"nsincom_{name}"]) <= 1e-12, name
    for r, name in zip(res, ("cont_rms", "mom_x_rms", "mom_y_rms", "tracer_rms")):
        got = (r ** 2).mean(dim=(1, 2, 3)).sqrt().numpy()
        assert np.allclose(got, Z[f"nsincom_{name}"], rtol=1e-12, atol=0), name
    # the pressure term of the specification (:109-111, :122-123): face-averaged = central difference
    p = torch.randn(vx.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    resp = O.residuals_ns_incom(vx, vy, p, d, fx, fy, **kw)
    s, pm = par["skip"], p[:, 1:-1]
    px = (pm[..., s + 1:-s + 1 or None, s:-s] - pm[..., s - 1:-s - 1, s:-s]) / (2 * par["dx"])
    py = (pm[..., s:-s, s + 1:-s + 1 or None] - pm[..., s:-s, s - 1:-s - 1]) / (2 * par["dy"])
    assert rel_l2((resp[1] - res[1]).numpy(), px.numpy()) <= 1e-10 and rel_l2((resp[2] - res[2]).numpy(), py.numpy()) <= 1e-10
    assert torch.equal(resp[0], res[0]) and torch.equal(resp[3], res[3])


def test_periodic_bc_loss():
    """tools/test_gt_periodic_bc.py:38-105"""
    u, s = T("pbc_u").requires_grad_(True), T("pbc_s").requires_grad_(True)
    total, d = O.periodic_bc_loss({"u": u, "s": s})
    total.backward()
    assert list(d) == J("pbc_keys")
    assert np.allclose([float(v) for v in d.values()], Z["pbc_values"], rtol=1e-13, atol=0)
    assert abs(float(total) - float(Z["pbc_total"])) <= 1e-13 * float(Z["pbc_total"])
    assert rel_l2(u.grad.numpy(), Z["pbc_grad_u"]) <= 1e-13 and rel_l2(s.grad.numpy(), Z["pbc_grad_s"]) <= 1e-13


def test_eval_nrmse_loop():
    """eval/eval_nrmse.py:74-212 run around a stub model, dataloader and accelerator"""
    data, mask = T("nrmse_data"), T("nrmse_mask")
    t_in = data.shape[2] - 1
    batches = []
    for i in range(data.shape[0]):
        pred = T("nrmse_out_norm")[i] * T("nrmse_std")[i] + T("nrmse_mean")[i]
        batches.append((pred, data[i][:, 1:t_in + 1], mask))
    got = GO.eval_nrmse(batches, per_timestep=True)
    chs = [int(c) for c in Z["nrmse_channels"]]
    assert sorted(got["per_channel_nrmse"]) == chs and got["num_samples"] == int(Z["nrmse_num_samples"])
    for k in ("per_channel_nrmse", "per_channel_rmse", "per_channel_gt_rms"):
        assert np.allclose([got[k][c] for c in chs], Z[f"nrmse_{k}"], rtol=1e-6, atol=0), k      # fp32 sums in the reference
    assert abs(got["overall_nrmse"] - float(Z["nrmse_overall"])) <= 1e-6 * float(Z["nrmse_overall"])
    assert np.allclose([got["per_timestep_nrmse"][c] for c in chs], Z["nrmse_per_timestep"], rtol=1e-6, atol=0)


def test_boundary_loss_on_6d_tensor():
    """finetune/train_burgers_3d_apebench_lora.py:115-150 applied to [B, T, X, Y, Z, C]"""
    pred, tgt, mask = T("bc6d_pred"), T("bc6d_target"), T("bc6d_mask")
    for w in (1, 2):
        q = pred.clone().requires_grad_(True)
        val = GO.boundary_3d_xy_strips(q, tgt, mask, w)
        val.backward()
        assert abs(float(val) - float(Z[f"bc6d_w{w}"])) <= 1e-13 * float(val)
        assert rel_l2(q.grad.numpy(), Z[f"bc6d_w{w}_grad"]) <= 1e-13


def test_rollout_loop():
    """tools/eval_rollout.py:94-146 (`evaluate`, extracted from the script and run around tests/helpers.py's stub model)"""
    from tests.helpers import RolloutStubModel
    model = RolloutStubModel()

    def predict(x):
        out, mean, std = model(x)
        return out * std + mean
    chans = [tuple(c) for c in J("rollout_channels")]
    one, roll = GO.rollout_evaluate(predict, T("rollout_data"), int(Z["rollout_t_input"]), chans)
    assert np.allclose([one[n] for _, n in chans], Z["rollout_one_step"], rtol=1e-12, atol=0)
    assert np.allclose([roll[n] for _, n in chans], Z["rollout_rollout"], rtol=1e-12, atol=0)
