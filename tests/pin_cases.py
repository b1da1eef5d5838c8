"""Shared bodies of the product tests against tests/golden/pins.npz (outputs of the reference's own verification code,
see tools/make_golden.py run_pins): run on the CPU interpreter backends (-m "not gpu") and on the HIP library (-m gpu)."""
import json
import os

import numpy as np
import torch

import pde_amd
from oracle import classes as OC
from oracle import pde_oracle as O
from tests.helpers import GOLDEN, rel_l2

TOL = 1e-5
Z = np.load(os.path.join(GOLDEN, "pins.npz"))
J = lambda k: json.loads(bytes(Z[k]).decode())
MODES = ("auto", "stored", "fused")


def _f32(k, dev):
    return torch.tensor(Z[k], dtype=torch.float32, device=dev)


def _against_oracle(make, oracle_fn, fields, extra, dev, modes=MODES):
    """loss, per-equation losses and gradients of the product class in every backward mode against the fp64 oracle"""
    f64 = [torch.tensor(f, dtype=torch.float64).float().double().requires_grad_(True) for f in fields]   # the fp32-rounded inputs
    e64 = {k: torch.tensor(v, dtype=torch.float64).float().double() for k, v in extra.items()}
    ot, ol = oracle_fn(*f64, **e64)
    ot.backward()
    for mode in modes:
        mod = make().to(dev)
        mod.backward_mode = mode
        fs = [torch.tensor(f, dtype=torch.float32, device=dev).requires_grad_(True) for f in fields]
        ex = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in extra.items()}
        total, losses = mod(*fs, **ex)
        total.backward()
        assert abs(float(total) - float(ot)) <= TOL * abs(float(ot)), mode
        assert sorted(losses) == sorted(ol)
        for k in ol:
            assert abs(float(losses[k]) - float(ol[k])) <= TOL * abs(float(ol[k])) + 1e-30, (mode, k)
        for a, b in zip(fs, f64):
            assert rel_l2(a.grad.cpu().numpy(), b.grad.numpy()) <= TOL, mode


def check_gray_scott_fd(dev):
    """the fused FD-c4 kernel: residual fields against tools/test_gt_pde_gray_scott.py's arrays, then loss and gradient
    against the oracle that reproduces those arrays to 1e-12"""
    par = J("gsfd_params")
    kw = dict(nx=12, ny=10, eq_scales={"A": 0.02, "B": 0.05}, eq_weights={"A": 1.0, "B": 0.7}, **par)
    mod = pde_amd.GrayScottPDELoss(laplacian="fd4", **kw).to(dev)
    res = mod.residuals(_f32("gsfd_A", dev), _f32("gsfd_B", dev))
    assert rel_l2(res["A_equation"].cpu().numpy(), Z["gsfd_R_A"]) <= TOL
    assert rel_l2(res["B_equation"].cpu().numpy(), Z["gsfd_R_B"]) <= TOL
    _against_oracle(lambda: pde_amd.GrayScottPDELoss(laplacian="fd4", **kw),
                    lambda a, b: OC.build("GrayScottPDELoss", **kw)(a, b, laplacian="fd4"), [Z["gsfd_A"], Z["gsfd_B"]], {}, dev)


def check_heat3d(dev):
    par = J("heat3d_params")
    kw = dict(nx=6, ny=5, nz=7, Lx=6 * par["dx"], Ly=5 * par["dy"], Lz=7 * par["dz"], dt=par["dt"])
    mod = pde_amd.Heat3DPDELoss(**kw).to(dev)
    with torch.no_grad():
        total, _ = mod(_f32("heat3d_u", dev), alpha=_f32("heat3d_alpha", dev))
    sites = 3 * 6 * 5 * 7
    per_sample = mod.last_per_sample[:, 0].cpu().numpy() * 2      # sums carry 1 / (B * sites): back to per-sample means
    assert np.allclose(per_sample, Z["heat3d_mse"], rtol=TOL, atol=0), (per_sample, Z["heat3d_mse"], sites)
    assert abs(float(total) - Z["heat3d_mse"].mean()) <= TOL * Z["heat3d_mse"].mean()
    kw2 = dict(kw, eq_scales={"heat": 3.0})
    _against_oracle(lambda: pde_amd.Heat3DPDELoss(**kw2), lambda u, alpha: OC.build("Heat3DPDELoss", **kw2)(u, alpha=alpha),
                    [Z["heat3d_u"]], dict(alpha=Z["heat3d_alpha"]), dev)


def check_tg3d(dev):
    par = J("tg3d_params")
    kw = dict(nx=6, ny=5, nz=7, Lx=6 * par["dx"], Ly=5 * par["dy"], Lz=7 * par["dz"], dt=par["dt"])
    mod = pde_amd.TaylorGreen3DPDELoss(**kw).to(dev)
    fields = [Z[f"tg3d_{n}"] for n in "uvwp"]
    with torch.no_grad():
        mod(*[_f32(f"tg3d_{n}", dev) for n in "uvwp"], nu=_f32("tg3d_nu", dev))
    per_sample = mod.last_per_sample.cpu().numpy() * 2
    assert np.allclose(per_sample[:, 1], Z["tg3d_u_momentum_mse"], rtol=TOL, atol=0)
    # the verifier's divergence covers every frame, the class the interior ones: compare through the oracle pinned to it
    div = O.tg3d_face_divergence(*[torch.tensor(f)[:, 1:-1] for f in fields[:3]], dx=par["dx"], dy=par["dy"], dz=par["dz"])
    assert np.allclose(per_sample[:, 0], (div ** 2).mean(dim=(1, 2, 3, 4)).numpy(), rtol=TOL, atol=0)
    kw2 = dict(kw, eq_scales={"continuity": 0.5, "u_momentum": 2.0, "v_momentum": 3.0, "w_momentum": 1.5},
               eq_weights={"v_momentum": 0.7})
    _against_oracle(lambda: pde_amd.TaylorGreen3DPDELoss(**kw2), lambda u, v, w, p, nu: OC.build("TaylorGreen3DPDELoss", **kw2)(u, v, w, p, nu=nu),
                    fields, dict(nu=Z["tg3d_nu"]), dev)


def check_ns_incom(dev):
    par = J("nsincom_params")
    kw = dict(nx=12, ny=10, Lx=12 * par["dx"], Ly=10 * par["dy"], dt=par["dt"], nu=par["nu"], skip_bl=par["skip"])
    mod = pde_amd.NSIncomPDELoss(**kw).to(dev)
    names = ("vx", "vy", None, "d")
    fields = [Z[f"nsincom_{n}"] if n else np.zeros_like(Z["nsincom_vx"]) for n in names]
    dfields = [torch.tensor(f, dtype=torch.float32, device=dev) for f in fields]
    res = mod.residuals(*dfields, _f32("nsincom_fx", dev), _f32("nsincom_fy", dev))
    s = par["skip"]
    for eq, key in zip(mod.EQS, ("div", "R_u", "R_v", "R_d")):
        got = res[eq][..., s:-s, s:-s].cpu().numpy()
        assert rel_l2(got, Z[f"nsincom_{key}"]) <= TOL, eq
        # zero outside the lattice
        outside = res[eq].clone()
        outside[..., s:-s, s:-s] = 0
        assert not outside.any()
    # with a pressure field and non-unit scales: loss and gradients against the oracle
    g = torch.Generator().manual_seed(9)
    fields[2] = torch.randn(fields[0].shape, generator=g, dtype=torch.float64).numpy()
    kw2 = dict(kw, eq_weights={"tracer": 0.6})
    fx, fy = torch.tensor(Z["nsincom_fx"]), torch.tensor(Z["nsincom_fy"])
    f32r = lambda t: t.float().double()
    _against_oracle(lambda: _Bound(pde_amd.NSIncomPDELoss(**kw2), Z["nsincom_fx"], Z["nsincom_fy"]),
                    lambda u, v, p, d: OC.build("NSIncomPDELoss", **kw2)(u, v, p, d, f32r(fx), f32r(fy)), fields, {}, dev)
    kw3 = dict(kw, use_div_correction=False)
    _against_oracle(lambda: _Bound(pde_amd.NSIncomPDELoss(**kw3), Z["nsincom_fx"], Z["nsincom_fy"]),
                    lambda u, v, p, d: OC.build("NSIncomPDELoss", **kw3)(u, v, p, d, f32r(fx), f32r(fy)), fields, {}, dev, modes=("auto",))


class _Bound(torch.nn.Module):
    """a loss module with its force fields bound (they are data: no gradient)"""

    def __init__(self, mod, fx, fy):
        super().__init__()
        self.mod = mod
        self.register_buffer("fx", torch.tensor(fx, dtype=torch.float32))
        self.register_buffer("fy", torch.tensor(fy, dtype=torch.float32))

    @property
    def backward_mode(self):
        return self.mod.backward_mode

    @backward_mode.setter
    def backward_mode(self, m):
        self.mod.backward_mode = m

    def forward(self, u, v, p, d):
        return self.mod(u, v, p, d, self.fx, self.fy)


def check_taylor_green_2d_alias(dev):
    """TaylorGreenPDELoss (imported by two reference scripts, defined by none): its successor's numbers, own nu fallback"""
    g = torch.Generator().manual_seed(4)
    fields = [torch.randn(2, 5, 12, 16, generator=g).to(dev) for _ in range(3)]
    old = pde_amd.TaylorGreenPDELoss(nx=16, ny=12, dt=0.001, nu=0.003).to(dev)
    new = pde_amd.TaylorGreen2DPDELoss(nx=16, ny=12, dt=0.001).to(dev)
    with torch.no_grad():
        a, da = old(*fields)
        b, db = new(*fields, nu=torch.full((2,), 0.003, device=dev))
        c, _ = old(*fields, nu=torch.tensor([0.01, 0.02], device=dev))
        d, _ = new(*fields, nu=torch.tensor([0.01, 0.02], device=dev))
    assert float(a) == float(b) and float(c) == float(d) and sorted(da) == sorted(db)


def check_periodic_bc(dev):
    u, s = _f32("pbc_u", dev).requires_grad_(True), _f32("pbc_s", dev).requires_grad_(True)
    total, d = pde_amd.PeriodicBCLoss(dx=0.1, dy=0.2)({"u": u, "s": s})
    total.backward()
    assert list(d) == J("pbc_keys")
    assert np.allclose([float(v.detach()) for v in d.values()], Z["pbc_values"], rtol=TOL, atol=0)
    assert rel_l2(u.grad.cpu().numpy(), Z["pbc_grad_u"]) <= TOL and rel_l2(s.grad.cpu().numpy(), Z["pbc_grad_s"]) <= TOL
    assert np.array_equal(u.grad.cpu().numpy() != 0, Z["pbc_grad_u"] != 0)
    # [T, H, W] input, as the reference's script feeds it
    t3, _ = pde_amd.PeriodicBCLoss(dx=0.1, dy=0.2)({"u": u.detach()[0]})
    ref3 = O.periodic_bc_loss({"u": torch.tensor(Z["pbc_u"])[0]})[0]
    assert abs(float(t3) - float(ref3)) <= TOL * float(ref3)


def check_eval_nrmse(dev):
    """NRMSEAccumulator fed what eval/eval_nrmse.py's loop sees, against that loop's own result"""
    data, mask = _f32("nrmse_data", dev), _f32("nrmse_mask", dev)
    t_in = data.shape[2] - 1
    chs = [int(c) for c in Z["nrmse_channels"]]
    for affine in (False, True):
        acc = pde_amd.NRMSEAccumulator(per_timestep=True)
        for i in range(data.shape[0]):
            on, mean, std = _f32("nrmse_out_norm", dev)[i], _f32("nrmse_mean", dev)[i], _f32("nrmse_std", dev)[i]
            pred = pde_amd.Denormalised(on, mean, std) if affine else on * std + mean
            acc.update(pred, data[i][:, 1:t_in + 1], mask)
        got = acc.result()
        assert sorted(got["per_channel_nrmse"]) == chs and got["num_samples"] == int(Z["nrmse_num_samples"])
        for k in ("per_channel_nrmse", "per_channel_rmse", "per_channel_gt_rms"):
            assert np.allclose([got[k][c] for c in chs], Z[f"nrmse_{k}"], rtol=TOL, atol=0), k
        assert abs(got["overall_nrmse"] - float(Z["nrmse_overall"])) <= TOL * float(Z["nrmse_overall"])
        assert np.allclose([got["per_timestep_nrmse"][c] for c in chs], Z["nrmse_per_timestep"], rtol=TOL, atol=0)


def check_boundary_6d(dev):
    mask = torch.tensor(Z["bc6d_mask"], device=dev)
    for w in (1, 2):
        p = _f32("bc6d_pred", dev).requires_grad_(True)
        val = pde_amd.compute_boundary_loss(p, _f32("bc6d_target", dev), mask, w, variant="xy_strips")
        val.backward()
        assert abs(float(val.detach()) - float(Z[f"bc6d_w{w}"])) <= TOL * float(Z[f"bc6d_w{w}"])
        assert rel_l2(p.grad.cpu().numpy(), Z[f"bc6d_w{w}_grad"]) <= TOL
        assert np.array_equal(p.grad.cpu().numpy() != 0, Z[f"bc6d_w{w}_grad"] != 0)


def check_rollout_loop(dev):
    """evaluate_rollout against the result of the reference's own loop (tools/eval_rollout.py:94-146) on the same stub model"""
    import json
    from pde_amd import rollout as PR
    from tests.helpers import RolloutStubModel
    chans = [tuple(c) for c in json.loads(bytes(Z["rollout_channels"]).decode())]
    data = _f32("rollout_data", dev)
    one, roll = PR.evaluate_rollout(PR.denormalised(RolloutStubModel()), data, int(Z["rollout_t_input"]), chans, device=dev)
    assert np.allclose([one[n] for _, n in chans], Z["rollout_one_step"], rtol=TOL, atol=0)
    assert np.allclose([roll[n] for _, n in chans], Z["rollout_rollout"], rtol=10 * TOL, atol=0)      # fp32 predictions fed back


CHECKS = {f.__name__[6:]: f for f in (check_gray_scott_fd, check_heat3d, check_tg3d, check_ns_incom, check_taylor_green_2d_alias,
                                      check_periodic_bc, check_eval_nrmse, check_boundary_6d, check_rollout_loop)}
