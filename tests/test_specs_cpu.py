"""Specs, symbolic adjoints and the host-side loss classes, checked on the CPU.

pde_amd's classes run here on the NumPy spec interpreter (tests/spec_interp.py) instead of the HIP
kernels; results are compared with the golden vectors of the reference and with the CPU oracle.
"""
import json
import os

import numpy as np
import pytest
import torch

import pde_amd.losses as PL
from oracle import classes as OC
from pde_amd import functional as PF
from tests.helpers import GOLDEN, backprop, close, golden_cases, load_case, rel_l2
from tests.spec_interp import InterpBackend


@pytest.fixture(autouse=True)
def interp_backend():
    old = PF.set_backend(InterpBackend())
    yield
    PF.set_backend(old)


def _run(mod, fields, extra):
    fields = [f.clone().requires_grad_(True) for f in fields]
    total, losses = mod(*fields, **extra)
    backprop(total)
    return total, losses, [f.grad for f in fields]


@pytest.mark.parametrize("name", golden_cases())
def test_classes_match_reference_golden(name):
    meta, kwargs, fields, extra, z = load_case(name, torch.float32)
    mod = getattr(PL, meta["cls"])(**kwargs)
    total, losses, grads = _run(mod, fields, extra)
    assert close(total, z["f32_total"], 2e-5)
    ref_keys = sorted(k[9:] for k in z.files if k.startswith("f32_loss_"))
    assert sorted(losses) == ref_keys
    for k in ref_keys:
        assert close(losses[k], z[f"f64_loss_{k}"], 2e-5, 1e-30), k
    for n, g in zip(meta["fields"], grads):
        ref = z[f"f64_grad_{n}"]
        got = g.numpy() if g is not None else np.zeros_like(ref)
        assert rel_l2(got, ref) <= 2e-5, n


@pytest.mark.parametrize("name", golden_cases())
def test_classes_match_oracle_tightly(name):
    """Same fp32-rounded inputs through the fp64 oracle: differences are fp32 constant rounding only."""
    meta, kwargs, fields, extra, z = load_case(name, torch.float32)
    mod = getattr(PL, meta["cls"])(**kwargs)
    total, losses, grads = _run(mod, fields, extra)
    f64 = [f.double().requires_grad_(True) for f in fields]
    e64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in extra.items()}
    ototal, olosses = OC.build(meta["cls"], **kwargs)(*f64, **e64)
    backprop(ototal)
    assert close(total, ototal.detach().numpy(), 3e-6)
    for n, g, f in zip(meta["fields"], grads, f64):
        ref = f.grad.numpy() if f.grad is not None else np.zeros(f.shape)
        got = g.numpy() if g is not None else np.zeros_like(ref)
        assert rel_l2(got, ref) <= 3e-6, n


def test_ghost_burgers_function():
    z = np.load(os.path.join(GOLDEN, "pde_burgers2d_ghost.npz"))
    m = json.loads(bytes(z["meta"]).decode())
    pred = torch.tensor(z["in_pred"], dtype=torch.float32, requires_grad=True)
    total, lu, lv, ru, rv = PL.burgers_pde_loss(pred, m["nu"], dt=m["dt"], dx=m["dx"], dy=m["dy"])
    total.backward()
    assert abs(float(total) - float(z["f64_total"])) <= 2e-5 * abs(float(z["f64_total"]))
    assert abs(float(lu) - float(z["f64_loss_u"])) <= 2e-5 * abs(float(z["f64_loss_u"]))
    assert ru.shape == z["f64_res_u"].shape
    assert ru.requires_grad and rel_l2(ru.detach().numpy(), z["f64_res_u"]) <= 2e-5
    assert rel_l2(rv.detach().numpy(), z["f64_res_v"]) <= 2e-5
    assert rel_l2(pred.grad.numpy(), z["f64_grad_pred"]) <= 2e-5


def test_ghost_burgers_residuals_are_in_graph():
    """The returned residual fields are differentiable, as the reference's are (finetune/pde_loss.py:216): an objective
    built from them, alone or mixed with the losses, against autograd through the oracle."""
    g = torch.Generator().manual_seed(12)
    base = torch.randn(2, 4, 9, 12, 2, generator=g)
    wgt = torch.randn(2, 3, 7, 10, generator=g)
    objectives = (lambda t, lu, ru, rv, w: (ru * w).sum() + 0.5 * (rv ** 2).mean(),
                  lambda t, lu, ru, rv, w: 2.0 * t + lu + (rv * w).sum())
    for use in objectives:
        pred = base.clone().requires_grad_(True)
        t, lu, lv, ru, rv = PL.burgers_pde_loss(pred, 0.03, dt=0.02, dx=0.11, dy=0.13)
        use(t, lu, ru, rv, wgt).backward()
        p64 = base.double().requires_grad_(True)
        ot, olu, olv, oru, orv = OC.burgers_pde_loss(p64, 0.03, dt=0.02, dx=0.11, dy=0.13)
        use(ot, olu, oru, orv, wgt.double()).backward()
        assert rel_l2(pred.grad.numpy(), p64.grad.numpy()) <= 3e-5


def test_heat3d_matches_oracle():
    g = torch.Generator().manual_seed(3)
    u = torch.randn(2, 5, 6, 5, 7, generator=g)
    alpha = torch.tensor([0.01, 0.03])
    kw = dict(nx=6, ny=5, nz=7, Lx=1.0, Ly=1.5, Lz=2.0, dt=0.01, eq_scales={"heat": 3.0})
    mod = PL.Heat3DPDELoss(**kw)
    total, losses, grads = _run(mod, [u], dict(alpha=alpha))
    u64 = u.double().requires_grad_(True)
    ot, _ = OC.build("Heat3DPDELoss", **kw)(u64, alpha.double())
    ot.backward()
    assert sorted(losses) == ["heat", "total"]
    assert abs(float(total) - float(ot)) <= 3e-6 * abs(float(ot))
    assert rel_l2(grads[0].numpy(), u64.grad.numpy()) <= 3e-6


def test_gray_scott_fd4_matches_oracle():
    g = torch.Generator().manual_seed(4)
    A, B = torch.rand(2, 5, 12, 10, generator=g), torch.rand(2, 5, 12, 10, generator=g)
    kw = dict(nx=12, ny=10, dx=0.1, dy=0.15, dt=10.0)
    mod = PL.GrayScottPDELoss(laplacian="fd4", **kw)
    total, losses, grads = _run(mod, [A, B], {})
    a64, b64 = A.double().requires_grad_(True), B.double().requires_grad_(True)
    ot, _ = OC.build("GrayScottPDELoss", **kw)(a64, b64, laplacian="fd4")
    ot.backward()
    assert abs(float(total) - float(ot)) <= 3e-6 * abs(float(ot))
    assert rel_l2(grads[0].numpy(), a64.grad.numpy()) <= 3e-6
    assert rel_l2(grads[1].numpy(), b64.grad.numpy()) <= 3e-6


def test_fused_mode_host_logic():
    """backward_mode='fused': same loss and gradients as the default mode, for the assumed upstream value, another one,
    a per-equation upstream (falls back to the ordinary adjoint) and a second backward through the same graph"""
    import pde_amd
    gen = torch.Generator().manual_seed(3)
    fields0 = [torch.randn(2, 5, 8, 8, generator=gen) for _ in range(3)]
    nu = torch.tensor([0.02, 0.05])

    def run(mode, hint, use):
        fn = pde_amd.TaylorGreen2DPDELoss(nx=8, ny=8, dt=0.01)
        fn.backward_mode, fn.upstream_hint = mode, hint
        fs = [f.clone().requires_grad_(True) for f in fields0]
        total, losses = fn(*fs, nu=nu)
        obj = use(total, losses)
        obj.backward(retain_graph=True)
        g1 = [f.grad.clone() for f in fs]
        for f in fs:
            f.grad = None
        obj.backward()
        return float(total), g1, [f.grad.clone() for f in fs]

    uses = [lambda t, l: t, lambda t, l: 0.37 * t, lambda t, l: 2.0 * l["x_momentum"] + t]
    for use in uses:
        t0, ga, gb = run("auto", 1.0, use)
        for hint in (1.0, 0.37):
            t1, g1, g2 = run("fused", hint, use)
            assert t1 == t0
            for a, b, c in zip(ga, g1, g2):
                assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())
                assert float((a - c).abs().max()) <= 1e-6 * float(a.abs().max())


def test_scales_and_weights_changed_after_the_first_call_take_effect():
    """eq_scales / eq_weights / the per-timestep buffers are re-read on every call, as in the reference"""
    g = torch.Generator().manual_seed(8)
    fields = [torch.randn(2, 5, 8, 8, generator=g) for _ in range(3)]
    nu = torch.tensor([0.02, 0.05])
    per_t = {"continuity": torch.tensor([1.0, 2.0, 3.0, 4.0])}
    mod = PL.TaylorGreen2DPDELoss(nx=8, ny=8, dt=0.01, eq_scales_per_t={k: v.clone() for k, v in per_t.items()})
    ref = OC.build("TaylorGreen2DPDELoss", nx=8, ny=8, dt=0.01, eq_scales_per_t=per_t)
    f64 = [f.double() for f in fields]

    def same():
        a, _ = mod(*fields, nu=nu)
        b, _ = ref(*f64, nu=nu.double())
        assert abs(float(a) - float(b)) <= 3e-6 * abs(float(b))
        return float(a)

    v0 = same()
    mod.eq_scales["x_momentum"] = ref.scales["x_momentum"] = 3.0
    v1 = same()
    mod.eq_weights["y_momentum"] = ref.weights["y_momentum"] = 0.0      # now a skipped (guarded) equation
    v2 = same()
    mod.scales_t_continuity.mul_(2.0)
    ref.per_t_bufs["continuity"] = ref.per_t_bufs["continuity"] * 2.0
    v3 = same()
    assert len({v0, v1, v2, v3}) == 4


def test_flow_mixing_v2_golden():
    """flow_mixing_pde_loss_v2 against the reference's values (eval/pde_loss_flow_v2.py run on the same inputs): the three
    losses, the residual field, and the gradients w.r.t. the prediction and all four boundary tensors."""
    z = np.load(os.path.join(GOLDEN, "flow_v2.npz"))
    m = json.loads(bytes(z["meta"]).decode())
    names = ("pred", "left", "right", "bottom", "top")
    ins = [torch.tensor(z[f"in_{n}"], dtype=torch.float32, device="cpu", requires_grad=True) for n in names]
    loss, lt, la, res = PL.flow_mixing_pde_loss_v2(*ins, **m)
    (loss + 0.5 * lt + 0.25 * la + 0.01 * res.sum()).backward()
    for n, x in (("loss", loss), ("loss_time", lt), ("loss_adv", la), ("residual", res)):
        assert close(x, z[f"f64_{n}"], 2e-5), n
    for n, x in zip(names, ins):
        assert rel_l2(x.grad.cpu().numpy(), z[f"f64_grad_{n}"]) <= 2e-5, n


def test_scalars_out_receives_the_losses():
    """`scalars_out`: the losses land in the caller's vector and the returned scalars are views of it, same values and
    gradients as without; a vector of the wrong shape or dtype is refused."""
    g = torch.Generator().manual_seed(17)
    fields = [torch.randn(2, 5, 8, 8, generator=g) for _ in range(3)]
    nu = torch.tensor([0.02, 0.05])
    for mode in ("auto", "stored", "fused"):
        ref = PL.TaylorGreen2DPDELoss(nx=8, ny=8, dt=0.01)
        ref.backward_mode = mode
        t0, l0, g0 = _run(ref, fields, dict(nu=nu))
        mod = PL.TaylorGreen2DPDELoss(nx=8, ny=8, dt=0.01)
        mod.backward_mode = mode
        vec = torch.full((4,), -1.0)
        mod.scalars_out = vec
        t1, l1, g1 = _run(mod, fields, dict(nu=nu))
        assert t1.data_ptr() == vec[3].data_ptr() and l1["continuity"].data_ptr() == vec[0].data_ptr()
        assert vec.tolist() == [float(l0["continuity"]), float(l0["x_momentum"]), float(l0["y_momentum"]), float(t0)]
        for a, b in zip(g0, g1):
            assert torch.equal(a, b)
    for bad in (torch.zeros(3), torch.zeros(4, dtype=torch.float64), torch.zeros(8)[::2]):
        mod.scalars_out = bad
        with pytest.raises(ValueError):
            mod(*fields, nu=nu)


def test_mode_resolution_rules():
    """'auto' / 'fused' -> kernels: the measured per-spec flags, and small problems on the stored-residual kernels"""
    from pde_amd import functional as PF
    from tests.spec_interp import InterpBackend
    old = PF.set_backend(InterpBackend())
    small, PF.SMALL_FIELD_ELEMENTS = PF.SMALL_FIELD_ELEMENTS, 2_000_000
    try:
        big = 16 * 16 * 256 * 256
        assert PF.resolve_mode("fused", "tg2d", big) == "fused" and PF.resolve_mode("auto", "tg2d", big) == "recompute"
        assert PF.resolve_mode("fused", "tg2d", 1000) == "stored" and PF.resolve_mode("auto", "tg2d", 1000) == "stored"
        assert PF.resolve_mode("fused", "tg2d") == "fused"                       # size unknown: no rule
        assert PF.resolve_mode("fused", "burgers1d", 1000) == "fused"            # 1-D clips are launch-bound either way
        assert PF.resolve_mode("recompute", "tg2d", 1000) == "recompute"         # an explicit mode is kept
        assert PF.resolve_mode("fused", "heat3d", big) == "stored"               # fused_slower and auto_stored
        assert PF.resolve_mode("auto", "heat3d", big) == "stored"                # auto_stored
    finally:
        PF.SMALL_FIELD_ELEMENTS = small
        PF.set_backend(old)


def test_capture_needs_device_tensors():
    import pde_amd
    with pytest.raises(RuntimeError, match="HIP device"):
        pde_amd.Heat1DPDELoss().capture(torch.zeros(2, 9, 64), alpha=torch.ones(2))
