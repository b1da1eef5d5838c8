"""loss_fn.capture(): a replayed forward + backward gives what the eager step gives, bit for bit, also after the inputs
were overwritten in place; the outputs keep their addresses."""
import pytest
import torch

import pde_amd

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cases():
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g)
    return {
        "tg2d_fused": (pde_amd.TaylorGreen2DPDELoss(nx=64, ny=32, dt=0.01), "fused", [r(2, 6, 32, 64) for _ in range(3)], dict(nu=torch.tensor([0.02, 0.05]))),
        "tg2d_auto": (pde_amd.TaylorGreen2DPDELoss(nx=64, ny=32, dt=0.01), "auto", [r(2, 6, 32, 64) for _ in range(3)], dict(nu=torch.tensor([0.02, 0.05]))),
        "burgers1d": (pde_amd.Burgers1DPDELoss(), "fused", [r(4, 9, 256)], dict(nu=torch.tensor([0.01, 0.02, 0.03, 0.04]))),
        "gray_scott_fft": (pde_amd.GrayScottPDELoss(nx=32, ny=32, dx=2.0 / 32, dy=2.0 / 32), "fused", [r(2, 5, 32, 32).abs() for _ in range(2)], {}),
        "gray_scott_fd": (pde_amd.GrayScottPDELoss(nx=32, ny=32, dx=2.0 / 32, dy=2.0 / 32, laplacian="fd4"), "fused", [r(2, 5, 32, 32).abs() for _ in range(2)], {}),
    }


def _eager(mod, fields, kw):
    fs = [f.detach().clone().requires_grad_(True) for f in fields]
    total, losses = mod(*fs, **kw)
    total.backward()
    return total.detach().clone(), {k: v.detach().clone() for k, v in losses.items()}, [f.grad.clone() for f in fs]


@pytest.mark.parametrize("name", sorted(_cases()))
def test_replay_matches_eager(name):
    mod, mode, fields, kw = _cases()[name]
    mod = mod.to(DEV)
    mod.backward_mode = mode
    fields = [f.to(DEV).requires_grad_(True) for f in fields]
    kw = {k: v.to(DEV) for k, v in kw.items()}
    step = mod.capture(*fields, **kw)
    assert isinstance(step, pde_amd.CapturedLossStep)
    ptrs = [g.data_ptr() for g in step.grads] + [step.total.data_ptr()]
    for trial in range(3):
        if trial:
            with torch.no_grad():
                for i, f in enumerate(fields):      # new inputs go INTO the captured tensors
                    f.copy_(torch.roll(f, trial + i, -1) * (1.0 + 0.25 * trial))
        total, losses = step()
        want_total, want_losses, want_grads = _eager(mod, fields, kw)
        assert torch.equal(total, want_total)
        assert set(losses) == set(want_losses) and all(torch.equal(losses[k], want_losses[k]) for k in losses)
        for g, w in zip(step.grads, want_grads):
            assert torch.equal(g, w)
        assert [g.data_ptr() for g in step.grads] + [step.total.data_ptr()] == ptrs
        assert all(f.grad is g for f, g in zip(fields, step.grads))


def test_capture_without_gradients_is_a_forward_graph():
    mod = pde_amd.Heat1DPDELoss().to(DEV)
    u = torch.randn(3, 9, 256, device=DEV)
    alpha = torch.tensor([0.01, 0.02, 0.03], device=DEV)
    step = mod.capture(u, alpha=alpha)
    total, _ = step()
    want, _ = mod(u, alpha=alpha)
    assert torch.equal(total, want) and step.grads == [None]


def test_capture_rejects_non_leaf_fields():
    mod = pde_amd.Heat1DPDELoss().to(DEV)
    w = torch.randn(3, 9, 256, device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError, match="leaf"):
        mod.capture(w * 2.0, alpha=torch.full((3,), 0.01, device=DEV))
