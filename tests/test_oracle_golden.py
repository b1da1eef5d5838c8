"""The CPU oracle against golden vectors produced by the reference classes (tools/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import classes as OC
from tests.helpers import GOLDEN, backprop, close, golden_cases, load_case, rel_l2


@pytest.mark.parametrize("name", golden_cases())
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_matches_reference(name, prec):
    dtype = torch.float64 if prec == "f64" else torch.float32
    tol = 1e-11 if prec == "f64" else 2e-5
    meta, kwargs, fields, extra, z = load_case(name, dtype)
    fields = [f.requires_grad_(True) for f in fields]
    fn = OC.build(meta["cls"], **kwargs)
    total, losses = fn(*fields, **extra)
    backprop(total)
    assert close(total, z[f"{prec}_total"], tol)
    ref_keys = sorted(k[len(prec) + 6:] for k in z.files if k.startswith(f"{prec}_loss_"))
    assert sorted(losses) == ref_keys
    for k in ref_keys:
        assert close(losses[k], z[f"{prec}_loss_{k}"], tol, 1e-300), k
    for n, f in zip(meta["fields"], fields):
        ref = z[f"{prec}_grad_{n}"]
        got = f.grad.numpy() if f.grad is not None else np.zeros_like(ref)
        assert rel_l2(got, ref) <= (1e-10 if prec == "f64" else 2e-5), n


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_ghost_burgers(prec):
    z = np.load(os.path.join(GOLDEN, "pde_burgers2d_ghost.npz"))
    m = json.loads(bytes(z["meta"]).decode())
    dtype = torch.float64 if prec == "f64" else torch.float32
    pred = torch.tensor(z["in_pred"], dtype=dtype, requires_grad=True)
    total, lu, lv, ru, rv = OC.burgers_pde_loss(pred, m["nu"], dt=m["dt"], dx=m["dx"], dy=m["dy"])
    total.backward()
    tol = 1e-11 if prec == "f64" else 2e-5
    assert abs(float(total) - float(z[f"{prec}_total"])) <= tol * abs(float(z[f"{prec}_total"]))
    assert rel_l2(ru.detach().numpy(), z[f"{prec}_res_u"]) <= tol
    assert rel_l2(rv.detach().numpy(), z[f"{prec}_res_v"]) <= tol
    assert rel_l2(pred.grad.numpy(), z[f"{prec}_grad_pred"]) <= (1e-10 if prec == "f64" else 2e-5)


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_flow_mixing_v2(prec):
    """eval/pde_loss_flow_v2.py: losses, residual field and the gradients w.r.t. the prediction and the boundary values"""
    z = np.load(os.path.join(GOLDEN, "flow_v2.npz"))
    m = json.loads(bytes(z["meta"]).decode())
    dtype = torch.float64 if prec == "f64" else torch.float32
    names = ("pred", "left", "right", "bottom", "top")
    ins = [torch.tensor(z[f"in_{n}"], dtype=dtype, requires_grad=True) for n in names]
    loss, lt, la, res = OC.flow_mixing_pde_loss_v2(*ins, **m)
    (loss + 0.5 * lt + 0.25 * la + 0.01 * res.sum()).backward()
    tol = 1e-11 if prec == "f64" else 2e-5
    for n, x in (("loss", loss), ("loss_time", lt), ("loss_adv", la), ("residual", res)):
        assert close(x, z[f"{prec}_{n}"], tol), n
    for n, x in zip(names, ins):
        assert rel_l2(x.grad.numpy(), z[f"{prec}_grad_{n}"]) <= (1e-10 if prec == "f64" else 2e-5), n
