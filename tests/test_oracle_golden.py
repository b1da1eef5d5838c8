"""The CPU oracle against golden vectors produced by the reference classes (tools/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import classes as OC
from tests.helpers import GOLDEN, golden_cases, load_case, rel_l2


@pytest.mark.parametrize("name", golden_cases())
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_oracle_matches_reference(name, prec):
    dtype = torch.float64 if prec == "f64" else torch.float32
    tol = 1e-11 if prec == "f64" else 2e-5
    meta, kwargs, fields, extra, z = load_case(name, dtype)
    fields = [f.requires_grad_(True) for f in fields]
    fn = OC.build(meta["cls"], **kwargs)
    total, losses = fn(*fields, **extra)
    total.backward()
    assert abs(float(total) - float(z[f"{prec}_total"])) <= tol * abs(float(z[f"{prec}_total"]))
    ref_keys = sorted(k[len(prec) + 6:] for k in z.files if k.startswith(f"{prec}_loss_"))
    assert sorted(losses) == ref_keys
    for k in ref_keys:
        ref = float(z[f"{prec}_loss_{k}"])
        assert abs(float(losses[k]) - ref) <= tol * max(abs(ref), 1e-300), k
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
