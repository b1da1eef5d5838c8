"""Shared fixtures: golden-case loading and input reconstruction."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "pde_*.npz"))
                  if not p.endswith("pde_burgers2d_ghost.npz"))


def _untensor(v):
    if isinstance(v, dict):
        if "__tensor__" in v:
            return torch.tensor(v["__tensor__"])
        return {k: _untensor(x) for k, x in v.items()}
    return v


def load_case(name, dtype=torch.float64, device="cpu"):
    """-> (meta, ctor kwargs, [field tensors], {forward kwargs}, npz)"""
    z = np.load(os.path.join(GOLDEN, f"pde_{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    kwargs = _untensor(meta["kwargs"])
    fields = [torch.tensor(z[f"in_{n}"], dtype=dtype, device=device) for n in meta["fields"]]
    extra = {}
    for k in meta["extra_keys"]:
        if f"arg_{k}" not in z.files:      # a plain keyword such as reduction='sum'
            extra[k] = meta["plain_extra"][k]
            continue
        a = torch.tensor(z[f"arg_{k}"])
        extra[k] = a.to(dtype=dtype, device=device) if a.is_floating_point() else a.to(device)
    return meta, kwargs, fields, extra, z


def backprop(total):
    """backward of a loss; with reduction='none' the loss is a field and its sum is differentiated (as the fixtures did)"""
    (total.sum() if total.dim() else total).backward()


def close(got, ref, tol, floor=0.0):
    """a scalar within tol relative (plus floor), or a field within tol in relative L2"""
    g = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    r = np.asarray(ref)
    if r.size == 1:
        return abs(float(g) - float(r)) <= tol * abs(float(r)) + floor
    return g.shape == r.shape and rel_l2(g, r) <= tol


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d
