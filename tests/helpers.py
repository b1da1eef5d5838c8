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


class RolloutStubModel:
    """A deterministic stand-in for the trained model of tools/eval_rollout.py (called as model(x, return_normalized=True),
    returns the normalised output with its per-channel mean and std): its output depends on its input, so a rollout that
    feeds predictions back differs from the one-step evaluation.  Used to produce the rollout fixture and to replay it."""

    def eval(self):
        return self

    def __call__(self, x, return_normalized=True):
        C = x.shape[-1]
        ramp = torch.linspace(0.0, 1.0, C, dtype=x.dtype, device=x.device).view(1, 1, 1, 1, C)
        mean, std = 0.4 * ramp - 0.2, 0.8 + 0.4 * ramp
        out = 0.55 * torch.roll(x, shifts=1, dims=2) + 0.35 * torch.tanh(x) + 0.1 * torch.roll(x, shifts=-1, dims=3)
        return (out - mean) / std, mean, std
