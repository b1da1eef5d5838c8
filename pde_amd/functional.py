"""Differentiable residual-loss primitive over the C ABI.

`residual_sums(problem, fields, ...)` returns the fp64 tensor sums[B, n_eqs] of weighted squared
residuals; its autograd backward launches the adjoint stencil kernel.  Everything runs on the
current torch HIP stream; there is no host synchronisation and no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .stencil.specs import PDE_IDS, SPECS


@dataclass
class StencilProblem:
    """What the kernels need to know besides the tensors: PDE, constants, lattice crop."""
    spec: str
    consts: List[float]
    lo: Tuple[int, int, int] = (0, 0, 0)
    hi: Tuple[int, int, int] = (0, 0, 0)

    def c_struct(self, shape: Sequence[int]) -> "_lib.Problem":
        s = SPECS[self.spec]
        nd = s.ndim
        if len(shape) != 2 + nd:
            raise ValueError(f"{self.spec}: expected a [B, T, {'*, ' * (nd - 1)}*] field, got shape {tuple(shape)}")
        p = _lib.Problem()
        p.pde_id = PDE_IDS[self.spec]
        p.batch, p.frames = int(shape[0]), int(shape[1])
        n = [1] * (3 - nd) + [int(v) for v in shape[2:]]
        for a in range(3):
            p.n[a], p.lo[a], p.hi[a] = n[a], int(self.lo[a]), int(self.hi[a])
        p.n_consts = len(self.consts)
        for i, v in enumerate(self.consts):
            p.consts[i] = float(v)
        return p


class HipBackend:
    """Launches the gfx950 kernels. Tensors must live on a HIP device."""

    name = "hip"

    @staticmethod
    def _stream(t: torch.Tensor):
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def forward(self, prob: StencilProblem, fields, aux, sparams, weights, store_residuals: bool):
        f0 = fields[0]
        if not f0.is_cuda:
            raise RuntimeError("pde_amd: fields must be on a HIP device (there is no CPU fallback)")
        L = _lib.lib()
        s = SPECS[prob.spec]
        cp = prob.c_struct(f0.shape)
        B, T = f0.shape[0], f0.shape[1]
        TR = max(T - s.tlo - s.thi, 0)
        ne = len(s.eqs)
        sums = torch.empty((B, ne), dtype=torch.float64, device=f0.device)
        ws_bytes = L.pde_amd_residual_workspace_bytes(C.byref(cp))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=f0.device)
        resid = None
        rptr = None
        if store_residuals:
            resid = [torch.empty((B, TR) + tuple(f0.shape[2:]), dtype=torch.float32, device=f0.device) for _ in range(ne)]
            rptr = _lib.ptr_array(resid, ne)
        with torch.cuda.device(f0.device):
            st = L.pde_amd_residual_forward(C.byref(cp), _lib.ptr_array(fields, len(fields)), _lib.ptr_array(aux, len(aux)),
                                            _lib.ptr_array(sparams, len(sparams)), C.c_void_p(weights.data_ptr()),
                                            C.c_void_p(sums.data_ptr()), rptr, C.c_void_p(ws.data_ptr()), ws_bytes,
                                            self._stream(f0))
        _lib.check(st, f"residual_forward[{prob.spec}]")
        return sums, resid

    def backward(self, prob: StencilProblem, fields, aux, sparams, coef, resid):
        f0 = fields[0]
        L = _lib.lib()
        cp = prob.c_struct(f0.shape)
        grads = [torch.empty_like(f) for f in fields]
        rptr = _lib.ptr_array(resid, len(resid)) if resid is not None else None
        with torch.cuda.device(f0.device):
            st = L.pde_amd_residual_backward(C.byref(cp), _lib.ptr_array(fields, len(fields)), _lib.ptr_array(aux, len(aux)),
                                             _lib.ptr_array(sparams, len(sparams)), C.c_void_p(coef.data_ptr()), rptr,
                                             _lib.ptr_array(grads, len(grads)), None, 0, self._stream(f0))
        _lib.check(st, f"residual_backward[{prob.spec}]")
        return grads


_BACKEND = HipBackend()


def set_backend(b):
    """Test hook: swap the kernel launcher (the CPU test-suite installs a spec interpreter)."""
    global _BACKEND
    old, _BACKEND = _BACKEND, b
    return old


class _ResidualSums(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, aux, sparams, weights, mode, *fields):
        need_grad = any(ctx.needs_input_grad[5:])
        store = need_grad and mode == "stored"
        sums, resid = _BACKEND.forward(prob, fields, aux, sparams, weights, store)
        ctx.prob, ctx.aux, ctx.sparams, ctx.weights, ctx.resid = prob, aux, sparams, weights, resid
        ctx.save_for_backward(*fields)
        return sums

    @staticmethod
    def backward(ctx, gsums):
        fields = ctx.saved_tensors
        # d loss / d R_k(b, t, site) = gsums[b, k] * 2 * w[k, t] * R_k
        coef = (2.0 * gsums.to(torch.float64)[:, :, None] * ctx.weights.to(torch.float64)[None]).to(torch.float32).contiguous()
        grads = _BACKEND.backward(ctx.prob, fields, ctx.aux, ctx.sparams, coef, ctx.resid)
        out = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[5:])]
        return (None, None, None, None, None, *out)


def residual_sums(prob: StencilProblem, fields: Sequence[torch.Tensor], weights: torch.Tensor,
                  sparams: Sequence[torch.Tensor] = (), aux: Sequence[torch.Tensor] = (),
                  mode: str = "stored") -> torch.Tensor:
    """sums[b, k] = sum_{t, site} weights[k, t] * R_k(b, t, site)^2   (fp64, differentiable w.r.t. fields)."""
    s = SPECS[prob.spec]
    if len(fields) != len(s.fields) or len(sparams) != len(s.sparams) or len(aux) != len(s.aux):
        raise ValueError(f"{prob.spec}: expected {len(s.fields)} fields, {len(s.sparams)} per-sample parameters, {len(s.aux)} aux fields")
    shape = fields[0].shape
    fs = []
    for f in fields:
        if f.shape != shape:
            raise ValueError(f"{prob.spec}: all fields must share one shape, got {tuple(f.shape)} vs {tuple(shape)}")
        fs.append(f.to(torch.float32).contiguous())
    sp = [p.to(device=fs[0].device, dtype=torch.float32).contiguous() for p in sparams]
    ax = [x.to(device=fs[0].device, dtype=torch.float32).contiguous() for x in aux]
    for p in sp:
        if p.shape != (shape[0],):
            raise ValueError(f"{prob.spec}: per-sample parameters must have shape [{shape[0]}], got {tuple(p.shape)}")
    w = weights.to(device=fs[0].device, dtype=torch.float32).contiguous()
    return _ResidualSums.apply(prob, tuple(ax), tuple(sp), w, mode, *fs)
