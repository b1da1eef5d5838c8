"""PDE-residual loss modules with the reference's constructor / forward signatures.

Each class mirrors one `nn.Module` of the reference's finetune/pde_loss_verified.py (cited per class):
same constructor arguments and defaults, same `forward` arguments, same returned
`(total, {equation: loss})` with the same dict keys, same public attributes (`eq_scales`,
`eq_weights`, `use_per_t_scales` / `has_per_t_scales`, `scales_t_<eq>` buffers).  The arithmetic runs in
the gfx950 stencil kernels through `pde_amd.functional.residual_sums`; the host code below only turns
scales / weights / per-timestep scales into the `[n_eqs, T_res]` weight table the kernels consume.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .functional import (StencilProblem, residual_fields, residual_losses, residual_losses_and_fields, spectral_laplacian_pair,
                         spectral_pair_supported, totals_into)
from .stencil.specs import SPECS

Tensor = torch.Tensor


class StencilPDELoss(nn.Module):
    """Shared host logic: weight table, per-sample parameter defaults, result dict."""

    SPEC: str = ""
    EQS: Sequence[str] = ()                 # dict keys, in the reference's insertion order
    GUARDED: Sequence[str] = ()             # equations skipped (exact 0.0, no gradient) when their weight is <= 0
    PER_T: str = "none"                     # 'helper': _get_per_t_scale(+t_offset); 'slice': scales_t_X[:T_res]; 'none'
    PER_T_FLAG: str = "use_per_t_scales"    # attribute name the reference exposes
    PER_T_EQS: Optional[Sequence[str]] = None   # equations that honour per-t scales (None = all)
    PER_T_ALIAS: Dict[str, str] = {}        # equation -> buffer suffix when they differ
    TOTAL_KEY: bool = False                 # whether the dict carries 'total'
    SCALE_KEYS: Dict[str, str] = {}         # equation -> key looked up in eq_scales / eq_weights when they differ
    backward_mode = "auto"                  # 'recompute' | 'stored' | 'fused' | 'auto' (recompute where the PDE supports it)
    # 'fused': when the fields need gradients, forward computes the loss AND the gradient of upstream_hint * total in
    # one pass (the adjoint kernel recomputes every residual anyway); backward hands that gradient out.  If autograd
    # then delivers another upstream value (lambda_pde, a loss scaler) the gradient is rescaled on the device, one more
    # pass over it: set upstream_hint to the factor the training loop applies to `total`.  No host synchronisation.
    upstream_hint = 1.0
    # A contiguous fp32 vector of len(EQS) + 1 elements on the fields' device, or None.  When set, the mean-reduction forward
    # writes [loss of every equation..., their sum] into it and the returned scalars are views of it: a training loop that
    # logs through dist.ScalarWindow hands a row of the window here and pays no copy launch per step.  Entries of skipped
    # equations hold the kernel's value (0), and the total is the kernel's sum over the kept equations.
    scalars_out: Optional[Tensor] = None

    def _init_common(self, default_scales, eq_scales, eq_weights, eq_scales_per_t, per_t_supported=True):
        self.eq_scales = {**default_scales, **(eq_scales or {})}
        self.eq_weights = {**{k: 1.0 for k in default_scales}, **(eq_weights or {})}
        if per_t_supported:
            setattr(self, self.PER_T_FLAG, eq_scales_per_t is not None)
            if eq_scales_per_t is not None:
                for name, scales in eq_scales_per_t.items():
                    self.register_buffer(f"scales_t_{name}", torch.as_tensor(scales).float())
        self._wt_cache = {}

    # -- to be provided by subclasses -------------------------------------------------------------
    def _params(self) -> Dict[str, float]:
        raise NotImplementedError

    def _lattice(self, spatial: Sequence[int]) -> Tuple[Tuple[int, int, int], Tuple[int, int, int]]:
        return (0, 0, 0), (0, 0, 0)

    # -- helpers ----------------------------------------------------------------------------------
    def _problem(self, spatial) -> StencilProblem:
        vals = tuple(self._params().values())      # the geometry attributes are public and mutable: part of the key
        try:
            key = ("problem", *spatial, vals)
            hit = self._wt_cache.get(key)
        except TypeError:                          # an unhashable parameter value
            key = ("problem", *spatial, repr(vals))
            hit = self._wt_cache.get(key)
        if hit is None:
            hit = self._wt_cache[key] = self._build_problem(spatial)
        return hit

    def _build_problem(self, spatial) -> StencilProblem:
        spec = SPECS[self.SPEC]
        p = self._params()
        consts = [fn(p) for _, fn in spec.consts]
        lo, hi = self._lattice(spatial)
        return StencilProblem(self.SPEC, consts, lo, hi)

    def _per_t_scale(self, eq: str, T_res: int, t_offset: Optional[Tensor]) -> Optional[Tensor]:
        """The [T_res] scale vector the reference would divide by, or None for the fixed scale."""
        if self.PER_T == "none" or not getattr(self, self.PER_T_FLAG, False):
            return None
        if self.PER_T_EQS is not None and eq not in self.PER_T_EQS:
            return None
        buf = getattr(self, f"scales_t_{self.PER_T_ALIAS.get(eq, eq)}", None)
        if buf is None:
            return None
        if self.PER_T == "slice":                      # e.g. :2487-2490
            if buf.shape[0] < T_res:
                raise ValueError(f"scales_t_{eq} has {buf.shape[0]} entries, fewer than the {T_res} residual frames")
            return buf[:T_res].clamp(min=1e-8)
        # 'helper' = _get_per_t_scale, :854-901
        T_full = buf.shape[0]
        if t_offset is not None:
            off = t_offset.to(device=buf.device, dtype=torch.long).view(-1)
            ok = (off >= 0) & (off + T_res <= T_full)
            clamped = off.clamp(max=T_full - T_res).clamp(min=0)
            start = torch.where(ok, off, clamped)
            idx = start[:, None] + torch.arange(T_res, device=buf.device)[None, :]
            scale = buf[idx].mean(dim=0)
        else:
            scale = buf[:T_res] if T_res <= T_full else buf.mean().expand(T_res)
        return scale.clamp(min=1e-8)

    def _weights(self, T_res: int, n_sites: int, device, t_offset) -> Tuple[Tensor, List[bool]]:
        """wt[k, t] = w_k / (s^2 N) with s the fixed or per-timestep scale; zero rows for skipped equations."""
        # eq_scales, eq_weights and the per-timestep buffers are public and mutable (a schedule on a weight, load_state_dict):
        # their current values are part of the key, so a change takes effect on the next call, as in the reference
        key = (T_res, n_sites, str(device)) if t_offset is None else None
        state = self._scale_state() if key is not None else None
        if key is not None and key in self._wt_cache and self._wt_cache[key][0] == state:
            return self._wt_cache[key][1]
        rows, skipped = [], []
        for eq in self.EQS:
            k = self.SCALE_KEYS.get(eq, eq)
            w = float(self.eq_weights[k])
            skip = (eq in self.GUARDED) and not (w > 0)
            skipped.append(skip)
            if skip or T_res <= 0 or n_sites <= 0:
                rows.append(torch.zeros(max(T_res, 0), dtype=torch.float64, device=device))
                continue
            st = self._per_t_scale(eq, T_res, t_offset)
            if st is not None:
                rows.append(w / (st.to(device=device, dtype=torch.float64) ** 2 * n_sites))
            else:
                s = self._fixed_scale(eq)
                rows.append(torch.full((T_res,), w / (s ** 2 * n_sites), dtype=torch.float64, device=device))
        wt = torch.stack(rows).to(torch.float32) if rows else torch.zeros((0, max(T_res, 0)), device=device)
        out = (wt, skipped)
        if key is not None:
            self._wt_cache[key] = (state, out)      # one entry per shape: a newer state replaces the older one
        return out

    def _scale_state(self) -> tuple:
        bufs = tuple((n, b._version, b.data_ptr()) for n, b in self._buffers.items() if b is not None and n.startswith("scales_t_"))
        flag = bool(getattr(self, self.PER_T_FLAG, False))
        return (tuple(self.eq_scales), tuple(map(float, self.eq_scales.values())), tuple(self.eq_weights),
                tuple(map(float, self.eq_weights.values())), flag, bufs)

    def _fixed_scale(self, eq: str) -> float:
        return float(self.eq_scales[self.SCALE_KEYS.get(eq, eq)])

    def _sample_param(self, value, default, B: int, ref: Tensor) -> Tensor:
        if value is None:
            value = default
        dt = torch.float64 if ref.dtype is torch.float64 else torch.float32      # fp64 fields: the fp64 residual evaluation
        if torch.is_tensor(value):
            if value.dtype is dt and value.dim() == 1 and value.shape[0] == B and value.device == ref.device:
                return value
            v = value.to(device=ref.device, dtype=dt).reshape(-1)
            return v.expand(B) if v.numel() == 1 else v
        return torch.full((B,), float(value), dtype=dt, device=ref.device)

    def upstream_scalar(self, device) -> Tensor:
        """The device scalar holding `upstream_hint`, the upstream gradient 'fused' mode assumes for `total`.  Passing this
        very tensor to `total.backward(gradient=...)` tells the backward that the gradient computed in forward is the one
        asked for, without a launch that compares the two values on the device."""
        hint = float(self.upstream_hint)
        return self._buf(("assumed", hint), lambda: torch.tensor(hint, dtype=torch.float32), torch.device(device))

    def _evaluate(self, fields: Sequence[Tensor], sparams: Sequence[Tensor] = (), aux: Sequence[Tensor] = (),
                  t_offset: Optional[Tensor] = None, reduction: str = "mean"):
        if reduction != "mean":
            return self._evaluate_unreduced(fields, sparams, aux, reduction)
        spec = SPECS[self.SPEC]
        f0 = fields[0]
        if f0.dim() != 2 + spec.ndim:
            raise ValueError(f"{type(self).__name__}: fields must be [B, T, {spec.ndim} spatial dims], got {tuple(f0.shape)}")
        B, T = f0.shape[0], f0.shape[1]
        spatial = list(f0.shape[2:])
        prob = self._problem(spatial)
        full = [1] * (3 - spec.ndim) + spatial
        ext = [max(full[a] - prob.lo[a] - prob.hi[a], 0) for a in range(3)]
        T_res = T - spec.tlo - spec.thi
        n_sites = B * max(T_res, 0) * ext[0] * ext[1] * ext[2]
        wt, skipped = self._weights(T_res, n_sites, f0.device, t_offset)
        if getattr(self, "_capture", None) is not None:      # residuals(): the fields themselves, no loss
            self._capture.update(zip(self.EQS, residual_fields(prob, fields, wt, sparams, aux)))
            self._capture["__sites_per_frame__"] = B * ext[0] * ext[1] * ext[2]
            return None, {}
        mode = self.backward_mode
        if mode == "fused":
            hint = float(self.upstream_hint)
            mode = ("fused", self._buf(("assumed", hint), lambda: torch.tensor(hint, dtype=torch.float32), f0.device))
        with totals_into(self.scalars_out):
            totals, sums = residual_losses(prob, fields, wt, sparams, aux, mode=mode)
        self.last_per_sample = sums.detach()        # [B, n_eqs] fp64 diagnostic: each sample's share of every loss
        losses: Dict[str, Tensor] = {}
        all_kept = not any(skipped) and n_sites > 0
        for k, eq in enumerate(self.EQS):
            if skipped[k]:
                losses[eq] = self._zero(f0.device)
            elif n_sites == 0:
                losses[eq] = totals[k] * float("nan")      # mean over an empty lattice
            else:
                losses[eq] = totals[k]
        # the kernel's fp32 left-to-right sum is the reference's sum(losses.values()) when nothing was skipped
        total = totals[-1] if all_kept else sum(losses.values())
        if self.TOTAL_KEY:
            losses["total"] = total
        return total, losses

    def _evaluate_unreduced(self, fields, sparams, aux, reduction: str):
        """reduction 'sum' or 'none' with the fixed scales, as the reference's classes that take the argument do
        (finetune/pde_loss_verified.py:286-294): w_k * sum(R_k^2) / s_k^2, or the fields w_k * R_k^2 / s_k^2 themselves
        (on the residual lattice, in the autograd graph) and their sum as the total."""
        if reduction not in ("sum", "none"):
            raise ValueError(f"reduction must be 'mean', 'sum' or 'none', got {reduction!r}")
        spec = SPECS[self.SPEC]
        f0 = fields[0]
        if f0.dim() != 2 + spec.ndim:
            raise ValueError(f"{type(self).__name__}: fields must be [B, T, {spec.ndim} spatial dims], got {tuple(f0.shape)}")
        spatial = list(f0.shape[2:])
        prob = self._problem(spatial)
        T_res = f0.shape[1] - spec.tlo - spec.thi
        coef = [float(self.eq_weights[self.SCALE_KEYS.get(eq, eq)]) / self._fixed_scale(eq) ** 2 for eq in self.EQS]
        wt = torch.tensor(coef, dtype=torch.float32, device=f0.device)[:, None].expand(len(coef), max(T_res, 0)).contiguous()
        losses: Dict[str, Tensor] = {}
        if reduction == "sum":
            totals, _ = residual_losses(prob, fields, wt, sparams, aux, mode=self.backward_mode if self.backward_mode != "fused" else "auto")
            losses.update(zip(self.EQS, totals[:-1]))
            total = totals[-1]
        else:
            _, _, resid = residual_losses_and_fields(prob, fields, wt, sparams, aux)
            full = [1] * (3 - spec.ndim) + spatial
            crop = (slice(None), slice(None)) + tuple(slice(prob.lo[a], full[a] - prob.hi[a]) for a in range(3 - spec.ndim, 3))
            for eq, c, r in zip(self.EQS, coef, resid):
                losses[eq] = c * r[crop] ** 2
            total = sum(losses.values())
        if self.TOTAL_KEY:
            losses["total"] = total
        return total, losses

    @torch.no_grad()
    def residuals(self, *args, **kwargs) -> Dict[str, Tensor]:
        """The raw residual field of every equation for the inputs forward() takes: {equation: [B, T_res, *spatial]}
        (zero outside the residual lattice), plus '__sites_per_frame__' = lattice sites per residual frame.  fp32, or fp64
        when the fields are fp64: then the residuals are evaluated in double precision, as the reference's offline scale
        tools do (tools/compute_gt_scales_per_t.py:66-175)."""
        self._capture = {}
        try:
            self.forward(*args, **kwargs)
            return self._capture
        finally:
            self._capture = None

    def capture(self, *fields: Tensor, warmup: int = 2, pool=None, **kwargs) -> "CapturedLossStep":
        """Record `total, losses = self(*fields, **kwargs); total.backward()` as one HIP graph and return the object that
        replays it: the small configurations are bound by the host's launch work (a few launches and the autograd
        bookkeeping per step), a replay costs one launch.  See CapturedLossStep for what stays fixed between replays."""
        return CapturedLossStep(self, fields, kwargs, warmup=warmup, pool=pool)

    def _buf(self, name, make, device):
        key = (name, str(device))
        if key not in self._wt_cache:
            self._wt_cache[key] = make().to(device)
        return self._wt_cache[key]

    def _zero(self, device) -> Tensor:
        key = ("zero", str(device))
        if key not in self._wt_cache:
            self._wt_cache[key] = torch.zeros((), dtype=torch.float32, device=device)
        return self._wt_cache[key]


class CapturedLossStep:
    """`step = loss_fn.capture(u, v, p, nu=nu)`; then `total, losses = step()` replays forward + backward.

    The graph reads the tensors it was captured with, at their addresses: write new inputs INTO them
    (`u.copy_(new_u)`, or let the producer write there), do not rebind them.  Fields that require gradients are leaves of
    the captured step; after a replay their gradients of `upstream_hint * total` stand in `step.grads` (the same tensors
    every time, also reachable as `field.grad`), and `total` / `losses` are the same tensors every time with new values.
    A model whose output feeds the loss continues with `prediction.backward(gradient=step.grads[i])`.  The module's
    settings at capture time (backward_mode, scales, weights, scalars_out) are part of the graph; capture again after
    changing them.  Replays on one GPU must not overlap (they share the gradient and workspace buffers)."""

    def __init__(self, loss_fn: "StencilPDELoss", fields: Sequence[Tensor], kwargs: dict, warmup: int = 2, pool=None):
        if not fields or not all(torch.is_tensor(f) and f.is_cuda for f in fields):
            raise RuntimeError("capture(): the fields must be tensors on a HIP device (a graph replays GPU work only)")
        self.loss_fn, self.fields, self.kwargs = loss_fn, list(fields), dict(kwargs)
        dev = fields[0].device
        self._upstream = loss_fn.upstream_scalar(dev)
        needs = [f for f in self.fields if f.requires_grad]
        if any(not f.is_leaf for f in needs):
            raise RuntimeError("capture(): fields that require gradients must be leaf tensors (detach the model's output into a "
                               "buffer, capture on that buffer, and pass step.grads on to prediction.backward)")

        def run():
            for f in needs:
                f.grad = None
            total, losses = loss_fn(*self.fields, **self.kwargs)
            if needs:
                total.backward(gradient=self._upstream)
            return total, losses

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):      # whatever a first call loads or allocates happens outside the graph
                run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
            self.total, self.losses = run()
        self.grads = [f.grad for f in self.fields]
        self.total, self.losses = self.total.detach(), {k: v.detach() for k, v in self.losses.items()}

    def __call__(self):
        self.graph.replay()
        return self.total, self.losses

    replay = __call__


# =================================================================================================
# 1-D periodic
# =================================================================================================
class Burgers1DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:2924-3027"""
    SPEC, EQS, PER_T, TOTAL_KEY = "burgers1d", ("momentum",), "slice", True
    DEFAULT_EQ_SCALES = {"momentum": 1.0}

    def __init__(self, dx: float = 2 * np.pi / 256, dt: float = 0.01, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.dx, self.dt = dx, dt
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt)

    def forward(self, u: Tensor, nu: Tensor):
        return self._evaluate([u], [self._sample_param(nu, None, u.shape[0], u)])


class Advection1DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3034-3127"""
    SPEC, EQS, PER_T, TOTAL_KEY = "advection1d", ("advection",), "slice", True
    DEFAULT_EQ_SCALES = {"advection": 1.0}

    def __init__(self, dx: float = 2 * np.pi / 256, dt: float = 0.01, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.dx, self.dt = dx, dt
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt)

    def forward(self, u: Tensor, a: Tensor):
        return self._evaluate([u], [self._sample_param(a, None, u.shape[0], u)])


class Heat1DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3134-3229"""
    SPEC, EQS, PER_T, TOTAL_KEY = "heat1d", ("heat",), "slice", True
    DEFAULT_EQ_SCALES = {"heat": 1.0}

    def __init__(self, dx: float = 2 * np.pi / 256, dt: float = 0.01, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.dx, self.dt = dx, dt
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt)

    def forward(self, u: Tensor, alpha: Tensor):
        return self._evaluate([u], [self._sample_param(alpha, None, u.shape[0], u)])


class APEBenchBurgers1DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3348-3419"""
    SPEC, EQS = "ape_burgers1d", ("momentum",)
    DEFAULT_EQ_SCALES = {"momentum": 1.0}

    def __init__(self, nx: int = 160, Lx: float = 1.0, dt: float = 1.0, alpha_2: float = 2.9297e-5,
                 delta_conv: float = -4.6875e-3, eq_scales=None, eq_weights=None):
        super().__init__()
        self.dx, self.dt, self.alpha_2, self.delta_conv = Lx / nx, dt, alpha_2, delta_conv
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt, alpha_2=self.alpha_2, delta_conv=self.delta_conv)

    def forward(self, u: Tensor):
        return self._evaluate([u])


class KS1DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3535-3649"""
    SPEC, EQS = "ks1d", ("momentum",)
    DEFAULT_EQ_SCALES = {"momentum": 1.0}

    def __init__(self, nx: int = 160, Lx: float = 1.0, dt: float = 1.0, alpha_2: float = -2.34375e-5,
                 alpha_4: float = -2.8610229492e-9, beta_2: float = -2.34375e-4, eq_scales=None, eq_weights=None):
        super().__init__()
        self.dx, self.dt = Lx / nx, dt
        self.alpha_2, self.alpha_4, self.beta_2 = alpha_2, alpha_4, beta_2
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt, alpha_2=self.alpha_2, alpha_4=self.alpha_4, beta_2=self.beta_2)

    def forward(self, u: Tensor):
        return self._evaluate([u])


class KdV1DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3656-3775"""
    SPEC, EQS, PER_T, TOTAL_KEY = "kdv1d", ("kdv",), "slice", True
    DEFAULT_EQ_SCALES = {"kdv": 1.0}

    def __init__(self, nx: int = 160, Lx: float = 1.0, dt: float = 1.0, alpha_3: float = -8.5449e-7,
                 alpha_4: float = -1.7166e-9, beta_1: float = -1.2500e-2, eq_scales=None, eq_weights=None,
                 eq_scales_per_t=None):
        super().__init__()
        self.dx, self.dt = Lx / nx, dt
        self.alpha_3, self.alpha_4, self.beta_1 = alpha_3, alpha_4, beta_1
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt, alpha_3=self.alpha_3, alpha_4=self.alpha_4, beta_1=self.beta_1)

    def forward(self, u: Tensor):
        return self._evaluate([u])


# =================================================================================================
# 2-D periodic, central differences
# =================================================================================================
class TaylorGreen2DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:1717-1923 (dx = Lx / nx runs along dim -1, dy = Ly / ny along dim -2)"""
    SPEC, EQS = "tg2d", ("continuity", "x_momentum", "y_momentum")
    GUARDED, PER_T, PER_T_FLAG = ("x_momentum", "y_momentum"), "helper", "has_per_t_scales"
    DEFAULT_EQ_SCALES = {"continuity": 1.0, "x_momentum": 1.0, "y_momentum": 1.0}

    def __init__(self, nx: int = 256, ny: int = 256, Lx: float = 6.283185307179586, Ly: float = 6.283185307179586,
                 dt: float = 0.01, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt = Lx / nx, Ly / ny, dt
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt)

    def forward(self, u: Tensor, v: Tensor, p: Tensor, nu: Optional[Tensor] = None, t_offset: Optional[Tensor] = None):
        return self._evaluate([u, v, p], [self._sample_param(nu, 0.01, u.shape[0], u)], t_offset=t_offset)   # :1853-1856


class TaylorGreenPDELoss(TaylorGreen2DPDELoss):
    """The 2-D class finetune/train_taylor_green_lora_v3.py:66 and tools/visualize_taylor_green_lora.py:34 import; the
    reference snapshot no longer defines it.  Constructor and call as at those sites (:341-348 `nx, ny, Lx, Ly, dt, nu`,
    :190 `forward(u, v, p, nu=nu)`); the arithmetic is that of its in-tree successor TaylorGreen2DPDELoss, with the
    constructor's `nu` as the fallback when forward gets none.  No reference class pins it."""

    def __init__(self, nx: int = 256, ny: int = 256, Lx: float = 6.283185307179586, Ly: float = 6.283185307179586,
                 dt: float = 0.001, nu: float = 0.001, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__(nx, ny, Lx, Ly, dt, eq_scales, eq_weights, eq_scales_per_t)
        self.nu = nu

    def forward(self, u: Tensor, v: Tensor, p: Tensor, nu: Optional[Tensor] = None, t_offset: Optional[Tensor] = None):
        return self._evaluate([u, v, p], [self._sample_param(nu, self.nu, u.shape[0], u)], t_offset=t_offset)


class NSIncomPDELoss(StencilPDELoss):
    """Forced incompressible Navier-Stokes with a tracer between no-slip walls (PDEBench ns_incom).

    The reference snapshot imports this class (finetune/train_ns_incom_lora_v3.py:68; constructed at :417-428, called at
    :223 as `forward(u, v, p, d, fx, fy)` with the time-invariant force fx, fy [B, H, W]) but does not define it.  The
    arithmetic follows the n-PINN form ("Method B") of tools/test_gt_pde_ns_incom.py:258-378, which the trainer's header
    (:4-8) names: face-velocity divergence; conservative 2nd-order upwind convection, c2 Laplacian, force d * f and the
    div correction for u, v and the tracer; plus the face-averaged pressure gradient of that file's specification
    (:109-111, :122-123), which its code leaves out because the data set stores no pressure.  x runs along dim -2, y along
    dim -1; only sites `skip_bl` (>= 2) cells away from every wall enter.  Default scales: the values quoted at
    configs/finetune_ns_incom_v3.yaml:119-126.  Pinned by that file's residual arrays, not by a reference class."""
    SPEC, EQS = "ns_incom", ("continuity", "u_momentum", "v_momentum", "tracer")
    GUARDED = ("u_momentum", "v_momentum", "tracer")
    DEFAULT_EQ_SCALES = {"continuity": 7e-4, "u_momentum": 1.34, "v_momentum": 1.31, "tracer": 0.24}

    def __init__(self, nx: int = 512, ny: int = 512, Lx: float = 1.0, Ly: float = 1.0, dt: float = 0.005, nu: float = 0.01,
                 skip_bl: int = 2, use_div_correction: bool = True, eq_scales=None, eq_weights=None):
        super().__init__()
        if int(skip_bl) < 2:
            raise ValueError("NSIncomPDELoss: skip_bl must be at least 2 (the upwind fluxes reach two cells)")
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.nu, self.skip_bl = Lx / nx, Ly / ny, dt, nu, int(skip_bl)
        self.use_div_correction = use_div_correction
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, nu=self.nu, div_corr=1.0 if self.use_div_correction else 0.0)

    def _lattice(self, spatial):
        return (0, self.skip_bl, self.skip_bl), (0, self.skip_bl, self.skip_bl)

    def forward(self, u: Tensor, v: Tensor, p: Tensor, d: Tensor, fx: Tensor, fy: Tensor):
        return self._evaluate([u, v, p, d], aux=[fx, fy])


class AdvDiff2DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:2369-2495"""
    SPEC, EQS, PER_T, TOTAL_KEY = "advdiff2d", ("advdiff",), "slice", True
    DEFAULT_EQ_SCALES = {"advdiff": 1.0}

    def __init__(self, nx: int = 256, ny: int = 256, dx: float = 2 * np.pi / 256, dy: float = 2 * np.pi / 256,
                 dt: float = 0.01, a: float = 1.0, b: float = 0.5, nu: float = 0.05, eq_scales=None, eq_weights=None,
                 eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny, self.dx, self.dy, self.dt, self.a, self.b, self.nu = nx, ny, dx, dy, dt, a, b, nu
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, a=self.a, b=self.b, nu=self.nu)

    def forward(self, u: Tensor):
        return self._evaluate([u])


class AnisoDiff2DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:2502-2643"""
    SPEC, EQS, PER_T, TOTAL_KEY = "anisodiff2d", ("aniso_diff",), "slice", True
    DEFAULT_EQ_SCALES = {"aniso_diff": 1.0}

    def __init__(self, nx: int = 160, ny: int = 160, dx: float = 1.0 / 160, dy: float = 1.0 / 160, dt: float = 0.1,
                 A_xx: float = 0.001, A_xy: float = 0.0005, A_yx: float = 0.0005, A_yy: float = 0.002,
                 eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny, self.dx, self.dy, self.dt = nx, ny, dx, dy, dt
        self.A_xx, self.A_xy, self.A_yx, self.A_yy = A_xx, A_xy, A_yx, A_yy
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, A_xx=self.A_xx, A_xy=self.A_xy, A_yx=self.A_yx, A_yy=self.A_yy)

    def forward(self, u: Tensor):
        return self._evaluate([u])


class Wave2DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:2650-2789"""
    SPEC, EQS, PER_T, TOTAL_KEY = "wave2d", ("u_equation", "w_equation"), "slice", True
    DEFAULT_EQ_SCALES = {"u_equation": 1.0, "w_equation": 1.0}

    def __init__(self, dx: float = 2 * np.pi / 256, dy: float = 2 * np.pi / 256, dt: float = 0.01, eq_scales=None,
                 eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.dx, self.dy, self.dt = dx, dy, dt
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt)

    def forward(self, u: Tensor, w: Tensor, c: Tensor):
        c = self._sample_param(c, None, u.shape[0], u)
        return self._evaluate([u, w], [c * c])          # c^2 in fp32, as :2747


class APEBenchBurgers2DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3241-3341"""
    SPEC, EQS = "ape_burgers2d", ("u_momentum", "v_momentum")
    DEFAULT_EQ_SCALES = {"u_momentum": 1.0, "v_momentum": 1.0}

    def __init__(self, nx: int = 160, ny: int = 160, Lx: float = 1.0, Ly: float = 1.0, dt: float = 1.0,
                 alpha_2: float = 1.4648e-5, delta_conv: float = -2.3438e-3, eq_scales=None, eq_weights=None):
        super().__init__()
        self.dx, self.dy, self.dt, self.alpha_2, self.delta_conv = Lx / nx, Ly / ny, dt, alpha_2, delta_conv
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, alpha_2=self.alpha_2, delta_conv=self.delta_conv)

    def forward(self, u: Tensor, v: Tensor):
        return self._evaluate([u, v])


class NSPwCPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:1576-1710"""
    SPEC, EQS, GUARDED = "nspwc", ("divergence", "vorticity", "tracer"), ("vorticity", "tracer")
    DEFAULT_EQ_SCALES = {"divergence": 1.0, "vorticity": 1.0, "tracer": 1.0}

    def __init__(self, nx: int = 128, ny: int = 128, Lx: float = 1.0, Ly: float = 1.0, dt: float = 0.05,
                 nu: float = 4e-4, kappa: float = 4e-4, eq_scales=None, eq_weights=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.nu, self.kappa = Lx / nx, Ly / ny, dt, nu, kappa
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, nu=self.nu, kappa=self.kappa)

    def forward(self, u: Tensor, v: Tensor, s: Tensor):
        return self._evaluate([u, v, s])


class KP2DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:1930-2072"""
    SPEC, EQS, PER_T = "kp2d", ("kp2_equation",), "slice"
    DEFAULT_EQ_SCALES = {"kp2_equation": 1.0}

    def __init__(self, nx: int = 256, ny: int = 256, dx: float = 20.0 / 256, dy: float = 20.0 / 256, dt: float = 0.01,
                 skip_boundary: int = 20, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny, self.dx, self.dy, self.dt, self.skip_boundary = nx, ny, dx, dy, dt, skip_boundary
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt)

    def _lattice(self, spatial):
        sb = max(int(self.skip_boundary), 0)
        return (0, sb, sb), (0, sb, sb)

    def forward(self, u: Tensor):
        return self._evaluate([u])


# =================================================================================================
# 2-D periodic, conservative upwind (n-PINN)
# =================================================================================================
class ShearFlowPDELossNPINN(StencilPDELoss):
    """finetune/pde_loss_verified.py:1146-1470"""
    SPEC, EQS = "shear_flow", ("continuity", "x_momentum", "y_momentum", "tracer")
    GUARDED, PER_T, PER_T_FLAG = ("x_momentum", "y_momentum", "tracer"), "helper", "has_per_t_scales"
    DEFAULT_EQ_SCALES = {"continuity": 1.0, "x_momentum": 1.0, "y_momentum": 1.0, "tracer": 1.0}

    def __init__(self, nx: int = 256, ny: int = 512, Lx: float = 1.0, Ly: float = 2.0, dt: float = 0.1,
                 nu: float = 1e-4, D: float = 1e-3, use_div_correction: bool = True, eq_scales=None, eq_weights=None,
                 eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.nu, self.D = Lx / nx, Ly / ny, dt, nu, D
        self.use_div_correction = use_div_correction
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, nu=self.nu, D=self.D, div_corr=1.0 if self.use_div_correction else 0.0)

    def forward(self, u: Tensor, v: Tensor, p: Tensor, s: Tensor, reduction: str = "mean", t_offset: Optional[Tensor] = None):
        return self._evaluate([u, v, p, s], t_offset=t_offset)


class ShearFlowPDELoss(StencilPDELoss):
    """Central-difference residuals of the periodic shear-flow system (continuity, u_momentum, v_momentum, tracer).

    The reference's tools import this class from finetune/pde_loss_verified.py and construct it as
    ShearFlowPDELoss(nx, ny, Lx, Ly, dt, nu, D) (tools/test_gt_npinn_vs_central.py:82-85, tools/test_gt_fp32_pde_bc.py:52-55,
    tools/test_gt_pde_all_samples.py:193), call it as pde_fn(u, v, p, s, reduction='mean' | 'none') and read the four keys
    above (test_gt_fp32_pde_bc.py:117-121, test_gt_pde_all_samples.py:104-123), but the snapshot holds neither its
    definition nor a verification of its arithmetic.  Constructor, call and keys follow those call sites; the arithmetic is
    the advective form with the file's 4th-order periodic central stencils (x along dim -2, y along dim -1, like the n-PINN
    class of the same system).  Parity with the original class is therefore unpinned; the kernels are checked against the
    CPU restatement of this same arithmetic."""
    SPEC, EQS = "shear_flow_central", ("continuity", "u_momentum", "v_momentum", "tracer")
    GUARDED, PER_T, PER_T_FLAG = ("u_momentum", "v_momentum", "tracer"), "helper", "has_per_t_scales"
    TOTAL_KEY = True      # tools/test_gt_pde_all_samples.py:104 reads losses['total'] beside the four equations
    DEFAULT_EQ_SCALES = {"continuity": 1.0, "u_momentum": 1.0, "v_momentum": 1.0, "tracer": 1.0}

    def __init__(self, nx: int = 256, ny: int = 512, Lx: float = 1.0, Ly: float = 2.0, dt: float = 0.1,
                 nu: float = 1e-4, D: float = 1e-3, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.nu, self.D = Lx / nx, Ly / ny, dt, nu, D
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, nu=self.nu, D=self.D)

    def forward(self, u: Tensor, v: Tensor, p: Tensor, s: Tensor, reduction: str = "mean", t_offset: Optional[Tensor] = None):
        return self._evaluate([u, v, p, s], t_offset=t_offset, reduction=reduction)


class Burgers2DCHPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:2079-2362"""
    SPEC, EQS = "burgers2d_ch", ("u_momentum", "v_momentum", "irrotational")
    GUARDED, PER_T, PER_T_FLAG = ("u_momentum", "v_momentum", "irrotational"), "helper", "has_per_t_scales"
    DEFAULT_EQ_SCALES = {"u_momentum": 1.0, "v_momentum": 1.0, "irrotational": 1.0}

    def __init__(self, nx: int = 256, ny: int = 256, Lx: float = 6.283185307179586, Ly: float = 6.283185307179586,
                 dt: float = 0.01, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt = Lx / nx, Ly / ny, dt
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt)

    def _fixed_scale(self, eq):
        s = float(self.eq_scales[eq])
        if eq == "irrotational" and s < 1e-10:      # :2350-2356
            return 1.0
        return s

    def forward(self, u: Tensor, v: Tensor, nu: Tensor, reduction: str = "mean", t_offset: Optional[Tensor] = None):
        return self._evaluate([u, v], [self._sample_param(nu, None, u.shape[0], u)], t_offset=t_offset)


class ActiveMatterNPINNPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:525-742 (the Dxx equation never uses per-timestep scales, :736-737)"""
    SPEC, EQS = "active_matter", ("continuity", "concentration", "Dxx")
    GUARDED, PER_T, PER_T_EQS = ("concentration", "Dxx"), "slice", ("continuity", "concentration")
    DEFAULT_EQ_SCALES = {"continuity": 1.0, "concentration": 1.0, "Dxx": 1.0}

    def __init__(self, nx: int = 256, ny: int = 256, Lx: float = 10.0, Ly: float = 10.0, dt: float = 0.25,
                 d_T: float = 0.05, use_div_correction: bool = True, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.d_T = Lx / nx, Ly / ny, dt, d_T
        self.use_div_correction = use_div_correction
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, d_T=self.d_T, div_corr=1.0 if self.use_div_correction else 0.0)

    def forward(self, u: Tensor, v: Tensor, c: Tensor, Dxx: Tensor):
        return self._evaluate([u, v, c, Dxx])


# =================================================================================================
# 2-D mixed periodic / wall
# =================================================================================================
class RayleighBenardFullPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:749-1001 (fields [B, T, Nx, Ny]: dim -2 periodic, dim -1 between walls)"""
    SPEC, EQS = "rayleigh_benard", ("continuity", "buoyancy")
    GUARDED, PER_T, PER_T_FLAG = ("buoyancy",), "helper", "has_per_t_scales"
    DEFAULT_EQ_SCALES = {"continuity": 1.0, "buoyancy": 1.0}

    def __init__(self, nx: int = 512, ny: int = 128, Lx: float = 4.0, Ly: float = 1.0, dt: float = 0.25,
                 kappa: float = 1e-5, skip_bl: int = 15, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.kappa, self.skip_bl = Lx / nx, Ly / (ny - 1), dt, kappa, skip_bl
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt)

    def _lattice(self, spatial):
        sk = max(int(self.skip_bl), 0)
        return (0, 0, sk), (0, 0, sk)

    def forward(self, u: Tensor, v: Tensor, b: Tensor, kappa: Optional[Tensor] = None, t_offset: Optional[Tensor] = None):
        return self._evaluate([u, v, b], [self._sample_param(kappa, self.kappa, u.shape[0], u)], t_offset=t_offset)


class TurbulentRadiativePDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:1008-1139"""
    SPEC, EQS = "turbulent_radiative", ("mass", "x_momentum", "y_momentum")
    DEFAULT_EQ_SCALES = {"mass": 1.0, "x_momentum": 1.0, "y_momentum": 1.0}

    def __init__(self, nx: int = 128, ny: int = 384, Lx: float = 1.0, Ly: float = 3.0, dt: float = 1.597033,
                 skip_bl: int = 10, eq_scales=None, eq_weights=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.skip_bl = Lx / nx, Ly / (ny - 1), dt, skip_bl
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt)

    def _lattice(self, spatial):
        sk = 1 + max(int(self.skip_bl), 0)
        return (0, 0, sk), (0, 0, sk)

    def forward(self, vx: Tensor, vy: Tensor, rho: Tensor, P: Tensor):
        return self._evaluate([vx, vy, rho, P])


class WaveGaussPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:1477-1569 (wave speed c is data: it receives no gradient)"""
    SPEC, EQS, TOTAL_KEY = "wave_gauss", ("wave",), True
    DEFAULT_EQ_SCALES = {"wave": 1.0}

    def __init__(self, nx: int = 128, ny: int = 128, Lx: float = 1.0, Ly: float = 1.0, dt: float = 1.0 / 14,
                 skip_boundary: int = 2, eq_scales=None, eq_weights=None):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.dx, self.dy, self.dt, self.skip_boundary = Lx / nx, Ly / ny, dt, skip_boundary
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt)

    def _lattice(self, spatial):
        sb = 1 + max(int(self.skip_boundary), 0)
        return (0, sb, sb), (0, sb, sb)

    def forward(self, u: Tensor, c: Tensor):
        if c.dim() == 4:
            c = c[:, 0]
        return self._evaluate([u], aux=[c])


# =================================================================================================
# Gray-Scott
# =================================================================================================
class GrayScottPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:148-302.

    laplacian='fft' (default) reproduces the reference: spectral Laplacian (torch.fft -> rocFFT) feeding a
    fused pointwise residual + reduction kernel.  laplacian='fd4' is the 4th-order periodic stencil of
    tools/test_gt_pde_gray_scott.py:27-79 fused into one kernel (no FFT); it is a different discretisation
    and agrees with the reference only on band-limited fields.

    The reference reads its fixed scales and weights under the keys 'A' / 'B' (:161-164, :271, :283-288)
    while the per-timestep buffers are named '*_equation' (:266-280); both conventions are kept.
    """
    EQS, PER_T, TOTAL_KEY = ("A_equation", "B_equation"), "slice", True
    SCALE_KEYS = {"A_equation": "A", "B_equation": "B"}
    DEFAULT_EQ_SCALES = {"A": 1.0, "B": 1.0}

    def __init__(self, nx: int = 128, ny: int = 128, dx: float = 2.0 / 128, dy: float = 2.0 / 128, dt: float = 10.0,
                 F: float = 0.098, k: float = 0.057, D_A: float = 1.81e-5, D_B: float = 1.39e-5, eq_scales=None,
                 eq_weights=None, eq_scales_per_t=None, laplacian: str = "fft"):
        super().__init__()
        if laplacian not in ("fft", "fd4"):
            raise ValueError("laplacian must be 'fft' or 'fd4'")
        self.laplacian = laplacian
        self.SPEC = "gray_scott_pw" if laplacian == "fft" else "gray_scott_fd"
        self.nx, self.ny, self.dx, self.dy, self.dt = nx, ny, dx, dy, dt
        self.F, self.k, self.D_A, self.D_B = F, k, D_A, D_B
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)
        kx = torch.fft.fftfreq(nx, d=dx) * 2 * np.pi
        ky = torch.fft.fftfreq(ny, d=dy) * 2 * np.pi
        KX, KY = torch.meshgrid(kx, ky, indexing="ij")
        self.register_buffer("K2", KX ** 2 + KY ** 2)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dt=self.dt, F=self.F, k=self.k, D_A=self.D_A, D_B=self.D_B)

    def fft_laplacian(self, f: Tensor) -> Tensor:
        """Spectral Laplacian over the last two axes (:197-214).  The fields are real, so the half-spectrum transforms
        give the same result as the reference's fft2 / ifft2(...).real with half the work and no strided real-part view."""
        nk = self.K2.shape[1] // 2 + 1
        if f.dtype is torch.float64:      # the fp64 residual evaluation: wavenumbers in double as well
            def k2():
                kx = torch.fft.fftfreq(self.nx, d=self.dx, dtype=torch.float64) * 2 * np.pi
                ky = torch.fft.fftfreq(self.ny, d=self.dy, dtype=torch.float64) * 2 * np.pi
                return -(kx[:, None] ** 2 + ky[None, :nk] ** 2)
            negK2 = self._buf("negK2h64", k2, f.device)
        else:
            negK2 = self._buf("negK2h", lambda: -self.K2[:, :nk].contiguous(), f.device)
        return torch.fft.irfft2(negK2 * torch.fft.rfft2(f), s=f.shape[-2:])

    def forward(self, A: Tensor, B: Tensor, reduction: str = "mean"):
        if A.dim() == 3:
            A, B = A.unsqueeze(0), B.unsqueeze(0)
        if self.laplacian == "fd4":
            return self._evaluate([A, B], reduction=reduction)
        dt = torch.float64 if A.dtype is torch.float64 and getattr(self, "_capture", None) is not None else torch.float32
        mult = self._pair_multiplier(A) if dt is torch.float32 and A.is_cuda and A.shape == B.shape else None
        if mult is not None:      # both fields of a sample and frame as one complex plane, transformed inside one workgroup's LDS
            lapA, lapB = spectral_laplacian_pair(A.to(dt), B.to(dt), mult)
            return self._evaluate([A, B, lapA, lapB], reduction=reduction)
        lap = self.fft_laplacian(torch.stack([A.to(dt), B.to(dt)]))      # rocFFT: both fields in one batched transform
        return self._evaluate([A, B, lap[0], lap[1]], reduction=reduction)

    def _pair_multiplier(self, f: Tensor) -> Optional[Tensor]:
        """-K2 on the fields' device when the LDS-resident transform kernel applies: plane sizes it is built for, and a
        table that is even in both wavenumbers (the reference's always is), which the kernel's way of transforming two
        real fields as one complex field and its backward pass rely on.  None: the rocFFT path."""
        key = ("pair_mult", str(f.device), self.K2._version, self.K2.data_ptr())
        hit = self._wt_cache.get(key)
        if hit is None:
            k2 = self.K2.detach().float().cpu()
            nx, ny = k2.shape
            ok = tuple(f.shape[-2:]) == (nx, ny) and spectral_pair_supported(nx, ny)
            if ok:
                mirror = torch.roll(torch.flip(k2, (0, 1)), (1, 1), (0, 1))      # K2[-i mod nx, -j mod ny]
                ok = bool(torch.equal(k2, mirror))
            hit = self._wt_cache[key] = ((-k2).contiguous().to(f.device),) if ok else (None,)
        m = hit[0]
        return m if m is not None and tuple(f.shape[-2:]) == tuple(m.shape) else None


def _half_wavenumbers(n_full: int, d_full: float, n_half: int, d_half: float):
    """(KX, KY) of a real 2-D transform over (dim -2, dim -1): all of axis -2, the non-negative half of axis -1, times i
    left to the caller.  The Nyquist rows are zeroed: for a first derivative the reference's `ifft2(1j * K * fft2(f)).real`
    drops exactly that part (it is not conjugate-symmetric), and a half-spectrum inverse would keep half of it."""
    kx = torch.fft.fftfreq(n_full, d=d_full) * 2 * np.pi
    ky = torch.fft.rfftfreq(n_half, d=d_half) * 2 * np.pi
    if n_full % 2 == 0:
        kx[n_full // 2] = 0.0
    if n_half % 2 == 0:
        ky[-1] = 0.0
    return torch.meshgrid(kx, ky, indexing="ij")


class ActiveMatterPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:28-146 — the spectral (doubly periodic) active-matter class: div(u) = 0 and
    dc/dt + u . grad c = 0 with a forward time difference; forward(c, u, v, reduction).  The four spectral derivatives
    are rocFFT transforms (torch.fft, half spectrum); the residuals, their reduction and the adjoint of the pointwise part
    run in the stencil kernels, autograd carries the gradient back through the transforms."""
    SPEC, EQS, TOTAL_KEY = "active_matter_fft", ("continuity", "advection"), True

    def __init__(self, nx: int = 256, ny: int = 256, dx: float = 10.0 / 256, dy: float = 10.0 / 256, dt: float = 1.0):
        super().__init__()
        self.nx, self.ny, self.dx, self.dy, self.dt = nx, ny, dx, dy, dt
        self._init_common({"continuity": 1.0, "advection": 1.0}, None, None, None, per_t_supported=False)
        KX, KY = torch.meshgrid(torch.fft.fftfreq(nx, d=dx) * 2 * np.pi, torch.fft.fftfreq(ny, d=dy) * 2 * np.pi, indexing="ij")
        self.register_buffer("KX", KX)      # the reference's buffers (:54-61); the transforms below use their half-spectrum forms
        self.register_buffer("KY", KY)

    def _params(self):
        return dict(dt=self.dt)

    def _derivatives(self, f: Tensor):
        """(df/dx, df/dy) over the last two axes, one forward transform (:63-77)"""
        kxh, kyh = self._buf("Kh", lambda: torch.stack(_half_wavenumbers(self.nx, self.dx, self.ny, self.dy)), f.device)
        fh = torch.fft.rfft2(f)
        return torch.fft.irfft2(1j * kxh * fh, s=f.shape[-2:]), torch.fft.irfft2(1j * kyh * fh, s=f.shape[-2:])

    def forward(self, c: Tensor, u: Tensor, v: Tensor, reduction: str = "mean"):
        if c.dim() == 3:
            c, u, v = c.unsqueeze(0), u.unsqueeze(0), v.unsqueeze(0)
        c, u, v = c.to(torch.float32), u.to(torch.float32), v.to(torch.float32)
        cx, cy = self._derivatives(c)
        ux, _ = self._derivatives(u)
        _, vy = self._derivatives(v)
        return self._evaluate([c, u * cx + v * cy, ux + vy], reduction=reduction)


class ViscoelasticContinuityLoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:309-393 — continuity only: spectral du/dx along the periodic x (dim -2), central
    dv/dy on the interior of the wall-bounded y (dim -1), every frame; forward(u, v, reduction)."""
    SPEC, EQS, TOTAL_KEY = "viscoelastic_cont", ("continuity",), True

    def __init__(self, nx: int = 512, ny: int = 512, dx: float = 2 * np.pi / 512, dy: float = 2.0 / 512):
        super().__init__()
        self.nx, self.ny, self.dx, self.dy = nx, ny, dx, dy
        self._init_common({"continuity": 1.0}, None, None, None, per_t_supported=False)
        self.register_buffer("kx", (torch.fft.fftfreq(nx, d=dx) * 2 * np.pi).view(nx, 1))      # :341-342

    def _params(self):
        return dict(dy=self.dy)

    def _lattice(self, spatial):
        return (0, 0, 1), (0, 0, 1)

    def fft_dx(self, f: Tensor) -> Tensor:
        """spectral derivative along dim -2 (:344-349); the half-spectrum inverse drops the Nyquist term as `.real` does"""
        kh = self._buf("kxh", lambda: (torch.fft.rfftfreq(self.nx, d=self.dx) * 2 * np.pi).view(-1, 1), f.device)
        return torch.fft.irfft(1j * kh * torch.fft.rfft(f, dim=-2), n=f.shape[-2], dim=-2)

    def forward(self, u: Tensor, v: Tensor, reduction: str = "mean"):
        unbatched = u.dim() == 3      # [T, X, Y]: the reference never adds a batch axis (:352-389), so 'none' keeps that shape
        if unbatched:
            u, v = u.unsqueeze(0), v.unsqueeze(0)
        total, losses = self._evaluate([self.fft_dx(u.to(torch.float32)), v], reduction=reduction)
        if unbatched and reduction == "none":
            losses = {k: x.squeeze(0) for k, x in losses.items()}
            total = total.squeeze(0)
        return total, losses


# =================================================================================================
# 3-D periodic
# =================================================================================================
class Advection3DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:2796-2917"""
    SPEC, EQS, PER_T, TOTAL_KEY = "advection3d", ("advection",), "slice", True
    DEFAULT_EQ_SCALES = {"advection": 1.0}

    def __init__(self, dx: float = 2 * np.pi / 64, dy: float = 2 * np.pi / 64, dz: float = 2 * np.pi / 64,
                 dt: float = 0.05, eq_scales=None, eq_weights=None, eq_scales_per_t=None):
        super().__init__()
        self.dx, self.dy, self.dz, self.dt = dx, dy, dz, dt
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dz=self.dz, dt=self.dt)

    def forward(self, u: Tensor, a: Tensor, b: Tensor, c: Tensor):
        B = u.shape[0]
        return self._evaluate([u], [self._sample_param(x, None, B, u) for x in (a, b, c)])


class APEBenchBurgers3DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3426-3528"""
    SPEC, EQS = "ape_burgers3d", ("u_momentum", "v_momentum", "w_momentum")
    DEFAULT_EQ_SCALES = {"u_momentum": 1.0, "v_momentum": 1.0, "w_momentum": 1.0}

    def __init__(self, nx: int = 32, Lx: float = 1.0, dt: float = 1.0, alpha_2: float = 2.4414e-4,
                 beta_1: float = 1.5625e-2, eq_scales=None, eq_weights=None):
        super().__init__()
        self.dx, self.dt, self.alpha_2, self.beta_1 = Lx / nx, dt, alpha_2, beta_1
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt, alpha_2=self.alpha_2, beta_1=self.beta_1)

    def forward(self, u: Tensor, v: Tensor, w: Tensor):
        return self._evaluate([u, v, w])


class SwiftHohenberg3DPDELoss(StencilPDELoss):
    """finetune/pde_loss_verified.py:3782-3876"""
    SPEC, EQS, PER_T, TOTAL_KEY = "swift_hohenberg3d", ("swift_hohenberg",), "slice", True
    DEFAULT_EQ_SCALES = {"swift_hohenberg": 1.0}

    def __init__(self, domain_extent: float = 31.41592653589793, num_points: int = 32, dt: float = 0.1,
                 reactivity: float = 0.7, critical_number: float = 1.0, eq_scales=None, eq_weights=None,
                 eq_scales_per_t=None):
        super().__init__()
        self.dx, self.dt = domain_extent / num_points, dt
        self.r, self.k = reactivity, critical_number
        self.linear_coeff = reactivity - critical_number ** 2
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dt=self.dt, linear_coeff=self.linear_coeff, k=self.k)

    def forward(self, u: Tensor):
        return self._evaluate([u])


class Heat3DPDELoss(StencilPDELoss):
    """3-D heat equation u_t = alpha (u_xx + u_yy + u_zz), periodic, per-sample alpha.

    The reference snapshot imports this class (finetune/train_heat_3d_lora_v3.py:69, constructed at :322-331,
    called at :170) but does not define it.  The arithmetic follows the in-tree verifier
    `verify_3d_heat` (finetune/generate_analytical_3d.py:160-178): 7-point 2nd-order Laplacian, central time
    difference, mean of squares; dict keys follow Heat1DPDELoss (:3229).  No reference class pins it.
    """
    SPEC, EQS, PER_T, TOTAL_KEY = "heat3d", ("heat",), "slice", True
    DEFAULT_EQ_SCALES = {"heat": 1.0}

    def __init__(self, nx: int = 128, ny: int = 128, nz: int = 128, Lx: float = 2 * np.pi, Ly: float = 2 * np.pi,
                 Lz: float = 2 * np.pi, dt: float = 0.01, alpha: float = 0.01, eq_scales=None, eq_weights=None,
                 eq_scales_per_t=None):
        super().__init__()
        self.nx, self.ny, self.nz = nx, ny, nz
        self.dx, self.dy, self.dz, self.dt, self.alpha = Lx / nx, Ly / ny, Lz / nz, dt, alpha
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, eq_scales_per_t)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dz=self.dz, dt=self.dt)

    def forward(self, u: Tensor, alpha: Optional[Tensor] = None):
        return self._evaluate([u], [self._sample_param(alpha, self.alpha, u.shape[0], u)])


class TaylorGreen3DPDELoss(StencilPDELoss):
    """3-D Taylor-Green vortex: face-velocity divergence and the three c2 primitive-form momentum equations, periodic,
    per-sample nu; fields u, v, w, p [B, T, X, Y, Z].

    The reference has no such class; the arithmetic is that of its in-tree verifier `verify_3d_taylor_green`
    (finetune/generate_analytical_3d.py:97-157), which spells out the divergence and the u momentum (v and w are the same
    with the axes permuted); all four are evaluated on the interior frames.  Pinned by that function's outputs."""
    SPEC, EQS = "tg3d", ("continuity", "u_momentum", "v_momentum", "w_momentum")
    DEFAULT_EQ_SCALES = {"continuity": 1.0, "u_momentum": 1.0, "v_momentum": 1.0, "w_momentum": 1.0}

    def __init__(self, nx: int = 64, ny: int = 64, nz: int = 64, Lx: float = 2 * np.pi, Ly: float = 2 * np.pi,
                 Lz: float = 2 * np.pi, dt: float = 0.01, nu: float = 0.01, eq_scales=None, eq_weights=None):
        super().__init__()
        self.nx, self.ny, self.nz = nx, ny, nz
        self.dx, self.dy, self.dz, self.dt, self.nu = Lx / nx, Ly / ny, Lz / nz, dt, nu
        self._init_common(self.DEFAULT_EQ_SCALES, eq_scales, eq_weights, None, per_t_supported=False)

    def _params(self):
        return dict(dx=self.dx, dy=self.dy, dz=self.dz, dt=self.dt)

    def forward(self, u: Tensor, v: Tensor, w: Tensor, p: Tensor, nu: Optional[Tensor] = None):
        return self._evaluate([u, v, w, p], [self._sample_param(nu, self.nu, u.shape[0], u)])


# =================================================================================================
# ghost-cell Burgers (finetune/pde_loss.py)
# =================================================================================================
def burgers_pde_loss(pred: Tensor, nu: float, dt: float = 1 / 999, dx: float = 1 / 127, dy: float = 1 / 127):
    """finetune/pde_loss.py:87-216 — returns (total, loss_u, loss_v, residual_u, residual_v).

    pred is channel-last [B, T, H, W, 2]; residuals are [B, T-1, H-2, W-2], differentiable like the losses.
    """
    B, T, H, W, C2 = pred.shape
    assert T >= 2, "T must be at least 2"
    assert H >= 4 and W >= 4, "H, W must be at least 4"
    u, v = pred[..., 0], pred[..., 1]
    spec = SPECS["burgers2d_ghost"]
    p = dict(nu=float(nu), dt=dt, dx=dx, dy=dy)
    prob = StencilProblem("burgers2d_ghost", [fn(p) for _, fn in spec.consts], (0, 1, 1), (0, 1, 1))
    n = B * (T - 1) * (H - 2) * (W - 2)
    wt = torch.full((2, T - 1), 1.0 / n, dtype=torch.float32, device=pred.device)
    totals, _, resid = residual_losses_and_fields(prob, [u, v], wt)      # one forward launch; the residuals stay in the graph (:216)
    ru, rv = (r[:, :, 1:H - 1, 1:W - 1] for r in resid)
    return totals[2], totals[0], totals[1], ru, rv


burgers_pde_loss_upwind = burgers_pde_loss


# =================================================================================================
# flow mixing, central differences with ghost cells (eval/pde_loss_flow_v2.py)
# =================================================================================================
def compute_flow_coefficients(H: int, W: int, vtmax: float, device, dtype=torch.float32):
    """eval/pde_loss_flow_v2.py:18-54 — the advection coefficients a = -omega y, b = omega x on the interior grid [H, W]."""
    x = torch.linspace(1 / 256, 255 / 256, W, device=device, dtype=dtype)
    y = torch.linspace(1 / 256, 255 / 256, H, device=device, dtype=dtype)
    Y, X = torch.meshgrid(y, x, indexing="ij")
    r = torch.sqrt(X ** 2 + Y ** 2).clamp(min=1e-10)
    rc = r.clamp(max=20.0)
    omega = (1.0 / torch.cosh(rc)) ** 2 * torch.tanh(rc) / (r * vtmax + 1e-10)
    return -omega * Y, omega * X


def pad_with_boundaries_1x(interior: Tensor, boundary_left: Tensor, boundary_right: Tensor, boundary_bottom: Tensor,
                           boundary_top: Tensor) -> Tensor:
    """eval/pde_loss_flow_v2.py:57-111 — one layer of linearly extrapolated ghost cells (2 boundary - first interior
    value), corners the mean of their two neighbours' ghosts: [B, T, H, W] -> [B, T, H+2, W+2], differentiable."""
    gl = 2 * boundary_left.squeeze(-1) - interior[..., 0]
    gr = 2 * boundary_right.squeeze(-1) - interior[..., -1]
    gb = 2 * boundary_bottom.squeeze(-2) - interior[..., 0, :]
    gt = 2 * boundary_top.squeeze(-2) - interior[..., -1, :]
    mid = torch.cat([gl.unsqueeze(-1), interior, gr.unsqueeze(-1)], dim=-1)
    bottom = torch.cat([(gl[..., :1] + gb[..., :1]) / 2, gb, (gr[..., :1] + gb[..., -1:]) / 2], dim=-1)
    top = torch.cat([(gl[..., -1:] + gt[..., :1]) / 2, gt, (gr[..., -1:] + gt[..., -1:]) / 2], dim=-1)
    return torch.cat([bottom.unsqueeze(-2), mid, top.unsqueeze(-2)], dim=-2)


_flow_coef_cache = {}


def flow_mixing_pde_loss_v2(pred: Tensor, boundary_left: Tensor, boundary_right: Tensor, boundary_bottom: Tensor,
                            boundary_top: Tensor, vtmax: float, dt: float = 1 / 999, Lx: float = 1.0, Ly: float = 1.0):
    """eval/pde_loss_flow_v2.py:179-243 — returns (pde_loss, loss_time, loss_advection, residual [B, T-1, H, W]).

    pred [B, T, H, W, 1]; the ghost layer is assembled with tensor ops (a few strips), the three mean squares and the
    residual field come from one stencil launch on the padded field, and everything stays differentiable."""
    B, T, H, W, _ = pred.shape
    u = pred[..., 0].to(torch.float32)
    padded = pad_with_boundaries_1x(u, boundary_left[..., 0].to(torch.float32), boundary_right[..., 0].to(torch.float32),
                                    boundary_bottom[..., 0].to(torch.float32), boundary_top[..., 0].to(torch.float32))
    key = (H, W, float(vtmax), B, str(pred.device))
    coef = _flow_coef_cache.get(key)
    if coef is None:
        a, b = compute_flow_coefficients(H, W, vtmax, pred.device)
        coef = _flow_coef_cache[key] = tuple(torch.nn.functional.pad(x, (1, 1, 1, 1)).expand(B, H + 2, W + 2).contiguous() for x in (a, b))
    spec = SPECS["flow_mixing_v2"]
    p = dict(dt=dt, dx=Lx / W, dy=Ly / H)
    prob = StencilProblem("flow_mixing_v2", [fn(p) for _, fn in spec.consts], (0, 1, 1), (0, 1, 1))
    n = B * (T - 1) * H * W
    wt = torch.full((3, T - 1), 1.0 / n, dtype=torch.float32, device=pred.device)
    totals, _, resid = residual_losses_and_fields(prob, [padded], wt, (), coef)
    return totals[2], totals[0], totals[1], resid[2][:, :, 1:H + 1, 1:W + 1]
