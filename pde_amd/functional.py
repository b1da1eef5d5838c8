"""Differentiable residual-loss primitive over the C ABI.

`residual_losses(problem, fields, weights, ...)` returns `(totals, sums)`:
  totals  fp32 [n_eqs + 1]  per-equation losses and, last, their sum  (what the loss modules hand out)
  sums    fp64 [B, n_eqs]   per-sample weighted squared-residual sums
Both are differentiable w.r.t. the fields; the backward launches ONE adjoint-stencil kernel that reads
the upstream gradients on the device (no host synchronisation, no elementwise glue kernels).
Everything runs on the current torch HIP stream.  There is no CPU path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from dataclasses import dataclass
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
        key = tuple(shape)
        cache = self.__dict__.setdefault("_cstructs", {})
        if key not in cache:
            cache[key] = self._c_struct(shape)
        return cache[key]

    def op_args(self):
        """(pde id, constants, lo, hi, equations, t_lo, t_hi) as the C++ operator takes them"""
        a = self.__dict__.get("_op_args")
        if a is None:
            s = SPECS[self.spec]
            a = self.__dict__["_op_args"] = (PDE_IDS[self.spec], [float(v) for v in self.consts], [int(v) for v in self.lo],
                                             [int(v) for v in self.hi], len(s.eqs), s.tlo, s.thi)
        return a

    def _c_struct(self, shape: Sequence[int]) -> "_lib.Problem":
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
    # Optional instrumentation: when set to a list, every backward launch appends a (start, end) pair of events recorded
    # on the launch stream by the library itself, right around its adjoint kernel launch (pde_amd_time_next_adjoint).
    backward_events = None
    backward_event_pool = None       # optional list of pre-created (start, end) event pairs to draw from
    forward_events = None            # same for the forward launches (residual kernel + the two tiny reductions)
    forward_event_pool = None
    last_adjoint_kernel = ""         # with backward_events set: the adjoint kernel of the last timed call, as a profiler names it

    @staticmethod
    def _stream(t: torch.Tensor):
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    _ws = {}

    def _workspace(self, device, nbytes: int) -> torch.Tensor:
        """Scratch for partial sums, reused across calls: every use is ordered on the stream it was launched on, and a
        buffer is kept per (device, stream)."""
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _lib.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    # Diagnostic (PDE_AMD_POISON_LDS=1): NaNs into every CU's LDS before each residual launch, so that a kernel that reads
    # LDS it did not write fails the parity tests instead of living on finite leftovers.
    poison_lds = os.environ.get("PDE_AMD_POISON_LDS") == "1"

    def _poison(self, t: torch.Tensor):
        if self.poison_lds:
            with self._on(t.device):
                _lib.check(_lib.lib().pde_amd_diag_poison_lds(self._stream(t)), "diag_poison_lds")

    @staticmethod
    def _on(device):
        """Context that makes `device` current for the launch; free when it already is."""
        return contextlib.nullcontext() if torch.cuda.current_device() == device.index else torch.cuda.device(device)

    def _arm_adjoint_timer(self):
        """With backward_events set: a pair of events that the library records right around its next adjoint kernel launch."""
        if self.backward_events is None:
            return None
        ev = self.backward_event_pool.pop() if self.backward_event_pool else (
            torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for e in ev:
            if not e.cuda_event:      # the handle exists only after a first record
                e.record()
        _lib.lib().pde_amd_time_next_adjoint(C.c_void_p(ev[0].cuda_event), C.c_void_p(ev[1].cuda_event))
        return ev

    def _disarm_adjoint_timer(self, ev):
        if ev is not None:
            L = _lib.lib()
            L.pde_amd_time_next_adjoint(None, None)
            # asked here, on the thread that made the call (autograd runs backward on a thread of its own, and the library
            # keeps the name per thread)
            HipBackend.last_adjoint_kernel = L.pde_amd_last_adjoint_kernel().decode()
            self.backward_events.append(ev)

    def can_recompute(self, spec: str) -> bool:
        return bool(_lib.pde_info(spec).can_recompute)

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
        sums = _lib.empty((B, ne), dtype=torch.float64, device=f0.device)
        totals = new_totals(ne, f0.device)
        ws_bytes = L.pde_amd_residual_workspace_bytes(C.byref(cp))
        ws = self._workspace(f0.device, ws_bytes)
        resid = None
        rptr = None
        if store_residuals:
            resid = [_lib.empty((B, TR) + tuple(f0.shape[2:]), dtype=torch.float32, device=f0.device) for _ in range(ne)]
            rptr = _lib.ptr_array(resid, ne)
        ev = None
        if self.forward_events is not None:
            ev = self.forward_event_pool.pop() if self.forward_event_pool else (
                torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._poison(f0)
        with self._on(f0.device):
            st = L.pde_amd_residual_forward(C.byref(cp), _lib.ptr_array(fields, len(fields)), _lib.ptr_array(aux, len(aux)),
                                            _lib.ptr_array(sparams, len(sparams)), C.c_void_p(weights.data_ptr()),
                                            C.c_void_p(sums.data_ptr()), C.c_void_p(totals.data_ptr()), rptr,
                                            C.c_void_p(ws.data_ptr()), ws_bytes, self._stream(f0))
        if ev is not None:
            ev[1].record()
            self.forward_events.append(ev)
        _lib.check(st, f"residual_forward[{prob.spec}]")
        return totals, sums, resid

    def forward_f64(self, prob: StencilProblem, fields, aux, sparams, weights):
        """The residual forward in double precision on double tensors: (totals [n_eqs + 1], sums [B, n_eqs], residual fields)."""
        f0 = fields[0]
        if not f0.is_cuda:
            raise RuntimeError("pde_amd: fields must be on a HIP device (there is no CPU fallback)")
        L = _lib.lib()
        s = SPECS[prob.spec]
        cp = prob.c_struct(f0.shape)
        B, T = f0.shape[0], f0.shape[1]
        TR, ne = max(T - s.tlo - s.thi, 0), len(s.eqs)
        sums = _lib.empty((B, ne), dtype=torch.float64, device=f0.device)
        totals = _lib.empty((ne + 1,), dtype=torch.float64, device=f0.device)
        resid = [_lib.empty((B, TR) + tuple(f0.shape[2:]), dtype=torch.float64, device=f0.device) for _ in range(ne)]
        ws_bytes = L.pde_amd_residual_workspace_bytes(C.byref(cp))
        ws = self._workspace(f0.device, ws_bytes)
        consts = (C.c_double * max(len(prob.consts), 1))(*[float(v) for v in prob.consts])
        with self._on(f0.device):
            st = L.pde_amd_residual_forward_f64(C.byref(cp), consts, _lib.ptr_array(fields, len(fields)), _lib.ptr_array(aux, len(aux)),
                                                _lib.ptr_array(sparams, len(sparams)), C.c_void_p(weights.data_ptr()),
                                                C.c_void_p(sums.data_ptr()), C.c_void_p(totals.data_ptr()), _lib.ptr_array(resid, ne),
                                                C.c_void_p(ws.data_ptr()), ws_bytes, self._stream(f0))
        _lib.check(st, f"residual_forward_f64[{prob.spec}]")
        return totals, sums, resid

    def backward(self, prob: StencilProblem, fields, aux, sparams, weights, upstream, upstream_sums, resid):
        f0 = fields[0]
        L = _lib.lib()
        s = SPECS[prob.spec]
        cp = prob.c_struct(f0.shape)
        grads = [_lib.empty_like(f) for f in fields]
        rptr = _lib.ptr_array(resid, len(resid)) if resid is not None else None
        uptr = _lib.ptr_array(upstream, len(s.eqs) + 1) if upstream is not None else None
        gs = C.c_void_p(upstream_sums.data_ptr()) if upstream_sums is not None else None
        ev = self._arm_adjoint_timer()
        self._poison(f0)
        with self._on(f0.device):
            st = L.pde_amd_residual_backward(C.byref(cp), _lib.ptr_array(fields, len(fields)), _lib.ptr_array(aux, len(aux)),
                                             _lib.ptr_array(sparams, len(sparams)), C.c_void_p(weights.data_ptr()), uptr, gs,
                                             rptr, _lib.ptr_array(grads, len(grads)), None, 0, self._stream(f0))
        self._disarm_adjoint_timer(ev)
        _lib.check(st, f"residual_backward[{prob.spec}]")
        return grads


    def forward_backward(self, prob: StencilProblem, fields, aux, sparams, weights, assumed: torch.Tensor):
        """One pass: (totals, sums, gradients of assumed * totals[-1]).  `assumed` is a device scalar."""
        f0 = fields[0]
        if not f0.is_cuda:
            raise RuntimeError("pde_amd: fields must be on a HIP device (there is no CPU fallback)")
        L = _lib.lib()
        s = SPECS[prob.spec]
        cp = prob.c_struct(f0.shape)
        ne = len(s.eqs)
        sums = _lib.empty((f0.shape[0], ne), dtype=torch.float64, device=f0.device)
        totals = new_totals(ne, f0.device)
        grads = [_lib.empty_like(f) for f in fields]
        ws_bytes = L.pde_amd_residual_workspace_bytes(C.byref(cp))
        ws = self._workspace(f0.device, ws_bytes)
        ev = self._arm_adjoint_timer()      # the adjoint kernel does the whole step here
        self._poison(f0)
        with self._on(f0.device):
            st = L.pde_amd_residual_forward_backward(
                C.byref(cp), _lib.ptr_array(fields, len(fields)), _lib.ptr_array(aux, len(aux)),
                _lib.ptr_array(sparams, len(sparams)), C.c_void_p(weights.data_ptr()), C.c_void_p(assumed.data_ptr()),
                C.c_void_p(sums.data_ptr()), C.c_void_p(totals.data_ptr()), _lib.ptr_array(grads, len(grads)),
                C.c_void_p(ws.data_ptr()), ws_bytes, self._stream(f0))
        self._disarm_adjoint_timer(ev)
        _lib.check(st, f"residual_forward_backward[{prob.spec}]")
        return totals, sums, grads

    def rescale(self, grads, upstream: torch.Tensor, assumed: torch.Tensor):
        """grads *= upstream / assumed in place; a no-op decided on the device when the two scalars are equal."""
        g0 = grads[0]
        with self._on(g0.device):
            st = _lib.lib().pde_amd_rescale_gradients(_lib.ptr_array(grads, len(grads)), len(grads), g0.numel(),
                                                      C.c_void_p(upstream.data_ptr()), C.c_void_p(assumed.data_ptr()),
                                                      self._stream(g0))
        _lib.check(st, "rescale_gradients")


_BACKEND = HipBackend()
# Host path of residual_losses on the HIP backend: "op" = the C++ operator of libpde_amd_torch.so (default),
# "ctypes" = the Python autograd function below over the ctypes binding (also used while launches are being timed)
_HOST = os.environ.get("PDE_AMD_HOST", "auto")
if _HOST not in ("auto", "op", "ctypes"):
    raise ValueError(f"PDE_AMD_HOST must be 'auto', 'op' or 'ctypes', got {_HOST!r}")
_OP_OK = None


def _use_op() -> bool:
    """Whether the residual losses go through the C++ operator.  'op' insists on it (and raises where the operator
    library is missing or does not bind), 'ctypes' never uses it, 'auto' (default) uses it where it loads and otherwise
    says so once and drives the same kernels from the Python autograd function.  The kernel library itself is required
    either way."""
    global _OP_OK
    if _HOST == "ctypes" or HipBackend.poison_lds:      # the diagnostic hooks sit in the Python launcher
        return False
    if _OP_OK is None:
        try:
            _lib.torch_op()
            _OP_OK = True
        except (RuntimeError, OSError) as e:
            if _HOST == "op":
                raise
            _lib.lib()      # a missing kernel library is an error, not a reason to change the host path
            import warnings
            warnings.warn(f"pde_amd: the C++ torch operator is unavailable ({e}); using the Python autograd function over the same kernels")
            _OP_OK = False
    return _OP_OK
_MODES = {"recompute": 0, "stored": 1, "fused": 2}


_DEST = threading.local()


class totals_into:
    """Context: the residual-loss calls inside write their n_eqs + 1 loss scalars into `vec` (a contiguous fp32 vector on
    the fields' device) instead of a fresh tensor, and the scalars they return are views of it.  For callers that collect
    the scalars of many steps in one buffer (dist.ScalarWindow) without a copy launch per step.  None: no effect."""

    def __init__(self, vec: Optional[torch.Tensor]):
        self.vec = vec

    def __enter__(self):
        self.old = getattr(_DEST, "vec", None)
        _DEST.vec = self.vec
        return self.vec

    def __exit__(self, *exc):
        _DEST.vec = self.old
        return False


def new_totals(ne: int, device, dtype=torch.float32) -> torch.Tensor:
    """The vector a residual-loss call writes its scalars to: the one named by an enclosing totals_into, else a fresh one."""
    vec = getattr(_DEST, "vec", None)
    if vec is None or dtype is not torch.float32:
        return _lib.empty((ne + 1,), dtype=dtype, device=device)
    if vec.dtype is not torch.float32 or vec.dim() != 1 or vec.numel() != ne + 1 or not vec.is_contiguous() or vec.device != device \
            or vec.requires_grad:
        raise ValueError(f"totals_into: need a contiguous fp32 vector of {ne + 1} elements on {device} that requires no gradient, "
                         f"got {tuple(vec.shape)} {vec.dtype} on {vec.device}")
    return vec


def set_backend(b):
    """Test hook: swap the kernel launcher (the CPU test-suite installs a spec interpreter)."""
    global _BACKEND
    old, _BACKEND = _BACKEND, b
    return old


def get_backend():
    return _BACKEND


# Fields of fewer elements than this take the per-frame stored-residual kernels in 'auto' and 'fused': a marching kernel
# walks a clip's frames one after the other, and on a problem this small the chip has too few workgroups to hide that
# chain (Gray-Scott FD-c4 at 8 samples, [8, 9, 128, 128]: stored 0.062 ms, recompute 0.085, one pass 0.081).
SMALL_FIELD_ELEMENTS = 2_000_000


def resolve_mode(mode: str, spec: str, field_elements: Optional[int] = None) -> str:
    """'auto' and 'fused' -> the kernels that run: 'recompute' (forward kernel, then an adjoint that re-evaluates the
    residuals), 'stored' (forward writes the residuals, the adjoint reads them) or 'fused' (one pass).  'auto' recomputes
    where the PDE allows it, unless its spec says the stored kernels measured faster (`auto_stored`); 'fused' needs the
    recomputing adjoint and yields to what 'auto' takes where the spec says the one-pass step measured slower
    (`fused_slower`).  Small problems (`field_elements`, the elements of one field, below SMALL_FIELD_ELEMENTS) take the
    stored kernels in both."""
    if mode not in ("auto", "fused"):
        return mode
    if mode == "fused" and os.environ.get("PDE_AMD_FORCE_FUSED") == "1" and _BACKEND.can_recompute(spec):
        return "fused"      # diagnostic: measure the one-pass step where the spec prefers other kernels (tools/mode_sweep.py)
    s = SPECS[spec]
    can = _BACKEND.can_recompute(spec)
    if not can:
        return "stored"
    if field_elements is not None and field_elements < SMALL_FIELD_ELEMENTS and s.ndim > 1:
        return "stored"
    if mode == "fused" and not s.fused_slower:
        return "fused"
    return "stored" if s.auto_stored else "recompute"


class _ResidualLosses(torch.autograd.Function):
    """outputs: totals[0], ..., totals[n_eqs] as 0-dim tensors, then sums[B, n_eqs]."""

    @staticmethod
    def forward(ctx, prob, aux, sparams, weights, mode, *fields):
        need_grad = any(ctx.needs_input_grad[5:])
        if isinstance(mode, tuple) and mode[0] == "nograd":      # called under torch.no_grad(): nothing to prepare for a backward
            need_grad, mode = False, mode[1]
        assumed = None
        if isinstance(mode, tuple):      # ('fused', device scalar): the upstream gradient of the total to assume
            mode, assumed = mode
        mode = resolve_mode(mode, prob.spec, fields[0].numel())
        ctx.ghat = ctx.assumed = None
        if need_grad and mode == "fused":
            # loss and gradient in one pass; the gradient is handed out in backward, rescaled if autograd delivers
            # another upstream value than the assumed one
            totals, sums, ctx.ghat = _BACKEND.forward_backward(prob, fields, aux, sparams, weights, assumed)
            ctx.assumed, resid = assumed, None
        else:
            store = need_grad and mode == "stored"
            totals, sums, resid = _BACKEND.forward(prob, fields, aux, sparams, weights, store)
        ctx.prob, ctx.aux, ctx.sparams, ctx.weights, ctx.resid = prob, aux, sparams, weights, resid
        ctx.save_for_backward(*fields)
        ctx.set_materialize_grads(False)
        return (*totals.unbind(0), sums)

    @staticmethod
    def backward(ctx, *gouts):
        fields = ctx.saved_tensors
        *gtot, gsums = gouts
        ups = [g.to(torch.float32).contiguous() if g is not None else None for g in gtot]
        if all(u is None for u in ups):
            ups = None
        gs = gsums.to(torch.float64).contiguous() if gsums is not None else None
        if ups is None and gs is None:
            return (None,) * (5 + len(fields))
        ghat, ctx.ghat = ctx.ghat, None      # handed out once; a second backward recomputes
        if ghat is not None and gs is None and ups[-1] is not None and all(u is None for u in ups[:-1]):
            if ups[-1].data_ptr() != ctx.assumed.data_ptr():      # handed the assumed scalar itself: nothing to compare
                _BACKEND.rescale(ghat, ups[-1], ctx.assumed)
            grads = ghat
        else:
            del ghat
            grads = _BACKEND.backward(ctx.prob, fields, ctx.aux, ctx.sparams, ctx.weights, ups, gs, ctx.resid)
        out = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[5:])]
        return (None, None, None, None, None, *out)


class _ResidualsInGraph(torch.autograd.Function):
    """outputs: totals[0..n_eqs] as 0-dim tensors, sums[B, n_eqs], then the residual fields R_k [B, T_res, *spatial]
    (zero outside the lattice), all differentiable w.r.t. the fields.  One forward launch (the residual-storing one).
    backward: the stored-residual adjoint for the upstream of totals and sums; for an upstream g_k of a residual field the
    same kernel with g_k in the place of the stored residuals and weights 1/2, which makes it the plain product J^T g."""

    @staticmethod
    def forward(ctx, prob, aux, sparams, weights, *fields):
        totals, sums, resid = _BACKEND.forward(prob, fields, aux, sparams, weights, True)
        ctx.prob, ctx.aux, ctx.sparams, ctx.weights, ctx.nf = prob, aux, sparams, weights, len(fields)
        ctx.save_for_backward(*fields, *resid)
        ctx.set_materialize_grads(False)
        return (*totals.unbind(0), sums, *resid)

    @staticmethod
    def backward(ctx, *gouts):
        saved = ctx.saved_tensors
        fields, resid = saved[:ctx.nf], saved[ctx.nf:]
        ne = len(resid)
        gtot, gsums, gres = gouts[:ne + 1], gouts[ne + 1], gouts[ne + 2:]
        ups = [g.to(torch.float32).contiguous() if g is not None else None for g in gtot]
        gs = gsums.to(torch.float64).contiguous() if gsums is not None else None
        grads = None
        if gs is not None or any(u is not None for u in ups):
            grads = _BACKEND.backward(ctx.prob, fields, ctx.aux, ctx.sparams, ctx.weights,
                                      ups if any(u is not None for u in ups) else None, gs, list(resid))
        if any(g is not None for g in gres):
            f0 = fields[0]
            g = [x.to(torch.float32).contiguous() if x is not None else torch.zeros_like(resid[0]) for x in gres]
            half = torch.full_like(ctx.weights, 0.5)
            one = torch.ones((), dtype=torch.float32, device=f0.device)
            extra = _BACKEND.backward(ctx.prob, fields, ctx.aux, ctx.sparams, half, [None] * ne + [one], None, g)
            grads = extra if grads is None else [a.add_(b) for a, b in zip(grads, extra)]
        if grads is None:
            return (None,) * (4 + ctx.nf)
        out = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[4:])]
        return (None, None, None, None, *out)


def spectral_pair_supported(nx: int, ny: int) -> bool:
    """whether the LDS-resident spectral Laplacian kernel takes planes of nx x ny points"""
    return bool(_lib.lib().pde_amd_spectral_pair_supported(int(nx), int(ny)))


def _spectral_pair_apply(a: torch.Tensor, b: torch.Tensor, mult: torch.Tensor):
    if not (a.is_cuda and b.is_cuda and mult.is_cuda):
        raise RuntimeError("pde_amd: the spectral Laplacian kernel needs tensors on a HIP device; there is no CPU fallback")
    a, b = a.contiguous(), b.contiguous()
    nx, ny = a.shape[-2], a.shape[-1]
    if a.shape != b.shape or a.dtype is not torch.float32 or b.dtype is not torch.float32 or tuple(mult.shape) != (nx, ny):
        raise ValueError(f"spectral Laplacian: two fp32 fields of one shape and a multiplier [{nx}, {ny}] expected")
    la, lb = _lib.empty_like(a), _lib.empty_like(b)
    planes = a.numel() // (nx * ny)
    with torch.cuda.device(a.device):
        stream = C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        if HipBackend.poison_lds:
            _lib.check(_lib.lib().pde_amd_diag_poison_lds(stream), "diag_poison_lds")
        _lib.check(_lib.lib().pde_amd_spectral_laplacian_pair(a.data_ptr(), b.data_ptr(), la.data_ptr(), lb.data_ptr(), mult.data_ptr(),
                                                              planes, nx, ny, stream), "spectral_laplacian_pair")
    return la, lb


class _SpectralLaplacianPair(torch.autograd.Function):
    """(A, B) -> (ifft2(m * fft2(A)).real, ifft2(m * fft2(B)).real) for a real multiplier m that is even in both
    wavenumbers (the reference's -K2, finetune/pde_loss_verified.py:200-214): such an operator is symmetric, so the
    backward pass applies it to the two upstream gradients."""

    @staticmethod
    def forward(ctx, a, b, mult):
        ctx.save_for_backward(mult)
        return _spectral_pair_apply(a.detach(), b.detach(), mult)

    @staticmethod
    def backward(ctx, ga, gb):
        (mult,) = ctx.saved_tensors
        if ga is None and gb is None:
            return None, None, None
        ga = torch.zeros_like(gb) if ga is None else ga
        gb = torch.zeros_like(ga) if gb is None else gb
        xa, xb = _spectral_pair_apply(ga, gb, mult)
        return xa, xb, None


def spectral_laplacian_pair(a: torch.Tensor, b: torch.Tensor, multiplier: torch.Tensor):
    """Spectral Laplacians of two fp32 fields [..., nx, ny] in one launch (both planes of a sample and frame share one
    complex transform in LDS); differentiable.  `multiplier`: [nx, ny] fp32, real and even in both wavenumbers."""
    return _SpectralLaplacianPair.apply(a, b, multiplier)


def residual_losses_and_fields(prob: StencilProblem, fields: Sequence[torch.Tensor], weights: torch.Tensor,
                               sparams: Sequence[torch.Tensor] = (), aux: Sequence[torch.Tensor] = ()):
    """residual_losses plus the residual fields themselves, in the autograd graph like the reference's
    (e.g. finetune/pde_loss.py:216 returns them next to the losses): (totals, sums, [R_k [B, T_res, *spatial]])."""
    fs, sp, ax, w = _prepare(prob, fields, weights, sparams, aux)
    ne = len(SPECS[prob.spec].eqs)
    out = _ResidualsInGraph.apply(prob, tuple(ax), tuple(sp), w, *fs)
    return list(out[:ne + 1]), out[ne + 1], list(out[ne + 2:])


def _f32c(t, device):
    """fp32, contiguous, on `device`; the tensor itself when it already is (the usual case, and the cheap one)"""
    if t.dtype is not torch.float32 or t.device != device:
        t = t.to(device=device, dtype=torch.float32)
    return t if t.is_contiguous() else t.contiguous()


def _prepare(prob, fields, weights, sparams, aux):
    """Shape checks and fp32-contiguous copies shared by the entry points below."""
    s = SPECS[prob.spec]
    if len(fields) != len(s.fields) or len(sparams) != len(s.sparams) or len(aux) != len(s.aux):
        raise ValueError(f"{prob.spec}: expected {len(s.fields)} fields, {len(s.sparams)} per-sample parameters, {len(s.aux)} aux fields")
    f0 = fields[0]
    shape, dev = f0.shape, f0.device
    fs = []
    for f in fields:
        if f.shape != shape:
            raise ValueError(f"{prob.spec}: all fields must share one shape, got {tuple(f.shape)} vs {tuple(shape)}")
        fs.append(_f32c(f, f.device))
    sp = [_f32c(p, dev) for p in sparams]
    ax = [_f32c(x, dev) for x in aux]
    for p in sp:
        if p.dim() != 1 or p.shape[0] != shape[0]:
            raise ValueError(f"{prob.spec}: per-sample parameters must have shape [{shape[0]}], got {tuple(p.shape)}")
    for x in ax:
        if x.shape[0] != shape[0] or x.shape[1:] != shape[2:]:
            raise ValueError(f"{prob.spec}: auxiliary fields must be [B, *spatial], got {tuple(x.shape)}")
    return fs, sp, ax, _f32c(weights, dev)


@torch.no_grad()
def residual_fields(prob: StencilProblem, fields: Sequence[torch.Tensor], weights: torch.Tensor,
                    sparams: Sequence[torch.Tensor] = (), aux: Sequence[torch.Tensor] = ()):
    """The raw residual of every equation as [B, T_res, *spatial] fp32 tensors (zero outside the residual lattice),
    written by the forward kernel; no autograd.  Double-precision fields are evaluated in double precision (per-sample
    parameters, auxiliary fields and weights are converted to match) and give double-precision residuals."""
    if fields[0].dtype is torch.float64:
        dev = fields[0].device
        f64 = lambda t: t.detach().to(device=dev, dtype=torch.float64).contiguous()
        s = SPECS[prob.spec]
        if len(fields) != len(s.fields) or len(sparams) != len(s.sparams) or len(aux) != len(s.aux):
            raise ValueError(f"{prob.spec}: expected {len(s.fields)} fields, {len(s.sparams)} per-sample parameters, {len(s.aux)} aux fields")
        if any(f.shape != fields[0].shape for f in fields):
            raise ValueError(f"{prob.spec}: all fields must share one shape")
        return _BACKEND.forward_f64(prob, [f64(f) for f in fields], tuple(f64(x) for x in aux), tuple(f64(p) for p in sparams), f64(weights))[2]
    fs, sp, ax, w = _prepare(prob, fields, weights, sparams, aux)
    _, _, resid = _BACKEND.forward(prob, [f.detach() for f in fs], tuple(ax), tuple(sp), w, True)
    return resid


def residual_losses(prob: StencilProblem, fields: Sequence[torch.Tensor], weights: torch.Tensor,
                    sparams: Sequence[torch.Tensor] = (), aux: Sequence[torch.Tensor] = (),
                    mode: str = "auto"):
    """-> (list of n_eqs + 1 scalar fp32 tensors: per-equation losses then their sum, sums[B, n_eqs] fp64).

    loss_k = sum_{b, t, site} weights[k, t] * R_k(b, t, site)^2.
    mode: 'recompute' (adjoint kernel re-evaluates the residuals; nothing is stored between forward and
    backward), 'stored' (forward writes the residuals, backward reads them), 'auto', or ('fused', assumed): loss and
    gradient of assumed * total in one pass at forward time when the fields need gradients (`assumed` is a device
    scalar); backward hands the gradient out, rescaled on the device if the upstream value differs.
    """
    be = _BACKEND
    # The C++ operator: same kernels, same results; checks, conversions and the autograd node in C++.  While launches are
    # being timed it still serves the fused step, whose adjoint is launched from this thread, inside the call (the
    # library records the events around that launch); the other modes launch their adjoint from autograd's thread, where
    # only the Python function can arm the timer.
    timed = be.backward_events is not None or be.forward_events is not None if type(be) is HipBackend else False
    fused_timed = timed and type(mode) is tuple and resolve_mode(mode[0], prob.spec, fields[0].numel()) == "fused" and be.forward_events is None \
        and torch.is_grad_enabled() and any(f.requires_grad for f in fields)
    if type(be) is HipBackend and (not timed or fused_timed) and _use_op():
        assumed = None
        if type(mode) is tuple:
            mode, assumed = mode
        mode = resolve_mode(mode, prob.spec, fields[0].numel())
        ev = be._arm_adjoint_timer() if fused_timed else None
        try:
            out = _lib.torch_op()(fields, aux, sparams, weights, *prob.op_args(), _MODES[mode], assumed, getattr(_DEST, "vec", None))
        finally:
            be._disarm_adjoint_timer(ev)
        return out[:-1], out[-1]
    fs, sp, ax, w = _prepare(prob, fields, weights, sparams, aux)
    if not torch.is_grad_enabled():      # inside Function.forward the grad mode is always off, so it is passed in
        mode = ("nograd", mode)
    out = _ResidualLosses.apply(prob, tuple(ax), tuple(sp), w, mode, *fs)
    return list(out[:-1]), out[-1]
