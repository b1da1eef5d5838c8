"""Boundary-strip, wall and patch-seam losses on the model's channel-last tensors.

Each function mirrors one reference helper (same name where the reference has one name per trainer the
variant is spelled out).  They all reduce to sums of squares over a handful of "terms" (Cartesian
products of index lists, one per slice the reference concatenates), evaluated by ONE forward launch
per term set and differentiated by one launch per term (one for the whole set where its terms touch no common element); the tiny sqrt / mean epilogue is ordinary
torch on [n_terms] numbers, so autograd chains it.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import reductions as R
from .sink import sink_of, unwrap

CH_VX, CH_VY = 0, 1      # finetune/train_ns_incom_lora_v3.py:74-75


_last_mask = [None, -1, None]      # weak reference to the mask tensor last read, its version counter, the channel list


def valid_channels(channel_mask: torch.Tensor) -> List[int]:
    """`torch.where(mask > 0)[0]` of the first row (finetune/train_taylor_green_2d_lora_v3.py:141-144).  Reading the list
    synchronises with the device, as in the reference; the losses and metrics of one batch share the same mask tensor, so
    the answer for that very tensor (same object, not modified in place since) is remembered and the batch pays once."""
    ref, version, ch = _last_mask
    if ref is not None and ref() is channel_mask and version == channel_mask._version:
        return list(ch)
    m = channel_mask if channel_mask.dim() == 1 else channel_mask[0]
    ch = [int(i) for i in torch.where(m > 0)[0].tolist()]
    try:
        _last_mask[:] = [weakref.ref(channel_mask), channel_mask._version, ch]
    except TypeError:
        _last_mask[:] = [None, -1, None]
    return list(ch)


class TermSet:
    """Host description of a set of terms plus the cached device copy of their index lists."""

    _cache: Dict[tuple, "TermSet"] = {}

    def __init__(self, dims: Sequence[int]):
        self.dims = [int(d) for d in dims]          # d0, d1, d2, C
        self.terms: List[_lib.Term] = []
        self.groups: List[int] = []                  # which mean a term belongs to
        self.pool: List[int] = []
        self.elems: List[int] = []                   # selected elements per outer index
        self._dev: Dict[torch.device, torch.Tensor] = {}
        self._means: Dict[tuple, torch.Tensor] = {}
        self.foot: List[List[set]] = []
        self.disjoint = True                         # no two terms touch a common element: their backward may run as one launch      # group-mean matrices by (device, outer, groups)

    def add(self, lists: Sequence[Sequence[int]], group: int = 0, use_target: bool = True, axis: int = 0,
            taps: Sequence[Tuple[int, float]] = ((0, 1.0),)):
        t = _lib.Term()
        n = 1
        for k in range(4):
            t.n[k] = len(lists[k])
            t.idx_off[k] = len(self.pool)
            self.pool.extend(int(i) for i in lists[k])
            n *= len(lists[k])
        t.axis, t.n_taps = axis, len(taps)
        for j, (s, c) in enumerate(taps):
            t.shift[j], t.coef[j] = int(s), float(c)
        t.use_target = int(use_target)
        # elements of pred the term touches, per axis (the taps widen the list of their axis): two terms can only write the
        # same gradient element if these sets meet on every axis
        foot = [set(int(i) for i in lists[k]) for k in range(4)]
        foot[axis] = {i + int(sh) for i in foot[axis] for sh, _ in taps}
        self.disjoint = self.disjoint and all(any(not (foot[k] & other[k]) for k in range(4)) for other in self.foot)
        self.foot.append(foot)
        self.terms.append(t)
        self.groups.append(group)
        self.elems.append(n)

    def indices(self, device) -> torch.Tensor:
        if device not in self._dev:
            self._dev[device] = torch.tensor(self.pool if self.pool else [0], dtype=torch.int32, device=device)
        return self._dev[device]


class _TermSums(torch.autograd.Function):
    """sums[t] = sum of squared term values, fp64 [n_terms]; differentiable w.r.t. pred."""

    @staticmethod
    def forward(ctx, pred, target, ts: TermSet, outer: int, sink, affine):
        be = R.get_backend()
        ctx.ts, ctx.outer, ctx.sink, ctx.affine = ts, outer, sink, affine
        ctx.save_for_backward(pred, target if target is not None else pred.new_empty(0))
        ctx.has_target = target is not None
        return be.terms_forward(pred, target, outer, ts.dims, ts.terms, ts.indices(pred.device), affine)

    @staticmethod
    def backward(ctx, gsums):
        pred, target = ctx.saved_tensors
        be = R.get_backend()
        args = (pred, target if ctx.has_target else None, ctx.outer, ctx.ts.dims, ctx.ts.terms,
                ctx.ts.indices(pred.device), gsums.to(torch.float32).contiguous(), ctx.affine, ctx.ts.disjoint)
        if ctx.sink is not None:      # added into the shared buffer of `pred` once that exists
            ctx.sink.terms.append(args)
            return None, None, None, None, None, None
        grad = torch.zeros_like(pred)
        be.terms_backward(*args, grad)
        return grad, None, None, None, None, None


def _prep(x: torch.Tensor, what: str) -> torch.Tensor:
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()


def _term_means(pred, target, ts: TermSet, n_groups: int) -> List[torch.Tensor]:
    """mean of the squared term values of each group, as fp64 0-dim tensors (nan for an empty group, like torch.mean)."""
    pred, affine = unwrap(pred)      # a Denormalised prediction: std and mean are applied inside the kernels
    pred = _prep(pred, "pred")
    target = _prep(target, "target") if target is not None else None
    outer = int(pred.shape[0]) * int(pred.shape[1])
    sums = _TermSums.apply(pred, target, ts, outer, sink_of(pred), affine)
    counts = [outer * sum(ts.elems[i] for i, gg in enumerate(ts.groups) if gg == g) for g in range(n_groups)]
    if all(c > 0 for c in counts):
        # one small matrix product: means[g] = sum_i [term i in group g] / count_g * sums[i]
        key = (pred.device, outer, n_groups)
        m = ts._means.get(key)
        if m is None:
            m = torch.zeros(n_groups, len(ts.groups), dtype=torch.float64)
            for i, g in enumerate(ts.groups):
                m[g, i] = 1.0 / counts[g]
            m = ts._means[key] = m.to(pred.device)
        return list((m @ sums).unbind(0))
    out = []
    for g in range(n_groups):
        sel = [i for i, gg in enumerate(ts.groups) if gg == g]
        tot = sums[sel].sum()
        out.append(tot / counts[g] if counts[g] > 0 else tot * float("nan"))
    return out


def _sl(n: int, s: slice) -> List[int]:
    return list(range(n))[s]


def _dims4(shape: Sequence[int], nd: int) -> List[int]:
    """[d0, d1, d2, C] of a [B, T, *spatial(nd), C] tensor (leading spatial axes padded with 1)."""
    sp = [int(v) for v in shape[2:2 + nd]]
    return [1] * (3 - nd) + sp + [int(shape[-1])]


def _cached(key, build) -> TermSet:
    ts = TermSet._cache.get(key)
    if ts is None:
        if len(TermSet._cache) > 64:
            TermSet._cache.clear()
        ts = build()
        TermSet._cache[key] = ts
    return ts


def _rmse(pred, target, ts) -> torch.Tensor:
    mse = _term_means(pred, target, ts, 1)[0]
    return torch.sqrt(mse + 1e-8).to(torch.float32)


def boundary_loss_1d(pred, target, channel_mask, bc_width: int = 1) -> torch.Tensor:
    """finetune/train_burgers_1d_lora_v3.py:111-139 — [B, T, X, C]; left and right strips of width bc_width."""
    ch, w = valid_channels(channel_mask), bc_width
    X = pred.shape[2]

    def build():
        ts = TermSet(_dims4(pred.shape, 1))
        ts.add([[0], [0], _sl(X, slice(None, w)), ch])
        ts.add([[0], [0], _sl(X, slice(-w, None)), ch])
        return ts
    return _rmse(pred, target, _cached(("1d", tuple(pred.shape[2:]), tuple(ch), w), build))


def boundary_loss_2d_strips(pred, target, channel_mask, bc_width: int = 1) -> torch.Tensor:
    """finetune/train_taylor_green_2d_lora_v3.py:132-166 (same in train_rayleigh_benard_lora_v3.py:138-175) —
    [B, T, H, W, C]; top/bottom full rows, left/right without corners.  train_burgers_3d_apebench_lora.py:115-150 is the
    same text applied to a 6-D tensor, which gives different arithmetic: see boundary_loss_3d_xy_strips."""
    ch, w = valid_channels(channel_mask), bc_width
    H, W = pred.shape[2], pred.shape[3]

    def build():
        ts = TermSet(_dims4(pred.shape, 2))
        allw = list(range(W))
        ts.add([[0], _sl(H, slice(None, w)), allw, ch])
        ts.add([[0], _sl(H, slice(-w, None)), allw, ch])
        mid = _sl(H, slice(w, -w))
        ts.add([[0], mid, _sl(W, slice(None, w)), ch])
        ts.add([[0], mid, _sl(W, slice(-w, None)), ch])
        return ts
    return _rmse(pred, target, _cached(("2ds", tuple(pred.shape[2:]), tuple(ch), w), build))


def boundary_loss_2d_edges(pred, target, channel_mask) -> torch.Tensor:
    """finetune/train_shear_flow_lora_v3.py:111-146 — single-pixel edges; the four corners count twice."""
    ch = valid_channels(channel_mask)
    H, W = pred.shape[2], pred.shape[3]

    def build():
        ts = TermSet(_dims4(pred.shape, 2))
        allh, allw = list(range(H)), list(range(W))
        ts.add([[0], allh, [0], ch])
        ts.add([[0], allh, [W - 1], ch])
        ts.add([[0], [0], allw, ch])
        ts.add([[0], [H - 1], allw, ch])
        return ts
    return _rmse(pred, target, _cached(("2de", tuple(pred.shape[2:]), tuple(ch)), build))


def boundary_loss_3d_faces(pred, target, channel_mask) -> torch.Tensor:
    """finetune/train_heat_3d_lora_v3.py:109-141 — [B, T, X, Y, Z, C]; six single-cell faces.  As in the reference,
    the two z faces index the Y axis with the channel list and keep all C channels (:133-134), so Y must exceed
    the largest valid channel index."""
    ch = valid_channels(channel_mask)
    X, Y, Z, Cn = pred.shape[2], pred.shape[3], pred.shape[4], pred.shape[5]
    if ch and max(ch) >= Y:
        raise IndexError(f"index {max(ch)} is out of bounds for dimension 3 with size {Y}")

    def build():
        ts = TermSet(_dims4(pred.shape, 3))
        ax, ay, az, ac = list(range(X)), list(range(Y)), list(range(Z)), list(range(Cn))
        ts.add([[0], ay, az, ch]); ts.add([[X - 1], ay, az, ch])
        ts.add([ax, [0], az, ch]); ts.add([ax, [Y - 1], az, ch])
        ts.add([ax, ch, [0], ac]); ts.add([ax, ch, [Z - 1], ac])
        return ts
    return _rmse(pred, target, _cached(("3df", tuple(pred.shape[2:]), tuple(ch)), build))


def boundary_loss_3d_shells(pred, target, channel_mask, bc_width: int = 1) -> torch.Tensor:
    """finetune/train_advection_3d_lora_v3.py:110-159 — six face slabs of width bc_width without double counting."""
    ch, w = valid_channels(channel_mask), bc_width
    X, Y, Z = pred.shape[2], pred.shape[3], pred.shape[4]

    def build():
        ts = TermSet(_dims4(pred.shape, 3))
        ay, az = list(range(Y)), list(range(Z))
        mx, my = _sl(X, slice(w, -w)), _sl(Y, slice(w, -w))
        ts.add([_sl(X, slice(None, w)), ay, az, ch]); ts.add([_sl(X, slice(-w, None)), ay, az, ch])
        ts.add([mx, _sl(Y, slice(None, w)), az, ch]); ts.add([mx, _sl(Y, slice(-w, None)), az, ch])
        ts.add([mx, my, _sl(Z, slice(None, w)), ch]); ts.add([mx, my, _sl(Z, slice(-w, None)), ch])
        return ts
    return _rmse(pred, target, _cached(("3ds", tuple(pred.shape[2:]), tuple(ch), w), build))


def boundary_loss_3d_xy_strips(pred, target, channel_mask, bc_width: int = 1) -> torch.Tensor:
    """finetune/train_burgers_3d_apebench_lora.py:115-150 as it behaves on that trainer's [B, T, X, Y, Z, C] tensors: the
    2-D strip slicing runs over X and Y (top/bottom full rows of X, left/right strips of Y without the corners), the
    channel list indexes dim 4, i.e. selects Z planes, and all C channels are kept.  Reproduced because that trainer's
    loss values depend on it; Z must exceed the largest valid channel index, as in the reference."""
    ch, w = valid_channels(channel_mask), bc_width
    X, Y, Z, Cn = pred.shape[2], pred.shape[3], pred.shape[4], pred.shape[5]
    if ch and max(ch) >= Z:
        raise IndexError(f"index {max(ch)} is out of bounds for dimension 4 with size {Z}")

    def build():
        ts = TermSet(_dims4(pred.shape, 3))
        ay, ac = list(range(Y)), list(range(Cn))
        ts.add([_sl(X, slice(None, w)), ay, ch, ac])
        ts.add([_sl(X, slice(-w, None)), ay, ch, ac])
        mid = _sl(X, slice(w, -w))
        ts.add([mid, _sl(Y, slice(None, w)), ch, ac])
        ts.add([mid, _sl(Y, slice(-w, None)), ch, ac])
        return ts
    return _rmse(pred, target, _cached(("3dxy", tuple(pred.shape[2:]), tuple(ch), w), build))


class PeriodicBCLoss(torch.nn.Module):
    """tools/test_gt_periodic_bc.py:38-105 (tools/test_gt_fp32_pde_bc.py:17 imports it from the loss library, which does not
    define it): for every field [B, T, H, W] or [T, H, W], the mean squared mismatch of the linear extrapolation across the
    periodic wrap, (2 f[-1] - f[-2] - f[0])^2, along dim -2 ("x") and dim -1 ("y").
    forward({name: field}) -> (total, {name_val_x, name_val_y, name_total, ..., total})."""

    def __init__(self, dx: float, dy: float):
        super().__init__()
        self.dx, self.dy = dx, dy

    def forward(self, fields: Dict[str, torch.Tensor], reduction: str = "mean"):
        losses: Dict[str, torch.Tensor] = {}
        total = None
        taps = lambda n: ((0, -1.0), (1, 2.0), (-(n - 2), -1.0))      # anchored at index n - 2: -f[n-2] + 2 f[n-1] - f[0]
        for name, f in fields.items():
            x = f if f.dim() == 4 else f.unsqueeze(0)
            x = x.unsqueeze(-1)                                        # one channel
            H, W = x.shape[2], x.shape[3]
            if H < 2 or W < 2:
                raise IndexError("PeriodicBCLoss needs at least two points along both axes")

            def build():
                ts = TermSet(_dims4(x.shape, 2))
                ts.add([[0], [H - 2], list(range(W)), [0]], group=0, use_target=False, axis=1, taps=taps(H))
                ts.add([[0], list(range(H)), [W - 2], [0]], group=1, use_target=False, axis=2, taps=taps(W))
                return ts
            vx, vy = _term_means(x, None, _cached(("pbc", H, W), build), 2)
            vx, vy = vx.to(torch.float32), vy.to(torch.float32)
            losses[f"{name}_val_x"], losses[f"{name}_val_y"], losses[f"{name}_total"] = vx, vy, vx + vy
            total = vx + vy if total is None else total + (vx + vy)
        if total is None:
            raise StopIteration("PeriodicBCLoss: no fields")
        losses["total"] = total
        return total, losses


def compute_boundary_loss(pred, target, channel_mask, bc_width: int = 1, variant: Optional[str] = None) -> torch.Tensor:
    """The reference has one `compute_boundary_loss` per trainer; `variant` picks which
    ('1d', 'strips', 'edges', 'faces', 'shells', 'xy_strips'); by default the tensor rank decides: 4 -> '1d', 5 -> 'strips',
    6 -> 'shells'.  'xy_strips' is what the 3-D Burgers trainer's copy computes on its 6-D tensors
    (finetune/train_burgers_3d_apebench_lora.py:115-150); it is never picked by rank."""
    if variant is None:
        variant = {4: "1d", 5: "strips", 6: "shells"}.get(pred.dim())
    if variant == "1d":
        return boundary_loss_1d(pred, target, channel_mask, bc_width)
    if variant == "strips":
        return boundary_loss_2d_strips(pred, target, channel_mask, bc_width)
    if variant == "edges":
        return boundary_loss_2d_edges(pred, target, channel_mask)
    if variant == "faces":
        return boundary_loss_3d_faces(pred, target, channel_mask)
    if variant == "shells":
        return boundary_loss_3d_shells(pred, target, channel_mask, bc_width)
    if variant == "xy_strips":
        return boundary_loss_3d_xy_strips(pred, target, channel_mask, bc_width)
    raise ValueError(f"compute_boundary_loss: unknown variant {variant!r} for a {pred.dim()}-d tensor")


def compute_noslip_loss(output, ch_vx: int = CH_VX, ch_vy: int = CH_VY) -> torch.Tensor:
    """finetune/train_ns_incom_lora_v3.py:150-176 — mean(u_wall^2) + mean(v_wall^2) over the four walls (corners twice)."""
    H, W = output.shape[2], output.shape[3]

    def build():
        ts = TermSet(_dims4(output.shape, 2))
        allh, allw = list(range(H)), list(range(W))
        for g, c in enumerate((ch_vx, ch_vy)):
            ts.add([[0], [0], allw, [c]], group=g, use_target=False)
            ts.add([[0], [H - 1], allw, [c]], group=g, use_target=False)
            ts.add([[0], allh, [0], [c]], group=g, use_target=False)
            ts.add([[0], allh, [W - 1], [c]], group=g, use_target=False)
        return ts
    mu, mv = _term_means(output, None, _cached(("noslip", tuple(output.shape[2:]), ch_vx, ch_vy), build), 2)
    return (mu + mv).to(torch.float32)


def compute_patch_boundary_loss(output, channel_mask, patch_size: int = 16) -> torch.Tensor:
    """finetune/train_rayleigh_benard_lora_v3.py:114-135 — mean squared curvature x[P-2] - 2 x[P-1] + x[P] across every
    patch seam, along W and along H."""
    ch, P = valid_channels(channel_mask), patch_size
    H, W = output.shape[2], output.shape[3]

    def seams(n):
        a, b, c = _sl(n, slice(P - 2, -2, P)), _sl(n, slice(P - 1, -1, P)), _sl(n, slice(P, None, P))
        if not (len(a) == len(b) == len(c)):
            raise RuntimeError(f"The size of tensor a ({len(a)}) must match the size of tensor b ({len(b)}) at a patch seam axis of length {n}")
        return a

    def build():
        ts = TermSet(_dims4(output.shape, 2))
        taps = ((0, 1.0), (1, -2.0), (2, 1.0))
        ts.add([[0], list(range(H)), seams(W), ch], group=0, use_target=False, axis=2, taps=taps)
        ts.add([[0], seams(H), list(range(W)), ch], group=1, use_target=False, axis=1, taps=taps)
        return ts
    mh, mv = _term_means(output, None, _cached(("patch", tuple(output.shape[2:]), tuple(ch), P), build), 2)
    return (mh + mv).to(torch.float32)
