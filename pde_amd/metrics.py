"""Evaluation metrics on channel-last tensors, one device pass each.

Every function mirrors a reference helper (same name, arguments, epsilons, return types).  The device work is
`reductions.pair_stats`: per (group, channel) sums of (p-g)^2, g^2, g, p, (p-g)^2/(g^2+eps) and max|p-g| in
fp64; what remains is arithmetic on [groups, channels] numbers.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import reductions as R
from .sink import Denormalised
from .boundary import valid_channels


def _flat(pred, target, n_lead: int):
    """(groups, rows, channels) of a [lead..., *spatial, C] pair with the first n_lead axes as groups."""
    if pred.shape != target.shape:
        raise RuntimeError(f"The size of tensor a {tuple(pred.shape)} must match the size of tensor b {tuple(target.shape)}")
    C = int(pred.shape[-1])
    G = 1
    for v in pred.shape[:n_lead]:
        G *= int(v)
    rows = pred.numel() // max(G * C, 1)
    return G, rows, C


def compute_vrmse(output, target, channel_mask) -> Tuple[torch.Tensor, torch.Tensor]:
    """finetune/train_taylor_green_2d_lora_v3.py:204-232 — (vrmse, rmse) over the valid channels:
    rmse = sqrt(mse + 1e-8); vrmse = mean_c sqrt(mse_c / (var_c + 1e-8))."""
    ch = valid_channels(channel_mask)
    G, rows, C = _flat(output, target, 0)
    if isinstance(output, Denormalised):
        # one group per sample, std and mean applied while reading; the groups' sums are then added
        B = int(output.shape[0])
        per = R.get_backend().pair_stats(output.base.detach(), target, B, rows // B, C, affine=(output.scale, output.shift))
        st = per.sum(0)
    else:
        st = R.get_backend().pair_stats(output, target, 1, rows, C)[0]        # [C, NSTAT]
    if len(ch) <= 64:      # one tiny launch instead of a dozen element-wise ones
        out = R.get_backend().vrmse_from_stats(st, ch, rows)
        return out[0], out[1]
    sel = st[ch]
    n = float(rows)
    mse = sel[:, R.S_D2].sum() / (n * len(ch)) if ch else sel[:, R.S_D2].sum() * float("nan")
    rmse = torch.sqrt(mse + 1e-8)
    mse_c = sel[:, R.S_D2] / n
    mean_g = sel[:, R.S_G] / n
    var_c = sel[:, R.S_G2] / n - mean_g * mean_g
    vrmse = torch.sqrt(mse_c / (var_c + 1e-8)).sum() / len(ch) if ch else mse * float("nan")
    return vrmse.to(torch.float32), rmse.to(torch.float32)


@torch.no_grad()
def validation_scalars(output, input_data, target, channel_mask, pde_loss_fn, field_channels: Sequence[int], bc_width: int = 1,
                       **loss_kwargs):
    """The four numbers validate() adds up per batch (finetune/train_taylor_green_2d_lora_v3.py:252-275):
    (bc_loss, pde_loss, rmse, vrmse) = compute_boundary_loss, compute_pde_loss[0] and compute_vrmse of one model output,
    with ONE sweep over that output: the statistics pass also writes the fields of the residual loss (the gather), and the
    boundary strips are a few rows.  `output` may be a Denormalised (std and mean applied while reading).  loss_kwargs are
    handed to pde_loss_fn after the fields (nu=..., t_offset=...).  bc_loss and pde_loss equal the separate functions' bit for
    bit, rmse and vrmse to the rounding of their fp64 sums (another summation order)."""
    from .boundary import compute_boundary_loss
    from .sink import unwrap
    out, affine = unwrap(output)
    ch = valid_channels(channel_mask)
    stats, fields = R.get_backend().pair_stats_gather(out.detach(), target, input_data.detach().float().contiguous(), field_channels, affine)
    rows = out.numel() // int(out.shape[-1])
    st = stats.sum(0)
    if len(ch) <= 64:
        vr = R.get_backend().vrmse_from_stats(st, ch, rows)
        vrmse, rmse = vr[0], vr[1]
    else:
        vrmse, rmse = compute_vrmse(output, target, channel_mask)
    pde_loss, _ = pde_loss_fn(*fields, **loss_kwargs)
    bc_loss = compute_boundary_loss(output, target, channel_mask, bc_width)
    return bc_loss, pde_loss, rmse, vrmse


def nrmse_per_sample(pred, gt) -> torch.Tensor:
    """tools/eval_rollout.py:60-72 — [M, ...] -> [M]: sqrt(sum (p-g)^2 / (sum g^2 + 1e-12)) over everything but M."""
    M = int(pred.shape[0])
    rows = pred.numel() // max(M, 1)
    st = R.get_backend().pair_stats(pred.reshape(M, rows, 1), gt.reshape(M, rows, 1), M, rows, 1)
    return torch.sqrt(st[:, 0, R.S_D2] / (st[:, 0, R.S_G2] + 1e-12)).to(torch.float32)


def nrmse_per_channel(pred, gt, valid_channels_: Sequence[Tuple[int, str]]) -> Dict[str, torch.Tensor]:
    """tools/eval_rollout.py:75-90 — {name: [M]} for (channel index, name) pairs; one pass for all channels."""
    M, rows, C = _flat(pred, gt, 1)
    st = R.get_backend().pair_stats(pred, gt, M, rows, C)
    return {name: torch.sqrt(st[:, ci, R.S_D2] / (st[:, ci, R.S_G2] + 1e-12)).to(torch.float32) for ci, name in valid_channels_}


def compute_pdebench_nrmse(pred, target) -> Tuple[float, List[float]]:
    """eval/eval_omni.py:243-278 — per channel sqrt(mean(((p - g) / (std(g) + 1e-8))^2)) with the unbiased std of the whole
    target channel; returns (mean over channels, per-channel list) as Python floats, from ONE pass over the two tensors
    and one host read (the reference takes a `.item()` per channel)."""
    G, rows, C = _flat(pred, target, 0)
    st = R.get_backend().pair_stats(pred, target, 1, rows, C)[0]            # [C, NSTAT] fp64
    n = float(rows)
    var = (st[:, R.S_G2] - st[:, R.S_G] * st[:, R.S_G] / n) / (n - 1.0)
    sigma = torch.sqrt(var.clamp_min(0.0)) + 1e-8
    per = (torch.sqrt(st[:, R.S_D2] / n) / sigma).tolist()
    return float(np.mean(per)), [float(v) for v in per]


class LpLoss:
    """eval/metrics.py:599-623 — relative Lp error per example, ||x - y||_p / (eps + ||y||_p) over everything but axis 0.
    p = 2 is one pass of the statistics kernel over both tensors; other p use device-side torch reductions."""

    def __init__(self, p=2, reduction="mean"):
        if not p > 0:
            raise AssertionError("the norm order p must be positive")
        self.p, self.reduction = p, reduction

    def _reduce(self, r):
        """per-example ratios -> their mean / sum, or the ratios themselves for any other reduction string"""
        fold = {"mean": torch.mean, "sum": torch.sum}.get(self.reduction)
        return r if fold is None else fold(r)

    def _ratio(self, x, y, eps):
        n = x.size()[0]
        if self.p == 2 and x.dtype == y.dtype and x.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
            rows = x.numel() // max(n, 1)
            st = R.get_backend().pair_stats(x.reshape(n, rows, 1), y.reshape(n, rows, 1), n, rows, 1)
            return (torch.sqrt(st[:, 0, R.S_D2]) / (eps + torch.sqrt(st[:, 0, R.S_G2]))).to(x.dtype)
        d = torch.norm(x.reshape(n, -1) - y.reshape(n, -1), self.p, 1)
        return d / (eps + torch.norm(y.reshape(n, -1), self.p, 1))

    def __call__(self, x, y, eps=1e-20):
        return self._reduce(self._ratio(x, y, eps))


class FftLpLoss(LpLoss):
    """eval/metrics.py:626-678 — the same ratio on the Fourier coefficients (rocFFT through torch.fft), optionally on
    the band flow <= k < fhigh of every transformed axis; |z|^2 sums for p = 2 come from the statistics kernel on the
    real view of the complex coefficients."""

    def __call__(self, x, y, flow=None, fhigh=None, eps=1e-20):
        dims = list(range(1, len(x.shape)))
        xf = torch.fft.fftn(x, dim=dims)
        yf = torch.fft.fftn(y, dim=dims)
        lo = 0 if flow is None else flow
        hi = int(np.max(xf.shape[1:])) if fhigh is None else fhigh
        if 1 <= len(dims) <= 4:
            sl = (slice(None),) + (slice(lo, hi),) * len(dims)
            xf, yf = xf[sl], yf[sl]
        yf = yf.reshape(xf.shape)
        if self.p == 2:
            r = self._ratio(torch.view_as_real(xf.contiguous()), torch.view_as_real(yf.contiguous()), eps)
        else:
            n = x.size()[0]
            r = torch.norm((xf - yf).reshape(n, -1), self.p, 1) / (eps + torch.norm(yf.reshape(n, -1), self.p, 1))
        return self._reduce(r)


def compute_rse(pred, target, eps: float = 1e-8) -> torch.Tensor:
    """eval/eval_autoregressive.py:197-213 — the elementwise map (p-g)^2 / (g^2 + eps); callers that only need its
    mean should use compute_mean_rse, which never materialises it."""
    return (pred - target) ** 2 / (target ** 2 + eps)


def compute_mean_rse(pred, target, eps: float = 1e-8) -> float:
    """eval/eval_autoregressive.py:216-219."""
    n = pred.numel()
    C = int(pred.shape[-1]) if pred.dim() > 0 else 1
    st = R.get_backend().pair_stats(pred, target, 1, n // max(C, 1), C, rse_eps=eps)
    return float(st[0, :, R.S_RSE].sum() / n)


class NRMSEAccumulator:
    """The per-channel (and per-frame) accumulators of eval/eval_nrmse.py:96-209.

    update(pred, target, channel_mask) adds one batch [B, T, *spatial, C] without any host synchronisation; result()
    sums across ranks (one packed all-reduce when torch.distributed is initialised) and returns the reference's dict."""

    def __init__(self, per_timestep: bool = False):
        self.per_timestep = per_timestep
        self.sums = None          # fp64 [T, C, 2]: sum (p-g)^2, sum g^2
        self.count = 0            # elements per (frame, channel)
        self.num_samples = 0
        self.channels: List[int] = []

    def update(self, pred, target, channel_mask):
        B, T = int(pred.shape[0]), int(pred.shape[1])
        G, rows, C = _flat(pred, target, 2)
        if isinstance(pred, Denormalised):      # std and mean applied while reading: one value per (sample, channel), repeated per frame
            affine = (pred.scale.repeat_interleave(T, 0), pred.shift.repeat_interleave(T, 0))
            st = R.get_backend().pair_stats(pred.base.detach(), target, G, rows, C, affine=affine).view(B, T, C, R.NSTAT)
        else:
            st = R.get_backend().pair_stats(pred, target, G, rows, C).view(B, T, C, R.NSTAT)
        s = st[..., [R.S_D2, R.S_G2]].sum(0)                     # [T, C, 2]
        if self.sums is None:
            self.sums = s
        elif self.sums.shape == s.shape:
            self.sums = self.sums + s
        else:
            raise ValueError("NRMSEAccumulator: batches with different numbers of frames or channels")
        self.count += B * rows
        self.num_samples += B
        for c in valid_channels(channel_mask):
            if c not in self.channels:
                self.channels.append(c)

    def result(self) -> dict:
        import torch.distributed as dist
        if self.sums is None:
            return {"per_channel_nrmse": {}, "per_channel_rmse": {}, "per_channel_gt_rms": {}, "overall_nrmse": 0.0, "num_samples": 0}
        T, C = self.sums.shape[0], self.sums.shape[1]
        packed = torch.cat([self.sums.reshape(-1), torch.tensor([float(self.count), float(self.num_samples)], dtype=torch.float64,
                                                                device=self.sums.device)])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        host = packed.cpu()
        sums = host[:-2].view(T, C, 2)
        count_t, num_samples = float(host[-2]), int(host[-1])
        per_nrmse, per_rmse, per_rms = {}, {}, {}
        for ch in sorted(self.channels):
            mse_mean = float(sums[:, ch, 0].sum()) / (count_t * T)
            gt_mean = float(sums[:, ch, 1].sum()) / (count_t * T)
            per_rmse[ch] = mse_mean ** 0.5
            per_rms[ch] = gt_mean ** 0.5
            per_nrmse[ch] = (mse_mean / (gt_mean + 1e-10)) ** 0.5
        out = {"per_channel_nrmse": per_nrmse, "per_channel_rmse": per_rmse, "per_channel_gt_rms": per_rms,
               "overall_nrmse": sum(per_nrmse.values()) / max(len(per_nrmse), 1), "num_samples": num_samples}
        if self.per_timestep:
            out["per_timestep_nrmse"] = {
                ch: [((float(sums[t, ch, 0]) / count_t) / (float(sums[t, ch, 1]) / count_t + 1e-10)) ** 0.5 for t in range(T)]
                for ch in sorted(self.channels)}
        return out


_BIN_CACHE: Dict[tuple, tuple] = {}


def _radial_bins(shape: Sequence[int], device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """The radial bins of the low half-spectrum box as a CSR list: (bin_start[nbins + 1], point_index, nbins), where a
    point is the flat index of a wavenumber in the full spectrum grid and its bin is floor(|k|); wavenumbers whose bin
    lies past the last one are left out, as the reference skips them (eval/metrics.py:271-276, :284-290)."""
    key = (tuple(shape), str(device))
    nbins = min(n // 2 for n in shape)
    if key not in _BIN_CACHE:
        axes = [np.arange(n // 2) for n in shape]
        grids = np.meshgrid(*axes, indexing="ij")
        r2 = sum(g.astype(np.float64) ** 2 for g in grids)
        it = np.floor(np.sqrt(r2)).astype(np.int64).reshape(-1)
        flat = np.zeros_like(grids[0])
        for g, n in zip(grids, shape):
            flat = flat * n + g
        flat = flat.reshape(-1)
        keep = it <= nbins - 1
        it, flat = it[keep], flat[keep]
        order = np.argsort(it, kind="stable")      # members of a bin in grid order
        start = np.zeros(nbins + 1, dtype=np.int64)
        np.cumsum(np.bincount(it, minlength=nbins), out=start[1:])
        _BIN_CACHE[key] = (torch.from_numpy(start.astype(np.int32)).to(device),
                           torch.from_numpy(flat[order].astype(np.int32)).to(device))
    return (*_BIN_CACHE[key], nbins)


def metric_func(pred, target, if_mean: bool = True, Lx: float = 1.0, Ly: float = 1.0, Lz: float = 1.0,
                iLow: int = 4, iHigh: int = 12, initial_step: int = 1):
    """eval/metrics.py:164-308 — (RMSE, nRMSE, conserved-sum error, max error, boundary RMSE, Fourier-band error,
    MPP-nRMSE) of [nb, nx(, ny(, nz)), nt, nc] tensors, the first `initial_step` frames dropped.

    RMSE, nRMSE, MPP-nRMSE, the conserved-sum error and the max error come from one pair_stats pass with
    (frame, channel) as the channel axis; the boundary RMSE runs the same kernel on the boundary slabs.  The
    Fourier-band term uses rocFFT through torch.fft and the band_power kernel for the radial bins."""
    if pred.dim() not in (4, 5, 6):
        raise ValueError("metric_func expects [nb, nx(, ny(, nz)), nt, nc]")
    be = R.get_backend()
    nd = pred.dim() - 3
    nb = int(pred.shape[0])
    sp = [int(v) for v in pred.shape[1:1 + nd]]
    nt_all, nc = int(pred.shape[-2]), int(pred.shape[-1])
    S = 1
    for v in sp:
        S *= v
    t0 = initial_step
    nt = len(range(nt_all)[t0:])

    def stats(p, g, rows):
        st = be.pair_stats(p, g, nb, rows, nt_all * nc).view(nb, nt_all, nc, R.NSTAT)[:, t0:]
        return st.permute(0, 2, 1, 3)                                  # [nb, nc, nt, NSTAT]

    st = stats(pred, target, S)
    f32 = lambda x: x.to(torch.float32)
    err_mean = torch.sqrt(st[..., R.S_D2] / S)                         # :190-194
    err_RMSE = err_mean.mean(0)
    nrm = torch.sqrt(st[..., R.S_G2] / S)
    err_nRMSE = (err_mean / nrm).mean(0)
    err_MPP = torch.sqrt((err_mean ** 2 / (nrm ** 2 + 1e-8)).mean(0))
    err_CSV = torch.sqrt(((st[..., R.S_P] - st[..., R.S_G]) ** 2).mean(0)) / S
    err_Max = st[..., R.S_MAXD].amax(0)

    # boundary RMSE (:228-256)
    def face_sums(axis):
        idx = torch.tensor([0, sp[axis] - 1], device=pred.device)
        p = pred.index_select(1 + axis, idx).contiguous()
        g = target.index_select(1 + axis, idx).contiguous()
        return stats(p, g, 2 * S // sp[axis])[..., R.S_D2]              # [nb, nc, nt]

    if nd == 1:
        err_BD = torch.sqrt(face_sums(0) / 2.0).mean(0)
    elif nd == 2:
        err_BD = torch.sqrt((face_sums(0) + face_sums(1)) / (2 * sp[0] + 2 * sp[1])).mean(0)
    else:
        # the reference folds the channel axis into the sum here and keeps the batch axis (:250-256)
        tot = (face_sums(0) + face_sums(1) + face_sums(2)).sum(1)       # [nb, nt]
        err_BD = torch.sqrt(tot / (2 * sp[0] * sp[1] + 2 * sp[1] * sp[2] + 2 * sp[2] * sp[0]))

    # Fourier-band error (:258-296)
    perm = [0, nd + 2] + list(range(1, nd + 1)) + [nd + 1]
    d = (pred[..., t0:, :] - target[..., t0:, :]).permute(perm)         # [nb, nc, *sp, nt]
    if nd == 1:
        dF = torch.fft.rfft(d, dim=2)
        _err_F = torch.sqrt((dF.abs() ** 2).mean(0)) / sp[0] * Lx       # [nc, nx/2+1, nt]
    else:
        dF = torch.fft.fftn(d, dim=list(range(2, 2 + nd)))
        start, members, nbins = _radial_bins(sp, pred.device)
        acc = be.band_power(dF.reshape(nb * nc, S, nt), start, members, nbins).view(nb, nc, nbins, nt)
        vol = float(S)
        L = Lx * Ly if nd == 2 else Lx * Ly * Lz
        _err_F = torch.sqrt(acc.mean(0)) / vol * L                       # [nc, nbins, nt]
    err_F = torch.zeros(nc, 3, nt, dtype=torch.float32, device=pred.device)
    err_F[:, 0] += _err_F[:, :iLow].mean(1)
    err_F[:, 1] += _err_F[:, iLow:iHigh].mean(1)
    err_F[:, 2] += _err_F[:, iHigh:].mean(1)

    res = (f32(err_RMSE), f32(err_nRMSE), f32(err_CSV), f32(err_Max), f32(err_BD), err_F, f32(err_MPP))
    if if_mean:
        return tuple(r.mean(dim=[0, -1]) for r in res)
    return res
