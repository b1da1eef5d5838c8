"""CPU restatement of the reference's boundary / wall / seam losses, small metrics and trainer glue.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else) as the checker for the HIP path.  Plain torch on
whatever dtype it is given, written from the cited lines; pinned by tests/golden/glue.npz and metric_func.npz,
which tools/make_golden.py produced by executing the reference's own functions.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _valid(mask):
    m = mask if mask.dim() == 1 else mask[0]
    return torch.nonzero(m > 0).flatten()


def _rmse_of(pairs):
    """sqrt(mean over the concatenation of all slices of (p - g)^2 + 1e-8)"""
    d = torch.cat([(p - g).reshape(-1) for p, g in pairs])
    return torch.sqrt((d * d).mean() + 1e-8)


def boundary_1d(pred, target, mask, w=1):
    # finetune/train_burgers_1d_lora_v3.py:128-139
    ch = _valid(mask)
    cut = lambda x, s: x[:, :, s][..., ch]
    return _rmse_of([(cut(pred, s), cut(target, s)) for s in (slice(None, w), slice(-w, None))])


def boundary_2d_strips(pred, target, mask, w=1):
    # finetune/train_taylor_green_2d_lora_v3.py:146-166
    ch = _valid(mask)
    boxes = [(slice(None, w), slice(None)), (slice(-w, None), slice(None)), (slice(w, -w), slice(None, w)), (slice(w, -w), slice(-w, None))]
    return _rmse_of([(pred[:, :, a, b][..., ch], target[:, :, a, b][..., ch]) for a, b in boxes])


def boundary_2d_edges(pred, target, mask):
    # finetune/train_shear_flow_lora_v3.py:127-146
    ch = _valid(mask)
    pick = [lambda x: x[:, :, :, 0], lambda x: x[:, :, :, -1], lambda x: x[:, :, 0], lambda x: x[:, :, -1]]
    return _rmse_of([(f(pred)[..., ch], f(target)[..., ch]) for f in pick])


def boundary_3d_faces(pred, target, mask):
    # finetune/train_heat_3d_lora_v3.py:124-141; the z faces take rows `ch` of the y axis and every channel (:133-134)
    ch = _valid(mask)
    pairs = []
    for x in (0, -1):
        pairs.append((pred[:, :, x][..., ch], target[:, :, x][..., ch]))
    for y in (0, -1):
        pairs.append((pred[:, :, :, y][..., ch], target[:, :, :, y][..., ch]))
    for z in (0, -1):
        pairs.append((pred[:, :, :, :, z][:, :, :, ch], target[:, :, :, :, z][:, :, :, ch]))
    return _rmse_of(pairs)


def boundary_3d_shells(pred, target, mask, w=1):
    # finetune/train_advection_3d_lora_v3.py:127-159
    ch = _valid(mask)
    lo, hi, mid, al = slice(None, w), slice(-w, None), slice(w, -w), slice(None)
    boxes = [(lo, al, al), (hi, al, al), (mid, lo, al), (mid, hi, al), (mid, mid, lo), (mid, mid, hi)]
    return _rmse_of([(pred[:, :, a, b, c][..., ch], target[:, :, a, b, c][..., ch]) for a, b, c in boxes])


def boundary_3d_xy_strips(pred, target, mask, w=1):
    # finetune/train_burgers_3d_apebench_lora.py:128-150 on [B, T, X, Y, Z, C]: the 2-D strip slicing over X and Y, with the
    # channel list applied to dim 4 (the Z axis) and every channel kept
    ch = _valid(mask)
    boxes = [(slice(None, w), slice(None)), (slice(-w, None), slice(None)), (slice(w, -w), slice(None, w)), (slice(w, -w), slice(-w, None))]
    return _rmse_of([(pred[:, :, a, b][:, :, :, :, ch], target[:, :, a, b][:, :, :, :, ch]) for a, b in boxes])


def noslip(output, ch_vx=0, ch_vy=1):
    # finetune/train_ns_incom_lora_v3.py:159-176
    tot = 0.0
    for c in (ch_vx, ch_vy):
        f = output[..., c]
        wall = torch.cat([f[:, :, 0].reshape(-1), f[:, :, -1].reshape(-1), f[:, :, :, 0].reshape(-1), f[:, :, :, -1].reshape(-1)])
        tot = tot + (wall * wall).mean()
    return tot


def patch_seams(output, mask, P=16):
    # finetune/train_rayleigh_benard_lora_v3.py:125-135
    x = output[..., _valid(mask)]
    curv_w = x[:, :, :, P - 2:-2:P] - 2 * x[:, :, :, P - 1:-1:P] + x[:, :, :, P::P]
    curv_h = x[:, :, P - 2:-2:P] - 2 * x[:, :, P - 1:-1:P] + x[:, :, P::P]
    return (curv_w ** 2).mean() + (curv_h ** 2).mean()


def vrmse(output, target, mask):
    # finetune/train_taylor_green_2d_lora_v3.py:214-232
    ch = _valid(mask)
    d = output[..., ch] - target[..., ch]
    rmse = torch.sqrt((d * d).mean() + 1e-8)
    acc = 0.0
    for c in ch.tolist():
        g = target[..., c]
        mse_c = ((output[..., c] - g) ** 2).mean()
        var_c = ((g - g.mean()) ** 2).mean()
        acc = acc + torch.sqrt(mse_c / (var_c + 1e-8))
    return acc / len(ch), rmse


def nrmse_per_sample(pred, gt):
    # tools/eval_rollout.py:67-72
    M = pred.shape[0]
    p, g = pred.reshape(M, -1), gt.reshape(M, -1)
    return torch.sqrt(((p - g) ** 2).sum(-1) / ((g * g).sum(-1) + 1e-12))


def nrmse_per_channel(pred, gt, named):
    # tools/eval_rollout.py:81-90
    return {name: nrmse_per_sample(pred[..., c], gt[..., c]) for c, name in named}


def rse(pred, target, eps=1e-8):
    # eval/eval_autoregressive.py:211-213
    return (pred - target) ** 2 / (target ** 2 + eps)


def eval_nrmse(batches, per_timestep=False):
    """eval/eval_nrmse.py:96-209 on a list of (pred, target, mask) batches, single rank."""
    acc, acc_t, n = {}, {}, 0
    for pred, target, mask in batches:
        n += pred.shape[0]
        for c in _valid(mask).tolist():
            d2, g2 = (pred[..., c] - target[..., c]) ** 2, target[..., c] ** 2
            a = acc.setdefault(c, [0.0, 0.0, 0])
            a[0] += float(d2.sum()); a[1] += float(g2.sum()); a[2] += d2.numel()
            for t in range(pred.shape[1]):
                b = acc_t.setdefault(c, {}).setdefault(t, [0.0, 0.0, 0])
                b[0] += float(d2[:, t].sum()); b[1] += float(g2[:, t].sum()); b[2] += d2[:, t].numel()
    out = {"per_channel_nrmse": {}, "per_channel_rmse": {}, "per_channel_gt_rms": {}, "num_samples": n}
    for c in sorted(acc):
        m, g = acc[c][0] / acc[c][2], acc[c][1] / acc[c][2]
        out["per_channel_rmse"][c], out["per_channel_gt_rms"][c] = m ** 0.5, g ** 0.5
        out["per_channel_nrmse"][c] = (m / (g + 1e-10)) ** 0.5
    out["overall_nrmse"] = sum(out["per_channel_nrmse"].values()) / max(len(out["per_channel_nrmse"]), 1)
    if per_timestep:
        out["per_timestep_nrmse"] = {c: [((acc_t[c][t][0] / acc_t[c][t][2]) / (acc_t[c][t][1] / acc_t[c][t][2] + 1e-10)) ** 0.5
                                         for t in sorted(acc_t[c])] for c in sorted(acc_t)}
    return out


def gather(output, input_data, channels, transpose=False):
    # finetune/train_taylor_green_2d_lora_v3.py:185-195, train_rayleigh_benard_lora_v3.py:209-211
    fields = []
    for c in channels:
        f = torch.cat([input_data[:, 0:1, ..., c].float(), output[..., c].float()], dim=1)
        fields.append(f.transpose(-1, -2) if transpose else f)
    return fields


def metric_func(pred, target, if_mean=True, Lx=1.0, Ly=1.0, Lz=1.0, iLow=4, iHigh=12, initial_step=1):
    """eval/metrics.py:174-308 restated on [nb, *space, nt, nc] -> (RMSE, nRMSE, CSV, Max, BD, F, MPP-nRMSE)."""
    p = pred[..., initial_step:, :].movedim(-1, 1)       # [nb, nc, *space, nt]
    g = target[..., initial_step:, :].movedim(-1, 1)
    nb, nc, nt = p.shape[0], p.shape[1], p.shape[-1]
    sp = list(p.shape[2:-1])
    nd, S = len(sp), math.prod(sp)
    pf, gf = p.reshape(nb, nc, S, nt), g.reshape(nb, nc, S, nt)
    e = torch.sqrt(((pf - gf) ** 2).mean(2))
    nrm = torch.sqrt((gf ** 2).mean(2))
    rmse, nrmse = e.mean(0), (e / nrm).mean(0)
    mpp = torch.sqrt((e ** 2 / (nrm ** 2 + 1e-8)).mean(0))
    csv = torch.sqrt(((pf.sum(2) - gf.sum(2)) ** 2).mean(0)) / S
    mx = (pf - gf).abs().amax(2).amax(0)
    d2 = (p - g) ** 2
    if nd == 1:
        bd = torch.sqrt((d2[:, :, 0] + d2[:, :, -1]) / 2.0).mean(0)
    elif nd == 2:
        bx, by = d2[:, :, 0] + d2[:, :, -1], d2[:, :, :, 0] + d2[:, :, :, -1]
        bd = torch.sqrt((bx.sum(-2) + by.sum(-2)) / (2 * sp[0] + 2 * sp[1])).mean(0)
    else:
        bx, by, bz = d2[:, :, 0] + d2[:, :, -1], d2[:, :, :, 0] + d2[:, :, :, -1], d2[:, :, :, :, 0] + d2[:, :, :, :, -1]
        tot = sum(b.reshape(nb, -1, nt).sum(-2) for b in (bx, by, bz))
        bd = torch.sqrt(tot / (2 * sp[0] * sp[1] + 2 * sp[1] * sp[2] + 2 * sp[2] * sp[0]))
    if nd == 1:
        F = torch.sqrt((torch.fft.rfft(p - g, dim=2).abs() ** 2).mean(0)) / sp[0] * Lx
    else:
        e2 = torch.fft.fftn(p - g, dim=list(range(2, 2 + nd))).abs() ** 2
        nbin = min(n // 2 for n in sp)
        acc = torch.zeros(nb, nc, nbin, nt, dtype=e2.dtype)
        import itertools
        for idx in itertools.product(*[range(n // 2) for n in sp]):
            it = math.floor(math.sqrt(sum(i * i for i in idx)))
            if it <= nbin - 1:
                acc[:, :, it] += e2[(slice(None), slice(None)) + idx]
        F = torch.sqrt(acc.mean(0)) / S * (Lx * Ly if nd == 2 else Lx * Ly * Lz)
    band = torch.stack([F[:, :iLow].mean(1), F[:, iLow:iHigh].mean(1), F[:, iHigh:].mean(1)], dim=1)
    res = (rmse, nrmse, csv, mx, bd, band, mpp)
    return tuple(r.mean(dim=[0, -1]) for r in res) if if_mean else res


def pdebench_nrmse(pred, target):
    """eval/eval_omni.py:243-278 — per channel sqrt(mean(((p - g) / (std(g) + 1e-8))^2)), std over everything (unbiased);
    returns (mean over channels, list)."""
    per = []
    for c in range(pred.shape[-1]):
        p, g = pred[..., c], target[..., c]
        sigma = g.std() + 1e-8
        per.append(float(torch.sqrt(torch.mean(((p - g) / sigma) ** 2))))
    return float(np.mean(per)), per


def lp_loss(x, y, p=2, reduction="mean", eps=1e-20):
    """eval/metrics.py:599-623 — per example ||x - y||_p / (eps + ||y||_p) over everything but axis 0."""
    n = x.shape[0]
    d = torch.norm(x.reshape(n, -1) - y.reshape(n, -1), p, 1)
    r = d / (eps + torch.norm(y.reshape(n, -1), p, 1))
    return r.mean() if reduction == "mean" else r.sum() if reduction == "sum" else r


def fft_lp_loss(x, y, p=2, reduction="mean", flow=None, fhigh=None, eps=1e-20):
    """eval/metrics.py:626-678 — the same ratio on the n-dimensional Fourier coefficients, optionally on the band
    flow <= k < fhigh of every transformed axis."""
    n = x.shape[0]
    dims = list(range(1, x.dim()))
    xf, yf = torch.fft.fftn(x, dim=dims), torch.fft.fftn(y, dim=dims)
    lo = 0 if flow is None else flow
    hi = max(xf.shape[1:]) if fhigh is None else fhigh
    if 1 <= len(dims) <= 4:
        sl = (slice(None),) + (slice(lo, hi),) * len(dims)
        xf, yf = xf[sl], yf[sl]
    d = torch.norm((xf - yf).reshape(n, -1), p, 1)
    r = d / (eps + torch.norm(yf.reshape(n, -1), p, 1))
    return r.mean() if reduction == "mean" else r.sum() if reduction == "sum" else r


def rollout_evaluate(predict, data, t_input, valid_channels):
    """tools/eval_rollout.py:93-146 as written there: per step the last output frame against the next true frame,
    per-channel per-sample nRMSE, mean over the samples; the rollout's window is rebuilt with a cat per step."""
    T_pred = data.shape[1] - t_input
    one = {n: [] for _, n in valid_channels}
    for t in range(T_pred):
        out = predict(data[:, t:t + t_input])
        for n, v in nrmse_per_channel(out[:, -1], data[:, t + t_input], valid_channels).items():
            one[n].append(float(v.mean()))
    roll = {n: [] for _, n in valid_channels}
    cur = data[:, :t_input].clone()
    for t in range(T_pred):
        pred = predict(cur)[:, -1]
        for n, v in nrmse_per_channel(pred, data[:, t_input + t], valid_channels).items():
            roll[n].append(float(v.mean()))
        cur = torch.cat([cur[:, 1:], pred.unsqueeze(1)], dim=1)
    return one, roll


def rollout_summary(one, roll):
    """tools/eval_rollout.py:212-234"""
    g = lambda v: float(np.exp(np.mean(np.log(np.asarray(v) + 1e-12))))
    og, rg = {n: g(v) for n, v in one.items()}, {n: g(v) for n, v in roll.items()}
    return {"one_step_gmean": og, "rollout_gmean": rg,
            "one_step_overall": float(np.mean(list(og.values()))), "rollout_overall": float(np.mean(list(rg.values())))}


def gt_scales_per_t(residual_batches, sites_per_sample):
    """tools/compute_gt_scales_per_t.py:109-152 — residual_batches: per chunk a list of residual tensors [n, T_res, ...]
    (one per equation, float64); sums of squares per frame over chunks, samples and space, divided by the number of
    samples x lattice sites, square root; returned as float32 like the saved file."""
    n_eq = len(residual_batches[0])
    T_res = residual_batches[0][0].shape[1]
    sums = torch.zeros(n_eq, T_res, dtype=torch.float64)
    n = 0
    for batch in residual_batches:
        n += batch[0].shape[0]
        for k, r in enumerate(batch):
            for t in range(T_res):
                sums[k, t] += torch.sum(r[:, t].double() ** 2).item()
    return [torch.sqrt(sums[k] / (n * sites_per_sample)).float() for k in range(n_eq)]
