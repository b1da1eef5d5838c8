"""One-step and autoregressive rollout nRMSE with the trajectories and every intermediate resident on the device.

tools/eval_rollout.py:93-146 moves each predicted frame to the host and reads one Python float per channel and
step; its sliding window is rebuilt with a `cat` per step.  Here the window is a view into one preallocated buffer,
every step's per-channel values stay on the device, and the host reads once at the end.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import reductions as R


def denormalised(model) -> Callable[[torch.Tensor], torch.Tensor]:
    """predict(x) for a model with the reference's calling convention (tools/eval_rollout.py:119-120)."""
    def predict(x):
        out, mean, std = model(x, return_normalized=True)
        return out * std + mean
    return predict


def _step_values(pred_frame, gt_frame, cols: Sequence[int]) -> torch.Tensor:
    """mean over samples of sqrt(sum (p-g)^2 / (sum g^2 + 1e-12)) for the listed channels (:75-90, :123-125): one
    pass of the statistics kernel over the frame pair."""
    M, C = int(pred_frame.shape[0]), int(pred_frame.shape[-1])
    rows = pred_frame[0].numel() // C
    st = R.get_backend().pair_stats(pred_frame, gt_frame, M, rows, C)            # [M, C, NSTAT] fp64
    per = torch.sqrt(st[:, :, R.S_D2] / (st[:, :, R.S_G2] + 1e-12)).to(torch.float32)
    return per.mean(0)[list(cols)]


@torch.no_grad()
def evaluate_rollout(predict: Callable[[torch.Tensor], torch.Tensor], data: torch.Tensor, t_input: int,
                     valid_channels: Sequence[Tuple[int, str]], device=None) -> Tuple[Dict[str, List[float]], Dict[str, List[float]]]:
    """tools/eval_rollout.py:93-146.  data [M, T_total, *spatial, C]; predict maps [M, t_input, *spatial, C] input
    frames to the model's de-normalised output of the same shape, whose last frame is the prediction.
    Returns (one_step, rollout): {channel name: [T_pred] mean nRMSE over the M samples}."""
    if device is not None:
        data = data.to(device)
    M, T_total = int(data.shape[0]), int(data.shape[1])
    T_pred = T_total - t_input
    if T_pred <= 0:
        raise ValueError(f"evaluate_rollout: {T_total} frames leave nothing to predict after {t_input} input frames")
    cols = [int(c) for c, _ in valid_channels]
    names = [n for _, n in valid_channels]
    vals = torch.empty((2, T_pred, len(cols)), dtype=torch.float32, device=data.device)
    for t in range(T_pred):                                   # one-step: ground truth in, one frame out
        out = predict(data[:, t:t + t_input])
        vals[0, t] = _step_values(out[:, -1], data[:, t + t_input], cols)
    window = torch.empty_like(data)                           # frames [t, t + t_input) are the model input of step t
    window[:, :t_input] = data[:, :t_input]
    for t in range(T_pred):                                   # rollout: predictions fed back
        out = predict(window[:, t:t + t_input])
        window[:, t_input + t] = out[:, -1]
        vals[1, t] = _step_values(window[:, t_input + t], data[:, t_input + t], cols)
    host = vals.cpu()                                         # the only host read
    one = {n: host[0, :, i].tolist() for i, n in enumerate(names)}
    roll = {n: host[1, :, i].tolist() for i, n in enumerate(names)}
    return one, roll


def summarize_rollout(one_step: Dict[str, List[float]], rollout: Dict[str, List[float]]) -> Dict[str, object]:
    """tools/eval_rollout.py:212-234 — geometric mean over the steps per channel (with its 1e-12), then the mean of
    the channel values."""
    gm = lambda v: float(np.exp(np.mean(np.log(np.asarray(v, dtype=np.float64) + 1e-12))))
    os_g = {n: gm(v) for n, v in one_step.items()}
    ro_g = {n: gm(v) for n, v in rollout.items()}
    return {"one_step_gmean": os_g, "rollout_gmean": ro_g,
            "one_step_overall": float(np.mean(list(os_g.values()))), "rollout_overall": float(np.mean(list(ro_g.values())))}
