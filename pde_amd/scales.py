"""Residual scales of ground-truth trajectories: the `eq_scales` / `eq_scales_per_t` the loss classes divide by.

The reference produces them offline, in float64 torch on the CPU with a `.item()` per frame and chunk
(tools/compute_gt_scales_per_t.py:66-175 for Gray-Scott, :177-283 for active matter; the test_gt_pde_* tools print
the scalar RMS the configs carry).  Here the residual fields come from the same forward kernels the loss uses in
training (fp32 arithmetic), their squares are added in float64 on the device, and the host reads once, in result().
Near an exact solution the fp32 residual has a floor of about 1e-7 of its largest term; a scale below that floor
is not resolved by this accumulator.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


class ResidualScaleAccumulator:
    """acc = ResidualScaleAccumulator(loss_fn); acc.update(*fields, **params) per batch; acc.result().

    result() -> {'eq_scales': {eq: float}, 'eq_scales_per_t': {eq: float32 tensor [T_res]}} with
    per_t[eq][t] = sqrt(sum over samples and lattice sites of R_eq[:, t]^2 / their number)
    (tools/compute_gt_scales_per_t.py:141-152) and eq_scales[eq] the same root mean square over all frames.
    With torch.distributed initialised the sums of all ranks are added first (one all-reduce)."""

    def __init__(self, loss_fn):
        self.loss_fn = loss_fn
        self.sumsq: Optional[torch.Tensor] = None      # [n_eqs, T_res] float64, device
        self.count = 0                                   # samples x lattice sites per frame, this rank
        self.eqs = None

    @torch.no_grad()
    def update(self, *fields, **params) -> None:
        res = self.loss_fn.residuals(*fields, **params)
        sites = res.pop("__sites_per_frame__")
        eqs = list(res.keys())
        rows = []
        for eq in eqs:
            r = res[eq]
            rows.append(r.to(torch.float64).square().sum(dim=[d for d in range(r.dim()) if d != 1]))
        s = torch.stack(rows)
        if self.sumsq is None:
            self.sumsq, self.eqs = s, eqs
        else:
            if s.shape != self.sumsq.shape:
                raise ValueError(f"batches must share the number of residual frames: {tuple(s.shape)} vs {tuple(self.sumsq.shape)}")
            self.sumsq += s
        self.count += int(sites)

    def result(self) -> Dict[str, dict]:
        if self.sumsq is None:
            raise RuntimeError("ResidualScaleAccumulator.result(): no batch was added")
        packed = torch.cat([self.sumsq.reshape(-1), torch.tensor([float(self.count)], dtype=torch.float64, device=self.sumsq.device)])
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        host = packed.cpu()
        n = float(host[-1])
        sums = host[:-1].view_as(self.sumsq)
        T_res = sums.shape[1]
        return {
            "eq_scales": {eq: float(torch.sqrt(sums[k].sum() / (n * T_res))) for k, eq in enumerate(self.eqs)},
            "eq_scales_per_t": {eq: torch.sqrt(sums[k] / n).to(torch.float32) for k, eq in enumerate(self.eqs)},
        }
