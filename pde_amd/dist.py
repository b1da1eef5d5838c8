"""Data-parallel helpers: batch sharding and ONE packed all-reduce of the step's scalars.

The residual losses never exchange field data between ranks: every residual site depends on its own
sample only, so ranks own disjoint samples and gradients stay local.  What the reference reduces per
step are 1-element tensors, one collective each (accelerator.reduce on bc_loss and pde_loss,
finetune/train_taylor_green_2d_lora_v3.py:522-523; nRMSE sums, eval/eval_nrmse.py:160-174); here they
travel in one buffer through one RCCL all-reduce.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.distributed as dist


def shard_batch(n: int, rank: int, world: int):
    """Contiguous [start, stop) slice of n samples owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reduce_scalars(values: Dict[str, torch.Tensor], op: str = "mean", group=None) -> Dict[str, torch.Tensor]:
    """All-reduce a dict of 0-dim tensors with a single collective; identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {k: v.detach() for k, v in values.items()}
    keys = list(values)
    buf = torch.stack([values[k].detach().to(torch.float64).reshape(()) for k in keys])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if op == "mean":
        buf = buf / dist.get_world_size(group)
    elif op != "sum":
        raise ValueError("op must be 'mean' or 'sum'")
    return {k: buf[i] for i, k in enumerate(keys)}
