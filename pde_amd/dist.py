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


class PendingScalars:
    """Handle of a packed all-reduce in flight; result() waits for it (on the current stream) and unpacks."""

    def __init__(self, keys, buf, work, scale, values=None):
        self._keys, self._buf, self._work, self._scale, self._values = keys, buf, work, scale, values

    def join(self) -> None:
        """Make the current stream wait for the collective; no other device work (the mean's division is left to result())."""
        if self._work is not None:
            self._work.wait()
            self._work = None

    def result(self) -> Dict[str, torch.Tensor]:
        if self._values is not None:      # single process: nothing was packed
            return self._values
        self.join()
        if self._scale != 1.0:
            self._buf, self._scale = self._buf * self._scale, 1.0
        return {k: self._buf[i] for i, k in enumerate(self._keys)}


def reduce_scalars_async(values: Dict[str, torch.Tensor], op: str = "mean", group=None, force: bool = False) -> PendingScalars:
    """Start ONE all-reduce over a dict of 0-dim tensors and return at once; the collective runs beside whatever is enqueued
    next (the following step), and `.result()` joins it.  With more than one rank the values are copied into the packed buffer first, so the
    caller may overwrite them; a single process gets its own values back, unpacked and uncopied (force=True runs the collective
    even then: a rehearsal of the multi-rank path on one device)."""
    if op not in ("mean", "sum"):
        raise ValueError("op must be 'mean' or 'sum'")
    keys = list(values)
    if not keys:
        return PendingScalars(keys, torch.empty(0, dtype=torch.float64), None, 1.0)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return PendingScalars(keys, None, None, 1.0, {k: v.detach() for k, v in values.items()})
    buf = torch.stack([values[k].detach().reshape(()) for k in keys]).to(torch.float64)     # two small kernels, whatever the count
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return PendingScalars(keys, buf, work, 1.0 / dist.get_world_size(group) if op == "mean" else 1.0)


def pack_scalars(values: Dict[str, torch.Tensor]) -> torch.Tensor:
    """The dict's 0-dim tensors as one fp64 vector, in key order (two small kernels, capturable in a HIP graph)."""
    return torch.stack([v.detach().reshape(()) for v in values.values()]).to(torch.float64)


def pack_scalars_named(values: Dict[str, torch.Tensor]):
    """(vector, names): pack_scalars without any kernel when the scalars already sit side by side in one buffer, as the
    per-equation losses and the total of a residual loss do (one small tensor written by the reduction kernel): the vector
    is then a view of that buffer, in the buffer's order and dtype, and `names` says which entry is which.  The caller
    owns the buffer: an in-place all-reduce on the view rewrites the step's own scalars, which is what logging wants.
    Anything else takes pack_scalars."""
    names = list(values)
    vs = [v.detach() for v in values.values()]
    if vs and all(v.dim() == 0 and v.dtype == vs[0].dtype and v.device == vs[0].device for v in vs):
        store = vs[0].untyped_storage().data_ptr()
        offs = [v.storage_offset() for v in vs]
        lo = min(offs)
        if all(v.untyped_storage().data_ptr() == store for v in vs) and sorted(offs) == list(range(lo, lo + len(vs))):
            order = sorted(range(len(vs)), key=lambda i: offs[i])
            return vs[0].as_strided((len(vs),), (1,), lo), [names[i] for i in order]
    return pack_scalars(values), names


def reduce_packed_async(keys, buf: torch.Tensor, op: str = "mean", group=None, force: bool = False) -> PendingScalars:
    """reduce_scalars_async for a vector made by pack_scalars: the all-reduce runs IN PLACE on `buf`, which the caller
    must leave alone until join() / result().  Nothing else is launched on the current stream, so a step that produces
    `buf` inside a replayed graph pays no eager kernels for its logging."""
    if op not in ("mean", "sum"):
        raise ValueError("op must be 'mean' or 'sum'")
    keys = list(keys)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return PendingScalars(keys, buf, None, 1.0)
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return PendingScalars(keys, buf, work, 1.0 / dist.get_world_size(group) if op == "mean" else 1.0)


def reduce_scalars(values: Dict[str, torch.Tensor], op: str = "mean", group=None) -> Dict[str, torch.Tensor]:
    """All-reduce a dict of 0-dim tensors with a single collective; identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if op not in ("mean", "sum"):
            raise ValueError("op must be 'mean' or 'sum'")
        return {k: v.detach() for k, v in values.items()}
    return reduce_scalars_async(values, op, group).result()
