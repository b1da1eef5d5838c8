"""Data-parallel helpers: batch sharding and ONE packed all-reduce of the step's scalars.

The residual losses never exchange field data between ranks: every residual site depends on its own
sample only, so ranks own disjoint samples and gradients stay local.  What the reference reduces per
step are 1-element tensors, one collective each (accelerator.reduce on bc_loss and pde_loss,
finetune/train_taylor_green_2d_lora_v3.py:522-523; nRMSE sums, eval/eval_nrmse.py:160-174); here they
travel in one buffer through one RCCL all-reduce.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

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


class ScalarWindow:
    """The scalars of `steps` consecutive steps in one buffer, reduced over the ranks with ONE all-reduce per window.

    The reference reduces its logging scalars with a collective each, every step (accelerator.reduce on bc_loss and pde_loss,
    finetune/train_taylor_green_2d_lora_v3.py:522-523).  A 0.13 ms step cannot afford even one: on MI355X the stream
    hand-over around a small asynchronous all-reduce takes about 21 us of the step's timeline, whatever the message size.
    Here step i leaves its scalars in `row(i)` (hand that row to the loss module as `scalars_out`, or copy into it), and
    `after(i)`, called once step i is enqueued, starts the all-reduce of a window when step i completes it; the collective
    runs in place, beside the steps of the next window.  Two windows alternate, and `before(i)`, called ahead of step i,
    joins the collective that still reads the rows the coming window overwrites and keeps its result: `drain()` returns
    the windows finished since the last call, oldest first, as (first_step, {name: tensor[steps of that window]}).
    `flush()` reduces a partly filled last window.  A single process runs no collective (force=True does, for rehearsal).
    """

    def __init__(self, names: Sequence[str], steps: int, device, dtype=torch.float32, op: str = "mean", group=None, force: bool = False):
        if op not in ("mean", "sum"):
            raise ValueError("op must be 'mean' or 'sum'")
        if steps < 1 or not names:
            raise ValueError("ScalarWindow needs at least one name and one step per window")
        self.names, self.steps, self.op, self.group, self.force = list(names), int(steps), op, group, force
        self.buf = torch.zeros((2, self.steps, len(self.names)), dtype=dtype, device=device)
        self._pending: List[Optional[tuple]] = [None, None]      # per half: (first step, rows filled, handle)
        self._done: List[tuple] = []
        self._filled = 0

    def row(self, i: int) -> torch.Tensor:
        """Where step i leaves its scalars: a contiguous vector of len(names) elements."""
        return self.buf[(i // self.steps) % 2, i % self.steps]

    def before(self, i: int) -> None:
        """Ahead of step i: when it opens a window, join the collective that still owns the window's rows."""
        if i % self.steps == 0:
            self._finish((i // self.steps) % 2)

    def after(self, i: int) -> None:
        """Once step i is enqueued: when it completes a window, start that window's all-reduce."""
        self._filled = i % self.steps + 1
        if self._filled == self.steps:
            self._start(i - self.steps + 1, self.steps)
            self._filled = 0

    def flush(self, next_step: int) -> None:
        """Reduce the rows of an unfinished window (next_step = the index the next step would have had) and join everything."""
        if self._filled:
            self._start(next_step - self._filled, self._filled)
            self._filled = 0
        # older window first: by the first step each pending half recorded (which half is the older one depends on
        # whether the last window was a whole one)
        for h in sorted((h for h in (0, 1) if self._pending[h] is not None), key=lambda h: self._pending[h][0]):
            self._finish(h)

    def drain(self) -> List[tuple]:
        out, self._done = self._done, []
        return out

    def _start(self, first: int, rows: int) -> None:
        h = (first // self.steps) % 2
        self._finish(h)
        vec = self.buf[h, :rows].reshape(-1)      # contiguous: a view
        handle = reduce_packed_async(range(vec.numel()), vec, self.op, self.group, self.force)
        self._pending[h] = (first, rows, handle)

    def _finish(self, h: int) -> None:
        if self._pending[h] is None:
            return
        first, rows, handle = self._pending[h]
        self._pending[h] = None
        handle.join()
        vals = self.buf[h, :rows]
        vals = vals * handle._scale if handle._scale != 1.0 else vals.clone()      # the rows are about to be overwritten
        self._done.append((first, {n: vals[:, j] for j, n in enumerate(self.names)}))


def reduce_scalars(values: Dict[str, torch.Tensor], op: str = "mean", group=None) -> Dict[str, torch.Tensor]:
    """All-reduce a dict of 0-dim tensors with a single collective; identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if op not in ("mean", "sum"):
            raise ValueError("op must be 'mean' or 'sum'")
        return {k: v.detach() for k, v in values.items()}
    return reduce_scalars_async(values, op, group).result()
