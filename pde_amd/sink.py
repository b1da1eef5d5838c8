"""One gradient buffer for every loss taken on the same model output.

A trainer adds several losses on `output[B, T, *spatial, C]` (residual loss through `gather_fields`, boundary strips,
wall and seam terms).  Each of their backward passes would produce a full-size gradient (a scatter of the field
gradients; zeros plus a few strips), and autograd would add those tensors: about five passes over a tensor that
only needs to be written once.  `shared_grad(output)` returns the same data with a tag; the losses of this package
that are evaluated on the tagged tensor hand their contribution to the tag instead of materialising it, and the
gradient is then written in one pass (scatter) plus the in-place strip updates.

    out = pde_amd.shared_grad(output)
    pde_loss, losses = compute_pde_loss(out, input_data, pde_loss_fn, nu)
    bc_loss = compute_boundary_loss(out, target, channel_mask)
    (lambda_pde * pde_loss + lambda_bc * bc_loss).backward()

Other consumers of the tagged tensor keep working: their ordinary gradients are added to the shared buffer.
The reference has no counterpart; without the call everything behaves as plain autograd.
"""
from __future__ import annotations

import torch

from . import reductions as R


class GradSink:
    def __init__(self):
        self.gathers = []      # (field gradients, channels, transpose, scale or None)
        self.terms = []        # argument tuples of terms_backward without the gradient buffer


class _Shared(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sink):
        ctx.sink, ctx.shape, ctx.device = sink, tuple(x.shape), x.device
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        sink, be = ctx.sink, R.get_backend()
        gathers, terms = sink.gathers, sink.terms
        sink.gathers, sink.terms = [], []
        if not gathers and not terms:
            return g, None
        buf = None
        for gs, channels, transpose, scale in gathers:      # the first scatter writes every element
            part = be.scatter_field_grads(gs, tuple(ctx.shape), channels, transpose, scale)
            buf = part if buf is None else buf.add_(part)
        if buf is None:
            buf = torch.zeros(ctx.shape, dtype=torch.float32, device=ctx.device)
        for args in terms:
            be.terms_backward(*args, buf)
        if g is not None:
            buf = buf.add_(g.to(buf.dtype))
        return buf, None


def shared_grad(output: torch.Tensor) -> torch.Tensor:
    """Tag `output` (fp32, contiguous; other tensors are returned unchanged and untagged) so that the losses of this package
    accumulate their gradients w.r.t. it in one buffer."""
    if isinstance(output, Denormalised):      # already carries a shared buffer (or was built with share=False on purpose)
        return output
    if not (output.requires_grad and torch.is_grad_enabled()) or output.dtype != torch.float32 or not output.is_contiguous():
        return output
    sink = GradSink()
    out = _Shared.apply(output, sink)
    out._pde_amd_sink = sink
    return out


def sink_of(t: torch.Tensor):
    return getattr(t, "_pde_amd_sink", None)


class Denormalised:
    """The model output `output_norm * std + mean` without the product in memory.

    The trainers de-normalise the whole 18-channel output and take every loss on the result
    (finetune/train_taylor_green_2d_lora_v3.py:500-508); the multiply-add and its autograd backward are four passes
    over the full tensor, while the losses read three channels and a few border strips.  An object of this class
    stands for the product: `gather_fields` / `compute_pde_loss_*`, the boundary, wall and seam losses of this
    package apply std and mean (one value per sample and channel, pretrain/model_v3.py:427-431) inside their kernels,
    to exactly the fp32 values the product would hold, and return gradients w.r.t. output_norm.  It is not a tensor:
    anything else must call `.materialise()`.  With share=True (default) the losses also share one gradient buffer
    (see shared_grad)."""

    def __init__(self, output_norm: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, share: bool = True):
        B, Cn = int(output_norm.shape[0]), int(output_norm.shape[-1])
        full = (B,) + (1,) * (output_norm.dim() - 2) + (Cn,)

        def per_sample_channel(x, what):
            x = torch.as_tensor(x, device=output_norm.device).detach().to(torch.float32)
            try:
                return torch.broadcast_to(x, full).reshape(B, Cn).contiguous()
            except RuntimeError:
                raise ValueError(f"Denormalised: {what} must broadcast to one value per sample and channel {full}, got {tuple(x.shape)}")
        self.scale, self.shift = per_sample_channel(std, "std"), per_sample_channel(mean, "mean")
        self._raw, self._mean, self._std = output_norm, mean, std
        base = output_norm.to(torch.float32).contiguous()
        self.base = shared_grad(base) if share else base

    shape = property(lambda self: self.base.shape)
    device = property(lambda self: self.base.device)
    dtype = property(lambda self: self.base.dtype)
    requires_grad = property(lambda self: self.base.requires_grad)

    def dim(self) -> int:
        return self.base.dim()

    def numel(self) -> int:
        return self.base.numel()

    def materialise(self) -> torch.Tensor:
        """the product as an ordinary (differentiable) tensor, for consumers outside this package"""
        return self._raw * self._std + self._mean


def unwrap(x):
    """(tensor, affine) of a tensor or a Denormalised"""
    if isinstance(x, Denormalised):
        return x.base, (x.scale, x.shift)
    return x, None
