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
        self.gathers = []      # (field gradients, channels, transpose)
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
        for gs, channels, transpose in gathers:      # the first scatter writes every element
            part = be.scatter_field_grads(gs, tuple(ctx.shape), channels, transpose)
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
    if not (output.requires_grad and torch.is_grad_enabled()) or output.dtype != torch.float32 or not output.is_contiguous():
        return output
    sink = GradSink()
    out = _Shared.apply(output, sink)
    out._pde_amd_sink = sink
    return out


def sink_of(t: torch.Tensor):
    return getattr(t, "_pde_amd_sink", None)
