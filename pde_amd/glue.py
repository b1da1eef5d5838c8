"""Trainer-side glue: from the model's channel-last tensors to a residual loss and back.

`gather_fields` is the fused form of the reference's per-trainer preamble

    t0_u = input_data[:, 0:1, ..., CH].float();  u = torch.cat([t0_u, output[..., CH].float()], dim=1)

(finetune/train_taylor_green_2d_lora_v3.py:183-195 and its siblings): one launch writes every field the loss
needs, contiguous, with frame 0 of the input prepended; its backward is one launch that writes the whole
gradient w.r.t. `output`.  The `compute_pde_loss_*` functions keep the reference's signatures, the registry
mirrors finetune/train_fullft_physonly.py:65-223.
"""
from __future__ import annotations

import inspect
from typing import Dict, Optional, Sequence

import torch

from . import losses as L
from . import reductions as R
from .sink import sink_of, unwrap

# channel layout of the 18-channel tensors, as the trainers name them
CH_VX, CH_VY, CH_VZ = 0, 1, 2                       # finetune/train_burgers_3d_apebench_lora.py:76-78
CH_BUOY = 3                                          # finetune/train_rayleigh_benard_lora_v3.py:78
CH_A, CH_B = 5, 6                                    # finetune/train_gray_scott_lora_v3.py:70-71
CH_TRACER, CH_PRESS, CH_TEMP = 14, 15, 17            # finetune/train_shear_flow_lora_v3.py:73-74, train_heat_3d_lora_v3.py:74


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, input_data, channels, transpose, sink, affine):
        ctx.channels, ctx.transpose, ctx.shape, ctx.sink = tuple(channels), bool(transpose), tuple(output.shape), sink
        ctx.scale = affine[0] if affine is not None else None
        return tuple(R.get_backend().gather_fields(output, input_data, channels, transpose, affine))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g.contiguous() if g is not None else None for g in grads]
        ref = next(g for g in gs if g is not None)
        gs = [g if g is not None else torch.zeros_like(ref) for g in gs]
        if ctx.sink is not None:      # the shared buffer of `output` is written once, after every loss has reported
            ctx.sink.gathers.append((gs, ctx.channels, ctx.transpose, ctx.scale))
            return None, None, None, None, None, None
        return R.get_backend().scatter_field_grads(gs, ctx.shape, ctx.channels, ctx.transpose, ctx.scale), None, None, None, None, None


def gather_fields(output: torch.Tensor, input_data: torch.Tensor, channels: Sequence[int], transpose: bool = False):
    """[B, T_out, *spatial, C] and [B, T_in, *spatial, C] -> one contiguous fp32 [B, T_out + 1, *spatial] per channel, frame 0
    taken from `input_data[:, 0]`.  transpose=True swaps the last two spatial axes (the Rayleigh-Benard trainer's
    `.transpose(-1, -2)`, finetune/train_rayleigh_benard_lora_v3.py:209-211).  Differentiable w.r.t. `output` only, as the
    trainers use it (their inputs carry no gradient)."""
    output, affine = unwrap(output)      # a Denormalised output: std and mean are applied while gathering
    if output.shape[0] != input_data.shape[0] or output.shape[2:] != input_data.shape[2:]:
        raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Got {tuple(input_data.shape)} and {tuple(output.shape)}")
    out = output.float().contiguous()
    inp = input_data.detach().float().contiguous()
    return list(_Gather.apply(out, inp, tuple(int(c) for c in channels), transpose, sink_of(out), affine))


def compute_pde_loss_taylor_green_2d(output, input_data, pde_loss_fn, nu, t_offset=None):
    """finetune/train_taylor_green_2d_lora_v3.py:169-201"""
    u, v, p = gather_fields(output, input_data, (CH_VX, CH_VY, CH_PRESS))
    return pde_loss_fn(u, v, p, nu=nu.float(), t_offset=t_offset)


def compute_pde_loss_rayleigh_benard(output, input_data, pde_loss_fn, nu, t_offset=None):
    """finetune/train_rayleigh_benard_lora_v3.py:178-216 — spatial transpose, kappa = nu."""
    u, v, b = gather_fields(output, input_data, (CH_VX, CH_VY, CH_BUOY), transpose=True)
    return pde_loss_fn(u, v, b, kappa=nu.float(), t_offset=t_offset)


def compute_pde_loss_shear_flow(output, input_data, pde_loss_fn, t_offset=None):
    """finetune/train_shear_flow_lora_v3.py:149-189"""
    u, v, p, s = gather_fields(output, input_data, (CH_VX, CH_VY, CH_PRESS, CH_TRACER))
    return pde_loss_fn(u, v, p, s, t_offset=t_offset)


def compute_pde_loss_burgers_1d(output, input_data, pde_loss_fn, nu):
    """finetune/train_burgers_1d_lora_v3.py:142-172"""
    (u,) = gather_fields(output, input_data, (CH_VX,))
    return pde_loss_fn(u, nu=nu.float())


def compute_pde_loss_gray_scott(output, input_data, pde_loss_fn):
    """finetune/train_gray_scott_lora_v3.py:153-185"""
    A, B = gather_fields(output, input_data, (CH_A, CH_B))
    return pde_loss_fn(A, B)


def compute_pde_loss_burgers_3d(output, input_data, pde_loss_fn):
    """finetune/train_burgers_3d_apebench_lora.py:153-177"""
    u, v, w = gather_fields(output, input_data, (CH_VX, CH_VY, CH_VZ))
    return pde_loss_fn(u, v, w)


def compute_pde_loss_heat_3d(output, input_data, pde_loss_fn, alpha):
    """finetune/train_heat_3d_lora_v3.py:144-172"""
    (u,) = gather_fields(output, input_data, (CH_TEMP,))
    return pde_loss_fn(u, alpha=alpha.float())


# ---------------------------------------------------------------------------------------------------
# registry (finetune/train_fullft_physonly.py:65-85, :148-223)
# ---------------------------------------------------------------------------------------------------
PDE_LOSS_REGISTRY: Dict[str, type] = {
    "taylor_green_2d": L.TaylorGreen2DPDELoss,
    "kp2_2d": L.KP2DPDELoss,
    "wave_2d": L.Wave2DPDELoss,
    "advdiff_2d": L.AdvDiff2DPDELoss,
    "burgers_2d_ch": L.Burgers2DCHPDELoss,
}


def create_pde_loss(config: dict, device):
    """finetune/train_fullft_physonly.py:148-169 — class from config['pde_type'], kwargs = the entries of config['physics'] the
    constructor accepts, plus eq_scales / eq_weights."""
    pde_type = config.get("pde_type", "")
    physics = config.get("physics", {})
    if pde_type not in PDE_LOSS_REGISTRY:
        raise ValueError(f"Unknown pde_type: {pde_type}. Available: {list(PDE_LOSS_REGISTRY.keys())}")
    cls = PDE_LOSS_REGISTRY[pde_type]
    meta_keys = {"eq_scales", "eq_weights", "eq_scales_per_t_path", "pde_type"}
    valid = set(inspect.signature(cls.__init__).parameters.keys()) - {"self"}
    kwargs = {k: v for k, v in physics.items() if k in valid and k not in meta_keys}
    kwargs["eq_scales"] = physics.get("eq_scales", None)
    kwargs["eq_weights"] = physics.get("eq_weights", None)
    return cls(**kwargs).to(device)


def compute_pde_loss_generic(output, input_data, pde_loss_fn, config: dict, batch: dict):
    """finetune/train_fullft_physonly.py:172-223"""
    pde_type = config.get("pde_type", "")
    if pde_type == "taylor_green_2d":
        u, v, p = gather_fields(output, input_data, (0, 1, 15))
        nu = batch.get("nu", None)
        if nu is not None:
            nu = nu.to(output.device)
        return pde_loss_fn(u, v, p, nu=nu)
    if pde_type in ("kp2_2d", "advdiff_2d"):
        (u,) = gather_fields(output, input_data, (3,))
        return pde_loss_fn(u)
    if pde_type == "wave_2d":
        u, w = gather_fields(output, input_data, (3, 4))
        return pde_loss_fn(u, w, batch["nu"].to(output.device))
    if pde_type == "burgers_2d_ch":
        u, v = gather_fields(output, input_data, (0, 1))
        return pde_loss_fn(u, v, nu=batch["nu"].to(output.device))
    raise ValueError(f"Unknown pde_type: {pde_type}")
