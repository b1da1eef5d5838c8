"""Timing of the metric / boundary / glue kernels at rollout-evaluation sizes (HIP events, algorithmic bytes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pde_amd
from pde_amd import reductions as R

dev = torch.device("cuda:0")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


M, H, W, C = 64, 512, 128, 18
p = torch.randn(M, H, W, C, device=dev)
g = torch.randn(M, H, W, C, device=dev)
be = R.get_backend()
nbytes = 8 * p.numel()
for groups in (1, M):
    ms = timeit(lambda: be.pair_stats(p, g, groups, p.numel() // (groups * C), C))
    print(f"pair_stats [{M},{H},{W},{C}] groups={groups}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s algorithmic (8 B/element)")
mask = torch.zeros(C, device=dev); mask[[0, 1, 3]] = 1
ms = timeit(lambda: pde_amd.compute_vrmse(p, g, mask))
print(f"compute_vrmse incl. host epilogue: {ms:.3f} ms")
ms = timeit(lambda: torch.mean((p[..., [0, 1, 3]] - g[..., [0, 1, 3]]) ** 2))
print(f"torch mse on 3 gathered channels (one of the reference's 7 reductions): {ms:.3f} ms")

B, T = 16, 15
out = torch.randn(B, T, 256, 256, C, device=dev, requires_grad=True)
inp = torch.randn(B, 1, 256, 256, C, device=dev)
ms = timeit(lambda: pde_amd.gather_fields(out, inp, (0, 1, 15)))
rd = 4 * (out.numel() + inp.numel())
print(f"gather_fields 3 of 18 channels [{B},{T},256,256,18]: {ms:.3f} ms  ({rd / ms / 1e6:.0f} GB/s of full-row reads)")
fields = pde_amd.gather_fields(out, inp, (0, 1, 15))
gr = [torch.randn_like(f) for f in fields]
ms = timeit(lambda: R.get_backend().scatter_field_grads(gr, tuple(out.shape), (0, 1, 15), False))
print(f"scatter_field_grads: {ms:.3f} ms  ({4 * out.numel() / ms / 1e6:.0f} GB/s of gradient writes)")
tgt = torch.randn(B, T, 256, 256, C, device=dev)
m2 = torch.zeros(B, C, device=dev); m2[:, [0, 1, 15]] = 1
o2 = out.detach().requires_grad_(True)
def bc():
    o2.grad = None
    pde_amd.compute_boundary_loss(o2, tgt, m2, 1).backward()
ms = timeit(bc)
print(f"compute_boundary_loss fwd+bwd (includes zero-filling the {4 * out.numel() / 1e6:.0f} MB gradient): {ms:.3f} ms")

# the physics losses of one training step on the model output: residual loss through the channel gather + boundary strips
fn = pde_amd.TaylorGreen2DPDELoss(nx=256, ny=256, dt=0.01).to(dev)
nu = torch.rand(B, device=dev) * 0.09 + 0.01
def train_losses(shared):
    o2.grad = None
    o = pde_amd.shared_grad(o2) if shared else o2
    pde, _ = pde_amd.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)
    bc_l = pde_amd.compute_boundary_loss(o, tgt, m2, 1)
    (1e-3 * pde + 0.1 * bc_l).backward()
for shared, mode in ((False, "auto"), (True, "auto"), (True, "fused")):
    fn.backward_mode, fn.upstream_hint = mode, 1e-3
    ms = timeit(lambda: train_losses(shared))
    print(f"step losses (gather + TG residual + boundary strips) fwd+bwd, {'one shared gradient buffer' if shared else 'separate gradients added by autograd'}, backward_mode={mode}: {ms:.3f} ms")

# the same step starting from the model's normalised output, as the trainers do (output = output_norm * std + mean first)
mean_bc = torch.randn(B, 1, 1, 1, 18, device=dev)
std_bc = torch.rand(B, 1, 1, 1, 18, device=dev) + 0.5
def train_losses_from_norm(lazy):
    o2.grad = None
    o = pde_amd.Denormalised(o2, mean_bc, std_bc) if lazy else pde_amd.shared_grad(o2 * std_bc + mean_bc)
    pde, _ = pde_amd.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)
    bc_l = pde_amd.compute_boundary_loss(o, tgt, m2, 1)
    (1e-3 * pde + 0.1 * bc_l).backward()
fn.backward_mode, fn.upstream_hint = "fused", 1e-3
for lazy in (False, True):
    ms = timeit(lambda: train_losses_from_norm(lazy))
    print(f"step losses from output_norm, {'Denormalised (std, mean applied inside the kernels)' if lazy else 'product materialised by torch, shared gradient buffer'}, backward_mode=fused: {ms:.3f} ms")

# a validation batch from the normalised output (finetune/train_taylor_green_2d_lora_v3.py:263-269): boundary RMSE, residual loss, VRMSE
def validate_from_norm(lazy):
    with torch.no_grad():
        o = pde_amd.Denormalised(o2, mean_bc, std_bc) if lazy else o2 * std_bc + mean_bc
        bc_l = pde_amd.compute_boundary_loss(o, tgt, m2, 1)
        pde, _ = pde_amd.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)
        v, r = pde_amd.compute_vrmse(o, tgt, m2)
for lazy in (False, True):
    ms = timeit(lambda: validate_from_norm(lazy))
    print(f"validation scalars from output_norm, {'Denormalised' if lazy else 'product materialised by torch'}: {ms:.3f} ms")

# metric_func's Fourier-band error: rocFFT + the binning kernel, against the same bins through index_add
from pde_amd import metrics as PM
mp = torch.randn(8, 128, 128, 21, 4, device=dev)
mg = torch.randn(8, 128, 128, 21, 4, device=dev)
ms = timeit(lambda: PM.metric_func(mp, mg, if_mean=True))
print(f"metric_func [8,128,128,21,4] (all seven terms): {ms:.3f} ms")
dF = torch.fft.fftn((mp - mg)[..., 1:, :].permute(0, 4, 1, 2, 3), dim=[2, 3]).reshape(32, 128 * 128, 20).contiguous()
start, members, nbins = PM._radial_bins([128, 128], dev)
ms = timeit(lambda: PM.R.get_backend().band_power(dF, start, members, nbins))
binid = torch.repeat_interleave(torch.arange(nbins, device=dev), (start[1:] - start[:-1]).long())
def via_index_add():
    e2 = dF.abs() ** 2
    acc = torch.zeros(32, nbins, 20, device=dev)
    acc.index_add_(1, binid, e2[:, members.long()])
    return acc
ms2 = timeit(via_index_add)
print(f"radial binning of |F|^2, [32,128x128,20] -> [32,{nbins},20]: band_power kernel {ms:.3f} ms, abs^2 + gather + index_add {ms2:.3f} ms")
