"""Diagnostic: the scalars of one validation batch (boundary RMSE, residual loss, RMSE, VRMSE) from the normalised model
output, with the separate functions (two sweeps over the output) and with validation_scalars (one sweep)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pde_amd
dev = torch.device("cuda:0")
B, T = 16, 15
o2 = torch.randn(B, T, 256, 256, 18, device=dev)
inp = torch.randn(B, T, 256, 256, 18, device=dev)
tgt = torch.randn(B, T, 256, 256, 18, device=dev)
mask = torch.zeros(B, 18, device=dev); mask[:, [0, 1, 15]] = 1
mean = torch.randn(B, 1, 1, 1, 18, device=dev); std = torch.rand(B, 1, 1, 1, 18, device=dev) + 0.5
nu = torch.rand(B, device=dev) * 0.09 + 0.01
fn = pde_amd.TaylorGreen2DPDELoss(nx=256, ny=256, dt=0.01).to(dev)

@torch.no_grad()
def separate():
    o = pde_amd.Denormalised(o2, mean, std)
    return (pde_amd.compute_boundary_loss(o, tgt, mask, 1), pde_amd.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)[0],
            *reversed(pde_amd.compute_vrmse(o, tgt, mask)))

def fused():
    return pde_amd.validation_scalars(pde_amd.Denormalised(o2, mean, std), inp, tgt, mask, fn, (0, 1, 15), bc_width=1, nu=nu)

for name, f in (("separate functions", separate), ("validation_scalars", fused)):
    for _ in range(3):
        r = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        r = f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms   values {[float(x) for x in r]}")
