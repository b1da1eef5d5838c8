"""Diagnostic: the one-pass Taylor-Green step at several clip lengths (same batch and grid): the fixed cost of a workgroup's
first and last steps shows as the intercept of time against residual frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pde_amd
import tools.bench_all as BA
dev = BA.dev
rows = []
for T in (6, 9, 16, 23, 30):
    shape = (16, T, 256, 256)
    mod = pde_amd.TaylorGreen2DPDELoss(nx=256, ny=256, dt=0.01).to(dev)
    mod.backward_mode = "fused"
    fields = [(1.0 + 0.1 * torch.randn(shape, device=dev)).requires_grad_(True) for _ in range(3)]
    nu = 0.01 + 0.09 * torch.rand(16, device=dev)
    def step():
        for f in fields:
            f.grad = None
        mod(*fields, nu=nu)[0].backward()
    ms = BA.timeit(step)
    rows.append((T - 2, ms))
    print(f"T={T:3d} residual frames {T-2:3d}: {ms:.4f} ms  ({1e3 * ms / (T - 2):.2f} us per residual frame)")
(x0, y0), (x1, y1) = rows[2], rows[-1]
slope = (y1 - y0) / (x1 - x0)
print(f"slope {1e3 * slope:.2f} us per frame, intercept at T=16: {1e3 * (y0 - slope * x0):.1f} us")
