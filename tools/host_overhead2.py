"""Host enqueue time of the Taylor-Green forward and backward calls (GPU idle at the start of each measurement)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pde_amd
dev = torch.device("cuda:0")
fields = [torch.randn(16, 16, 256, 256, device=dev, requires_grad=True) for _ in range(3)]
nu = torch.rand(16, device=dev) * 0.1
mod = pde_amd.TaylorGreen2DPDELoss(nx=256, ny=256).to(dev)
tf = tb = 0.0
N = 200
for i in range(N + 20):
    for f in fields:
        f.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total, _ = mod(*fields, nu=nu)
    t1 = time.perf_counter()
    total.backward()
    t2 = time.perf_counter()
    if i >= 20:
        tf += t1 - t0; tb += t2 - t1
print(f"host enqueue: forward {tf / N * 1e6:.1f} us, backward {tb / N * 1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for i in range(200):
    for f in fields:
        f.grad = None
    total, _ = mod(*fields, nu=nu)
    total.backward()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
