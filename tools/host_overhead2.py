"""Host enqueue rate against GPU time for the Taylor-Green step, eager, both backward modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pde_amd
dev = torch.device("cuda:0")
fields = [torch.randn(16, 16, 256, 256, device=dev, requires_grad=True) for _ in range(3)]
nu = torch.rand(16, device=dev) * 0.1
one = torch.ones((), device=dev)
for mode in ("auto", "fused"):
    mod = pde_amd.TaylorGreen2DPDELoss(nx=256, ny=256).to(dev)
    mod.backward_mode = mode
    def step():
        for f in fields:
            f.grad = None
        t, _ = mod(*fields, nu=nu)
        t.backward(gradient=one)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host enqueue {(t1 - t0) / 300 * 1e6:.1f} us per step, with the GPU drained {(t2 - t0) / 300 * 1e6:.1f} us per step")

import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
