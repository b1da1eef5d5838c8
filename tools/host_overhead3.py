"""Where the host time of a fused-mode step goes: forward call, backward call (eager, Taylor-Green bench shape)."""
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
    tf = tb = tz = 0.0
    for i in range(320):
        a = time.perf_counter()
        for f in fields:
            f.grad = None
        b = time.perf_counter()
        t, _ = mod(*fields, nu=nu)
        c = time.perf_counter()
        t.backward(gradient=one)
        d = time.perf_counter()
        if i >= 20:
            tz += b - a; tf += c - b; tb += d - c
    torch.cuda.synchronize()
    print(f"{mode}: grad=None {tz / 300 * 1e6:.1f} us, forward call {tf / 300 * 1e6:.1f} us, backward call {tb / 300 * 1e6:.1f} us")
