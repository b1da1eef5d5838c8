"""Diagnostic: forward+backward time of the 3-D heat loss (eager, HIP events) for a given library build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pde_amd
dev = torch.device("cuda:0")
B = 4
u = torch.randn(B, 9, 128, 128, 128, device=dev, requires_grad=True)
alpha = torch.rand(B, device=dev) * 0.04 + 0.005
for mode in ("auto", "fused"):
    mod = pde_amd.Heat3DPDELoss().to(dev); mod.backward_mode = mode
    def step():
        u.grad = None
        t, _ = mod(u, alpha=alpha); t.backward()
    for _ in range(5): step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"heat3d B={B} {mode}: {ts[len(ts)//2]:.3f} ms")
