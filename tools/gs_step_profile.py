import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_amd
dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mod = pde_amd.GrayScottPDELoss().to(dev)
mod.backward_mode = "fused"
g = torch.Generator(device=dev).manual_seed(1)
fields = [(0.5 + 0.25 * torch.rand((b, 9, 128, 128), device=dev, generator=g)).requires_grad_(True) for _ in range(2)]
for i in range(30):
    for f in fields:
        f.grad = None
    t, _ = mod(*fields)
    t.backward()
torch.cuda.synchronize()
print(float(t))
