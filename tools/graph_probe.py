"""Can the whole loss step be captured into a HIP graph (ctypes launches, autograd, timing events) and replayed?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pde_amd
from pde_amd import functional as PF
dev = torch.device("cuda:0")
fields = [torch.randn(16, 16, 256, 256, device=dev, requires_grad=True) for _ in range(3)]
nu = torch.rand(16, device=dev) * 0.1
mod = pde_amd.TaylorGreen2DPDELoss(nx=256, ny=256).to(dev)
grads = [torch.zeros_like(f) for f in fields]
for f, g in zip(fields, grads):
    f.grad = g

def step():
    total, losses = mod(*fields, nu=nu)
    total.backward()
    return total

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        for f in fields:
            f.grad = None
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ref_total = float(step().detach()); ref_g = fields[0].grad.clone()
for f in fields:
    f.grad = None
g = torch.cuda.CUDAGraph()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
try:
    with torch.cuda.graph(g):
        PF.get_backend().backward_events = []
        PF.get_backend().backward_event_pool = [(ev[0], ev[1])]
        out = step()
    print("capture ok")
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300]); sys.exit(0)
PF.get_backend().backward_events = None
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
print(f"replay: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step; total {float(out):.6e} vs eager {ref_total:.6e}; grad equal {torch.equal(fields[0].grad, ref_g)}")
try:
    print("kernel events inside the graph:", ev[0].elapsed_time(ev[1]), "ms")
except Exception as e:
    print("event timing inside graph failed:", type(e).__name__, str(e)[:200])
