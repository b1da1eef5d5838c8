"""Diagnostic: the physics losses of one training step from the normalised model output (Denormalised + fused residual),
run a few times for rocprofv3 --kernel-trace --stats; prints the wall time per step by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pde_amd
dev = torch.device("cuda:0")
B, T = 16, 15
o2 = torch.randn(B, T, 256, 256, 18, device=dev, requires_grad=True)
inp = torch.randn(B, T, 256, 256, 18, device=dev)
tgt = torch.randn(B, T, 256, 256, 18, device=dev)
mask = torch.zeros(B, 18, device=dev); mask[:, [0, 1, 15]] = 1
mean = torch.randn(B, 1, 1, 1, 18, device=dev); std = torch.rand(B, 1, 1, 1, 18, device=dev) + 0.5
nu = torch.rand(B, device=dev) * 0.09 + 0.01
fn = pde_amd.TaylorGreen2DPDELoss(nx=256, ny=256, dt=0.01).to(dev)
fn.backward_mode, fn.upstream_hint = "fused", 1e-3
def step():
    o2.grad = None
    o = pde_amd.Denormalised(o2, mean, std)
    pde, _ = pde_amd.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)
    bc = pde_amd.compute_boundary_loss(o, tgt, mask, 1)
    (1e-3 * pde + 0.1 * bc).backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record(); torch.cuda.synchronize()
print(f"step losses: {e0.elapsed_time(e1) / 20:.3f} ms")

# the same step captured once and replayed as a HIP graph (no host work per step): needs a warmed-up step, since the first
# call reads the channel list from the mask on the host
if "--graph" in sys.argv:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    o2.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o = pde_amd.Denormalised(o2, mean, std)
        pde, _ = pde_amd.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)
        bc = pde_amd.compute_boundary_loss(o, tgt, mask, 1)
        (1e-3 * pde + 0.1 * bc).backward()
    g_live = o2.grad      # the buffer the replays write
    graph.replay(); torch.cuda.synchronize()
    g_ref = g_live.clone()
    e0.record()
    for _ in range(20):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    o2.grad = None
    step()
    print(f"step losses, graph replay: {e0.elapsed_time(e1) / 20:.3f} ms; gradient equals the eager one: {bool(torch.equal(g_ref, o2.grad))}")
