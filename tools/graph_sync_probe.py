"""Diagnostic: what a cross-stream dependency costs between replays of the benchmark step's HIP graph.
Variants: replays alone; + one event record per step; + a side stream that waits for the event and runs a tiny kernel;
+ the main stream waiting (one step later) for the side stream's event.  One rank, no collective."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pde_amd
import bench as BM
dev = torch.device("cuda:0")
fields = [f.requires_grad_(True) for f in BM.synth_fields(BM.B, BM.T, BM.H, BM.W, 1, dev)]
nu = 0.01 + 0.09 * torch.rand(BM.B, device=dev)
fn = pde_amd.TaylorGreen2DPDELoss(nx=BM.W, ny=BM.H, dt=0.01, eq_scales=BM.EQ_SCALES).to(dev)
fn.backward_mode = "fused"
one = torch.ones((), device=dev)
def loss_step():
    for f in fields:
        f.grad = None
    total, losses = fn(*fields, nu=nu)
    total.backward(gradient=one)
    return torch.stack([total.detach()] + [v.detach() for v in losses.values()]).to(torch.float64)
for _ in range(3):
    loss_step()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    loss_step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    buf = loss_step()
g.replay(); torch.cuda.synchronize()
aux = torch.cuda.Stream()
N = 200
def run(mode):
    evs = [torch.cuda.Event() for _ in range(N)]
    done = [torch.cuda.Event() for _ in range(N)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        g.replay()
        if mode >= 1:
            evs[i].record()
        if mode >= 2:
            aux.wait_event(evs[i])
            with torch.cuda.stream(aux):
                buf.add_(0.0)
                done[i].record()
        if mode >= 3 and i:
            torch.cuda.current_stream().wait_event(done[i - 1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3
names = ["replays alone", "+ event record per step", "+ side stream waits and runs a tiny kernel", "+ main stream waits for the side stream a step later"]
for rep in range(2):
    for m, nm in enumerate(names):
        print(f"{nm}: {run(m):.4f} ms/step")
