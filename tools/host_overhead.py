"""Host-side cost of one loss call on a tiny problem (where kernels are negligible): wall time and a cProfile top list."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pde_amd
dev = torch.device("cuda:0")
u = torch.randn(20, 9, 256, device=dev, requires_grad=True)
nu = torch.rand(20, device=dev) * 0.1
mod = pde_amd.Burgers1DPDELoss().to(dev)

def step():
    u.grad = None
    t, _ = mod(u, nu=nu)
    t.backward()

def fwd():
    with torch.no_grad():
        mod(u, nu=nu)
for name, fn in (("fwd+bwd", step), ("fwd (no_grad)", fwd)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
