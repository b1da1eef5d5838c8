"""replayed (loss_fn.capture) and eager step times of the small configurations"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_amd
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
cases = [
    ("C1 Burgers-1D [20,9,256]", pde_amd.Burgers1DPDELoss(), [R(20, 9, 256)], dict(nu=torch.full((20,), 0.01, device=dev))),
    ("C3 Gray-Scott FD 8/GPU", pde_amd.GrayScottPDELoss(laplacian="fd4"), [R(8, 9, 128, 128).abs() for _ in range(2)], {}),
    ("C3 Gray-Scott FFT 8/GPU", pde_amd.GrayScottPDELoss(), [R(8, 9, 128, 128).abs() for _ in range(2)], {}),
    ("C3 Gray-Scott FFT 64/GPU", pde_amd.GrayScottPDELoss(), [R(64, 9, 128, 128).abs() for _ in range(2)], {}),
    ("C2 Taylor-Green [16,16,256,256]", pde_amd.TaylorGreen2DPDELoss(), [R(16, 16, 256, 256) for _ in range(3)], dict(nu=torch.full((16,), 0.05, device=dev))),
]
def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for tag, mod, fields, kw in cases:
    mod = mod.to(dev)
    mod.backward_mode = "fused"
    fields = [f.requires_grad_(True) for f in fields]
    def eager():
        for f in fields:
            f.grad = None
        t, _ = mod(*fields, **kw)
        t.backward(gradient=mod.upstream_scalar(dev))
    e = timeit(eager)
    step = mod.capture(*fields, **kw)
    r = timeit(step)
    print("%-36s eager %.4f ms   replayed %.4f ms" % (tag, e, r), flush=True)
