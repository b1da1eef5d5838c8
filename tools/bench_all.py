"""Forward+backward timing of every residual loss at the reference's benchmark configurations (HIP events; algorithmic bytes
12 * E_in per step).  Prints one line per case; --json writes the table."""
import inspect
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pde_amd
from pde_amd.stencil.specs import SPECS

dev = torch.device("cuda:0")


def timeit(fn, n=15, warm=3, settle_ms=150.0):
    if "--quick" in sys.argv:      # counter collection: a few launches, no settling
        n, warm, settle_ms = 3, 1, 0.0
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    # an idle GPU reaches its working clocks only after a while of activity (the same kernel measures 10 % slower before)
    import time
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


# (class, ctor kwargs, field shape, per-sample kwargs builder)
U = lambda B, lo, hi: lo + (hi - lo) * torch.rand(B, device=dev)
CASES = [
    ("C1 Burgers1D", "Burgers1DPDELoss", dict(dt=0.01), (20, 9, 256), lambda B: dict(nu=U(B, 0.005, 0.1))),
    ("Burgers1D large", "Burgers1DPDELoss", dict(dx=2 * 3.141592653589793 / 4096, dt=0.01), (512, 17, 4096), lambda B: dict(nu=U(B, 0.005, 0.1))),
    ("Heat1D", "Heat1DPDELoss", dict(dx=2 * 3.141592653589793 / 4096, dt=0.01), (512, 17, 4096), lambda B: dict(alpha=U(B, 0.01, 0.1))),
    ("KdV1D", "KdV1DPDELoss", dict(), (512, 17, 4096), lambda B: {}),
    ("C2 TaylorGreen2D", "TaylorGreen2DPDELoss", dict(nx=256, ny=256, dt=0.01), (16, 16, 256, 256), lambda B: dict(nu=U(B, 0.01, 0.1))),
    ("C3 GrayScott fd4 (8/GPU)", "GrayScottPDELoss", dict(laplacian="fd4"), (8, 9, 128, 128), lambda B: {}),
    ("C3 GrayScott fft (8/GPU)", "GrayScottPDELoss", dict(), (8, 9, 128, 128), lambda B: {}),
    ("GrayScott fd4 B=64", "GrayScottPDELoss", dict(laplacian="fd4"), (64, 9, 128, 128), lambda B: {}),
    ("GrayScott fft B=64", "GrayScottPDELoss", dict(), (64, 9, 128, 128), lambda B: {}),
    ("C4 RayleighBenard B=16", "RayleighBenardFullPDELoss", dict(), (16, 9, 512, 128), lambda B: dict(kappa=U(B, 1e-4, 1e-2))),
    ("C4 RayleighBenard B=64", "RayleighBenardFullPDELoss", dict(), (64, 9, 512, 128), lambda B: dict(kappa=U(B, 1e-4, 1e-2))),
    ("C4 ShearFlow B=16", "ShearFlowPDELossNPINN", dict(), (16, 9, 256, 512), lambda B: {}),
    ("C4 ShearFlow B=64", "ShearFlowPDELossNPINN", dict(), (64, 9, 256, 512), lambda B: {}),
    ("AdvDiff2D", "AdvDiff2DPDELoss", dict(), (32, 9, 256, 256), lambda B: {}),
    ("Wave2D", "Wave2DPDELoss", dict(), (32, 9, 256, 256), lambda B: dict(c=U(B, 0.5, 2.0))),
    ("Burgers2DCH", "Burgers2DCHPDELoss", dict(), (32, 9, 256, 256), lambda B: dict(nu=U(B, 0.005, 0.05))),
    ("NSPwC", "NSPwCPDELoss", dict(), (32, 9, 256, 256), lambda B: {}),
    ("KP2D", "KP2DPDELoss", dict(), (32, 9, 256, 256), lambda B: {}),
    ("ActiveMatter", "ActiveMatterNPINNPDELoss", dict(), (16, 9, 256, 256), lambda B: {}),
    ("TurbulentRadiative", "TurbulentRadiativePDELoss", dict(), (16, 9, 128, 384), lambda B: {}),
    ("C5 Heat3D B=2", "Heat3DPDELoss", dict(), (2, 9, 128, 128, 128), lambda B: dict(alpha=U(B, 0.005, 0.05))),
    ("C5 Burgers3D B=1", "APEBenchBurgers3DPDELoss", dict(), (1, 9, 128, 128, 128), lambda B: {}),
    ("TaylorGreen3D B=1", "TaylorGreen3DPDELoss", dict(), (1, 9, 128, 128, 128), lambda B: dict(nu=U(B, 0.01, 0.1))),
    ("Advection3D B=2", "Advection3DPDELoss", dict(), (2, 9, 128, 128, 128), lambda B: dict(a=U(B, 0.1, 1), b=U(B, 0.1, 1), c=U(B, 0.1, 1))),
    ("SwiftHohenberg3D B=2", "SwiftHohenberg3DPDELoss", dict(), (2, 9, 64, 64, 64), lambda B: {}),
]


def pk_names(params):
    return set(params(1).keys())


def main():
    rows = []
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for tag, cls_name, kw, shape, params in CASES:
        if only and not any(o.lower() in tag.lower() for o in only):
            continue
        cls = getattr(pde_amd, cls_name)
        try:
            mod = cls(**kw).to(dev)
        except TypeError as e:
            print(f"{tag:28s} ctor failed: {e}")
            continue
        if "--fused" in sys.argv:
            mod.backward_mode = "fused"
        fwd_params = inspect.signature(mod.forward).parameters
        nf = sum(1 for q in fwd_params.values() if q.default is inspect.Parameter.empty and q.name not in pk_names(params))
        gen = torch.Generator(device=dev).manual_seed(1)
        fields = [(1.0 + 0.1 * torch.randn(shape, device=dev, generator=gen)).requires_grad_(True) for _ in range(nf)]
        pk = params(shape[0])

        def step():
            for f in fields:
                f.grad = None
            total, _ = mod(*fields, **pk)
            total.backward()

        def fwd_only():
            with torch.no_grad():
                mod(*fields, **pk)
        try:
            ms = timeit(step)
            ms_f = timeit(fwd_only)
        except Exception as e:
            print(f"{tag:28s} failed: {type(e).__name__}: {e}")
            continue
        e_in = nf
        for v in shape:
            e_in *= v
        # algorithmic bytes of a step: forward reads the fields (4 E_in), backward reads them again and writes the gradients
        # (8 E_in): 12 E_in.  The fused step is one pass, read + write = 8 E_in; where a PDE has no fused kernel the mode
        # falls back to forward + backward and 12 E_in stays the right count (the 8 E_in column is then an underestimate).
        fused = mod.backward_mode == "fused"
        alg = (8 if fused else 12) * e_in
        gbs = alg / ms / 1e6
        gbs_f = 4 * e_in / ms_f / 1e6
        rows.append(dict(case=tag, cls=cls_name, shape=list(shape), fields=nf, ms_fwd_bwd=ms, ms_fwd=ms_f, gbs_fwd_bwd=gbs, gbs_fwd=gbs_f,
                         mode=mod.backward_mode, algorithmic_bytes=alg, e_in_bytes_factor=8 if fused else 12))
        print(f"{tag:28s} {str(tuple(shape)):22s} F={nf}  fwd+bwd {ms:8.3f} ms {gbs:7.0f} GB/s of {'8' if fused else '12'} E_in ({gbs / 80:4.1f}% of 8 TB/s)   fwd {ms_f:7.3f} ms {gbs_f:6.0f} GB/s")
    if "--json" in sys.argv:
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(rows, open("gpurun_out/bench_all.json", "w"), indent=1)


if __name__ == "__main__":
    main()
