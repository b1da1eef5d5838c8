"""Poison the partial-sum workspace with NaNs before every call: a kernel path that leaves a slot unwritten, or a reduction
that reads past the slots written, shows up as a NaN total."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_amd
from pde_amd import functional as PF
PF.SMALL_FIELD_ELEMENTS = 0
dev = torch.device("cuda:0")
be = PF.get_backend()
g = torch.Generator().manual_seed(3)
def poison():
    torch.cuda.synchronize()
    for ws in be._ws.values():
        ws.view(torch.float64).fill_(float("nan"))
cases = [
  ("tg3d", lambda: pde_amd.TaylorGreen3DPDELoss(nx=64, ny=16, nz=20, Lx=2.0, Ly=1.0, Lz=3.0, dt=0.01), 4, (20, 16, 64), lambda B: dict(nu=torch.full((B,), 0.05, device=dev))),
  ("burgers3d", lambda: pde_amd.APEBenchBurgers3DPDELoss(nx=64, Lx=1.0, dt=1.0), 3, (20, 16, 64), lambda B: {}),
  ("heat3d", lambda: pde_amd.Heat3DPDELoss(nx=64, ny=16, nz=20, Lx=1.0, Ly=1.0, Lz=1.0, dt=0.01), 1, (20, 16, 64), lambda B: dict(alpha=torch.full((B,), 0.01, device=dev))),
  ("adv3d", lambda: pde_amd.Advection3DPDELoss(dx=0.1, dy=0.2, dz=0.3, dt=0.05), 1, (20, 16, 64), lambda B: dict(a=torch.ones(B, device=dev), b=torch.ones(B, device=dev), c=torch.ones(B, device=dev))),
  ("tg2d", lambda: pde_amd.TaylorGreen2DPDELoss(nx=64, ny=32, dt=0.01), 3, (32, 64), lambda B: dict(nu=torch.full((B,), 0.05, device=dev))),
]
be._workspace(dev, 1 << 26)
for name, make, nf, sp, kwf in cases:
    for mode in ("auto", "stored", "fused", "recompute"):
        for B in (1, 2):
            for T in (3, 4, 9, 12, 17):
                mod = make().to(dev)
                mod.backward_mode = mode
                fs = [(torch.randn((B, T) + sp, generator=g) * 0.5).to(dev).requires_grad_(True) for _ in range(nf)]
                poison()
                try:
                    t, _ = mod(*fs, **kwf(B))
                    t.backward()
                except Exception as e:
                    print(name, mode, B, T, "error", type(e).__name__, str(e)[:80]); continue
                bad = (not bool(torch.isfinite(t))) or any(not bool(torch.isfinite(f.grad).all()) for f in fs) or not bool(torch.isfinite(mod.last_per_sample).all())
                if bad:
                    print(name, mode, "B", B, "T", T, "NON-FINITE", float(t))
print("done")
