"""HIP kernels (through the C ABI) against the reference's golden vectors and the CPU oracle.

Tolerance: loss relative 1e-5, gradient relative L2 1e-5 (the bound the project states for fp32);
the golden references are the reference classes run in fp64 on the same inputs.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pde_amd.losses as PL
from oracle import classes as OC
from tests.helpers import GOLDEN, backprop, close, golden_cases, load_case, rel_l2

TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _run(mod, fields, extra):
    fields = [f.clone().requires_grad_(True) for f in fields]
    total, losses = mod(*fields, **extra)
    backprop(total)
    return total, losses, [f.grad for f in fields]


@pytest.mark.parametrize("mode", ["auto", "stored", "fused"])      # fused falls back to stored where the PDE cannot recompute
@pytest.mark.parametrize("name", golden_cases())
def test_golden(name, mode):
    dev = _dev()
    meta, kwargs, fields, extra, z = load_case(name, torch.float32, dev)
    mod = getattr(PL, meta["cls"])(**kwargs).to(dev)
    mod.backward_mode = mode
    total, losses, grads = _run(mod, fields, extra)
    assert close(total, z["f64_total"], TOL)
    ref_keys = sorted(k[9:] for k in z.files if k.startswith("f64_loss_"))
    assert sorted(losses) == ref_keys
    for k in ref_keys:
        assert close(losses[k], z[f"f64_loss_{k}"], TOL, 1e-30), k
    for n, g in zip(meta["fields"], grads):
        ref = z[f"f64_grad_{n}"]
        got = g.cpu().numpy() if g is not None else np.zeros_like(ref)
        assert rel_l2(got, ref) <= TOL, n


def test_ghost_burgers_golden():
    dev = _dev()
    z = np.load(os.path.join(GOLDEN, "pde_burgers2d_ghost.npz"))
    m = json.loads(bytes(z["meta"]).decode())
    pred = torch.tensor(z["in_pred"], dtype=torch.float32, device=dev, requires_grad=True)
    total, lu, lv, ru, rv = PL.burgers_pde_loss(pred, m["nu"], dt=m["dt"], dx=m["dx"], dy=m["dy"])
    total.backward()
    assert abs(float(total) - float(z["f64_total"])) <= TOL * abs(float(z["f64_total"]))
    assert ru.requires_grad and rel_l2(ru.detach().cpu().numpy(), z["f64_res_u"]) <= TOL
    assert rel_l2(rv.detach().cpu().numpy(), z["f64_res_v"]) <= TOL
    assert rel_l2(pred.grad.cpu().numpy(), z["f64_grad_pred"]) <= TOL


# moderate sizes with ragged tiles (extents that are not multiples of the tile) against the fp64 oracle
SIZED = [
    ("Burgers1DPDELoss", dict(dx=0.02, dt=0.01), (3, 7, 300), dict(nu=(0.005, 0.1))),
    ("TaylorGreen2DPDELoss", dict(nx=72, ny=50, Lx=6.28, Ly=6.28, dt=0.01), (3, 6, 50, 72), dict(nu=(0.01, 0.1))),
    ("ShearFlowPDELossNPINN", dict(nx=40, ny=70, Lx=1.0, Ly=2.0, dt=0.1), (2, 5, 40, 70), {}),
    ("ShearFlowPDELoss", dict(nx=40, ny=70, Lx=1.0, Ly=2.0, dt=0.1, nu=0.01, D=0.02), (2, 5, 40, 70), {}),
    ("RayleighBenardFullPDELoss", dict(nx=48, ny=40, Lx=4.0, Ly=1.0, dt=0.25, skip_bl=3), (2, 5, 48, 40), dict(kappa=(1e-3, 1e-2))),
    ("Burgers2DCHPDELoss", dict(nx=66, ny=20, dt=0.01), (2, 4, 20, 66), dict(nu=(0.01, 0.1))),
    ("KP2DPDELoss", dict(dx=0.1, dy=0.1, dt=0.01, skip_boundary=5), (2, 5, 30, 70), {}),
    ("NSPwCPDELoss", dict(nx=70, ny=33, dt=0.05), (2, 5, 33, 70), {}),
    ("TurbulentRadiativePDELoss", dict(nx=20, ny=70, dt=1.5, skip_bl=2), (2, 4, 20, 70), {}),
    ("APEBenchBurgers3DPDELoss", dict(nx=20, Lx=1.0, dt=1.0), (1, 4, 9, 10, 70), {}),
    ("Advection3DPDELoss", dict(dx=0.1, dy=0.2, dz=0.3, dt=0.05), (2, 4, 5, 9, 66), dict(a=(-1, 1), b=(-1, 1), c=(-1, 1))),
    ("SwiftHohenberg3DPDELoss", dict(domain_extent=6.0, num_points=12, dt=0.1), (1, 4, 6, 10, 65), {}),
    ("Heat3DPDELoss", dict(nx=6, ny=10, nz=65, Lx=1.0, Ly=1.0, Lz=1.0, dt=0.01), (2, 4, 6, 10, 65), dict(alpha=(0.005, 0.05))),
    ("GrayScottPDELoss", dict(nx=24, ny=70, dx=0.1, dy=0.1, dt=10.0), (2, 5, 24, 70), {}),
    ("Advection1DPDELoss", dict(dx=0.02, dt=0.01), (3, 6, 301), dict(a=(-1, 1))),
    ("Heat1DPDELoss", dict(dx=0.02, dt=0.01), (3, 6, 301), dict(alpha=(0.005, 0.05))),
    ("APEBenchBurgers1DPDELoss", dict(nx=301, Lx=1.0, dt=1.0), (2, 6, 301), {}),
    ("KS1DPDELoss", dict(nx=301, Lx=1.0, dt=1.0), (2, 6, 301), {}),
    ("KdV1DPDELoss", dict(nx=301, Lx=1.0, dt=1.0), (2, 6, 301), {}),
    ("AdvDiff2DPDELoss", dict(dx=0.2, dy=0.3, dt=0.01, a=1.0, b=0.5, nu=0.05), (2, 5, 21, 70), {}),
    ("AnisoDiff2DPDELoss", dict(dx=0.2, dy=0.3, dt=0.1), (2, 5, 21, 70), {}),
    ("Wave2DPDELoss", dict(dx=0.2, dy=0.3, dt=0.01), (2, 5, 21, 70), dict(c=(0.5, 1.5))),
    ("APEBenchBurgers2DPDELoss", dict(nx=70, ny=21, Lx=1.0, Ly=1.5, dt=1.0), (2, 5, 21, 70), {}),
    ("ActiveMatterNPINNPDELoss", dict(nx=70, ny=21, Lx=2.0, Ly=3.0, dt=0.25, d_T=0.05), (2, 5, 21, 70), {}),
    ("WaveGaussPDELoss", dict(nx=21, ny=70, Lx=1.0, Ly=1.0, dt=1.0 / 14, skip_boundary=1), (2, 5, 21, 70), dict(c=("field", 0.5, 1.5))),
    ("NSIncomPDELoss", dict(nx=40, ny=70, Lx=1.0, Ly=1.0, dt=0.005, nu=0.01, skip_bl=2), (2, 5, 40, 70),
     dict(fx=("field", -1.0, 1.0), fy=("field", -1.0, 1.0))),
    # 3-D grids of several chunks of planes (the last one short) and ragged (y, x) tiles
    ("TaylorGreen3DPDELoss", dict(nx=37, ny=18, nz=42, Lx=2.0, Ly=1.0, Lz=3.0, dt=0.01), (1, 4, 37, 18, 42), dict(nu=(0.01, 0.1))),
    ("APEBenchBurgers3DPDELoss", dict(nx=42, Lx=1.0, dt=1.0), (1, 4, 37, 18, 42), {}),
    ("Heat3DPDELoss", dict(nx=37, ny=18, nz=42, Lx=1.0, Ly=1.0, Lz=1.0, dt=0.01), (2, 4, 37, 18, 42), dict(alpha=(0.005, 0.05))),
    ("Advection3DPDELoss", dict(dx=0.1, dy=0.2, dz=0.3, dt=0.05), (1, 4, 37, 18, 42), dict(a=(-1, 1), b=(-1, 1), c=(-1, 1))),
    ("SwiftHohenberg3DPDELoss", dict(domain_extent=6.0, num_points=42, dt=0.1), (1, 4, 37, 18, 42), {}),
]


def _quad_variant(entry):
    """Same case with the fastest extent rounded up to a multiple of 4 (takes the register-blocked kernels)."""
    cls, kw, shape, ps = entry
    n = shape[-1]
    n4 = (n + 3) // 4 * 4 + 4
    kw = dict(kw)
    for key in ("nx", "ny", "nz", "num_points"):
        if kw.get(key) == n:
            kw[key] = n4
    return (cls, kw, tuple(shape[:-1]) + (n4,), ps)


SIZED = SIZED + [_quad_variant(e) for e in SIZED]


def _sized_extras(ps, shape, g):
    """per-sample parameters (lo, hi) -> [B]; ("field", lo, hi) -> a time-independent field [B, *spatial]"""
    out = {}
    for k, r in ps.items():
        size = (shape[0],) + tuple(shape[2:]) if r[0] == "field" else (shape[0],)
        lo, hi = r[-2:]
        out[k] = lo + (hi - lo) * torch.rand(size, generator=g)
    return out


@pytest.mark.parametrize("mode", ["auto", "stored", "fused"])
@pytest.mark.parametrize("cls,kw,shape,ps", SIZED, ids=[f"{s[0]}-{s[2][-1]}" for s in SIZED])
def test_ragged_sizes_vs_oracle(cls, kw, shape, ps, mode):
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    spec = PL.SPECS[getattr(PL, cls).SPEC] if getattr(PL, cls).SPEC else PL.SPECS["gray_scott_pw"]
    nf = len(spec.fields) if cls != "GrayScottPDELoss" else 2
    fields = [torch.randn(shape, generator=g) * 0.5 for _ in range(nf)]
    if cls == "TurbulentRadiativePDELoss":
        fields[2] = fields[2].abs() + 0.5
    extra = _sized_extras(ps, shape, g)
    mod = getattr(PL, cls)(**kw).to(dev)
    mod.backward_mode = mode
    total, losses, grads = _run(mod, [f.to(dev) for f in fields], {k: v.to(dev) for k, v in extra.items()})
    f64 = [f.double().requires_grad_(True) for f in fields]
    ot, ol = OC.build(cls, **kw)(*f64, **{k: v.double() for k, v in extra.items()})
    ot.backward()
    assert abs(float(total) - float(ot)) <= TOL * abs(float(ot))
    for k in ol:
        assert abs(float(losses[k]) - float(ol[k])) <= TOL * abs(float(ol[k])) + 1e-30, k
    for gr, f in zip(grads, f64):
        assert rel_l2(gr.cpu().numpy(), f.grad.numpy()) <= TOL


LONG3D = [
    ("TaylorGreen3DPDELoss", dict(nx=64, ny=16, nz=20, Lx=2.0, Ly=1.0, Lz=3.0, dt=0.01), (1, 17, 20, 16, 64), dict(nu=(0.01, 0.1))),
    ("APEBenchBurgers3DPDELoss", dict(nx=64, Lx=1.0, dt=1.0), (1, 17, 20, 16, 64), {}),
    ("Heat3DPDELoss", dict(nx=64, ny=16, nz=20, Lx=1.0, Ly=1.0, Lz=1.0, dt=0.01), (2, 17, 20, 16, 64), dict(alpha=(0.005, 0.05))),
    ("Advection3DPDELoss", dict(dx=0.1, dy=0.2, dz=0.3, dt=0.05), (2, 12, 20, 16, 64), dict(a=(-1, 1), b=(-1, 1), c=(-1, 1))),
]


@pytest.mark.parametrize("cls,kw,shape,ps", LONG3D, ids=[s[0] for s in LONG3D])
def test_long_clips_3d_vs_oracle(cls, kw, shape, ps):
    """Clips of 10 and 15 residual frames on grids the z-column kernels take: several groups of frames per tile, the last
    one shorter, with the frames between two groups staged by both."""
    test_ragged_sizes_vs_oracle(cls, kw, shape, ps, "auto")


def _whole_tile_variant(entry):
    """Same case on a grid of whole tiles (1-D: 512 points; 2-D: 32 x 128): uncropped periodic classes then take the
    adjoint instantiation that carries no lattice masks."""
    cls, kw, shape, ps = entry
    new = {1: (512,), 2: (32, 128)}[len(shape) - 2]
    kw = dict(kw)
    for old_n, new_n in zip(shape[2:], new):
        for key in ("nx", "ny", "num_points"):
            if kw.get(key) == old_n:
                kw[key] = new_n
    return (cls, kw, tuple(shape[:2]) + new, ps)


WHOLE = [_whole_tile_variant(e) for e in SIZED[:len(SIZED) // 2] if len(e[2]) - 2 in (1, 2)]


@pytest.mark.parametrize("mode", ["auto", "fused"])
@pytest.mark.parametrize("cls,kw,shape,ps", WHOLE, ids=[s[0] for s in WHOLE])
def test_whole_tile_grids_vs_oracle(cls, kw, shape, ps, mode):
    """Grids that are a whole number of tiles along every axis: the periodic, uncropped classes run the mask-free
    instantiation of the marching adjoint, which no ragged case reaches."""
    test_ragged_sizes_vs_oracle(cls, kw, shape, ps, mode)


@pytest.mark.parametrize("mode", ["auto", "stored", "fused"])
@pytest.mark.parametrize("width", [70, 72])
def test_per_t_scales_offsets_and_guards_at_tile_sizes(width, mode):
    """Per-timestep scale tables with sample offsets (finetune/pde_loss_verified.py:854-901), non-unit weights and an
    equation switched off by a zero weight, on grids that take the tiled kernels (the golden cases of these features are
    a few points wide)."""
    dev = _dev()
    g = torch.Generator().manual_seed(31)
    shape = (3, 7, 50, width)
    fields = [torch.randn(shape, generator=g) * 0.5 for _ in range(3)]
    nu = 0.01 + 0.09 * torch.rand(3, generator=g)
    per_t = {k: 0.5 + torch.rand(9, generator=g) for k in ("continuity", "x_momentum", "y_momentum")}
    base = dict(nx=width, ny=50, Lx=6.28, Ly=6.28, dt=0.01, eq_scales={"continuity": 0.5, "x_momentum": 2.0, "y_momentum": 3.0})
    cases = [
        (dict(base, eq_weights={"continuity": 1.0, "x_momentum": 0.7, "y_momentum": 1.1}, eq_scales_per_t=per_t),
         dict(t_offset=torch.tensor([2, 7, 0]))),
        (dict(base, eq_weights={"x_momentum": 0.7}, eq_scales_per_t=per_t), {}),
        (dict(base, eq_weights={"y_momentum": 0.0}), {}),
    ]
    for kw, ex in cases:
        mod = PL.TaylorGreen2DPDELoss(**kw).to(dev)
        mod.backward_mode = mode
        total, losses, grads = _run(mod, [f.to(dev) for f in fields], dict(nu=nu.to(dev), **{k: v.to(dev) for k, v in ex.items()}))
        f64 = [f.double().requires_grad_(True) for f in fields]
        ot, ol = OC.build("TaylorGreen2DPDELoss", **kw)(*f64, nu=nu.double(), **ex)
        ot.backward()
        assert abs(float(total) - float(ot)) <= TOL * abs(float(ot))
        for k in ol:
            assert abs(float(losses[k]) - float(ol[k])) <= TOL * abs(float(ol[k])) + 1e-30, k
        for gr, f in zip(grads, f64):
            assert rel_l2(gr.cpu().numpy(), f.grad.numpy()) <= TOL


@pytest.mark.parametrize("shape", [(2, 4, 40, 72, 2), (2, 3, 37, 70, 2)])
def test_ghost_burgers_at_tile_sizes(shape):
    """burgers_pde_loss (finetune/pde_loss.py:87-216) on grids of several tiles: losses, residual fields and gradient
    against the oracle"""
    dev = _dev()
    g = torch.Generator().manual_seed(37)
    base = torch.randn(shape, generator=g) * 0.5
    pred = base.to(dev).requires_grad_(True)
    total, lu, lv, ru, rv = PL.burgers_pde_loss(pred, 0.03, dt=0.02, dx=0.11, dy=0.13)
    (total + 0.5 * ru.sum()).backward()
    p64 = base.double().requires_grad_(True)
    ot, olu, olv, oru, orv = OC.burgers_pde_loss(p64, 0.03, dt=0.02, dx=0.11, dy=0.13)
    (ot + 0.5 * oru.sum()).backward()
    for a, b in ((total, ot), (lu, olu), (lv, olv)):
        assert abs(float(a) - float(b)) <= TOL * abs(float(b))
    assert rel_l2(ru.detach().cpu().numpy(), oru.detach().numpy()) <= TOL
    assert rel_l2(rv.detach().cpu().numpy(), orv.detach().numpy()) <= TOL
    assert rel_l2(pred.grad.cpu().numpy(), p64.grad.numpy()) <= TOL


@pytest.mark.parametrize("cls,kw,shape,ps", [SIZED[1], SIZED[len(SIZED) // 2 + 1], SIZED[len(SIZED) // 2 + 2], SIZED[len(SIZED) // 2 + 3]],
                         ids=["tg2d-ragged", "tg2d-quad", "shear-quad", "rb-quad"])
def test_fused_mode_upstream_values(cls, kw, shape, ps):
    """fused mode against the default mode: assumed upstream hit or missed, a per-equation upstream and a second
    backward (both take the ordinary adjoint), gradients wanted for some fields only"""
    dev = _dev()
    g = torch.Generator().manual_seed(23)
    nf = len(PL.SPECS[getattr(PL, cls).SPEC].fields)
    fields = [(torch.randn(shape, generator=g) * 0.5).to(dev) for _ in range(nf)]
    extra = {k: (lo + (hi - lo) * torch.rand(shape[0], generator=g)).to(dev) for k, (lo, hi) in ps.items()}

    def run(mode, hint, use, need=None):
        mod = getattr(PL, cls)(**kw).to(dev)
        mod.backward_mode, mod.upstream_hint = mode, hint
        fs = [f.clone().requires_grad_(need is None or i in need) for i, f in enumerate(fields)]
        total, losses = mod(*fs, **extra)
        obj = use(total, losses)
        obj.backward(retain_graph=True)
        g1 = [None if f.grad is None else f.grad.clone() for f in fs]
        for f in fs:
            f.grad = None
        obj.backward()
        return total.detach(), g1, [None if f.grad is None else f.grad.clone() for f in fs]

    first_eq = getattr(PL, cls).EQS[0]
    uses = [lambda t, l: t, lambda t, l: 0.37 * t, lambda t, l: 2.0 * l[first_eq] + t]
    for use in uses:
        t0, ga, _ = run("auto", 1.0, use)
        for hint in (1.0, 0.37):
            t1, g1, g2 = run("fused", hint, use)
            assert abs(float(t1) - float(t0)) <= 1e-6 * abs(float(t0))
            for a, b, c in zip(ga, g1, g2):
                assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) <= 1e-6
                assert rel_l2(c.cpu().numpy(), a.cpu().numpy()) <= 1e-6
    t0, ga, _ = run("auto", 1.0, uses[0], need={0})
    t1, g1, _ = run("fused", 1.0, uses[0], need={0})
    assert g1[1] is None and rel_l2(g1[0].cpu().numpy(), ga[0].cpu().numpy()) <= 1e-6


def test_no_grad_and_partial_requires_grad():
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    u, v, p = (torch.randn(2, 5, 20, 70, generator=g).to(dev) for _ in range(3))
    mod = PL.TaylorGreen2DPDELoss(nx=70, ny=20).to(dev)
    with torch.no_grad():
        t0, _ = mod(u, v, p, nu=torch.tensor([0.02, 0.05], device=dev))
    u.requires_grad_(True)
    t1, _ = mod(u, v, p, nu=torch.tensor([0.02, 0.05], device=dev))
    (3.0 * t1).backward()
    assert torch.allclose(t0, t1.detach())
    assert u.grad is not None and v.grad is None
    # frames 0 and T-1 only receive the time-difference contribution
    assert torch.isfinite(u.grad).all()
    # validation under no_grad on tensors that require gradients: no residuals stored, no adjoint launched
    from pde_amd import functional as PF
    be = PF.get_backend()
    calls = []
    orig_fb, orig_f, orig_host = be.forward_backward, be.forward, PF._HOST
    be.forward_backward = lambda *a, **k: calls.append("fb") or orig_fb(*a, **k)
    be.forward = lambda prob, fields, aux, sp, w, store: calls.append(("f", store)) or orig_f(prob, fields, aux, sp, w, store)
    PF._HOST = "ctypes"      # the Python autograd function is the one that goes through the backend object
    try:
        for mode in ("fused", "stored"):
            mod.backward_mode = mode
            with torch.no_grad():
                t2, _ = mod(u, v, p, nu=torch.tensor([0.02, 0.05], device=dev))
            assert torch.allclose(t0, t2)
    finally:
        be.forward_backward, be.forward, PF._HOST = orig_fb, orig_f, orig_host
    assert calls == [("f", False), ("f", False)]
    # the C++ operator under no_grad: no autograd node, and no residuals or gradients allocated
    for mode in ("fused", "stored"):
        mod.backward_mode = mode
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(dev)
        with torch.no_grad():
            t3, _ = mod(u, v, p, nu=torch.tensor([0.02, 0.05], device=dev))
        assert t3.grad_fn is None and torch.allclose(t0, t3)
        assert torch.cuda.memory_allocated(dev) - before < u.numel() * 4


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["recompute", "stored", "fused"])
def test_cpp_operator_matches_python_function(mode):
    """Both host paths drive the same kernels: loss, per-equation losses, per-sample sums and gradients are bit-identical,
    also with an upstream factor, a per-equation loss in the objective and a gradient w.r.t. the per-sample sums."""
    from pde_amd import functional as PF
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    base = [torch.randn(3, 6, 24, 40, generator=g).to(dev) for _ in range(3)]
    nu = torch.tensor([0.02, 0.05, 0.03], device=dev)
    results = {}
    orig = PF._HOST
    try:
        for host in ("op", "ctypes"):
            PF._HOST = host
            for objective in ("total", "mixed"):
                fields = [b.clone().requires_grad_(i != 1) for i, b in enumerate(base)]
                mod = PL.TaylorGreen2DPDELoss(nx=40, ny=24, eq_scales={"continuity": 0.5, "x_momentum": 2.0, "y_momentum": 3.0}).to(dev)
                mod.backward_mode, mod.upstream_hint = mode, 0.1
                total, losses = mod(*fields, nu=nu)
                obj = 0.1 * total if objective == "total" else 0.3 * total + 2.0 * losses["x_momentum"] + losses["continuity"]
                obj.backward()
                results[(host, objective)] = (total.detach().clone(), [losses[k].detach().clone() for k in mod.EQS],
                                              mod.last_per_sample.clone(), fields[0].grad.clone(), fields[2].grad.clone())
                assert fields[1].grad is None
    finally:
        PF._HOST = orig
    # a gradient w.r.t. the per-sample sums (what a sample-weighted objective sends back)
    prob = mod._problem([24, 40])
    wt = torch.rand(3, 4, device=dev)
    coef = torch.tensor([[1.0, 0.5, 0.0], [2.0, 0.0, 1.0], [0.0, 3.0, 1.5]], dtype=torch.float64, device=dev)
    sums_grads = {}
    try:
        for host in ("op", "ctypes"):
            PF._HOST = host
            fields = [b.clone().requires_grad_(True) for b in base]
            totals, sums = PF.residual_losses(prob, fields, wt, [nu], (), mode=mode if mode != "fused" else ("fused", torch.ones((), device=dev)))
            ((sums * coef).sum().float() + totals[1]).backward()
            sums_grads[host] = [f.grad.clone() for f in fields]
    finally:
        PF._HOST = orig
    assert all(torch.equal(x, y) for x, y in zip(sums_grads["op"], sums_grads["ctypes"]))
    for objective in ("total", "mixed"):
        a, b = results[("op", objective)], results[("ctypes", objective)]
        assert torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


def test_cpu_tensors_are_rejected():
    mod = PL.Burgers1DPDELoss()
    with pytest.raises(RuntimeError):
        mod(torch.randn(2, 5, 64), torch.tensor([0.01, 0.02]))


def test_two_frames_gives_empty_lattice():
    dev = _dev()
    mod = PL.Heat1DPDELoss().to(dev)
    total, _ = mod(torch.randn(2, 2, 64, device=dev), torch.tensor([0.01, 0.02], device=dev))
    assert torch.isnan(total)      # mean over an empty residual lattice, as in the reference


def test_upstream_gradients_and_long_clips():
    """lambda * total plus a separately weighted equation loss; T long enough for several marching chunks."""
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    shape = (1, 23, 20, 70)
    fields = [torch.randn(shape, generator=g) * 0.5 for _ in range(3)]
    kw = dict(nx=70, ny=20, dt=0.01, eq_scales={"continuity": 0.5, "x_momentum": 2.0, "y_momentum": 3.0})
    nu = torch.tensor([0.03])
    for mode in ("auto", "stored"):
        mod = PL.TaylorGreen2DPDELoss(**kw).to(dev)
        mod.backward_mode = mode
        df = [f.to(dev).requires_grad_(True) for f in fields]
        total, losses = mod(*df, nu=nu.to(dev))
        (0.37 * total + 2.5 * losses["x_momentum"]).backward()
        f64 = [f.double().requires_grad_(True) for f in fields]
        ot, ol = OC.build("TaylorGreen2DPDELoss", **kw)(*f64, nu=nu.double())
        (0.37 * ot + 2.5 * ol["x_momentum"]).backward()
        for a, b in zip(df, f64):
            assert rel_l2(a.grad.cpu().numpy(), b.grad.numpy()) <= TOL, mode


def test_per_sample_sums_are_differentiable():
    dev = _dev()
    g = torch.Generator().manual_seed(22)
    u = (torch.randn(3, 6, 300, generator=g) * 0.5)
    nu = torch.tensor([0.01, 0.02, 0.03])
    from pde_amd.functional import StencilProblem, residual_losses
    spec = PL.SPECS["burgers1d"]
    p = dict(dx=0.02, dt=0.01)
    prob = StencilProblem("burgers1d", [fn(p) for _, fn in spec.consts])
    wt = torch.full((1, 4), 1.0 / (3 * 4 * 300), device=dev)
    ud = u.to(dev).requires_grad_(True)
    totals, sums = residual_losses(prob, [ud], wt, [nu.to(dev)])
    coeff = torch.tensor([[1.0], [-2.0], [0.5]], dtype=torch.float64, device=dev)
    (sums * coeff).sum().backward()
    u64 = u.double().requires_grad_(True)
    from oracle import pde_oracle as O
    R = O.residuals_burgers1d(u64, nu.double(), dx=0.02, dt=0.01)[0]
    per = (R ** 2).sum(dim=(1, 2)) / (3 * 4 * 300)
    (per * coeff.cpu().view(-1)).sum().backward()
    assert rel_l2(ud.grad.cpu().numpy(), u64.grad.numpy()) <= TOL
    assert rel_l2(sums.detach().cpu().numpy().ravel(), per.detach().numpy()) <= TOL


def _full_size_properties(mod, fields, extra, shifts, modes=("auto",), odd_shift=True, fd_tol=2e-3):
    """size-independent checks at benchmark sizes, where no CPU oracle run is affordable:
    (1) a periodic shift of the fields (x by a multiple of 4, the lane width) shifts the gradients bit-exactly (every site
        sees the same neighbours, so tiles, halos, wrap-around and chunk ends must not matter) and leaves the per-sample sums unchanged up to summation order;
    (2) a permutation of the samples permutes per-sample sums and gradients bit-exactly;
    (3) the gradient matches a central difference of the fp64 per-sample sums along a random direction;
    (4) every backward mode gives the same numbers."""
    dims = tuple(range(2, fields[0].dim()))

    def run(fs, ex, mode):
        mod.backward_mode = mode
        fs = [f.detach().clone().requires_grad_(True) for f in fs]
        total, _ = mod(*fs, **ex)
        total.backward()
        return total.detach(), mod.last_per_sample.clone(), [f.grad for f in fs]

    t0, s0, g0 = run(fields, extra, modes[0])
    assert torch.isfinite(t0) and all(torch.isfinite(g).all() for g in g0)
    for mode in modes[1:]:
        t1, s1, g1 = run(fields, extra, mode)
        assert abs(float(t1) - float(t0)) <= 1e-6 * abs(float(t0))
        for a, b in zip(g0, g1):
            assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), mode
    # (1)
    t1, s1, g1 = run([torch.roll(f, shifts, dims) for f in fields], extra, modes[0])
    assert float(((s1 - s0).abs() / s0.abs().clamp_min(1e-300)).max()) <= 1e-9
    for a, b in zip(g0, g1):
        assert torch.equal(torch.roll(a, shifts, dims), b)
    # an x shift that is not a multiple of the lane width moves a site to another of the lane's points, whose multiply-adds
    # the compiler may contract differently: same values up to rounding
    if odd_shift:      # not for a last axis with walls
        odd = shifts[:-1] + (shifts[-1] + 1,)
        t1, s1, g1 = run([torch.roll(f, odd, dims) for f in fields], extra, modes[0])
        for a, b in zip(g0, g1):
            assert float((torch.roll(a, odd, dims) - b).abs().max()) <= 1e-5 * float(a.abs().max())
    # (2)
    B = fields[0].shape[0]
    if B > 1:
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(fields[0].device)
        t2, s2, g2 = run([f[perm] for f in fields], {k: v[perm] for k, v in extra.items()}, modes[0])
        assert torch.equal(s2, s0[perm])
        for a, b in zip(g0, g2):
            assert torch.equal(a[perm], b)
    # (3)
    gen = torch.Generator(device=fields[0].device).manual_seed(2)
    d = [torch.randn(f.shape, generator=gen, device=f.device) for f in fields]
    scale = float(fields[0].abs().mean())
    eps = 1e-3 * scale
    with torch.no_grad():
        mod(*[f + eps * x for f, x in zip(fields, d)], **extra)
        up = mod.last_per_sample.sum(0)
        mod(*[f - eps * x for f, x in zip(fields, d)], **extra)
        dn = mod.last_per_sample.sum(0)
    fd = float((up - dn).sum() / (2 * eps))
    an = float(sum((g.double() * x.double()).sum() for g, x in zip(g0, d)))
    assert abs(fd - an) <= fd_tol * abs(an), (fd, an)


def test_full_size_taylor_green_properties():
    """the benchmark configuration: B=16, T=16, 256 x 256, three fields"""
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(31)
    fields = [torch.randn(16, 16, 256, 256, generator=gen, device=dev) * 0.5 for _ in range(3)]
    nu = 0.01 + 0.09 * torch.rand(16, generator=gen, device=dev)
    mod = PL.TaylorGreen2DPDELoss(nx=256, ny=256, dt=0.01).to(dev)
    _full_size_properties(mod, fields, dict(nu=nu), (37, 100), modes=("auto", "fused", "stored"))
    # the analytic Taylor-Green vortex solves the equations: its residual loss is at the fp32 noise floor of the stencils
    x = torch.arange(256, device=dev, dtype=torch.float64) * (2 * torch.pi / 256)
    X, Y = x.view(1, 1, 1, 256), x.view(1, 1, 256, 1)
    t = (torch.arange(16, device=dev, dtype=torch.float64) * 0.01).view(1, 16, 1, 1)
    dec = torch.exp(-2.0 * nu.double().view(16, 1, 1, 1) * t)
    u = (torch.sin(X) * torch.cos(Y) * dec).float()
    v = (-torch.cos(X) * torch.sin(Y) * dec).float()
    p = (0.25 * (torch.cos(2 * X) + torch.cos(2 * Y)) * dec ** 2).float()
    with torch.no_grad():
        exact, _ = mod(u, v, p, nu=nu)
        noisy, _ = mod(u + 0.01 * fields[0], v, p, nu=nu)
    assert float(exact) < 1e-4 * float(noisy), (float(exact), float(noisy))


def test_full_size_burgers3d_properties():
    """the 3-D benchmark configuration: one sample of 128^3, three fields (stored-residual kernels)"""
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(32)
    fields = [torch.randn(1, 9, 128, 128, 128, generator=gen, device=dev) * 0.5 for _ in range(3)]
    mod = PL.APEBenchBurgers3DPDELoss(nx=128, Lx=1.0, dt=1.0).to(dev)
    _full_size_properties(mod, fields, {}, (5, 17, 68))


def test_full_size_shear_flow_properties():
    """shear flow at its benchmark shape: 16 samples of 9 frames, 256 x 512, four fields (upwind selects: the loss has kinks,
    so the finite-difference check is looser)"""
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(33)
    fields = [torch.randn(16, 9, 256, 512, generator=gen, device=dev) * 0.5 for _ in range(4)]
    mod = PL.ShearFlowPDELossNPINN(nx=256, ny=512).to(dev)
    _full_size_properties(mod, fields, {}, (37, 100), modes=("auto", "fused"), fd_tol=2e-2)


def test_full_size_rayleigh_benard_properties():
    """Rayleigh-Benard at its benchmark shape: 16 samples of 9 frames, 512 x 128 (x periodic, walls along the last axis, so
    only x is shifted), per-sample kappa"""
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(34)
    fields = [torch.randn(16, 9, 512, 128, generator=gen, device=dev) * 0.5 for _ in range(3)]
    kappa = 1e-4 + 1e-2 * torch.rand(16, generator=gen, device=dev)
    mod = PL.RayleighBenardFullPDELoss().to(dev)
    _full_size_properties(mod, fields, dict(kappa=kappa), (37, 0), modes=("auto", "fused"), odd_shift=False)


def test_full_size_gray_scott_fd_properties():
    """Gray-Scott with the finite-difference Laplacian at the per-GPU share of its benchmark: 8 samples, 9 frames, 128 x 128"""
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(35)
    fields = [torch.rand(8, 9, 128, 128, generator=gen, device=dev) for _ in range(2)]
    mod = PL.GrayScottPDELoss(laplacian="fd4").to(dev)
    _full_size_properties(mod, fields, {}, (37, 100), modes=("auto", "fused"))


def test_full_size_heat3d_properties():
    """3-D heat equation: two samples of 128^3, 9 frames, per-sample alpha"""
    dev = _dev()
    gen = torch.Generator(device=dev).manual_seed(36)
    fields = [torch.randn(2, 9, 128, 128, 128, generator=gen, device=dev) * 0.5]
    alpha = 0.005 + 0.045 * torch.rand(2, generator=gen, device=dev)
    mod = PL.Heat3DPDELoss().to(dev)
    _full_size_properties(mod, fields, dict(alpha=alpha), (5, 17, 68), modes=("auto", "fused"))


def test_flow_mixing_v2_golden():
    """flow_mixing_pde_loss_v2 against the reference's values (eval/pde_loss_flow_v2.py run on the same inputs): the three
    losses, the residual field, and the gradients w.r.t. the prediction and all four boundary tensors."""
    z = np.load(os.path.join(GOLDEN, "flow_v2.npz"))
    m = json.loads(bytes(z["meta"]).decode())
    names = ("pred", "left", "right", "bottom", "top")
    ins = [torch.tensor(z[f"in_{n}"], dtype=torch.float32, device=_dev(), requires_grad=True) for n in names]
    loss, lt, la, res = PL.flow_mixing_pde_loss_v2(*ins, **m)
    (loss + 0.5 * lt + 0.25 * la + 0.01 * res.sum()).backward()
    for n, x in (("loss", loss), ("loss_time", lt), ("loss_adv", la), ("residual", res)):
        assert close(x, z[f"f64_{n}"], TOL), n
    for n, x in zip(names, ins):
        assert rel_l2(x.grad.cpu().numpy(), z[f"f64_grad_{n}"]) <= TOL, n


def test_ghost_burgers_residuals_are_in_graph():
    """The residual fields burgers_pde_loss returns are differentiable (finetune/pde_loss.py:216): objectives built from
    them, alone or mixed with the losses, against autograd through the fp64 oracle."""
    dev = _dev()
    g = torch.Generator().manual_seed(12)
    base = torch.randn(2, 4, 9, 12, 2, generator=g)
    wgt = torch.randn(2, 3, 7, 10, generator=g)
    objectives = (lambda t, lu, ru, rv, w: (ru * w).sum() + 0.5 * (rv ** 2).mean(),
                  lambda t, lu, ru, rv, w: 2.0 * t + lu + (rv * w).sum())
    for use in objectives:
        pred = base.clone().to(dev).requires_grad_(True)
        t, lu, lv, ru, rv = PL.burgers_pde_loss(pred, 0.03, dt=0.02, dx=0.11, dy=0.13)
        use(t, lu, ru, rv, wgt.to(dev)).backward()
        p64 = base.double().requires_grad_(True)
        ot, olu, olv, oru, orv = OC.burgers_pde_loss(p64, 0.03, dt=0.02, dx=0.11, dy=0.13)
        use(ot, olu, oru, orv, wgt.double()).backward()
        assert rel_l2(pred.grad.cpu().numpy(), p64.grad.numpy()) <= TOL


@pytest.mark.parametrize("host", ["op", "ctypes"])
def test_scalars_out_on_the_device(host):
    """`scalars_out` through the C++ operator and through the Python function: the losses land in the caller's vector
    (a row of a larger buffer), the returned scalars are views of it, values and gradients as without it."""
    from pde_amd import functional as PF
    dev = _dev()
    g = torch.Generator().manual_seed(19)
    fields = [torch.randn(2, 5, 20, 72, generator=g).to(dev) for _ in range(3)]
    nu = torch.tensor([0.02, 0.05], device=dev)
    old, PF._HOST = PF._HOST, host
    try:
        for mode in ("auto", "stored", "fused"):
            ref = PL.TaylorGreen2DPDELoss(nx=72, ny=20, dt=0.01).to(dev)
            ref.backward_mode = mode
            t0, l0, g0 = _run(ref, fields, dict(nu=nu))
            mod = PL.TaylorGreen2DPDELoss(nx=72, ny=20, dt=0.01).to(dev)
            mod.backward_mode = mode
            window = torch.full((3, 4), -1.0, device=dev)
            mod.scalars_out = window[1]
            t1, l1, g1 = _run(mod, fields, dict(nu=nu))
            assert t1.data_ptr() == window[1, 3].data_ptr() and l1["x_momentum"].data_ptr() == window[1, 1].data_ptr()
            assert window[1].tolist() == [float(l0["continuity"]), float(l0["x_momentum"]), float(l0["y_momentum"]), float(t0)]
            assert window[0].tolist() == [-1.0] * 4 and window[2].tolist() == [-1.0] * 4
            for a, b in zip(g0, g1):
                assert torch.equal(a, b)
        mod.scalars_out = torch.zeros(3, device=dev)
        with pytest.raises((ValueError, RuntimeError)):
            mod(*fields, nu=nu)
    finally:
        PF._HOST = old
