"""Generate tests/golden/*.npz by running the reference implementation (build container only).

The reference tree (/root/reference) is imported here, fed small seeded inputs, and its outputs
(per-equation losses, total, gradients w.r.t. every field) are stored next to the inputs.  Only
data lands in the repository: no reference source is copied.  The GPU box never runs this script.

Usage:  python tools/make_golden.py            (rewrites every fixture)
"""
from __future__ import annotations

import ast
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _fields(shape, n, seed, dtype=torch.float64, smooth=True):
    """n seeded random fields: band-limited cosines plus white noise (keeps residuals O(1))."""
    g = torch.Generator().manual_seed(seed)
    out = []
    nd = len(shape) - 2
    for _ in range(n):
        f = 0.3 * torch.randn(shape, generator=g, dtype=dtype)
        if smooth:
            for ax in range(nd):
                N = shape[2 + ax]
                x = torch.arange(N, dtype=dtype) * (2 * np.pi / N)
                k = int(torch.randint(1, 3, (1,), generator=g))
                ph = float(torch.rand(1, generator=g)) * 6.28
                view = [1] * len(shape)
                view[2 + ax] = N
                f = f + torch.cos(k * x + ph).view(view)
            tt = torch.arange(shape[1], dtype=dtype).view(1, -1, *([1] * nd))
            f = f * torch.exp(-0.05 * tt)
        out.append(f)
    return out


def _uniform(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(n, generator=g, dtype=torch.float64)


# case table: name -> (reference class, ctor kwargs, field names, shape, extra forward inputs)
def cases():
    S1, S2, S3 = (2, 6, 24), (2, 5, 12, 10), (2, 5, 6, 5, 7)
    pert = lambda names, n, seed: {k: (0.5 + torch.rand(n, generator=torch.Generator().manual_seed(seed + i)))
                                   for i, k in enumerate(names)}
    C = {}
    C["burgers1d"] = ("Burgers1DPDELoss", dict(dx=0.26, dt=0.01, eq_scales={"momentum": 0.7}, eq_weights={"momentum": 1.3}),
                      ["u"], S1, dict(nu=_uniform(2, 0.005, 0.1, 1)))
    C["burgers1d_pert"] = ("Burgers1DPDELoss", dict(dx=0.26, dt=0.01, eq_scales_per_t=pert(["momentum"], 9, 7)),
                           ["u"], S1, dict(nu=_uniform(2, 0.005, 0.1, 1)))
    C["advection1d"] = ("Advection1DPDELoss", dict(dx=0.26, dt=0.01), ["u"], S1, dict(a=_uniform(2, -1, 1, 2)))
    C["heat1d"] = ("Heat1DPDELoss", dict(dx=0.26, dt=0.01, eq_scales={"heat": 2.0}), ["u"], S1, dict(alpha=_uniform(2, 0.005, 0.05, 3)))
    C["ape_burgers1d"] = ("APEBenchBurgers1DPDELoss", dict(nx=24, Lx=1.0, dt=1.0), ["u"], S1, {})
    C["ks1d"] = ("KS1DPDELoss", dict(nx=24, Lx=1.0, dt=1.0), ["u"], S1, {})
    C["kdv1d"] = ("KdV1DPDELoss", dict(nx=24, Lx=1.0, dt=1.0, eq_scales={"kdv": 0.5}), ["u"], S1, {})
    tgk = dict(nx=10, ny=12, Lx=2.0, Ly=3.0, dt=0.01,
               eq_scales={"continuity": 0.5, "x_momentum": 2.0, "y_momentum": 3.0},
               eq_weights={"continuity": 1.0, "x_momentum": 0.7, "y_momentum": 1.1})
    C["tg2d"] = ("TaylorGreen2DPDELoss", tgk, ["u", "v", "p"], S2, dict(nu=_uniform(2, 0.01, 0.1, 4)))
    C["tg2d_nonu"] = ("TaylorGreen2DPDELoss", tgk, ["u", "v", "p"], S2, {})
    C["tg2d_w0"] = ("TaylorGreen2DPDELoss", dict(tgk, eq_weights={"y_momentum": 0.0}), ["u", "v", "p"], S2,
                    dict(nu=_uniform(2, 0.01, 0.1, 4)))
    C["tg2d_pert"] = ("TaylorGreen2DPDELoss", dict(tgk, eq_scales_per_t=pert(["continuity", "x_momentum"], 8, 11)),
                      ["u", "v", "p"], S2, dict(nu=_uniform(2, 0.01, 0.1, 4)))
    C["tg2d_pert_off"] = ("TaylorGreen2DPDELoss", dict(tgk, eq_scales_per_t=pert(["continuity", "x_momentum", "y_momentum"], 8, 11)),
                          ["u", "v", "p"], S2, dict(nu=_uniform(2, 0.01, 0.1, 4), t_offset=torch.tensor([2, 7])))
    C["advdiff2d"] = ("AdvDiff2DPDELoss", dict(dx=0.2, dy=0.3, dt=0.01, a=1.0, b=0.5, nu=0.05), ["u"], S2, {})
    C["anisodiff2d"] = ("AnisoDiff2DPDELoss", dict(dx=0.2, dy=0.3, dt=0.1), ["u"], S2, {})
    C["wave2d"] = ("Wave2DPDELoss", dict(dx=0.2, dy=0.3, dt=0.01, eq_scales={"u_equation": 2.0, "w_equation": 5.0}),
                   ["u", "w"], S2, dict(c=_uniform(2, 0.5, 2.0, 5)))
    C["ape_burgers2d"] = ("APEBenchBurgers2DPDELoss", dict(nx=10, ny=12, Lx=1.0, Ly=1.5, dt=1.0), ["u", "v"], S2, {})
    C["nspwc"] = ("NSPwCPDELoss", dict(nx=10, ny=12, Lx=1.0, Ly=1.5, dt=0.05, nu=4e-4, kappa=3e-4), ["u", "v", "s"], S2, {})
    C["kp2d"] = ("KP2DPDELoss", dict(dx=0.3, dy=0.2, dt=0.01, skip_boundary=2), ["u"], (2, 5, 14, 12), {})
    sfk = dict(nx=12, ny=10, Lx=1.0, Ly=2.0, dt=0.1, nu=1e-2, D=1e-3,
               eq_scales={"continuity": 0.5, "x_momentum": 2.0, "y_momentum": 3.0, "tracer": 1.5})
    C["shear_flow"] = ("ShearFlowPDELossNPINN", sfk, ["u", "v", "p", "s"], S2, {})
    C["shear_flow_nocorr"] = ("ShearFlowPDELossNPINN", dict(sfk, use_div_correction=False), ["u", "v", "p", "s"], S2, {})
    C["burgers2d_ch"] = ("Burgers2DCHPDELoss", dict(nx=10, ny=12, Lx=2.0, Ly=3.0, dt=0.01), ["u", "v"], S2,
                         dict(nu=_uniform(2, 0.01, 0.1, 6)))
    C["active_matter"] = ("ActiveMatterNPINNPDELoss", dict(nx=10, ny=12, Lx=2.0, Ly=3.0, dt=0.25, d_T=0.05),
                          ["u", "v", "c", "Dxx"], S2, {})
    C["rayleigh_benard"] = ("RayleighBenardFullPDELoss", dict(nx=12, ny=10, Lx=4.0, Ly=1.0, dt=0.25, skip_bl=2),
                            ["u", "v", "b"], S2, dict(kappa=_uniform(2, 1e-3, 1e-2, 7)))
    C["rayleigh_benard_noskip"] = ("RayleighBenardFullPDELoss", dict(nx=12, ny=10, Lx=4.0, Ly=1.0, dt=0.25, skip_bl=0, kappa=2e-3),
                                   ["u", "v", "b"], S2, {})
    C["turbulent_radiative"] = ("TurbulentRadiativePDELoss", dict(nx=12, ny=10, Lx=1.0, Ly=3.0, dt=1.5, skip_bl=1),
                                ["vx", "vy", "rho", "P"], S2, {})
    C["wave_gauss"] = ("WaveGaussPDELoss", dict(nx=12, ny=10, Lx=1.0, Ly=1.0, dt=1 / 14, skip_boundary=1),
                       ["u"], S2, dict(c="__c_field__"))
    C["gray_scott"] = ("GrayScottPDELoss", dict(nx=12, ny=10, dx=0.1, dy=0.15, dt=10.0, eq_scales={"A": 2.0, "A_equation": 3.0}),
                       ["A", "B"], S2, {})
    C["advection3d"] = ("Advection3DPDELoss", dict(dx=0.2, dy=0.3, dz=0.25, dt=0.05), ["u"], S3,
                        dict(a=_uniform(2, -1, 1, 8), b=_uniform(2, -1, 1, 9), c=_uniform(2, -1, 1, 10)))
    C["ape_burgers3d"] = ("APEBenchBurgers3DPDELoss", dict(nx=6, Lx=1.0, dt=1.0, eq_scales={"u_momentum": 0.1}), ["u", "v", "w"], S3, {})
    # reductions other than the mean (:286-294), and the spectral legacy classes (:28-146, :309-393)
    C["gray_scott_sum"] = ("GrayScottPDELoss", dict(nx=12, ny=10, dx=0.2, dy=0.2, dt=10.0, eq_scales={"A": 0.5, "B": 2.0}, eq_weights={"A": 1.5, "B": 0.5}),
                           ["A", "B"], S2, dict(reduction="sum"))
    C["gray_scott_none"] = ("GrayScottPDELoss", dict(nx=12, ny=10, dx=0.2, dy=0.2, dt=10.0, eq_scales={"A": 0.5, "B": 2.0}),
                            ["A", "B"], S2, dict(reduction="none"))
    C["active_matter_fft"] = ("ActiveMatterPDELoss", dict(nx=12, ny=10, dx=0.3, dy=0.25, dt=0.5), ["c", "u", "v"], S2, {})
    C["active_matter_fft_sum"] = ("ActiveMatterPDELoss", dict(nx=12, ny=10, dx=0.3, dy=0.25, dt=0.5), ["c", "u", "v"], S2, dict(reduction="sum"))
    C["viscoelastic_cont"] = ("ViscoelasticContinuityLoss", dict(nx=12, ny=10, dx=0.3, dy=0.25), ["u", "v"], S2, {})
    C["viscoelastic_cont_none"] = ("ViscoelasticContinuityLoss", dict(nx=12, ny=10, dx=0.3, dy=0.25), ["u", "v"], S2, dict(reduction="none"))
    # [T, X, Y] inputs without a batch axis: the class never adds one, so reduction='none' returns [T, X, Y-2] fields
    C["viscoelastic_cont_none_3d"] = ("ViscoelasticContinuityLoss", dict(nx=12, ny=10, dx=0.3, dy=0.25), ["u", "v"], (1, 5, 12, 10),
                                      dict(reduction="none", __unbatched__=True))
    C["swift_hohenberg3d"] = ("SwiftHohenberg3DPDELoss", dict(domain_extent=6.0, num_points=6, dt=0.1), ["u"], S3, {})
    return C


def run_case(name, spec, mod):
    cls_name, kwargs, fnames, shape, extra = spec
    extra = dict(extra)
    unbatched = extra.pop("__unbatched__", False)      # the fields are generated with a batch of one and handed over without it
    seed = abs(hash(name)) % 10000 if False else sum(ord(ch) for ch in name)
    out = {}
    for dt_name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        fields = [(f[0] if unbatched else f).to(dtype).requires_grad_(True) for f in _fields(shape, len(fnames), seed)]
        kw = {}
        for k, v in extra.items():
            if isinstance(v, str) and v == "__c_field__":
                g = torch.Generator().manual_seed(seed + 99)
                v = 1.0 + torch.rand((shape[0],) + tuple(shape[2:]), generator=g, dtype=torch.float64)
            kw[k] = v.to(dtype) if (torch.is_tensor(v) and v.is_floating_point()) else v
        ck = dict(kwargs)
        loss_fn = getattr(mod, cls_name)(**ck)
        if dtype == torch.float64:
            loss_fn = loss_fn.double()
        total, d = loss_fn(*fields, **kw)
        (total.sum() if total.dim() else total).backward()      # reduction='none' returns fields: their sum is differentiated
        if dt_name == "f64":
            for n, f in zip(fnames, fields):
                out[f"in_{n}"] = f.detach().numpy()
            for k, v in kw.items():
                if torch.is_tensor(v):
                    out[f"arg_{k}"] = v.numpy()
        out[f"{dt_name}_total"] = total.detach().numpy()
        for k, v in d.items():
            out[f"{dt_name}_loss_{k}"] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)
        for n, f in zip(fnames, fields):
            out[f"{dt_name}_grad_{n}"] = f.grad.numpy() if f.grad is not None else np.zeros(tuple(f.shape))

    def jsonable(v):
        if torch.is_tensor(v):
            return {"__tensor__": v.tolist()}
        if isinstance(v, dict):
            return {k: jsonable(x) for k, x in v.items()}
        return v
    meta = dict(cls=cls_name, kwargs=jsonable(kwargs), fields=fnames, shape=list(shape), extra_keys=list(extra),
                plain_extra={k: v for k, v in extra.items() if isinstance(v, str) and v != "__c_field__"})
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out


def run_burgers_ghost():
    from finetune.pde_loss import burgers_pde_loss
    out = {}
    shape = (2, 4, 9, 8, 2)
    for dt_name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        g = torch.Generator().manual_seed(321)
        pred = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype).requires_grad_(True)
        total, lu, lv, ru, rv = burgers_pde_loss(pred, 0.03, dt=0.02, dx=0.11, dy=0.13)
        total.backward()
        if dt_name == "f64":
            out["in_pred"] = pred.detach().numpy()
        out[f"{dt_name}_total"] = total.detach().numpy()
        out[f"{dt_name}_loss_u"] = lu.detach().numpy()
        out[f"{dt_name}_loss_v"] = lv.detach().numpy()
        out[f"{dt_name}_res_u"] = ru.detach().numpy()
        out[f"{dt_name}_res_v"] = rv.detach().numpy()
        out[f"{dt_name}_grad_pred"] = pred.grad.numpy()
    meta = dict(nu=0.03, dt=0.02, dx=0.11, dy=0.13)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out


def run_flow_v2():
    """eval/pde_loss_flow_v2.py: the three losses, the residual field and the gradients w.r.t. the prediction and the
    boundary values (the module imports torch only)."""
    sys.path.insert(0, os.path.join(REF, "eval"))
    from pde_loss_flow_v2 import flow_mixing_pde_loss_v2
    out = {}
    B, T, H, W = 2, 5, 9, 12
    meta = dict(vtmax=0.4, dt=0.01, Lx=1.0, Ly=0.8)
    for dt_name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        g = torch.Generator().manual_seed(2024)
        mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dtype).requires_grad_(True)
        pred, bl, br, bb, bt = mk(B, T, H, W, 1), mk(B, T, H, 1, 1), mk(B, T, H, 1, 1), mk(B, T, 1, W, 1), mk(B, T, 1, W, 1)
        loss, lt, la, res = flow_mixing_pde_loss_v2(pred, bl, br, bb, bt, **meta)
        (loss + 0.5 * lt + 0.25 * la + 0.01 * res.sum()).backward()
        if dt_name == "f64":
            for n, x in (("pred", pred), ("left", bl), ("right", br), ("bottom", bb), ("top", bt)):
                out[f"in_{n}"] = x.detach().numpy()
        for n, x in (("loss", loss), ("loss_time", lt), ("loss_adv", la), ("residual", res)):
            out[f"{dt_name}_{n}"] = x.detach().numpy()
        for n, x in (("pred", pred), ("left", bl), ("right", br), ("bottom", bb), ("top", bt)):
            out[f"{dt_name}_grad_{n}"] = x.grad.numpy()
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    return out


def load_function(path, name, extra_globals=None):
    """Compile ONE top-level function out of a reference script whose module-level imports are unavailable."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            node.decorator_list = []
            m = ast.Module(body=[node], type_ignores=[])
            glb = {"torch": torch, "np": np, "Optional": __import__("typing").Optional,
                   "Tuple": __import__("typing").Tuple, "Dict": __import__("typing").Dict,
                   "List": __import__("typing").List}
            glb.update(extra_globals or {})
            exec(compile(m, path, "exec"), glb)
            return glb[name]
    raise KeyError(name)


def run_glue():
    """Boundary / no-slip / patch-seam losses and the small metric helpers, executed from the reference text."""
    out = {}
    g = torch.Generator().manual_seed(77)
    B, T, H, W, C = 2, 3, 12, 10, 18
    pred = torch.randn(B, T, H, W, C, generator=g, dtype=torch.float64)
    tgt = torch.randn(B, T, H, W, C, generator=g, dtype=torch.float64)
    mask = torch.zeros(B, C)
    mask[:, [0, 1, 15]] = 1.0
    out["pred"], out["target"], out["mask"] = pred.numpy(), tgt.numpy(), mask.numpy()

    def with_grad(fn, *a, **k):
        p = pred.clone().requires_grad_(True)
        val = fn(p, *a, **k)
        val = val if torch.is_tensor(val) else val[0]
        val.backward()
        return val.detach().numpy(), p.grad.numpy()

    f = load_function("finetune/train_taylor_green_2d_lora_v3.py", "compute_boundary_loss")
    for w in (1, 2):
        out[f"bc_strip_w{w}"], out[f"bc_strip_w{w}_grad"] = with_grad(f, tgt, mask, w)
    f = load_function("finetune/train_shear_flow_lora_v3.py", "compute_boundary_loss")
    out["bc_edge"], out["bc_edge_grad"] = with_grad(f, tgt, mask)
    f = load_function("finetune/train_taylor_green_2d_lora_v3.py", "compute_vrmse")
    v, r = f(pred, tgt, mask)
    out["vrmse"], out["rmse"] = v.numpy(), r.numpy()
    f = load_function("finetune/train_ns_incom_lora_v3.py", "compute_noslip_loss", dict(CH_VX=0, CH_VY=1))
    try:
        out["noslip"], out["noslip_grad"] = with_grad(f)
    except TypeError:
        pass
    # 1-D and 3-D boundary losses
    p1 = torch.randn(2, 4, 16, C, generator=g, dtype=torch.float64)
    t1 = torch.randn(2, 4, 16, C, generator=g, dtype=torch.float64)
    out["pred1d"], out["target1d"] = p1.numpy(), t1.numpy()
    f = load_function("finetune/train_burgers_1d_lora_v3.py", "compute_boundary_loss")
    q = p1.clone().requires_grad_(True)
    val = f(q, t1, mask)
    val.backward()
    out["bc_1d"], out["bc_1d_grad"] = val.detach().numpy(), q.grad.numpy()
    # Y >= 16: the 6-face variant indexes the Y axis with the channel list on its z faces
    p3 = torch.randn(2, 2, 4, 16, 5, C, generator=g, dtype=torch.float64)
    t3 = torch.randn(2, 2, 4, 16, 5, C, generator=g, dtype=torch.float64)
    out["pred3d"], out["target3d"] = p3.numpy(), t3.numpy()
    for tag, path in (("faces", "finetune/train_heat_3d_lora_v3.py"), ("shells", "finetune/train_advection_3d_lora_v3.py")):
        f = load_function(path, "compute_boundary_loss")
        q = p3.clone().requires_grad_(True)
        try:
            val = f(q, t3, mask)
        except TypeError:
            val = f(q, t3, mask, 1)
        val.backward()
        out[f"bc_3d_{tag}"], out[f"bc_3d_{tag}_grad"] = val.detach().numpy(), q.grad.numpy()
    # patch-seam curvature (Rayleigh-Benard trainer)
    ps = torch.randn(2, 3, 32, 48, C, generator=g, dtype=torch.float64)
    out["pred_patch"] = ps.numpy()
    try:
        f = load_function("finetune/train_rayleigh_benard_lora_v3.py", "compute_patch_boundary_loss")
        q = ps.clone().requires_grad_(True)
        import inspect
        sig = inspect.signature(f)
        out["patch_sig"] = np.frombuffer(str(sig).encode(), dtype=np.uint8)
        val = f(q, mask) if len(sig.parameters) >= 2 and "channel_mask" in sig.parameters else f(q)
        val.backward()
        out["patch"], out["patch_grad"] = val.detach().numpy(), q.grad.numpy()
    except Exception as e:  # recorded, not hidden
        out["patch_error"] = np.frombuffer(repr(e).encode(), dtype=np.uint8)
    # rollout nRMSE and RSE helpers
    f = load_function("tools/eval_rollout.py", "nrmse_per_sample")
    out["nrmse_per_sample"] = f(pred[:, 0, ..., 3], tgt[:, 0, ..., 3]).numpy()
    f2 = load_function("tools/eval_rollout.py", "nrmse_per_channel", dict(nrmse_per_sample=f))
    try:
        r = f2(pred[:, 0], tgt[:, 0], [(0, "vx"), (1, "vy"), (15, "p")])
        out["nrmse_per_channel"] = np.stack([r[k].numpy() for k in ("vx", "vy", "p")])
    except Exception as e:
        out["nrmse_per_channel_error"] = np.frombuffer(repr(e).encode(), dtype=np.uint8)
    f = load_function("eval/eval_autoregressive.py", "compute_rse")
    out["rse"] = f(pred[..., :3], tgt[..., :3]).numpy()
    f = load_function("eval/eval_autoregressive.py", "compute_mean_rse", dict(compute_rse=f))
    out["mean_rse"] = np.asarray(float(f(pred[..., :3], tgt[..., :3])))
    # PDEBench-style metrics: per-channel nRMSE with the global std (eval/eval_omni.py) and the Lp / Fourier-Lp losses
    f = load_function("eval/eval_omni.py", "compute_pdebench_nrmse")
    avg, per = f(pred[:, 0], tgt[:, 0])
    out["pdebench_nrmse_avg"], out["pdebench_nrmse"] = np.asarray(float(avg)), np.asarray([float(v) for v in per])
    from eval.metrics import LpLoss, FftLpLoss
    x, y = pred[:, 0, ..., 2], tgt[:, 0, ..., 2]              # [B, H, W]
    for pn in (1, 2, 3):
        out[f"lp{pn}_none"] = LpLoss(p=pn, reduction="none")(x, y).numpy()
        out[f"lp{pn}_mean"] = LpLoss(p=pn, reduction="mean")(x, y).numpy()
        out[f"lp{pn}_sum"] = LpLoss(p=pn, reduction="sum")(x, y).numpy()
    out["fftlp2_none"] = FftLpLoss(p=2, reduction="none")(x, y).numpy()
    out["fftlp2_band_mean"] = FftLpLoss(p=2, reduction="mean")(x, y, flow=1, fhigh=5).numpy()
    out["fftlp1_sum"] = FftLpLoss(p=1, reduction="sum")(x, y).numpy()
    return out


def run_metric_func():
    from eval.metrics import metric_func
    g = torch.Generator().manual_seed(5)
    out = {}
    for tag, shape in (("1d", (3, 16, 5, 2)), ("2d", (2, 12, 10, 5, 3)), ("3d", (2, 6, 5, 7, 4, 2))):
        pred = torch.randn(shape, generator=g)
        tgt = torch.randn(shape, generator=g)
        out[f"{tag}_pred"], out[f"{tag}_target"] = pred.numpy(), tgt.numpy()
        res = metric_func(pred.clone(), tgt.clone(), if_mean=True, Lx=1.0, Ly=1.0, Lz=1.0, iLow=2, iHigh=4)
        for i, r in enumerate(res):
            out[f"{tag}_mean_{i}"] = r.detach().cpu().numpy()
        res = metric_func(pred.clone(), tgt.clone(), if_mean=False, Lx=1.0, Ly=1.0, Lz=1.0, iLow=2, iHigh=4)
        for i, r in enumerate(res):
            out[f"{tag}_full_{i}"] = r.detach().cpu().numpy()
    return out


def load_class(path, name, extra_globals=None):
    """Compile ONE top-level class out of a reference script whose module-level imports are unavailable."""
    src = open(os.path.join(REF, path)).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == name:
            glb = {"torch": torch, "nn": torch.nn, "np": np, "Dict": __import__("typing").Dict,
                   "Tuple": __import__("typing").Tuple, "Optional": __import__("typing").Optional}
            glb.update(extra_globals or {})
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), glb)
            return glb[name]
    raise KeyError(name)


def run_pins():
    """Fixtures from the reference's own verification code for what has no class in the reference tree (or lives inside
    a driver loop): the FD-c4 Gray-Scott residuals, the 3-D heat and 3-D Taylor-Green verifiers, the n-PINN residuals of
    the inhomogeneous incompressible NS test, the periodic-BC loss, the eval_nrmse loop, the boundary loss that the
    3-D Burgers trainer applies to a 6-D tensor, and the one-step / rollout evaluation loop."""
    out = {}
    g = torch.Generator().manual_seed(2024)
    rnd = lambda *shape: torch.randn(*shape, generator=g, dtype=torch.float64)

    # -- Gray-Scott FD-c4: tools/test_gt_pde_gray_scott.py:27-79 (NumPy, one sample [T, H, W] at a time)
    lap = load_function("tools/test_gt_pde_gray_scott.py", "fd_laplacian_4th")
    res = load_function("tools/test_gt_pde_gray_scott.py", "compute_residuals", dict(fd_laplacian_4th=lap))
    A, Bq = _fields((2, 5, 12, 10), 2, 41)
    par = dict(dx=0.11, dy=0.07, dt=10.0, F=0.098, k=0.057, D_A=1.81e-5, D_B=1.39e-5)
    rs = [res(A[i].numpy(), Bq[i].numpy(), **par) for i in range(2)]
    out["gsfd_A"], out["gsfd_B"] = A.numpy(), Bq.numpy()
    out["gsfd_params"] = np.frombuffer(json.dumps(par).encode(), dtype=np.uint8)
    out["gsfd_R_A"] = np.stack([r["A_equation"] for r in rs])
    out["gsfd_R_B"] = np.stack([r["B_equation"] for r in rs])

    # -- heat 3-D: finetune/generate_analytical_3d.py:160-178 (mean squared residual of one sample [T, X, Y, Z])
    vh = load_function("finetune/generate_analytical_3d.py", "verify_3d_heat")
    (u,) = _fields((2, 5, 6, 5, 7), 1, 42)
    par = dict(dx=0.2, dy=0.3, dz=0.25, dt=0.01)
    alpha = [0.013, 0.041]
    out["heat3d_u"], out["heat3d_alpha"] = u.numpy(), np.asarray(alpha)
    out["heat3d_params"] = np.frombuffer(json.dumps(par).encode(), dtype=np.uint8)
    out["heat3d_mse"] = np.asarray([vh(u[i].numpy(), alpha[i], **par) for i in range(2)])

    # -- Taylor-Green 3-D: finetune/generate_analytical_3d.py:97-157 (face-velocity divergence on every frame,
    #    c2 primitive-form u momentum on interior frames)
    vt = load_function("finetune/generate_analytical_3d.py", "verify_3d_taylor_green")
    u3, v3, w3, p3 = _fields((2, 5, 6, 5, 7), 4, 43)
    nu3 = [0.02, 0.07]
    for n, f in zip("uvwp", (u3, v3, w3, p3)):
        out[f"tg3d_{n}"] = f.numpy()
    out["tg3d_nu"] = np.asarray(nu3)
    out["tg3d_params"] = np.frombuffer(json.dumps(par).encode(), dtype=np.uint8)
    r = [vt(u3[i].numpy(), v3[i].numpy(), w3[i].numpy(), p3[i].numpy(), nu3[i], **par) for i in range(2)]
    out["tg3d_div_mse"] = np.asarray([x["div"] for x in r])
    out["tg3d_u_momentum_mse"] = np.asarray([x["u_momentum"] for x in r])

    # -- incompressible NS with forcing, n-PINN form: tools/test_gt_pde_ns_incom.py:258-428.  The function only returns
    #    RMS values; its `rms` helper is replaced by one that also keeps the array it is given, in call order:
    #    div, R_u, du_dt, R_v, dv_dt, R_d, dd_dt, ...
    kept = []

    def rms(x):
        kept.append(np.array(x))
        return float(np.sqrt(np.mean(x ** 2)))
    mb = load_function("tools/test_gt_pde_ns_incom.py", "method_b_npinn_conservative", dict(rms=rms))
    vx, vy, dd = _fields((2, 5, 12, 10), 3, 44)
    fx, fy = rnd(2, 12, 10), rnd(2, 12, 10)
    par = dict(dx=0.09, dy=0.13, dt=0.005, nu=0.01, skip=2)
    for n, f in (("vx", vx), ("vy", vy), ("d", dd), ("fx", fx), ("fy", fy)):
        out[f"nsincom_{n}"] = f.numpy()
    out["nsincom_params"] = np.frombuffer(json.dumps(par).encode(), dtype=np.uint8)
    rows = {k: [] for k in ("div", "R_u", "R_v", "R_d")}
    scal = {k: [] for k in ("cont_rms", "mom_x_rms", "mom_y_rms", "tracer_rms")}
    for i in range(2):
        kept.clear()
        r = mb(vx[i].numpy(), vy[i].numpy(), dd[i].numpy(), fx[i].numpy(), fy[i].numpy(), **par)
        rows["div"].append(kept[0]); rows["R_u"].append(kept[1]); rows["R_v"].append(kept[3]); rows["R_d"].append(kept[5])
        for k in scal:
            scal[k].append(r[k])
    for k, v in rows.items():
        out[f"nsincom_{k}"] = np.stack(v)
    for k, v in scal.items():
        out[f"nsincom_{k}"] = np.asarray(v)

    # -- periodic-BC loss: tools/test_gt_periodic_bc.py:38-105
    PBC = load_class("tools/test_gt_periodic_bc.py", "PeriodicBCLoss")
    fa, fb = rnd(2, 3, 9, 8).requires_grad_(True), rnd(2, 3, 9, 8).requires_grad_(True)
    tot, d = PBC(dx=0.1, dy=0.2)({"u": fa, "s": fb})
    tot.backward()
    out["pbc_u"], out["pbc_s"] = fa.detach().numpy(), fb.detach().numpy()
    out["pbc_total"] = tot.detach().numpy()
    out["pbc_keys"] = np.frombuffer(json.dumps(list(d)).encode(), dtype=np.uint8)
    out["pbc_values"] = np.asarray([float(v) for v in d.values()])
    out["pbc_grad_u"], out["pbc_grad_s"] = fa.grad.numpy(), fb.grad.numpy()

    # -- eval_nrmse.evaluate: eval/eval_nrmse.py:74-212 around a stub model / dataloader / accelerator
    class _Acc:
        device = torch.device("cpu")

        def wait_for_everyone(self):
            pass

        def reduce(self, t, reduction="sum"):
            return t

    t_input, C = 3, 18
    batches, preds = [], []
    mask = torch.zeros(2, C)
    mask[:, [0, 4, 15]] = 1.0
    for _ in range(3):
        data = torch.randn(2, t_input + 1, 6, 5, C, generator=g)
        batches.append({"data": data, "channel_mask": mask})
        preds.append((torch.randn(2, t_input, 6, 5, C, generator=g), torch.randn(2, 1, 1, 1, C, generator=g),
                      0.5 + torch.rand(2, 1, 1, 1, C, generator=g)))

    class _Model:
        def __init__(self):
            self.i = 0

        def eval(self):
            pass

        def train(self):
            pass

        def __call__(self, x, return_normalized=True):
            r = preds[self.i]
            self.i += 1
            return r
    ev = load_function("eval/eval_nrmse.py", "evaluate", dict(Accelerator=object))
    r = ev(_Model(), batches, _Acc(), t_input, per_timestep=True)
    out["nrmse_data"] = np.stack([b["data"].numpy() for b in batches])
    out["nrmse_mask"] = mask.numpy()
    out["nrmse_out_norm"] = np.stack([p[0].numpy() for p in preds])
    out["nrmse_mean"] = np.stack([p[1].numpy() for p in preds])
    out["nrmse_std"] = np.stack([p[2].numpy() for p in preds])
    chs = sorted(r["per_channel_nrmse"])
    out["nrmse_channels"] = np.asarray(chs)
    for k in ("per_channel_nrmse", "per_channel_rmse", "per_channel_gt_rms"):
        out[f"nrmse_{k}"] = np.asarray([r[k][c] for c in chs])
    out["nrmse_overall"] = np.asarray(r["overall_nrmse"])
    out["nrmse_num_samples"] = np.asarray(r["num_samples"])
    out["nrmse_per_timestep"] = np.asarray([r["per_timestep_nrmse"][c] for c in chs])

    # -- the 2-D strip loss applied to a 6-D tensor: finetune/train_burgers_3d_apebench_lora.py:115-150
    #    (the channel list indexes the Z axis, all channels are kept; Z must exceed the largest valid channel)
    f = load_function("finetune/train_burgers_3d_apebench_lora.py", "compute_boundary_loss")
    m6 = torch.zeros(6)
    m6[[0, 1, 2]] = 1.0
    p6, t6 = rnd(2, 2, 6, 5, 4, 6), rnd(2, 2, 6, 5, 4, 6)
    out["bc6d_pred"], out["bc6d_target"], out["bc6d_mask"] = p6.numpy(), t6.numpy(), m6.numpy()
    for w in (1, 2):
        q = p6.clone().requires_grad_(True)
        val = f(q, t6, m6, w)
        val.backward()
        out[f"bc6d_w{w}"], out[f"bc6d_w{w}_grad"] = val.detach().numpy(), q.grad.numpy()

    # -- one-step and rollout evaluation loop: tools/eval_rollout.py:94-146 around a stub model whose output depends on
    #    its input (the script's own imports, h5py among them, are not needed by the function)
    import importlib.util      # by path: the reference tree on sys.path has a `tests` package of its own
    hspec = importlib.util.spec_from_file_location("pde_amd_test_helpers", os.path.join(os.path.dirname(OUT), "helpers.py"))
    helpers = importlib.util.module_from_spec(hspec)
    hspec.loader.exec_module(helpers)
    RolloutStubModel = helpers.RolloutStubModel
    nps = load_function("tools/eval_rollout.py", "nrmse_per_sample")
    npc = load_function("tools/eval_rollout.py", "nrmse_per_channel", dict(nrmse_per_sample=nps))
    ev = load_function("tools/eval_rollout.py", "evaluate", dict(nrmse_per_channel=npc))
    g2 = torch.Generator().manual_seed(515)
    data = torch.randn(3, 7, 6, 5, 18, generator=g2, dtype=torch.float64)
    chans = [(0, "vx"), (4, "c4"), (15, "p")]
    one, roll = ev(RolloutStubModel(), data, 3, chans, torch.device("cpu"))
    out["rollout_data"] = data.numpy()
    out["rollout_t_input"] = np.asarray(3)
    out["rollout_channels"] = np.frombuffer(json.dumps(chans).encode(), dtype=np.uint8)
    out["rollout_one_step"] = np.asarray([one[n] for _, n in chans])
    out["rollout_rollout"] = np.asarray([roll[n] for _, n in chans])
    return out


def main():
    sys.path.insert(0, REF)
    import finetune.pde_loss_verified as mod
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    if "--only" in sys.argv:      # --only name [name ...]: some fixtures of the class table, and/or flow_v2
        for name in sys.argv[sys.argv.index("--only") + 1:]:
            if name == "flow_v2":
                np.savez_compressed(os.path.join(OUT, "flow_v2.npz"), **run_flow_v2())
            else:
                np.savez_compressed(os.path.join(OUT, f"pde_{name}.npz"), **run_case(name, cases()[name], mod))
            print("wrote", name)
        return
    if "--only-pins" in sys.argv:
        np.savez_compressed(os.path.join(OUT, "pins.npz"), **run_pins())
        print("wrote pins")
        return
    if "--only-glue" in sys.argv:
        np.savez_compressed(os.path.join(OUT, "glue.npz"), **run_glue())
        print("wrote glue")
        return
    for name, spec in cases().items():
        np.savez_compressed(os.path.join(OUT, f"pde_{name}.npz"), **run_case(name, spec, mod))
        print("wrote", name)
    np.savez_compressed(os.path.join(OUT, "flow_v2.npz"), **run_flow_v2())
    print("wrote flow_v2")
    np.savez_compressed(os.path.join(OUT, "pde_burgers2d_ghost.npz"), **run_burgers_ghost())
    print("wrote burgers2d_ghost")
    np.savez_compressed(os.path.join(OUT, "glue.npz"), **run_glue())
    print("wrote glue")
    np.savez_compressed(os.path.join(OUT, "pins.npz"), **run_pins())
    print("wrote pins")
    try:
        np.savez_compressed(os.path.join(OUT, "metric_func.npz"), **run_metric_func())
        print("wrote metric_func")
    except Exception as e:
        print("metric_func fixture skipped:", repr(e))


if __name__ == "__main__":
    main()
