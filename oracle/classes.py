"""Reference-class-shaped front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

`build(cls_name, **ctor_kwargs)` returns a callable with the forward signature of the reference class
of that name (finetune/pde_loss_verified.py) and the same `(total, {equation: loss})` result, computed
by oracle/pde_oracle.py in the dtype of its inputs.  It restates the constructor defaults and the
per-class quirks of the epilogue (skipped equations, per-timestep scale selection, Gray-Scott's
scale keys) independently of pde_amd/losses.py.  Pinned by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import pde_oracle as O

TWO_PI = 2 * math.pi


def _per_t_helper(buf, T_res, t_offset):
    """finetune/pde_loss_verified.py:854-901"""
    if t_offset is not None:
        rows = []
        T_full = buf.shape[0]
        for off in t_offset.tolist():
            off = int(off)
            if off < 0 or off + T_res > T_full:
                off = max(0, min(off, T_full - T_res))
            rows.append(buf[off:off + T_res])
        st = torch.stack(rows).mean(0)
    else:
        st = buf[:T_res] if T_res <= buf.shape[0] else buf.mean().expand(T_res)
    return st.clamp(min=1e-8)


class _Cls:
    eqs = ()
    guarded = ()
    per_t = "none"            # 'helper' | 'slice' | 'none'
    per_t_eqs = None
    scale_key = {}
    total_key = False
    defaults: Dict[str, object] = {}
    default_scales: Dict[str, float] = {}

    def __init__(self, **kw):
        self.kw = {**self.defaults, **kw}
        keys = [self.scale_key.get(e, e) for e in self.eqs]
        self.scales = {**{k: 1.0 for k in keys}, **self.default_scales, **(self.kw.get("eq_scales") or {})}
        self.weights = {**{k: 1.0 for k in keys}, **(self.kw.get("eq_weights") or {})}
        self.per_t_bufs = {k: torch.as_tensor(v).float() for k, v in (self.kw.get("eq_scales_per_t") or {}).items()} \
            if self.kw.get("eq_scales_per_t") is not None else None

    def fixed_scale(self, eq):
        return float(self.scales[self.scale_key.get(eq, eq)])

    def finish(self, residuals, t_offset=None, reduction="mean"):
        out = {}
        if reduction != "mean":      # fixed scales only, e.g. :286-294
            for R, eq in zip(residuals, self.eqs):
                w = float(self.weights[self.scale_key.get(eq, eq)])
                sq = torch.sum(R ** 2) if reduction == "sum" else R ** 2
                out[eq] = w * sq / (self.fixed_scale(eq) ** 2)
            total = sum(out.values())
            if self.total_key:
                out["total"] = total
            return total, out
        for R, eq in zip(residuals, self.eqs):
            w = float(self.weights[self.scale_key.get(eq, eq)])
            if eq in self.guarded and not (w > 0):
                out[eq] = torch.tensor(0.0)
                continue
            st = None
            if self.per_t_bufs is not None and self.per_t != "none" and (self.per_t_eqs is None or eq in self.per_t_eqs) \
                    and eq in self.per_t_bufs:
                buf = self.per_t_bufs[eq].to(R.dtype)     # module.double() also converts the buffers
                T_res = R.shape[1]
                st = _per_t_helper(buf, T_res, t_offset) if self.per_t == "helper" else buf[:T_res].clamp(min=1e-8)
            if st is not None:
                st = st.to(R.dtype).view(1, -1, *([1] * (R.dim() - 2)))
                out[eq] = w * torch.mean((R / st) ** 2)
            else:
                out[eq] = w * torch.mean(R ** 2) / (self.fixed_scale(eq) ** 2)
        total = sum(out.values())
        if self.total_key:
            out["total"] = total
        return total, out


def _mk(name, eqs, defaults, call, **attrs):
    cls = type(name, (_Cls,), dict(eqs=eqs, defaults=defaults, __call__=call, **attrs))
    REGISTRY[name] = cls
    return cls


REGISTRY: Dict[str, type] = {}

_mk("Burgers1DPDELoss", ("momentum",), dict(dx=TWO_PI / 256, dt=0.01),
    lambda s, u, nu: s.finish(O.residuals_burgers1d(u, nu, dx=s.kw["dx"], dt=s.kw["dt"])), per_t="slice", total_key=True)
_mk("Advection1DPDELoss", ("advection",), dict(dx=TWO_PI / 256, dt=0.01),
    lambda s, u, a: s.finish(O.residuals_advection1d(u, a, dx=s.kw["dx"], dt=s.kw["dt"])), per_t="slice", total_key=True)
_mk("Heat1DPDELoss", ("heat",), dict(dx=TWO_PI / 256, dt=0.01),
    lambda s, u, alpha: s.finish(O.residuals_heat1d(u, alpha, dx=s.kw["dx"], dt=s.kw["dt"])), per_t="slice", total_key=True)
_mk("APEBenchBurgers1DPDELoss", ("momentum",), dict(nx=160, Lx=1.0, dt=1.0, alpha_2=2.9297e-5, delta_conv=-4.6875e-3),
    lambda s, u: s.finish(O.residuals_ape_burgers1d(u, dx=s.kw["Lx"] / s.kw["nx"], dt=s.kw["dt"], alpha_2=s.kw["alpha_2"],
                                                    delta_conv=s.kw["delta_conv"])))
_mk("KS1DPDELoss", ("momentum",), dict(nx=160, Lx=1.0, dt=1.0, alpha_2=-2.34375e-5, alpha_4=-2.8610229492e-9, beta_2=-2.34375e-4),
    lambda s, u: s.finish(O.residuals_ks1d(u, dx=s.kw["Lx"] / s.kw["nx"], dt=s.kw["dt"], alpha_2=s.kw["alpha_2"],
                                           alpha_4=s.kw["alpha_4"], beta_2=s.kw["beta_2"])))
_mk("KdV1DPDELoss", ("kdv",), dict(nx=160, Lx=1.0, dt=1.0, alpha_3=-8.5449e-7, alpha_4=-1.7166e-9, beta_1=-1.25e-2),
    lambda s, u: s.finish(O.residuals_kdv1d(u, dx=s.kw["Lx"] / s.kw["nx"], dt=s.kw["dt"], alpha_3=s.kw["alpha_3"],
                                            alpha_4=s.kw["alpha_4"], beta_1=s.kw["beta_1"])), per_t="slice", total_key=True)


def _tg(s, u, v, p, nu=None, t_offset=None):
    nu = 0.01 if nu is None else nu
    k = s.kw
    return s.finish(O.residuals_tg2d(u, v, p, nu, dx=k["Lx"] / k["nx"], dy=k["Ly"] / k["ny"], dt=k["dt"]), t_offset)


_mk("TaylorGreen2DPDELoss", ("continuity", "x_momentum", "y_momentum"),
    dict(nx=256, ny=256, Lx=TWO_PI, Ly=TWO_PI, dt=0.01), _tg, guarded=("x_momentum", "y_momentum"), per_t="helper")
_mk("AdvDiff2DPDELoss", ("advdiff",), dict(dx=TWO_PI / 256, dy=TWO_PI / 256, dt=0.01, a=1.0, b=0.5, nu=0.05),
    lambda s, u: s.finish(O.residuals_advdiff2d(u, **{k: s.kw[k] for k in ("dx", "dy", "dt", "a", "b", "nu")})),
    per_t="slice", total_key=True)
_mk("AnisoDiff2DPDELoss", ("aniso_diff",), dict(dx=1 / 160, dy=1 / 160, dt=0.1, A_xx=0.001, A_xy=0.0005, A_yx=0.0005, A_yy=0.002),
    lambda s, u: s.finish(O.residuals_anisodiff2d(u, **{k: s.kw[k] for k in ("dx", "dy", "dt", "A_xx", "A_xy", "A_yx", "A_yy")})),
    per_t="slice", total_key=True)
_mk("Wave2DPDELoss", ("u_equation", "w_equation"), dict(dx=TWO_PI / 256, dy=TWO_PI / 256, dt=0.01),
    lambda s, u, w, c: s.finish(O.residuals_wave2d(u, w, c, dx=s.kw["dx"], dy=s.kw["dy"], dt=s.kw["dt"])),
    per_t="slice", total_key=True)
_mk("APEBenchBurgers2DPDELoss", ("u_momentum", "v_momentum"),
    dict(nx=160, ny=160, Lx=1.0, Ly=1.0, dt=1.0, alpha_2=1.4648e-5, delta_conv=-2.3438e-3),
    lambda s, u, v: s.finish(O.residuals_ape_burgers2d(u, v, dx=s.kw["Lx"] / s.kw["nx"], dy=s.kw["Ly"] / s.kw["ny"], dt=s.kw["dt"],
                                                       alpha_2=s.kw["alpha_2"], delta_conv=s.kw["delta_conv"])))
_mk("NSPwCPDELoss", ("divergence", "vorticity", "tracer"), dict(nx=128, ny=128, Lx=1.0, Ly=1.0, dt=0.05, nu=4e-4, kappa=4e-4),
    lambda s, u, v, sc: s.finish(O.residuals_nspwc(u, v, sc, dx=s.kw["Lx"] / s.kw["nx"], dy=s.kw["Ly"] / s.kw["ny"], dt=s.kw["dt"],
                                                   nu=s.kw["nu"], kappa=s.kw["kappa"])), guarded=("vorticity", "tracer"))
_mk("KP2DPDELoss", ("kp2_equation",), dict(dx=20 / 256, dy=20 / 256, dt=0.01, skip_boundary=20),
    lambda s, u: s.finish(O.residuals_kp2d(u, dx=s.kw["dx"], dy=s.kw["dy"], dt=s.kw["dt"], skip_boundary=s.kw["skip_boundary"])),
    per_t="slice")


def _shear(s, u, v, p, sc, reduction="mean", t_offset=None):
    k = s.kw
    return s.finish(O.residuals_shear_flow(u, v, p, sc, dx=k["Lx"] / k["nx"], dy=k["Ly"] / k["ny"], dt=k["dt"], nu=k["nu"], D=k["D"],
                                           use_div_correction=k["use_div_correction"]), t_offset)


_mk("ShearFlowPDELossNPINN", ("continuity", "x_momentum", "y_momentum", "tracer"),
    dict(nx=256, ny=512, Lx=1.0, Ly=2.0, dt=0.1, nu=1e-4, D=1e-3, use_div_correction=True), _shear,
    guarded=("x_momentum", "y_momentum", "tracer"), per_t="helper")


def _shear_central(s, u, v, p, sc, reduction="mean", t_offset=None):
    k = s.kw
    return s.finish(O.residuals_shear_flow_central(u, v, p, sc, dx=k["Lx"] / k["nx"], dy=k["Ly"] / k["ny"], dt=k["dt"], nu=k["nu"],
                                                   D=k["D"]), t_offset)


# parity unpinned: see residuals_shear_flow_central
_mk("ShearFlowPDELoss", ("continuity", "u_momentum", "v_momentum", "tracer"),
    dict(nx=256, ny=512, Lx=1.0, Ly=2.0, dt=0.1, nu=1e-4, D=1e-3), _shear_central,
    guarded=("u_momentum", "v_momentum", "tracer"), per_t="helper", total_key=True)


class _B2CH(_Cls):
    def fixed_scale(self, eq):
        s = float(self.scales[eq])
        return 1.0 if (eq == "irrotational" and s < 1e-10) else s


def _b2ch(s, u, v, nu, reduction="mean", t_offset=None):
    k = s.kw
    return s.finish(O.residuals_burgers2d_ch(u, v, nu, dx=k["Lx"] / k["nx"], dy=k["Ly"] / k["ny"], dt=k["dt"]), t_offset)


REGISTRY["Burgers2DCHPDELoss"] = type("Burgers2DCHPDELoss", (_B2CH,), dict(
    eqs=("u_momentum", "v_momentum", "irrotational"), guarded=("u_momentum", "v_momentum", "irrotational"), per_t="helper",
    defaults=dict(nx=256, ny=256, Lx=TWO_PI, Ly=TWO_PI, dt=0.01), __call__=_b2ch))

_mk("ActiveMatterNPINNPDELoss", ("continuity", "concentration", "Dxx"),
    dict(nx=256, ny=256, Lx=10.0, Ly=10.0, dt=0.25, d_T=0.05, use_div_correction=True),
    lambda s, u, v, c, Dxx: s.finish(O.residuals_active_matter(u, v, c, Dxx, dx=s.kw["Lx"] / s.kw["nx"], dy=s.kw["Ly"] / s.kw["ny"],
                                                               dt=s.kw["dt"], d_T=s.kw["d_T"],
                                                               use_div_correction=s.kw["use_div_correction"])),
    guarded=("concentration", "Dxx"), per_t="slice", per_t_eqs=("continuity", "concentration"))


def _rb(s, u, v, b, kappa=None, t_offset=None):
    k = s.kw
    kappa = k["kappa"] if kappa is None else kappa
    return s.finish(O.residuals_rayleigh_benard(u, v, b, kappa, dx=k["Lx"] / k["nx"], dy=k["Ly"] / (k["ny"] - 1), dt=k["dt"],
                                                skip_bl=k["skip_bl"]), t_offset)


_mk("RayleighBenardFullPDELoss", ("continuity", "buoyancy"),
    dict(nx=512, ny=128, Lx=4.0, Ly=1.0, dt=0.25, kappa=1e-5, skip_bl=15), _rb, guarded=("buoyancy",), per_t="helper")
_mk("TurbulentRadiativePDELoss", ("mass", "x_momentum", "y_momentum"),
    dict(nx=128, ny=384, Lx=1.0, Ly=3.0, dt=1.597033, skip_bl=10),
    lambda s, vx, vy, rho, P: s.finish(O.residuals_turbulent_radiative(vx, vy, rho, P, dx=s.kw["Lx"] / s.kw["nx"],
                                                                       dy=s.kw["Ly"] / (s.kw["ny"] - 1), dt=s.kw["dt"],
                                                                       skip_bl=s.kw["skip_bl"])))
_mk("WaveGaussPDELoss", ("wave",), dict(nx=128, ny=128, Lx=1.0, Ly=1.0, dt=1 / 14, skip_boundary=2),
    lambda s, u, c: s.finish(O.residuals_wave_gauss(u, c, dx=s.kw["Lx"] / s.kw["nx"], dy=s.kw["Ly"] / s.kw["ny"], dt=s.kw["dt"],
                                                    skip_boundary=s.kw["skip_boundary"])), total_key=True)


def _gs(s, A, B, reduction="mean", laplacian="fft"):
    k = s.kw
    fn = O.residuals_gray_scott_fft if laplacian == "fft" else O.residuals_gray_scott_fd
    return s.finish(fn(A, B, **{n: k[n] for n in ("dx", "dy", "dt", "F", "k", "D_A", "D_B")}), reduction=reduction)


_mk("GrayScottPDELoss", ("A_equation", "B_equation"),
    dict(nx=128, ny=128, dx=2 / 128, dy=2 / 128, dt=10.0, F=0.098, k=0.057, D_A=1.81e-5, D_B=1.39e-5), _gs,
    per_t="slice", total_key=True, scale_key={"A_equation": "A", "B_equation": "B"})


_mk("Advection3DPDELoss", ("advection",), dict(dx=TWO_PI / 64, dy=TWO_PI / 64, dz=TWO_PI / 64, dt=0.05),
    lambda s, u, a, b, c: s.finish(O.residuals_advection3d(u, a, b, c, dx=s.kw["dx"], dy=s.kw["dy"], dz=s.kw["dz"], dt=s.kw["dt"])),
    per_t="slice", total_key=True)
_mk("APEBenchBurgers3DPDELoss", ("u_momentum", "v_momentum", "w_momentum"),
    dict(nx=32, Lx=1.0, dt=1.0, alpha_2=2.4414e-4, beta_1=1.5625e-2),
    lambda s, u, v, w: s.finish(O.residuals_ape_burgers3d(u, v, w, dx=s.kw["Lx"] / s.kw["nx"], dt=s.kw["dt"],
                                                          alpha_2=s.kw["alpha_2"], beta_1=s.kw["beta_1"])))
_mk("SwiftHohenberg3DPDELoss", ("swift_hohenberg",),
    dict(domain_extent=10 * math.pi, num_points=32, dt=0.1, reactivity=0.7, critical_number=1.0),
    lambda s, u: s.finish(O.residuals_swift_hohenberg3d(u, dx=s.kw["domain_extent"] / s.kw["num_points"], dt=s.kw["dt"],
                                                        r=s.kw["reactivity"], k=s.kw["critical_number"])),
    per_t="slice", total_key=True)
_mk("Heat3DPDELoss", ("heat",), dict(nx=128, ny=128, nz=128, Lx=TWO_PI, Ly=TWO_PI, Lz=TWO_PI, dt=0.01, alpha=0.01),
    lambda s, u, alpha=None: s.finish(O.residuals_heat3d(u, s.kw["alpha"] if alpha is None else alpha, dx=s.kw["Lx"] / s.kw["nx"],
                                                         dy=s.kw["Ly"] / s.kw["ny"], dz=s.kw["Lz"] / s.kw["nz"], dt=s.kw["dt"])),
    per_t="slice", total_key=True)


def _tg3(s, u, v, w, p, nu=None):
    k = s.kw
    nu = k["nu"] if nu is None else nu
    return s.finish(O.residuals_tg3d(u, v, w, p, nu, dx=k["Lx"] / k["nx"], dy=k["Ly"] / k["ny"], dz=k["Lz"] / k["nz"], dt=k["dt"]))


# no class of these names exists in the reference tree; arithmetic from its verification code (see pde_oracle.py)
_mk("TaylorGreen3DPDELoss", ("continuity", "u_momentum", "v_momentum", "w_momentum"),
    dict(nx=64, ny=64, nz=64, Lx=TWO_PI, Ly=TWO_PI, Lz=TWO_PI, dt=0.01, nu=0.01), _tg3)
_mk("NSIncomPDELoss", ("continuity", "u_momentum", "v_momentum", "tracer"),
    dict(nx=512, ny=512, Lx=1.0, Ly=1.0, dt=0.005, nu=0.01, skip_bl=2, use_div_correction=True),
    lambda s, u, v, p, d, fx, fy: s.finish(O.residuals_ns_incom(
        u, v, p, d, fx, fy, dx=s.kw["Lx"] / s.kw["nx"], dy=s.kw["Ly"] / s.kw["ny"], dt=s.kw["dt"], nu=s.kw["nu"],
        skip=s.kw["skip_bl"], use_div_correction=s.kw["use_div_correction"])),
    # defaults quoted at configs/finetune_ns_incom_v3.yaml:119-126
    default_scales={"continuity": 7e-4, "u_momentum": 1.34, "v_momentum": 1.31, "tracer": 0.24})
# the 2-D class finetune/train_taylor_green_lora_v3.py:66,341-348 imports: its successor's arithmetic, `nu` as fallback
_mk("TaylorGreenPDELoss", ("continuity", "x_momentum", "y_momentum"),
    dict(nx=256, ny=256, Lx=TWO_PI, Ly=TWO_PI, dt=0.001, nu=0.001),
    lambda s, u, v, p, nu=None: s.finish(O.residuals_tg2d(u, v, p, s.kw["nu"] if nu is None else nu, dx=s.kw["Lx"] / s.kw["nx"],
                                                          dy=s.kw["Ly"] / s.kw["ny"], dt=s.kw["dt"])),
    guarded=("x_momentum", "y_momentum"))


def _batched(*fs):
    return tuple(f.unsqueeze(0) if f.dim() == 3 else f for f in fs)


# the spectral legacy classes :28-146, :309-393 (the first adds a batch axis to a 3-D input, :101-104; the second never does)
_mk("ActiveMatterPDELoss", ("continuity", "advection"), dict(nx=256, ny=256, dx=10.0 / 256, dy=10.0 / 256, dt=1.0),
    lambda s, c, u, v, reduction="mean": s.finish(O.residuals_active_matter_fft(*_batched(c, u, v), dx=s.kw["dx"], dy=s.kw["dy"], dt=s.kw["dt"]),
                                                  reduction=reduction), total_key=True)
_mk("ViscoelasticContinuityLoss", ("continuity",), dict(nx=512, ny=512, dx=TWO_PI / 512, dy=2.0 / 512),
    lambda s, u, v, reduction="mean": s.finish(O.residuals_viscoelastic_cont(u, v, dx=s.kw["dx"], dy=s.kw["dy"]), reduction=reduction),      # no batch axis is added (:352-389)
    total_key=True)


def flow_mixing_pde_loss_v2(pred, left, right, bottom, top, vtmax, dt=1 / 999, Lx=1.0, Ly=1.0):
    """eval/pde_loss_flow_v2.py:179-243"""
    ut, adv, res = O.flow_mixing_v2(pred, left, right, bottom, top, vtmax=vtmax, dt=dt, Lx=Lx, Ly=Ly)
    return torch.mean(res ** 2), torch.mean(ut ** 2), torch.mean(adv ** 2), res


def build(cls_name: str, **kwargs):
    return REGISTRY[cls_name](**kwargs)


def burgers_pde_loss(pred, nu, dt=1 / 999, dx=1 / 127, dy=1 / 127):
    """finetune/pde_loss.py:87-216"""
    ru, rv = O.residuals_burgers2d_ghost(pred[..., 0], pred[..., 1], nu=nu, dt=dt, dx=dx, dy=dy)
    lu, lv = torch.mean(ru ** 2), torch.mean(rv ** 2)
    return lu + lv, lu, lv, ru, rv
