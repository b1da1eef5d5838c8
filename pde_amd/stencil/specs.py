"""Residual definitions of every supported PDE loss, as symbolic stencil expressions.

Axis numbering follows tensor dims: axis 0/1/2 = dim -3/-2/-1 of a field [B, T, (D0,) D1, D2].
Which axis a class calls "x" differs per class in the reference and is repeated here per spec.
Every citation is a line range of finetune/pde_loss_verified.py in the reference tree unless a
different file is named.

Per-instance constants are folded on the host in double precision and rounded once to fp32
(the reference divides fp32 tensors by Python doubles, which differs by at most 1 ulp per stencil).
"""
from __future__ import annotations

from dataclasses import dataclass, field as dc_field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from .expr import E, Ref, GRef, adjoint_exprs, collect, footprint
from .ops import (Q, C2_D1, C2_D2, C4_D1, C4_D2, D4_5PT, KDV_D3, KDV_D4, aux, const, dt_central,
                  face_div, field, qpsel, qsel, sh, sparam, stencil, tsh, upwind_flux_div)

ConstFn = Callable[[Dict[str, float]], float]


@dataclass
class Spec:
    name: str
    ndim: int
    fields: List[str]
    eqs: List[str]
    consts: List[Tuple[str, ConstFn]]
    sparams: List[str]
    periodic: Tuple[bool, bool, bool]      # per axis 0,1,2 (unused leading axes: True)
    residuals: List[E]
    aux: List[str] = dc_field(default_factory=list)
    doc: str = ""
    # tile shapes (axis0, axis1, axis2) and threads for the kernels
    tile_fwd: Tuple[int, int, int] = (1, 16, 64)
    tile_bwd: Tuple[int, int, int] = (1, 16, 64)
    threads: int = 256           # workgroup size of the per-frame kernels
    threads_march: int = 256     # workgroup size of the time-marching kernels
    waves_per_simd: int = 2      # occupancy the marching adjoint is compiled for (__launch_bounds__ 2nd argument)
    # register-blocked ("quad") marching kernels: a lane owns quad_width_* consecutive points of a row.
    # threads_quad / waves_per_simd_quad are quoted for width 4; width 2 covers the same tile with twice
    # the lanes at twice the waves per SIMD.
    quad_fwd: Tuple[int, int, int] = (1, 16, 64)
    quad_bwd: Tuple[int, int, int] = (1, 16, 64)
    threads_quad: int = 256
    waves_per_simd_quad: int = 2
    threads_quad_bwd: int = 0           # adjoint quad kernel: lanes at width 4 when they differ from threads_quad (0 = same)
    quad_width_fwd: int = 4
    quad_width_bwd: int = 4
    waves_per_simd_quad_fwd: int = 1    # forward quad kernel: compile for this many waves per SIMD (1 = no constraint)
    # plane-marching 3-D kernels (csrc/stencil_zmarch.hpp): (y, x) tile, lanes and lane width; (0, 0) = not used
    z_tile: Tuple[int, int] = (0, 0)
    z_threads: int = 256
    z_width: int = 2
    z_first: bool = False           # forward: prefer the plane-marching kernel to the time-marching one
    # z-column kernels (csrc/stencil_zcol.hpp): a workgroup marches a (y, x) tile through z for a whole group of frames
    zcol: bool = False              # stored-residual forward and adjoint take these kernels where the PDE qualifies
    zcol_plain_fwd: bool = False    # ... and so does the forward that stores no residuals (measured per PDE)
    zcol_stored_fwd: bool = True    # False: only the adjoint takes them, the forward that stores residuals keeps its kernel
    zc_tile: Tuple[int, int] = (8, 32)
    zc_tile_bwd: Optional[Tuple[int, int]] = None      # the adjoint's tile when it differs
    zc_width: int = 2               # points per lane of the forward ...
    zc_width_bwd: int = 2           # ... and of the adjoint
    auto_stored: bool = False       # backward modes 'auto' and 'fused' take the stored-residual kernels although the PDE can recompute
    fused_slower: bool = False      # the one-pass step measured slower than forward + recomputing adjoint: 'fused' takes those
    hoist_uniforms: bool = True     # precompute products/sums of constants and per-sample parameters once per sample
    quad_prefetch: int = 2          # adjoint: frames fetched this many steps ahead (2 costs one more register set)
    carry_max: int = 6              # adjoint: at most this many values per point handed from the residual stage to the gather
    factor_adjoint: bool = False    # emit the adjoint regrouped around the field values at the output point (expr.factor_pointwise)

    # derived
    def analyse(self):
        nf = len(self.fields)
        self.adjoint = adjoint_exprs(self.residuals, nf)
        self.fp_res = footprint(self.residuals, Ref)
        self.fp_adj_f = footprint(self.adjoint, Ref)
        self.fp_adj_g = footprint(self.adjoint, GRef)
        self.tlo = -self.fp_res["tmin"]
        self.thi = self.fp_res["tmax"]
        # every tap with an offset along axis 0 ("z") sits on the axis: no (y, x) offset, same frame
        pure = lambda nodes: all(n.o[0] == 0 or (n.o[1] == 0 and n.o[2] == 0 and n.ot == 0) for n in nodes)
        self.z_pure = self.ndim == 3 and pure([n for e in self.residuals for n in collect(e, Ref).values()]) and \
            pure([n for e in self.adjoint for n in collect(e, GRef).values()])
        return self


class _B:
    """Spec builder: hands out fields / constants / per-sample parameters as quantities."""

    def __init__(self, name, ndim, fields, eqs, periodic=(True, True, True), aux_names=(), doc=""):
        self.name, self.ndim, self.fields, self.eqs = name, ndim, list(fields), list(eqs)
        self.periodic, self.doc = periodic, doc
        self.consts: List[Tuple[str, ConstFn]] = []
        self.sparams: List[str] = []
        self.aux_names = list(aux_names)

    def F(self):
        qs = [field(i) for i in range(len(self.fields))]
        return qs[0] if len(qs) == 1 else qs

    def A(self, i=0):
        return aux(i)

    def c(self, name: str, fn: ConstFn) -> Q:
        for i, (n, _) in enumerate(self.consts):
            if n == name:
                return const(i, name)
        self.consts.append((name, fn))
        return const(len(self.consts) - 1, name)

    def s(self, name: str) -> Q:
        self.sparams.append(name)
        return sparam(len(self.sparams) - 1, name)

    def done(self, residuals: Sequence[Q], **kw) -> Spec:
        res = [r(0, (0, 0, 0)) for r in residuals]
        assert len(res) == len(self.eqs)
        return Spec(self.name, self.ndim, self.fields, self.eqs, self.consts, self.sparams,
                    self.periodic, res, aux=self.aux_names, doc=self.doc, **kw).analyse()


# ---- constant formulas -------------------------------------------------------------------------
def _inv2dt(p): return 1.0 / (2.0 * p["dt"])
def _k(name): return lambda p: float(p[name])
def _inv12(h): return lambda p: 1.0 / (12.0 * p[h])
def _inv12sq(h): return lambda p: 1.0 / (12.0 * p[h] ** 2)
def _inv(h, n=1): return lambda p: 1.0 / (p[h] ** n)
def _inv2(h): return lambda p: 1.0 / (2.0 * p[h])


def _d1c4(b, q, axis, h): return stencil(q, axis, C4_D1, b.c("inv12" + h, _inv12(h)))
def _d2c4(b, q, axis, h): return stencil(q, axis, C4_D2, b.c("inv12" + h + "2", _inv12sq(h)))
def _d2c2(b, q, axis, h): return stencil(q, axis, C2_D2, b.c("inv_" + h + "2", _inv(h, 2)))
def _d1c2(b, q, axis, h): return stencil(q, axis, C2_D1, b.c("inv2" + h, _inv2(h)))
def _ddt(b, q): return dt_central(q, b.c("inv2dt", _inv2dt))


def _lap_c4_grouped(b, q, coef, ax_x, hx, ax_y, hy):
    """coef * (d2c4_x + d2c4_y)(q) with the taps of equal weight added before they are scaled: four additions and five
    multiply-adds with per-sample coefficients instead of two five-tap sums, two scalings and the product with coef (the
    values differ from that form only in the order of rounding)."""
    cx, cy = b.c("inv12" + hx + "2", _inv12sq(hx)), b.c("inv12" + hy + "2", _inv12sq(hy))
    s1x, s2x = sh(q, ax_x, 1) + sh(q, ax_x, -1), sh(q, ax_x, 2) + sh(q, ax_x, -2)
    s1y, s2y = sh(q, ax_y, 1) + sh(q, ax_y, -1), sh(q, ax_y, 2) + sh(q, ax_y, -2)
    return ((16.0 * coef * cx) * s1x - (coef * cx) * s2x + (16.0 * coef * cy) * s1y - (coef * cy) * s2y
            - (30.0 * coef * (cx + cy)) * q)


SPECS: Dict[str, Spec] = {}


def _reg(s: Spec):
    SPECS[s.name] = s
    return s


T1D = dict(tile_fwd=(1, 1, 256), tile_bwd=(1, 1, 256), threads=256, quad_fwd=(1, 1, 256), quad_bwd=(1, 1, 256), threads_quad=64)
T3D = dict(tile_fwd=(2, 4, 64), tile_bwd=(4, 8, 64), threads=256, quad_fwd=(2, 8, 64), quad_bwd=(2, 8, 64), threads_quad=256)
Z3D = dict(z_tile=(16, 32), z_threads=256, z_width=2)      # plane-marching kernels: 16 x 32 tile, a pair of points per lane
T2D_WIDE = dict(tile_fwd=(1, 8, 64), tile_bwd=(1, 8, 64), quad_fwd=(1, 8, 64), quad_bwd=(1, 8, 64), threads_quad=128, quad_width_fwd=2, quad_width_bwd=2, quad_prefetch=1)   # 4-field PDEs


# ---- 1-D periodic family -----------------------------------------------------------------------
def _burgers1d():
    # Burgers1DPDELoss :2924-3027 — R = u_t + u u_x - nu u_xx, c4 periodic on dim -1
    b = _B("burgers1d", 1, ["u"], ["momentum"])
    u = b.F()
    nu = b.s("nu")
    return b.done([_ddt(b, u) + u * _d1c4(b, u, 2, "dx") - nu * _d2c4(b, u, 2, "dx")], **T1D)


def _advection1d():
    # Advection1DPDELoss :3034-3127 — R = u_t + a u_x
    b = _B("advection1d", 1, ["u"], ["advection"])
    u = b.F()
    a = b.s("a")
    return b.done([_ddt(b, u) + a * _d1c4(b, u, 2, "dx")], **T1D)


def _heat1d():
    # Heat1DPDELoss :3134-3229 — R = u_t - alpha u_xx
    b = _B("heat1d", 1, ["u"], ["heat"])
    u = b.F()
    al = b.s("alpha")
    return b.done([_ddt(b, u) - al * _d2c4(b, u, 2, "dx")], **T1D)


def _ape_burgers1d():
    # APEBenchBurgers1DPDELoss :3348-3419 — R = u_t - alpha_2 u_xx(c2) - delta u u_x(c4)
    b = _B("ape_burgers1d", 1, ["u"], ["momentum"])
    u = b.F()
    a2, dl = b.c("alpha_2", _k("alpha_2")), b.c("delta_conv", _k("delta_conv"))
    return b.done([_ddt(b, u) - a2 * _d2c2(b, u, 2, "dx") - dl * u * _d1c4(b, u, 2, "dx")], **T1D)


def _ks1d():
    # KS1DPDELoss :3535-3649 — R = u_t - a2 u_xx(c4) - a4 u_xxxx(5pt) - b2 * 0.5 * u_x^2
    b = _B("ks1d", 1, ["u"], ["momentum"])
    u = b.F()
    a2, a4, b2 = b.c("alpha_2", _k("alpha_2")), b.c("alpha_4", _k("alpha_4")), b.c("beta_2", _k("beta_2"))
    ux = _d1c4(b, u, 2, "dx")
    d4 = stencil(u, 2, D4_5PT, b.c("inv_dx4", _inv("dx", 4)))
    return b.done([_ddt(b, u) - a2 * _d2c4(b, u, 2, "dx") - a4 * d4 - b2 * (0.5 * ux * ux)], **T1D)


def _kdv1d():
    # KdV1DPDELoss :3656-3775 — R = u_t - b1 u u_x - a3 u_xxx - a4 u_xxxx (7-point d3, d4)
    b = _B("kdv1d", 1, ["u"], ["kdv"])
    u = b.F()
    a3, a4, b1 = b.c("alpha_3", _k("alpha_3")), b.c("alpha_4", _k("alpha_4")), b.c("beta_1", _k("beta_1"))
    d3 = stencil(u, 2, KDV_D3, b.c("inv_dx3", _inv("dx", 3)))
    d4 = stencil(u, 2, KDV_D4, b.c("inv_dx4", _inv("dx", 4)))
    return b.done([_ddt(b, u) - b1 * u * _d1c4(b, u, 2, "dx") - a3 * d3 - a4 * d4], **T1D)


# ---- 2-D periodic central family ---------------------------------------------------------------
def _tg2d():
    # TaylorGreen2DPDELoss :1717-1923 — x along dim -1 (dx), y along dim -2 (dy), c4, per-sample nu
    b = _B("tg2d", 2, ["u", "v", "p"], ["continuity", "x_momentum", "y_momentum"])
    u, v, p = b.F()
    nu = b.s("nu")
    dx = lambda q: _d1c4(b, q, 2, "dx")
    dy = lambda q: _d1c4(b, q, 1, "dy")
    visc = lambda q: _lap_c4_grouped(b, q, nu, 2, "dx", 1, "dy")
    cont = dx(u) + dy(v)
    xm = _ddt(b, u) + u * dx(u) + v * dy(u) + dx(p) - visc(u)
    ym = _ddt(b, v) + u * dx(v) + v * dy(v) + dy(p) - visc(v)
    return b.done([cont, xm, ym], threads_march=512, waves_per_simd=4, quad_prefetch=1)


def _advdiff2d():
    # AdvDiff2DPDELoss :2369-2495 — R = u_t + a u_x + b u_y - nu lap4(u); x dim -1, y dim -2
    b = _B("advdiff2d", 2, ["u"], ["advdiff"])
    u = b.F()
    a, bb, nu = b.c("a", _k("a")), b.c("b", _k("b")), b.c("nu", _k("nu"))
    lap = _d2c4(b, u, 2, "dx") + _d2c4(b, u, 1, "dy")
    return b.done([_ddt(b, u) + a * _d1c4(b, u, 2, "dx") + bb * _d1c4(b, u, 1, "dy") - nu * lap])


def _anisodiff2d():
    # AnisoDiff2DPDELoss :2502-2643 — u_xy = d_y(d_x u), chained c4
    b = _B("anisodiff2d", 2, ["u"], ["aniso_diff"])
    u = b.F()
    axx, axy, ayy = b.c("A_xx", _k("A_xx")), b.c("A_xy_sum", lambda p: p["A_xy"] + p["A_yx"]), b.c("A_yy", _k("A_yy"))
    uxy = _d1c4(b, _d1c4(b, u, 2, "dx"), 1, "dy")
    return b.done([_ddt(b, u) - axx * _d2c4(b, u, 2, "dx") - axy * uxy - ayy * _d2c4(b, u, 1, "dy")])


def _wave2d():
    # Wave2DPDELoss :2650-2789 — R_u = u_t - w ; R_w = w_t - c^2 lap4(u); x dim -2 (dx), y dim -1 (dy)
    b = _B("wave2d", 2, ["u", "w"], ["u_equation", "w_equation"])
    u, w = b.F()
    c2 = b.s("c2")
    lap = _d2c4(b, u, 1, "dx") + _d2c4(b, u, 2, "dy")
    return b.done([_ddt(b, u) - w, _ddt(b, w) - c2 * lap])      # [32, 9, 256^2] replayed: one pass 0.070 ms, forward + recomputing adjoint 0.092


def _ape_burgers2d():
    # APEBenchBurgers2DPDELoss :3241-3341 — c4 gradients, c2 Laplacian; x dim -1, y dim -2
    b = _B("ape_burgers2d", 2, ["u", "v"], ["u_momentum", "v_momentum"])
    u, v = b.F()
    a2, dl = b.c("alpha_2", _k("alpha_2")), b.c("delta_conv", _k("delta_conv"))
    dx = lambda q: _d1c4(b, q, 2, "dx")
    dy = lambda q: _d1c4(b, q, 1, "dy")
    lap = lambda q: _d2c2(b, q, 2, "dx") + _d2c2(b, q, 1, "dy")
    ru = _ddt(b, u) - a2 * lap(u) - dl * (u * dx(u) + v * dy(u))
    rv = _ddt(b, v) - a2 * lap(v) - dl * (u * dx(v) + v * dy(v))
    return b.done([ru, rv])


def _nspwc():
    # NSPwCPDELoss :1576-1710 — c2; vorticity w = v_x - u_y built at t-1, t, t+1; x dim -1, y dim -2
    b = _B("nspwc", 2, ["u", "v", "s"], ["divergence", "vorticity", "tracer"])
    u, v, s = b.F()
    nu, ka = b.c("nu", _k("nu")), b.c("kappa", _k("kappa"))
    dx = lambda q: _d1c2(b, q, 2, "dx")
    dy = lambda q: _d1c2(b, q, 1, "dy")
    lap = lambda q: _d2c2(b, q, 2, "dx") + _d2c2(b, q, 1, "dy")
    om = dx(v) - dy(u)
    div = dx(u) + dy(v)
    vort = _ddt(b, om) + u * dx(om) + v * dy(om) - nu * lap(om)
    trc = _ddt(b, s) + u * dx(s) + v * dy(s) - ka * lap(s)
    return b.done([div, vort, trc], tile_fwd=(1, 8, 64), tile_bwd=(1, 8, 64))


def _kp2d():
    # KP2DPDELoss :1930-2072 — R = (u_x)_t + 6 u_x^2 + 6 u u_xx + u_xxxx + 3 u_yy; x dim -2, y dim -1;
    # the crop `skip_boundary` is the runtime lattice
    b = _B("kp2d", 2, ["u"], ["kp2_equation"])
    u = b.F()
    ux = _d1c4(b, u, 1, "dx")
    d4 = stencil(u, 1, D4_5PT, b.c("inv_dx4", _inv("dx", 4)))
    R = _ddt(b, ux) + 6.0 * ux * ux + 6.0 * u * _d2c4(b, u, 1, "dx") + d4 + 3.0 * _d2c4(b, u, 2, "dy")
    return b.done([R], tile_fwd=(1, 8, 64), tile_bwd=(1, 8, 64))


# ---- 2-D n-PINN upwind family ------------------------------------------------------------------
def _lap_c2_grouped(b, q, coef, ax_x, hx, ax_y, hy):
    """coef * (d2c2_x + d2c2_y)(q) with each pair of neighbours added before it is scaled and one centre term: two
    additions and three multiply-adds with the coefficient folded into the grid constants, instead of two three-tap sums,
    their scalings and the product with coef."""
    cx, cy = b.c("inv_" + hx + "2", _inv(hx, 2)), b.c("inv_" + hy + "2", _inv(hy, 2))
    return ((coef * cx) * (sh(q, ax_x, 1) + sh(q, ax_x, -1)) + (coef * cy) * (sh(q, ax_y, 1) + sh(q, ax_y, -1))
            - (2.0 * coef * (cx + cy)) * q)


def _npinn_parts(b, u, v, ax_x, ax_y):
    idx, idy = b.c("inv_dx", _inv("dx")), b.c("inv_dy", _inv("dy"))
    div = face_div(u, v, ax_x, ax_y, idx, idy)
    conv = lambda f: upwind_flux_div(f, u, v, ax_x, ax_y, idx, idy)
    lap = lambda f, coef: _lap_c2_grouped(b, f, coef, ax_x, "dx", ax_y, "dy")      # coef * laplacian(f)
    return div, conv, lap


def _shear_flow():
    # ShearFlowPDELossNPINN :1146-1470 — x dim -2 (dx), y dim -1 (dy)
    b = _B("shear_flow", 2, ["u", "v", "p", "s"], ["continuity", "x_momentum", "y_momentum", "tracer"])
    u, v, p, s = b.F()
    div, conv, lap = _npinn_parts(b, u, v, 1, 2)
    nu, D, dc = b.c("nu", _k("nu")), b.c("D", _k("D")), b.c("div_corr", _k("div_corr"))
    ru = _ddt(b, u) + conv(u) + _d1c2(b, p, 1, "dx") - lap(u, nu) - dc * u * div
    rv = _ddt(b, v) + conv(v) + _d1c2(b, p, 2, "dy") - lap(v, nu) - dc * v * div
    rs = _ddt(b, s) + conv(s) - lap(s, D) - dc * s * div
    # adjoint tile 16 x 32: its residual region (tile + 2 each way) is 1.41x the tile instead of 1.59x for 8 x 64
    return b.done([div, ru, rv, rs], **{**T2D_WIDE, 'quad_bwd': (1, 16, 32)})


def _burgers2d_ch():
    # Burgers2DCHPDELoss :2079-2362 — x dim -1, y dim -2; irrotational constraint with c4
    b = _B("burgers2d_ch", 2, ["u", "v"], ["u_momentum", "v_momentum", "irrotational"])
    u, v = b.F()
    nu = b.s("nu")
    div, conv, lap = _npinn_parts(b, u, v, 2, 1)
    ru = _ddt(b, u) + conv(u) - u * div - lap(u, nu)
    rv = _ddt(b, v) + conv(v) - v * div - lap(v, nu)
    irr = _d1c4(b, v, 2, "dx") - _d1c4(b, u, 1, "dy")
    # adjoint: 171 VGPR at 2 points per lane; a 16 x 32 tile on 4 waves lets two workgroups share a CU (8 waves on 16 x 64 left room for one)
    # [32, 9, 256, 256]: forward + recomputing adjoint 0.205 ms, the one-pass step 0.185 ms
    return b.done([ru, rv, irr], quad_width_bwd=2, quad_prefetch=1, quad_bwd=(1, 16, 32), threads_quad_bwd=128)


def _active_matter():
    # ActiveMatterNPINNPDELoss :525-742 — x dim -1, y dim -2
    b = _B("active_matter", 2, ["u", "v", "c", "Dxx"], ["continuity", "concentration", "Dxx"])
    u, v, c, dxx = b.F()
    div, conv, lap = _npinn_parts(b, u, v, 2, 1)
    dT, dc = b.c("d_T", _k("d_T")), b.c("div_corr", _k("div_corr"))
    rc = _ddt(b, c) + conv(c) - lap(c, dT) - dc * c * div
    rd = _ddt(b, dxx) + conv(dxx) - lap(dxx, dT) - dc * dxx * div
    # [16, 9, 256, 256]: one-pass step 0.125 ms, stored residuals 0.127-0.134 ms, forward + recomputing adjoint 0.147 ms
    return b.done([div, rc, rd], **{**T2D_WIDE, "auto_stored": True})


# ---- 2-D mixed periodic / wall family ----------------------------------------------------------
def _rayleigh_benard():
    # RayleighBenardFullPDELoss :749-1001 — x dim -2 periodic c4; y dim -1 walls: c2 interior,
    # one-sided first derivative and zero second derivative on the wall rows (:823-852);
    # rows cropped by skip_bl = runtime lattice on axis 2
    b = _B("rayleigh_benard", 2, ["u", "v", "b"], ["continuity", "buoyancy"], periodic=(True, True, False))
    u, v, bb = b.F()
    ka = b.s("kappa")
    idy = b.c("inv_dy", _inv("dy"))

    def dyg(q):
        lo = (sh(q, 2, 1) - q) * idy
        hi = (q - sh(q, 2, -1)) * idy
        return qpsel(2, 0, False, lo, qpsel(2, 0, True, hi, _d1c2(b, q, 2, "dy")))

    def d2yg(q):
        return qpsel(2, 0, False, 0.0, qpsel(2, 0, True, 0.0, _d2c2(b, q, 2, "dy")))

    cont = _d1c4(b, u, 1, "dx") + dyg(v)
    buoy = _ddt(b, bb) + u * _d1c4(b, bb, 1, "dx") + v * dyg(bb) - ka * (_d2c4(b, bb, 1, "dx") + d2yg(bb))
    return b.done([cont, buoy])


def _turbulent_radiative():
    # TurbulentRadiativePDELoss :1008-1139 — conservative Euler; x dim -2 periodic c4, y dim -1 open:
    # c2 on interior rows only; lattice on axis 2 is [1+skip, N-1-skip)
    b = _B("turbulent_radiative", 2, ["vx", "vy", "rho", "P"], ["mass", "x_momentum", "y_momentum"],
           periodic=(True, True, False))
    vx, vy, rho, P = b.F()
    dx = lambda q: _d1c4(b, q, 1, "dx")
    dy = lambda q: _d1c2(b, q, 2, "dy")
    mass = _ddt(b, rho) + dx(rho * vx) + dy(rho * vy)
    xm = _ddt(b, rho * vx) + dx(rho * vx * vx + P) + dy(rho * vx * vy)
    ym = _ddt(b, rho * vy) + dx(rho * vx * vy) + dy(rho * vy * vy + P)
    # [16, 9, 128, 384] replayed: stored 0.093 ms, one pass 0.096, forward + recomputing adjoint 0.119
    return b.done([mass, xm, ym], **{**T2D_WIDE, 'quad_prefetch': 2, 'waves_per_simd_quad': 3, 'auto_stored': True, 'fused_slower': True})


def _wave_gauss():
    # WaveGaussPDELoss :1477-1569 — R = u_tt / c(x)^2 - lap2(u); both axes non-periodic;
    # lattice [1+sb, N-1-sb) on both axes; c is a time-independent field
    b = _B("wave_gauss", 2, ["u"], ["wave"], periodic=(True, False, False), aux_names=["c"])
    u = b.F()
    c = b.A(0)
    utt = (tsh(u, 1) - 2.0 * u + tsh(u, -1)) * b.c("inv_dt2", _inv("dt", 2))
    lap = _d2c2(b, u, 1, "dx") + _d2c2(b, u, 2, "dy")
    return b.done([utt / (c * c) - lap])


def _burgers2d_ghost():
    # burgers_pde_loss, finetune/pde_loss.py:87-216 — non-conservative 2nd-order upwind on sign(u), sign(v);
    # ghost cells 2 f[0] - f[1] beyond each wall (:125-135); backward time difference (:168-173);
    # residual lattice [1, N-1) on both axes; x dim -1, y dim -2
    b = _B("burgers2d_ghost", 2, ["u", "v"], ["u", "v"], periodic=(True, False, False))
    u, v = b.F()
    nu = b.c("nu", _k("nu"))
    idt = b.c("inv_dt", _inv("dt"))
    i2dx, i2dy = b.c("inv2dx", _inv2("dx")), b.c("inv2dy", _inv2("dy"))

    def nb(q, axis, k):
        """Neighbour k cells away along axis with the linear-extrapolation ghost one cell past a wall."""
        if abs(k) < 2:
            return sh(q, axis, k)
        if k == -2:   # site index 1 reaches the low ghost: 2 f[0] - f[1]
            return qpsel(axis, 1, False, 2.0 * sh(q, axis, -1) - q, sh(q, axis, -2))
        return qpsel(axis, 1, True, 2.0 * sh(q, axis, 1) - q, sh(q, axis, 2))

    def upw(phi, vel, axis, i2h):
        back = (3.0 * phi - 4.0 * nb(phi, axis, -1) + nb(phi, axis, -2)) * i2h
        fwd = (-3.0 * phi + 4.0 * nb(phi, axis, 1) - nb(phi, axis, 2)) * i2h
        return qsel(vel, back, fwd)

    lap = lambda q: _d2c2(b, q, 2, "dx") + _d2c2(b, q, 1, "dy")
    ddt = lambda q: (q - tsh(q, -1)) * idt
    ru = ddt(u) + u * upw(u, u, 2, i2dx) + v * upw(u, v, 1, i2dy) - nu * lap(u)
    rv = ddt(v) + u * upw(v, u, 2, i2dx) + v * upw(v, v, 1, i2dy) - nu * lap(v)
    return b.done([ru, rv])


# ---- Gray-Scott ---------------------------------------------------------------------------------
def _gray_scott_fd():
    # c4 periodic Laplacian restatement, tools/test_gt_pde_gray_scott.py:27-79 (x = axis -2);
    # reaction terms as GrayScottPDELoss :252-256
    b = _B("gray_scott_fd", 2, ["A", "B"], ["A_equation", "B_equation"])
    A, Bq = b.F()
    DA, DB = b.c("D_A", _k("D_A")), b.c("D_B", _k("D_B"))
    Fc, Fk = b.c("F", _k("F")), b.c("F_plus_k", lambda p: p["F"] + p["k"])
    lap = lambda q: _d2c4(b, q, 1, "dx") + _d2c4(b, q, 2, "dy")
    ab2 = A * Bq * Bq
    ra = _ddt(b, A) - DA * lap(A) + ab2 - Fc * (1.0 - A)
    rb = _ddt(b, Bq) - DB * lap(Bq) - ab2 + Fk * Bq
    return b.done([ra, rb])      # [64, 9, 128^2] replayed: one pass 0.051 ms, forward + recomputing adjoint 0.067


def _gray_scott_pw():
    # GrayScottPDELoss :216-302 with the spectral Laplacians supplied as two extra fields (the LDS-resident transform
    # kernel, or rocFFT for plane sizes it is not built for): pointwise residual + reduction only
    b = _B("gray_scott_pw", 2, ["A", "B", "lapA", "lapB"], ["A_equation", "B_equation"])
    A, Bq, lA, lB = b.F()
    DA, DB = b.c("D_A", _k("D_A")), b.c("D_B", _k("D_B"))
    Fc, Fk = b.c("F", _k("F")), b.c("F_plus_k", lambda p: p["F"] + p["k"])
    ab2 = A * Bq * Bq
    ra = _ddt(b, A) - DA * lA + ab2 - Fc * (1.0 - A)
    rb = _ddt(b, Bq) - DB * lB - ab2 + Fk * Bq
    return b.done([ra, rb])      # with the spectral Laplacian kernel, [64, 9, 128^2] replayed: one pass 0.223 ms, forward + adjoint 0.244


def _active_matter_fft():
    # ActiveMatterPDELoss (the spectral, non-n-PINN class) :28-146 — continuity = du/dx + dv/dy, advection =
    # (c[t+1] - c[t]) / dt + u dc/dx + v dc/dy at frame t; the spectral derivatives come in as fields: `div` is the
    # divergence, `adv` the transport term (rocFFT through torch.fft and two products); pointwise residual + reduction here
    b = _B("active_matter_fft", 2, ["c", "adv", "div"], ["continuity", "advection"])
    c, adv, div = b.F()
    return b.done([div, (tsh(c, 1) - c) * b.c("inv_dt", _inv("dt")) + adv])


def _viscoelastic_cont():
    # ViscoelasticContinuityLoss :309-393 — R = du/dx (spectral along x = dim -2, supplied as the field `ux`) +
    # (v[y+1] - v[y-1]) / (2 dy) on the interior of the wall-bounded y axis (dim -1); every frame, no time derivative
    b = _B("viscoelastic_cont", 2, ["ux", "v"], ["continuity"], periodic=(True, True, False))
    ux, v = b.F()
    return b.done([ux + _d1c2(b, v, 2, "dy")])


def _flow_mixing_v2():
    # flow_mixing_pde_loss_v2, eval/pde_loss_flow_v2.py:179-243 on the ghost-padded field [B, T, H+2, W+2] (:57-111):
    # time = (u[t] - u[t-1]) / dt, advection = a (uE - uW) / (2 dx) + b (uN - uS) / (2 dy) with the coefficient fields
    # a, b (:18-54), residual = their sum; lattice = the interior; x = dim -1, y = dim -2
    b = _B("flow_mixing_v2", 2, ["u"], ["time", "advection", "residual"], periodic=(True, False, False), aux_names=["a", "b"])
    u = b.F()
    ca, cb = b.A(0), b.A(1)
    rt = (u - tsh(u, -1)) * b.c("inv_dt", _inv("dt"))
    ra = ca * _d1c2(b, u, 2, "dx") + cb * _d1c2(b, u, 1, "dy")
    return b.done([rt, ra, rt + ra])


# ---- 3-D periodic family ------------------------------------------------------------------------
def _advection3d():
    # Advection3DPDELoss :2796-2917 — x,y,z = dims -3,-2,-1; per-sample a,b,c
    b = _B("advection3d", 3, ["u"], ["advection"])
    u = b.F()
    a, bb, c = b.s("a"), b.s("b"), b.s("c")
    R = _ddt(b, u) + a * _d1c4(b, u, 0, "dx") + bb * _d1c4(b, u, 1, "dy") + c * _d1c4(b, u, 2, "dz")
    # plane-marching forward and adjoint: measured 45 / 61 / 72 us against 84 / 93 / 105 us for the time-marching forward
    # (without / with residual output) and the per-frame adjoint on [2, 9, 128^3]
    return b.done([R], **{**T3D, **Z3D, "z_first": True, "zcol": True, "zc_width": 4, "zc_width_bwd": 4})


def _ape_burgers3d():
    # APEBenchBurgers3DPDELoss :3426-3528 — phi_t + beta_1 * 0.5 * div(phi (u,v,w)) - alpha_2 lap2(phi);
    # c4 derivative of the products, single grid spacing
    b = _B("ape_burgers3d", 3, ["u", "v", "w"], ["u_momentum", "v_momentum", "w_momentum"])
    u, v, w = b.F()
    a2, b1 = b.c("alpha_2", _k("alpha_2")), b.c("beta_1", _k("beta_1"))
    g = lambda q, ax: _d1c4(b, q, ax, "dx")
    lap = lambda q: _d2c2(b, q, 0, "dx") + _d2c2(b, q, 1, "dx") + _d2c2(b, q, 2, "dx")
    res = []
    for phi in (u, v, w):
        conv = 0.5 * (g(phi * u, 0) + g(phi * v, 1) + g(phi * w, 2))
        res.append(_ddt(b, phi) + b1 * conv - a2 * lap(phi))
    # per-frame kernels (the time ring of 3-D halo tiles does not fit LDS): a near-cubic tile reads 3.4 halo floats per
    # point instead of 5.1 and leaves room for three workgroups per CU
    return b.done(res, **{**T3D, "quad_fwd": (8, 8, 16), "quad_bwd": (8, 8, 16), **Z3D, "zcol": True, "zcol_plain_fwd": True, "zc_width": 4, "zc_width_bwd": 4, "zc_tile": (4, 64), "zc_tile_bwd": (8, 32), "factor_adjoint": True})


def _swift_hohenberg3d():
    # SwiftHohenberg3DPDELoss :3782-3876 — R = u_t - (r-k^2) u + 2k lap(u) + lap(lap(u)) - u^2 + u^3
    b = _B("swift_hohenberg3d", 3, ["u"], ["swift_hohenberg"])
    u = b.F()
    lin, twok = b.c("linear_coeff", _k("linear_coeff")), b.c("two_k", lambda p: 2.0 * p["k"])
    lap = lambda q: _d2c2(b, q, 0, "dx") + _d2c2(b, q, 1, "dx") + _d2c2(b, q, 2, "dx")
    lu = lap(u)
    R = _ddt(b, u) - lin * u + twok * lu + lap(lu) - u * u + u * u * u
    return b.done([R], **{**T3D, **Z3D, "z_first": True})


def _heat3d():
    # No class of this name exists in the reference snapshot; arithmetic follows verify_3d_heat,
    # finetune/generate_analytical_3d.py:160-178 (7-point c2 Laplacian, periodic, central time), with the
    # per-sample alpha and dict keys of the call site finetune/train_heat_3d_lora_v3.py:144-172
    b = _B("heat3d", 3, ["u"], ["heat"])
    u = b.F()
    al = b.s("alpha")
    lap = _d2c2(b, u, 0, "dx") + _d2c2(b, u, 1, "dy") + _d2c2(b, u, 2, "dz")
    # forward and adjoint on the plane-marching kernels, residuals stored: step 0.125 ms on [2, 9, 128^3] against 0.171 ms
    # for the time-marching forward + recomputing adjoint and 0.149 ms for the fused step
    return b.done([_ddt(b, u) - al * lap], **{**T3D, **Z3D, "z_first": True, "auto_stored": True, "fused_slower": True, "zcol": True, "zc_width": 4, "zc_width_bwd": 4, "zc_tile": (4, 64)})      # [2, 9, 128^3]: stored 0.125 ms, one pass 0.139, recompute 0.166


def _tg3d():
    # No class of this name exists in the reference snapshot; arithmetic follows verify_3d_taylor_green,
    # finetune/generate_analytical_3d.py:97-157 (x, y, z = dims -3, -2, -1): face-velocity divergence, and for each velocity
    # component the c2 primitive-form momentum R = phi_t + u phi_x + v phi_y + w phi_z + p_(own axis) - nu lap2(phi)
    # (the verifier spells out the u component only), per-sample nu
    b = _B("tg3d", 3, ["u", "v", "w", "p"], ["continuity", "u_momentum", "v_momentum", "w_momentum"])
    u, v, w, p = b.F()
    nu = b.s("nu")
    hs = ("dx", "dy", "dz")
    inv = [b.c("inv_" + h, _inv(h)) for h in hs]
    face = lambda q, ax: 0.5 * (sh(q, ax, 1) + q) - 0.5 * (sh(q, ax, -1) + q)
    div = face(u, 0) * inv[0] + face(v, 1) * inv[1] + face(w, 2) * inv[2]
    lap = lambda q: _d2c2(b, q, 0, "dx") + _d2c2(b, q, 1, "dy") + _d2c2(b, q, 2, "dz")
    res = [div]
    for ax, phi in enumerate((u, v, w)):
        conv = u * _d1c2(b, phi, 0, "dx") + v * _d1c2(b, phi, 1, "dy") + w * _d1c2(b, phi, 2, "dz")
        res.append(_ddt(b, phi) + conv + _d1c2(b, p, ax, hs[ax]) - nu * lap(phi))
    # [1, 9, 128^3], z-column form at 4 points per lane against plane-marching: forward 88 against 101 us, forward writing
    # residuals 137 against 153 us
    return b.done(res, **{**T3D, "quad_fwd": (8, 8, 16), "quad_bwd": (8, 8, 16), **Z3D, "zcol": True, "zcol_plain_fwd": True, "zc_width": 4, "zc_width_bwd": 2, "zc_tile": (4, 64)})


def _ns_incom():
    # No class of this name exists in the reference snapshot (finetune/train_ns_incom_lora_v3.py:68 imports it); arithmetic
    # follows "Method B" of tools/test_gt_pde_ns_incom.py:258-378 (n-PINN: face velocities, 2nd-order upwind conservative
    # fluxes, c2 Laplacian, div correction, time-invariant force fx, fy times the tracer; x = dim -2, y = dim -1) plus the
    # face-averaged pressure gradient of its specification (:109-111, :122-123).  Walls on both axes: the residual lattice
    # is [skip, N - skip), skip >= 2
    b = _B("ns_incom", 2, ["u", "v", "p", "d"], ["continuity", "u_momentum", "v_momentum", "tracer"],
           periodic=(True, False, False), aux_names=["fx", "fy"])
    u, v, p, d = b.F()
    fx, fy = b.A(0), b.A(1)
    div, conv, lap = _npinn_parts(b, u, v, 1, 2)
    idx, idy = b.c("inv_dx", _inv("dx")), b.c("inv_dy", _inv("dy"))
    nu, dc = b.c("nu", _k("nu")), b.c("div_corr", _k("div_corr"))
    px = (0.5 * (p + sh(p, 1, 1)) - 0.5 * (sh(p, 1, -1) + p)) * idx
    py = (0.5 * (p + sh(p, 2, 1)) - 0.5 * (sh(p, 2, -1) + p)) * idy
    ru = _ddt(b, u) + conv(u) + px - lap(u, nu) - d * fx - dc * u * div
    rv = _ddt(b, v) + conv(v) + py - lap(v, nu) - d * fy - dc * v * div
    rd = _ddt(b, d) + conv(d) - dc * d * div
    return b.done([div, ru, rv, rd], **{**T2D_WIDE, 'quad_bwd': (1, 16, 32)})


def _shear_flow_central():
    # ShearFlowPDELoss: the central-difference class the reference's tools construct (tools/test_gt_npinn_vs_central.py:82-85,
    # tools/test_gt_fp32_pde_bc.py:52-55, called as pde_fn(u, v, p, s, reduction=...) with the keys continuity, u_momentum,
    # v_momentum, tracer) but whose definition the snapshot does not hold.  Arithmetic chosen here: the advective form with the
    # 4th-order periodic central stencils of the file's other periodic classes (:1769-1790); x dim -2 (dx), y dim -1 (dy)
    # as in the n-PINN class of the same system.
    b = _B("shear_flow_central", 2, ["u", "v", "p", "s"], ["continuity", "u_momentum", "v_momentum", "tracer"])
    u, v, p, s = b.F()
    nu, D = b.c("nu", _k("nu")), b.c("D", _k("D"))
    dx = lambda q: _d1c4(b, q, 1, "dx")
    dy = lambda q: _d1c4(b, q, 2, "dy")
    visc = lambda q, c: _lap_c4_grouped(b, q, c, 1, "dx", 2, "dy")
    cont = dx(u) + dy(v)
    ru = _ddt(b, u) + u * dx(u) + v * dy(u) + dx(p) - visc(u, nu)
    rv = _ddt(b, v) + u * dx(v) + v * dy(v) + dy(p) - visc(v, nu)
    rs = _ddt(b, s) + u * dx(s) + v * dy(s) - visc(s, D)
    return b.done([cont, ru, rv, rs], **{**T2D_WIDE, 'quad_bwd': (1, 16, 32)})


for _f in (_burgers1d, _advection1d, _heat1d, _ape_burgers1d, _ks1d, _kdv1d, _tg2d, _advdiff2d, _anisodiff2d,
           _wave2d, _ape_burgers2d, _nspwc, _kp2d, _shear_flow, _burgers2d_ch, _active_matter,
           _rayleigh_benard, _turbulent_radiative, _wave_gauss, _burgers2d_ghost, _gray_scott_fd,
           _gray_scott_pw, _advection3d, _ape_burgers3d, _swift_hohenberg3d, _heat3d, _tg3d, _ns_incom,
           _active_matter_fft, _viscoelastic_cont, _flow_mixing_v2, _shear_flow_central):      # new PDEs go last: ids are part of the C ABI
    _reg(_f())

PDE_IDS = {name: i for i, name in enumerate(SPECS)}
