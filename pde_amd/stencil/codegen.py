"""Emit the per-PDE device code (`csrc/gen/pde_<name>.inc`) from the symbolic specs.

For every spec the generator prints a traits struct holding
  * sizes (fields, equations, constants, per-sample parameters),
  * the stencil footprints the tile loaders need (halo per axis, time range),
  * the tile shapes picked for the kernels,
  * `residual(acc, R)`: the residual of every equation at one lattice site,
  * `adjoint(acc, g)`: the gather-form adjoint, i.e. d(sum_k sum_sites G_k R_k)/d field at one grid point.
The accessor `acc` (hand-written, csrc/stencil_kernels.hpp) maps F/G/X/c/sp/at/ath to LDS tiles,
registers or global memory.  Run as `python -m pde_amd.stencil.codegen` to regenerate.
"""
from __future__ import annotations

import os
from typing import List

from .expr import (Emitter, Aux, Add, Car, GRef, Ref, ZERO, UniformTable, add, collect, factor_pointwise, find_carries, footprint,
                   hoist_uniforms, replace_subtrees, share_products)
from .specs import SPECS, PDE_IDS, Spec

GEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "gen")


def _emit_fn(name: str, outvar: str, n: int, exprs, acc="a", tab: UniformTable = None, share: bool = False) -> List[str]:
    if tab is not None:
        exprs = [hoist_uniforms(e, tab) for e in exprs]
    if share:
        exprs = share_products(list(exprs))
    em = Emitter(acc)
    em.prepare(exprs)
    finals = [em.expr(e) for e in exprs]
    out = [f"  template <class A> static __device__ __forceinline__ void {name}(const A& {acc}, typename A::real (&{outvar})[{n}]) {{",
           "    using R_ = typename A::real;"]
    out += ["    " + l for l in em.lines]
    for i, s in enumerate(finals):
        out.append(f"    {outvar}[{i}] = {s};")
    out.append("  }")
    return out


def _emit_carry_fns(residuals, space, ne: int, nf: int, limit: int, tab: UniformTable = None) -> (int, List[str]):
    """`residual_c` (the residuals plus the values the same-site adjoint can reuse) and `adjoint_space_c` (the same-frame
    adjoint reading those values instead of recomputing them).  Returns (number of carried values, lines)."""
    if tab is not None:
        residuals = [hoist_uniforms(e, tab) for e in residuals]
        space = [hoist_uniforms(e, tab) for e in space]
    carries = find_carries(residuals, space, limit=limit) if limit > 0 else []
    nc = len(carries)
    table = {c.key(): Car(i) for i, c in enumerate(carries)}
    out: List[str] = []
    shared = share_products(list(residuals) + list(carries))
    res_s, car_s = shared[:len(residuals)], shared[len(residuals):]
    em = Emitter("a")
    em.prepare(res_s + car_s)
    finals = [em.expr(e) for e in res_s]
    cvals = [em.expr(c) for c in car_s]
    out.append(f"  template <class A> static __device__ __forceinline__ void residual_c(const A& a, typename A::real (&R)[{ne}], typename A::real (&C)[{max(nc, 1)}]) {{")
    out.append("    using R_ = typename A::real;")
    out += ["    " + l for l in em.lines]
    for i, v in enumerate(finals):
        out.append(f"    R[{i}] = {v};")
    for i, v in enumerate(cvals):
        out.append(f"    C[{i}] = {v};")
    if not nc:
        out.append("    C[0] = R_(0.0);")
    out.append("  }")
    em = Emitter("a")
    sp = [replace_subtrees(e, table) for e in space]
    em.prepare(sp)
    finals = [em.expr(e) for e in sp]
    out.append(f"  template <class A> static __device__ __forceinline__ void adjoint_space_c(const A& a, const typename A::real (&C)[{max(nc, 1)}], typename A::real (&g)[{nf}]) {{")
    out.append("    using R_ = typename A::real;")
    out += ["    " + l for l in em.lines]
    for i, v in enumerate(finals):
        out.append(f"    g[{i}] = {v};")
    out.append("  }")
    return nc, out


def split_adjoint(e):
    """(terms whose scaled residual sits on the output frame, terms whose scaled residual sits on another frame)."""
    terms = e.terms if isinstance(e, Add) else (e,)
    space, time = [], []
    for t in terms:
        gs = list(collect(t, GRef).values())
        if not gs:
            continue
        assert len({g.ot for g in gs}) == 1
        (space if gs[0].ot == 0 else time).append(t)
    return (add(*space) if space else ZERO), (add(*time) if time else ZERO)


def split_time(e):
    """Other-frame adjoint terms -> (those reading scaled residuals of earlier frames, of later frames)."""
    terms = e.terms if isinstance(e, Add) else (e,)
    lo, hi = [], []
    for t in terms:
        gs = list(collect(t, GRef).values())
        if not gs:
            continue
        (lo if gs[0].ot < 0 else hi).append(t)
    return (add(*lo) if lo else ZERO), (add(*hi) if hi else ZERO)


def _max3(a, b):
    return tuple(max(x, y) for x, y in zip(a, b))


def emit_spec(s: Spec) -> str:
    nf, ne = len(s.fields), len(s.eqs)
    fr, fa, fg = s.fp_res, s.fp_adj_f, s.fp_adj_g
    fx = footprint(list(s.residuals) + list(s.adjoint), Aux)
    xh = fx["h0"]
    L: List[str] = []
    L.append(f"// GENERATED by pde_amd/stencil/codegen.py from spec '{s.name}' - do not edit by hand.")
    L.append(f"// fields: {', '.join(s.fields)}   equations: {', '.join(s.eqs)}")
    L.append(f"// constants: {', '.join(n for n, _ in s.consts) or '-'}   per-sample: {', '.join(s.sparams) or '-'}")
    L.append(f"struct Pde_{s.name} {{")
    L.append(f"  static constexpr int ID = {PDE_IDS[s.name]};")
    L.append(f"  static constexpr int ND = {s.ndim}, NF = {nf}, NE = {ne}, NA = {len(s.aux)}, NC = {len(s.consts)}, NSP = {len(s.sparams)};")
    L.append(f"  static constexpr int TLO = {s.tlo}, THI = {s.thi};")
    L.append(f"  static constexpr bool PER0 = {str(s.periodic[0]).lower()}, PER1 = {str(s.periodic[1]).lower()}, PER2 = {str(s.periodic[2]).lower()};")
    # residual footprint: halo of the ot==0 tile, halo needed at ot!=0 (0 -> pointwise, read from global)
    L.append("  // residual footprint")
    L.append(f"  static constexpr int RH0 = {fr['h0'][0]}, RH1 = {fr['h0'][1]}, RH2 = {fr['h0'][2]};")
    L.append(f"  static constexpr int RT0 = {fr['ht'][0]}, RT1 = {fr['ht'][1]}, RT2 = {fr['ht'][2]};")
    L.append("  // adjoint footprint: G refs (scaled residuals) and F refs, at ot==0 (H) and ot!=0 (T)")
    L.append(f"  static constexpr int GTLO = {-fg['tmin']}, GTHI = {fg['tmax']};")
    L.append(f"  static constexpr int GH0 = {fg['h0'][0]}, GH1 = {fg['h0'][1]}, GH2 = {fg['h0'][2]};")
    L.append(f"  static constexpr int GT0 = {fg['ht'][0]}, GT1 = {fg['ht'][1]}, GT2 = {fg['ht'][2]};")
    L.append(f"  static constexpr int ATLO = {-fa['tmin']}, ATHI = {fa['tmax']}, ANY_AF = {1 if fa['count'] else 0};")
    L.append(f"  static constexpr int AH0 = {fa['h0'][0]}, AH1 = {fa['h0'][1]}, AH2 = {fa['h0'][2]};")
    L.append(f"  static constexpr int AT0 = {fa['ht'][0]}, AT1 = {fa['ht'][1]}, AT2 = {fa['ht'][2]};")
    L.append(f"  static constexpr int XH0 = {xh[0]}, XH1 = {xh[1]}, XH2 = {xh[2]};")
    L.append(f"  static constexpr int FT0 = {s.tile_fwd[0]}, FT1 = {s.tile_fwd[1]}, FT2 = {s.tile_fwd[2]};")
    L.append(f"  static constexpr int BT0 = {s.tile_bwd[0]}, BT1 = {s.tile_bwd[1]}, BT2 = {s.tile_bwd[2]};")
    L.append(f"  static constexpr int NT = {s.threads}, NTM = {s.threads_march}, WPS = {s.waves_per_simd};")
    L.append(f"  static constexpr int QF0 = {s.quad_fwd[0]}, QF1 = {s.quad_fwd[1]}, QF2 = {s.quad_fwd[2]};")
    L.append(f"  static constexpr int QB0 = {s.quad_bwd[0]}, QB1 = {s.quad_bwd[1]}, QB2 = {s.quad_bwd[2]};")
    L.append(f"  static constexpr int NTQ = {s.threads_quad}, NTQB = {s.threads_quad_bwd or s.threads_quad}, WPSQ = {s.waves_per_simd_quad};")
    L.append(f"  static constexpr int QWF = {s.quad_width_fwd}, QWB = {s.quad_width_bwd}, QPD = {s.quad_prefetch}, WPSF = {s.waves_per_simd_quad_fwd};")
    L.append(f"  static constexpr int ZT1 = {s.z_tile[0]}, ZT2 = {s.z_tile[1]}, NTZ = {s.z_threads}, ZW = {s.z_width}, ZFIRST = {1 if s.z_first else 0};")
    L.append(f"  static constexpr int ZCOL = {1 if s.zcol else 0}, ZPURE = {1 if s.z_pure else 0}, ZCT1 = {s.zc_tile[0]}, ZCT2 = {s.zc_tile[1]}, ZCB1 = {(s.zc_tile_bwd or s.zc_tile)[0]}, ZCB2 = {(s.zc_tile_bwd or s.zc_tile)[1]}, ZCW = {s.zc_width}, ZCWB = {s.zc_width_bwd}, ZCOLV = {1 if s.zcol_plain_fwd else 0}, ZCOLF = {1 if s.zcol_stored_fwd else 0};")
    L.append(f"  static constexpr const char* NAME = \"{s.name}\";")
    parts = [split_adjoint(e) for e in s.adjoint]
    fa = (lambda e: factor_pointwise(e)) if s.factor_adjoint else (lambda e: e)
    time_f = footprint([p[1] for p in parts], Ref)["count"]
    tuse = [0] * nf
    for r in s.residuals:
        for node in collect(r, Ref).values():
            if node.ot != 0:
                tuse[node.f] = 1
    L.append(f"  // which fields the residuals read at other frames (the others need one LDS frame slot, not a ring)")
    L.append(f"  static __host__ __device__ constexpr int tuse(int f) {{ return " + " : ".join(f"f == {i} ? {v}" for i, v in enumerate(tuse)) + " : 0; }")
    L.append(f"  static constexpr int TIME_AF = {1 if time_f else 0};   // the other-frame adjoint terms read fields")
    arefs = [n for e in s.adjoint for n in collect(e, Ref).values()]
    afn = 1 + max((n.f for n in arefs), default=-1)
    apure = all(n.ot == 0 and sum(1 for o in n.o if o != 0) <= 1 for n in arefs)
    L.append(f"  static constexpr int AFN = {afn}, AFPURE = {1 if apure else 0};   // fields 0 .. AFN-1 are what the adjoint reads; "
             f"AFPURE: only at its own frame and at sites on the three axes through the output site")
    tab = UniformTable() if s.hoist_uniforms else None
    body = _emit_fn("residual", "R", ne, s.residuals, tab=tab, share=True)
    body += _emit_fn("adjoint", "g", nf, [fa(e) for e in s.adjoint], tab=tab)
    body += _emit_fn("adjoint_space", "g", nf, [fa(p[0]) for p in parts], tab=tab)
    ncar, lines = _emit_carry_fns(s.residuals, [p[0] for p in parts], ne, nf, s.carry_max, tab=tab)
    body += lines
    body += _emit_fn("adjoint_time", "g", nf, [p[1] for p in parts], tab=tab)
    # the same terms split by the side of the output frame their scaled residual sits on (earlier / later frames)
    sides = [split_time(p[1]) for p in parts]
    body += _emit_fn("adjoint_time_m", "g", nf, [sd[0] for sd in sides], tab=tab)
    body += _emit_fn("adjoint_time_p", "g", nf, [sd[1] for sd in sides], tab=tab)
    nu = len(tab.exprs) if tab is not None else 0
    L.append(f"  // uniform subexpressions of the code below, evaluated once per sample (an accessor only keeps the ones it reads)")
    L.append(f"  static constexpr int NU = {nu};")
    L.append(f"  static constexpr int NCAR = {ncar};   // values residual_c hands to adjoint_space_c at the same site and frame")
    L += _emit_fn("uniforms", "U", max(nu, 1), tab.exprs if nu else [ZERO])
    L += body
    L.append("};")
    return "\n".join(L) + "\n"


def generate(out_dir: str = GEN_DIR) -> List[str]:
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for name, s in SPECS.items():
        path = os.path.join(out_dir, f"pde_{name}.inc")
        txt = emit_spec(s)
        old = open(path).read() if os.path.exists(path) else None
        if old != txt:
            with open(path, "w") as f:
                f.write(txt)
        written.append(path)
    # registry: include list + X-macro over all PDE structs
    reg = ["// GENERATED by pde_amd/stencil/codegen.py - do not edit by hand."]
    for name in SPECS:
        reg.append(f'#include "pde_{name}.inc"')
    reg.append("#define PDE_AMD_FOR_EACH_PDE(X) \\")
    reg.append(" \\\n".join(f"  X(Pde_{name})" for name in SPECS))
    reg.append("")
    path = os.path.join(out_dir, "registry.inc")
    txt = "\n".join(reg)
    old = open(path).read() if os.path.exists(path) else None
    if old != txt:
        with open(path, "w") as f:
            f.write(txt)
    written.append(path)
    return written


if __name__ == "__main__":
    for p in generate():
        print(p)
