"""Test infrastructure: the HIP kernels of every symbolic spec against the NumPy interpreter of the same spec on random shapes
(ragged and tile-sized extents, short and long clips, random lattice crops, weights and constants), in every backward
mode the PDE supports.  Prints one line per case and the failures at the end; exit status 1 when anything differs.

    python -m tests.fuzz_parity [--cases N per spec] [--seed S] [--only spec ...]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pde_amd import functional as PF
from pde_amd.stencil.specs import SPECS
from pde_amd.stencil import expr as X
from tests.spec_interp import InterpBackend

TOL = 1e-5


def rel_l2(a, b):
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


def reach(spec):
    """largest spatial offset of any field, residual or auxiliary reference, per axis"""
    r = [0, 0, 0]
    for cls in (X.Ref, X.Aux, X.GRef):
        fp = X.footprint(list(spec.residuals) + list(spec.adjoint), cls)
        for a in range(3):
            r[a] = max(r[a], fp["h0"][a], fp["ht"][a])
    return r


def one_case(name, rng, dev, hip, interp, quad, whole=False):
    """quad: the fastest extent is a multiple of four (register-blocked kernels).  whole: extents that are whole numbers of
    the kernels' tiles and no crop on periodic axes, which selects the mask-free instantiations where the PDE has them."""
    s = SPECS[name]
    nd = s.ndim
    B = int(rng.integers(1, 4))
    T = s.tlo + s.thi + int(rng.choice([1, 2, 3, 5, 9, 17] if nd < 3 else [1, 2, 3, 5]))
    if nd == 1:
        sp = [int(rng.integers(12, 700))]
    elif nd == 2:
        sp = [int(rng.integers(6, 80)), int(rng.integers(8, 150))]
    else:
        sp = [int(rng.integers(5, 30)), int(rng.integers(5, 22)), int(rng.integers(8, 50))]      # the interpreter keeps every node's array
    if quad:
        sp[-1] = (sp[-1] + 3) // 4 * 4
    if whole:
        lcm = lambda a, b: a * b // np.gcd(a, b)
        unit = [int(lcm(s.quad_bwd[a], s.quad_fwd[a])) for a in range(3 - nd, 3)]
        sp = [u * max(int(rng.integers(1, 3 if u >= 32 else 4)), -(-8 // u)) for u in unit]      # at least 8 points: periodic halos
    full = [1] * (3 - nd) + sp
    rc = reach(s)
    lo, hi = [0, 0, 0], [0, 0, 0]
    for a in range(3 - nd, 3):
        extra_lo, extra_hi = (0, 0) if whole else (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        base = 0 if s.periodic[a] else rc[a]      # wall axes: keep the stencil inside the grid (zeros beyond it feed divisions)
        lo[a], hi[a] = base + extra_lo, base + extra_hi
        if lo[a] + hi[a] >= full[a]:
            lo[a], hi[a] = base, base
    shape = (B, T, *sp)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    fields = [torch.randn(shape, generator=g) * 0.5 for _ in s.fields]
    if name == "turbulent_radiative":
        fields[2] = fields[2].abs() + 0.5
    aux = [0.5 + torch.rand((B, *sp), generator=g) for _ in s.aux]
    sparams = [0.01 + 0.09 * torch.rand(B, generator=g) for _ in s.sparams]
    consts = [float(rng.uniform(0.2, 1.5)) * (1.0 if rng.random() < 0.8 else -1.0) for _ in s.consts]
    TR = T - s.tlo - s.thi
    wt = (0.2 + torch.rand(len(s.eqs), TR, generator=g)) / (B * TR * np.prod(sp))
    prob = PF.StencilProblem(name, consts, tuple(lo), tuple(hi))
    # reference: the interpreter, in fp64 on the fp32-rounded inputs
    tot_r, sums_r, _ = interp.forward(prob, fields, aux, sparams, wt, False)
    one = torch.ones(())
    grads_r = interp.backward(prob, fields, aux, sparams, wt, [None] * len(s.eqs) + [one], None, None)
    modes = ["stored"]
    if hip.can_recompute(name):
        modes += ["recompute", "fused"]
    bad = []
    for mode in modes:
        fs = [f.to(dev).requires_grad_(True) for f in fields]
        m = ("fused", torch.ones((), device=dev)) if mode == "fused" else mode
        totals, sums = PF.residual_losses(prob, fs, wt.to(dev), [p.to(dev) for p in sparams], [x.to(dev) for x in aux], mode=m)
        totals[-1].backward()
        t = torch.stack([x.detach() for x in totals]).cpu().numpy().astype(np.float64)
        tr = tot_r.numpy().astype(np.float64)
        errs = [float(np.max(np.abs(t - tr) / np.maximum(np.abs(tr), 1e-300))), rel_l2(sums.detach().cpu().numpy(), sums_r.numpy())]
        errs += [rel_l2(f.grad.cpu().numpy().astype(np.float64), gr.numpy().astype(np.float64)) for f, gr in zip(fs, grads_r)]
        worst = max(e if np.isfinite(e) else 1e9 for e in errs)
        if worst > TOL:
            bad.append((mode, worst))
    desc = f"{name:20s} {str(shape):26s} lo {tuple(lo)} hi {tuple(hi)}"
    return desc, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", nargs="*")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    PF.SMALL_FIELD_ELEMENTS = 0      # the modes as asked for, also on these small grids
    hip, interp = PF.get_backend(), InterpBackend()
    rng = np.random.default_rng(args.seed)
    failures = []
    t0 = time.time()
    n = 0
    for name in SPECS:
        if args.only and name not in args.only:
            continue
        for c in range(args.cases):
            desc, bad = one_case(name, rng, dev, hip, interp, quad=(c % 2 == 1), whole=(c % 3 == 2))
            n += 1
            print(f"[{time.time() - t0:6.1f}s] {desc}  {'ok' if not bad else 'FAIL ' + str(bad)}", flush=True)
            if bad:
                failures.append((desc, bad))
    print(f"{n} cases, {len(failures)} failing")
    for desc, bad in failures:
        print("FAIL", desc, bad)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
