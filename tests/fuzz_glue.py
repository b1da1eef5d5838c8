"""Test infrastructure: the boundary-term, seam, statistics and gather kernels against the CPU oracle on random shapes, channel masks
and strip widths (values and gradients).  Prints one line per case; exit status 1 when anything differs.

    python -m tests.fuzz_glue [--cases N] [--seed S]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pde_amd
from oracle import glue_oracle as GO

TOL = 1e-5


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d, n = np.linalg.norm((a - b).ravel()), np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


def rand_mask(rng, C):
    m = torch.zeros(C)
    k = int(rng.integers(1, min(C, 5) + 1))
    m[torch.tensor(rng.choice(C, size=k, replace=False))] = 1
    return m


def boundary_case(rng, dev):
    kind = rng.choice(["1d", "strips", "edges", "faces", "shells", "xy_strips"])
    B, T, C = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 20))
    w = int(rng.integers(1, 4))
    if kind == "1d":
        sp = [int(rng.integers(2 * w + 1, 300))]
    elif kind in ("strips", "edges"):
        sp = [int(rng.integers(2 * w + 1, 70)), int(rng.integers(2 * w + 1, 90))]
    else:
        sp = [int(rng.integers(2 * w + 1, 20)), int(rng.integers(2 * w + 1, 24)), int(rng.integers(2 * w + 1, 30))]
    if len(sp) == 3:      # two of the reference's 3-D variants index a spatial axis with the channel list
        C = min(C, sp[1], sp[2])
    shape = (B, T, *sp, C)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    pred, tgt = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    mask = rand_mask(rng, C)
    ofn = {"1d": GO.boundary_1d, "strips": GO.boundary_2d_strips, "edges": GO.boundary_2d_edges, "faces": GO.boundary_3d_faces,
           "shells": GO.boundary_3d_shells, "xy_strips": GO.boundary_3d_xy_strips}[kind]
    takes_w = kind in ("1d", "strips", "shells", "xy_strips")
    p64 = pred.double().requires_grad_(True)
    ref = ofn(p64, tgt.double(), mask, w) if takes_w else ofn(p64, tgt.double(), mask)
    ref.backward()
    pd = pred.to(dev).requires_grad_(True)
    got = pde_amd.compute_boundary_loss(pd, tgt.to(dev), mask.to(dev), w, variant=kind) if takes_w else \
        pde_amd.compute_boundary_loss(pd, tgt.to(dev), mask.to(dev), variant=kind)
    got.backward()
    errs = [abs(float(got) - float(ref)) / max(abs(float(ref)), 1e-300), rel(pd.grad.cpu().numpy(), p64.grad.numpy())]
    return f"boundary {kind:9s} {str(shape):30s} w={w} ch={int(mask.sum())}", max(errs)


def seam_case(rng, dev):
    B, T, C = int(rng.integers(1, 3)), int(rng.integers(1, 5)), int(rng.integers(2, 20))
    P = int(rng.choice([4, 8, 16]))
    sp = [P * int(rng.integers(2, 6)), P * int(rng.integers(2, 6))]      # the reference's slices assume whole patches
    shape = (B, T, *sp, C)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    out = torch.randn(shape, generator=g)
    mask = rand_mask(rng, C)
    o64 = out.double().requires_grad_(True)
    ref = GO.patch_seams(o64, mask, P) + GO.noslip(o64, 0, 1)
    ref.backward()
    od = out.to(dev).requires_grad_(True)
    got = pde_amd.compute_patch_boundary_loss(od, mask.to(dev), P) + pde_amd.compute_noslip_loss(od, 0, 1)
    got.backward()
    errs = [abs(float(got) - float(ref)) / max(abs(float(ref)), 1e-300), rel(od.grad.cpu().numpy(), o64.grad.numpy())]
    return f"seams+noslip       {str(shape):30s} P={P} ch={int(mask.sum())}", max(errs)


def stats_case(rng, dev):
    B, T, C = int(rng.integers(1, 5)), int(rng.integers(1, 6)), int(rng.integers(1, 22))
    sp = [int(rng.integers(3, 60)), int(rng.integers(3, 70))]
    shape = (B, T, *sp, C)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    pred, tgt = torch.randn(shape, generator=g), torch.randn(shape, generator=g) + 0.3
    mask = rand_mask(rng, C)
    v_ref, r_ref = GO.vrmse(pred.double(), tgt.double(), mask)
    v, r = pde_amd.compute_vrmse(pred.to(dev), tgt.to(dev), mask.to(dev))
    errs = [abs(float(v) - float(v_ref)) / abs(float(v_ref)), abs(float(r) - float(r_ref)) / abs(float(r_ref))]
    ps = pde_amd.nrmse_per_sample(pred.to(dev), tgt.to(dev))
    errs.append(rel(ps.cpu().numpy(), GO.nrmse_per_sample(pred.double(), tgt.double()).numpy()))
    mr = float(GO.rse(pred.double(), tgt.double()).mean())
    errs.append(abs(pde_amd.compute_mean_rse(pred.to(dev), tgt.to(dev)) - mr) / abs(mr))
    return f"vrmse/nrmse/rse    {str(shape):30s} ch={int(mask.sum())}", max(errs)


def gather_case(rng, dev):
    B, T, C = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(3, 24))
    sp = [int(rng.integers(3, 50)), int(rng.integers(3, 60))]
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    out = torch.randn((B, T, *sp, C), generator=g)
    inp = torch.randn((B, T + int(rng.integers(0, 3)), *sp, C), generator=g)
    chans = [int(c) for c in rng.choice(C, size=int(rng.integers(1, 4)), replace=False)]
    transpose = bool(rng.integers(0, 2))
    ref = GO.gather(out, inp, chans, transpose)
    od = out.to(dev).requires_grad_(True)
    got = pde_amd.gather_fields(od, inp.to(dev), chans, transpose=transpose)
    err = max(float((a.detach().cpu() - b).abs().max()) for a, b in zip(got, ref))      # copies: exact
    wts = [torch.randn(a.shape, generator=g) for a in ref]
    sum((a * w.to(dev)).sum() for a, w in zip(got, wts)).backward()
    o64 = out.clone().requires_grad_(True)
    sum((a * w).sum() for a, w in zip(GO.gather(o64, inp, chans, transpose), wts)).backward()
    err = max(err, float((od.grad.cpu() - o64.grad).abs().max()))      # the adjoint of a copy: exact too
    return f"gather             {str(tuple(out.shape)):30s} ch={chans} T={transpose}", err * 1e5 if err else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    t0, bad, n = time.time(), [], 0
    for i in range(args.cases):
        for fn in (boundary_case, seam_case, stats_case, gather_case):
            desc, err = fn(rng, dev)
            n += 1
            ok = np.isfinite(err) and err <= TOL
            print(f"[{time.time() - t0:6.1f}s] {desc}  {'ok' if ok else 'FAIL %.3g' % err}", flush=True)
            if not ok:
                bad.append((desc, err))
    print(f"{n} cases, {len(bad)} failing")
    for b in bad:
        print("FAIL", *b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
