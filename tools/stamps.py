"""Diagnostic: per-phase timeline of the marching adjoint (needs a library built with -DPDE_AMD_STAMPS)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pde_amd
from pde_amd import _lib, functional as PF

dev = torch.device("cuda:0")
B, T, H, W = 16, 16, 256, 256
fields = [torch.randn(B, T, H, W, device=dev) for _ in range(3)]
nu = torch.full((B,), 0.05, device=dev)
mod = pde_amd.TaylorGreen2DPDELoss(nx=W, ny=H)
prob = mod._problem([H, W])
wt, _ = mod._weights(T - 2, B * (T - 2) * H * W, dev, None)
L = _lib.lib()
cp = prob.c_struct(fields[0].shape)
grads = [torch.empty_like(f) for f in fields]
one = torch.ones((), device=dev)
ups = _lib.ptr_array([None, None, None, one], 4)
dbg = torch.zeros(16 * 8 * 20 * 8, dtype=torch.int64, device=dev)
for it in range(3):
    dbg.zero_()
    st = L.pde_amd_residual_backward(C.byref(cp), _lib.ptr_array(fields, 3), _lib.ptr_array([], 0), _lib.ptr_array([nu], 1),
                                     C.c_void_p(wt.data_ptr()), ups, None, None, _lib.ptr_array(grads, 3),
                                     C.c_void_p(dbg.data_ptr()), dbg.numel() * 8, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(16, 8, 20, 8).astype(np.float64)
names = ["top->loads issued", "G stage", "barrier1", "gather", "barrier2", "stage LDS store (waits loads)", "grad stores + barrier3"]
for blk in (0, 5):
    for wv in (0, 1, 2, 3):
        x = d[blk, wv]
        if x[:, 0].max() == 0:
            continue
        steps = [i for i in range(20) if x[i, 0] > 0 and x[i, 7] > 0]
        dur = np.array([[x[i, j + 1] - x[i, j] for j in range(7)] for i in steps[2:-2]])
        tot = np.array([x[i, 7] - x[i, 0] for i in steps[2:-2]])
        print(f"block {blk} wave {wv}: steps {len(steps)}  mean step {tot.mean():.0f} ticks")
        for j, nme in enumerate(names):
            print(f"    {nme:34s} {dur[:, j].mean():9.0f}  ({100 * dur[:, j].mean() / tot.mean():5.1f} %)")
t0 = d[:, :, 0, 0][d[:, :, 0, 0] > 0].min()
t1 = d[:, :, :, 7].max()
print("span of stamped blocks (ticks):", t1 - t0)
