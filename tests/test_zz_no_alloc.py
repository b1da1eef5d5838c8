"""The allocation-free step entry point (pde_amd_residual_forward_backward_static) against the workspace-taking one."""
import ctypes as C

import pytest
import torch

from pde_amd import _lib
from pde_amd.functional import StencilProblem
from pde_amd.stencil.specs import SPECS

pytestmark = pytest.mark.gpu


def _step(entry, prob, fields, nu, weights, ups, static):
    L = _lib.lib()
    cp = prob.c_struct(fields[0].shape)
    ne = len(SPECS[prob.spec].eqs)
    sums = torch.full((fields[0].shape[0], ne), -1.0, dtype=torch.float64, device=fields[0].device)
    totals = torch.full((ne + 1,), -1.0, device=fields[0].device)
    grads = [torch.full_like(f, float("nan")) for f in fields]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    common = (C.byref(cp), _lib.ptr_array(fields, len(fields)), _lib.ptr_array([], 0), _lib.ptr_array([nu], 1),
              C.c_void_p(weights.data_ptr()), C.c_void_p(ups.data_ptr()), C.c_void_p(sums.data_ptr()),
              C.c_void_p(totals.data_ptr()), _lib.ptr_array(grads, len(grads)))
    if static:
        st = L.pde_amd_residual_forward_backward_static(*common, stream)
    else:
        nbytes = L.pde_amd_residual_workspace_bytes(C.byref(cp))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=fields[0].device)
        st = L.pde_amd_residual_forward_backward(*common, C.c_void_p(ws.data_ptr()), nbytes, stream)
    torch.cuda.synchronize()
    return st, sums, totals, grads


def test_static_scratch_step_matches_workspace_step():
    dev = torch.device("cuda:0")
    prob = StencilProblem("tg2d", list(_tg_consts()))
    g = torch.Generator().manual_seed(5)
    # two calls in a row on the same GPU, the second on a smaller input: the scratch keeps the first call's contents
    for shape in [(3, 9, 64, 128), (2, 5, 32, 64)]:
        fields = [torch.randn(shape, generator=g).to(dev) for _ in range(3)]
        nu = (0.01 + 0.09 * torch.rand(shape[0], generator=g)).to(dev)
        weights = (0.5 + torch.rand(3, shape[1] - 2, generator=g)).to(dev)
        ups = torch.tensor(0.75, device=dev)
        st0, sums0, totals0, grads0 = _step(None, prob, fields, nu, weights, ups, static=False)
        st1, sums1, totals1, grads1 = _step(None, prob, fields, nu, weights, ups, static=True)
        assert st0 == 0 and st1 == 0
        assert torch.equal(sums0, sums1) and torch.equal(totals0, totals1)      # same kernels, same order: bit-exact
        for a, b in zip(grads0, grads1):
            assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")
def test_static_scratch_follows_the_buffers_not_the_current_device():
    """buffers and stream on cuda:1 while cuda:0 is the current device: the scratch copy used is cuda:1's"""
    dev = torch.device("cuda:1")
    prob = StencilProblem("tg2d", list(_tg_consts()))
    g = torch.Generator().manual_seed(6)
    shape = (2, 6, 64, 128)
    fields = [torch.randn(shape, generator=g).to(dev) for _ in range(3)]
    nu = (0.01 + 0.09 * torch.rand(shape[0], generator=g)).to(dev)
    weights = (0.5 + torch.rand(3, shape[1] - 2, generator=g)).to(dev)
    ups = torch.tensor(0.75, device=dev)
    with torch.cuda.device(dev):
        st0, sums0, totals0, grads0 = _step(None, prob, fields, nu, weights, ups, static=False)
    torch.cuda.set_device(0)
    L = _lib.lib()
    cp = prob.c_struct(shape)
    sums = torch.full((shape[0], 3), -1.0, dtype=torch.float64, device=dev)
    totals = torch.full((4,), -1.0, device=dev)
    grads = [torch.full_like(f, float("nan")) for f in fields]
    st = L.pde_amd_residual_forward_backward_static(
        C.byref(cp), _lib.ptr_array(fields, 3), _lib.ptr_array([], 0), _lib.ptr_array([nu], 1), C.c_void_p(weights.data_ptr()),
        C.c_void_p(ups.data_ptr()), C.c_void_p(sums.data_ptr()), C.c_void_p(totals.data_ptr()), _lib.ptr_array(grads, 3),
        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    assert st0 == 0 and st == 0
    assert torch.equal(sums0, sums) and torch.equal(totals0, totals)
    for a, b in zip(grads0, grads):
        assert torch.equal(a, b)


def test_static_scratch_rejects_oversized_problems():
    dev = torch.device("cuda:0")
    prob = StencilProblem("tg2d", list(_tg_consts()))
    shape = (4096, 16, 256, 256)          # its partial sums alone exceed the static scratch; nothing this large is allocated
    cp = prob.c_struct(shape)
    L = _lib.lib()
    assert L.pde_amd_residual_workspace_bytes(C.byref(cp)) > _lib.STATIC_SCRATCH_BYTES
    tiny = torch.zeros(16, device=dev)
    ptrs = _lib.ptr_array([tiny] * 3, 3)
    st = L.pde_amd_residual_forward_backward_static(
        C.byref(cp), ptrs, _lib.ptr_array([], 0), _lib.ptr_array([tiny], 1), C.c_void_p(tiny.data_ptr()),
        C.c_void_p(tiny.data_ptr()), C.c_void_p(tiny.data_ptr()), C.c_void_p(tiny.data_ptr()), ptrs,
        C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == -5
    assert "STATIC_SCRATCH" in L.pde_amd_status_string(st).decode()


def _tg_consts():
    import pde_amd
    fn = pde_amd.TaylorGreen2DPDELoss(nx=128, ny=64, dt=0.01)
    return fn._problem([64, 128]).consts
