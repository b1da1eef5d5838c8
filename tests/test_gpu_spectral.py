"""The LDS-resident spectral Laplacian (two real fields as one complex plane) against torch.fft in fp64: the reference's
ifft2(-K2 * fft2(f)).real (finetune/pde_loss_verified.py:200-214), its gradient, and the Gray-Scott loss built on it
against the same loss through the rocFFT path."""
import numpy as np
import pytest
import torch

import pde_amd
from pde_amd import functional as PF

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _k2(nx, ny, dx, dy):
    kx = torch.fft.fftfreq(nx, d=dx) * 2 * np.pi
    ky = torch.fft.fftfreq(ny, d=dy) * 2 * np.pi
    KX, KY = torch.meshgrid(kx, ky, indexing="ij")
    return KX ** 2 + KY ** 2


def _ref(f, k2):
    return torch.fft.ifft2(-k2.double() * torch.fft.fft2(f.double())).real


@pytest.mark.parametrize("nx,ny", [(16, 16), (32, 32), (64, 64), (128, 128), (128, 64), (16, 128), (64, 32), (32, 128)])
def test_against_fft2_fp64(nx, ny):
    g = torch.Generator().manual_seed(nx * 1000 + ny)
    a, b = torch.randn(2, 3, nx, ny, generator=g), torch.randn(2, 3, nx, ny, generator=g) * 3.0 + 1.0
    k2 = _k2(nx, ny, 2.0 / nx, 2.0 / ny)
    assert PF.spectral_pair_supported(nx, ny)
    la, lb = PF.spectral_laplacian_pair(a.to(DEV), b.to(DEV), (-k2).to(DEV))
    for got, f in ((la, a), (lb, b)):
        want = _ref(f, k2)
        # fp32 transforms of length <= 128 in each direction: a few ulp of the largest spectral term
        assert float((got.cpu().double() - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_unsupported_sizes_are_reported():
    assert not PF.spectral_pair_supported(96, 96) and not PF.spectral_pair_supported(256, 128)
    with pytest.raises(RuntimeError, match="spectral_laplacian_pair"):
        PF.spectral_laplacian_pair(torch.zeros(1, 96, 96, device=DEV), torch.zeros(1, 96, 96, device=DEV), torch.zeros(96, 96, device=DEV))


def test_gradient_is_the_operator_itself():
    nx = ny = 32
    g = torch.Generator().manual_seed(3)
    k2 = _k2(nx, ny, 2.0 / nx, 2.0 / ny)
    a = torch.randn(2, nx, ny, generator=g)
    b = torch.randn(2, nx, ny, generator=g)
    wa, wb = torch.randn(2, nx, ny, generator=g), torch.randn(2, nx, ny, generator=g)
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    la, lb = PF.spectral_laplacian_pair(ad, bd, (-k2).to(DEV))
    ((la * wa.to(DEV)).sum() + (lb * wb.to(DEV)).sum()).backward()
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    (( _ref(a64, k2) * wa.double()).sum() + (_ref(b64, k2) * wb.double()).sum()).backward()
    for got, want in ((ad.grad, a64.grad), (bd.grad, b64.grad)):
        assert float((got.cpu().double() - want).abs().max()) <= 2e-6 * float(want.abs().max())


@pytest.mark.parametrize("n,batch", [(32, 3), (128, 2)])
def test_gray_scott_loss_matches_the_rocfft_path(n, batch):
    g = torch.Generator().manual_seed(n)
    A = (0.5 + 0.2 * torch.rand(batch, 5, n, n, generator=g))
    B = (0.25 + 0.1 * torch.rand(batch, 5, n, n, generator=g))
    mod = pde_amd.GrayScottPDELoss(nx=n, ny=n, dx=2.0 / n, dy=2.0 / n).to(DEV)
    out = {}
    for path in ("lds", "rocfft"):
        if path == "rocfft":
            mod._pair_multiplier = lambda f: None      # what forward does for plane sizes the kernel is not built for
        a, b = A.to(DEV).requires_grad_(True), B.to(DEV).requires_grad_(True)
        total, losses = mod(a, b)
        total.backward()
        out[path] = (float(total), {k: float(v) for k, v in losses.items()}, a.grad.cpu(), b.grad.cpu())
    (t0, l0, ga0, gb0), (t1, l1, ga1, gb1) = out["lds"], out["rocfft"]
    assert abs(t0 - t1) <= 1e-5 * abs(t1) and all(abs(l0[k] - l1[k]) <= 1e-5 * abs(l1[k]) for k in l1)
    for x, y in ((ga0, ga1), (gb0, gb1)):
        assert float((x - y).norm() / y.norm()) <= 1e-5
