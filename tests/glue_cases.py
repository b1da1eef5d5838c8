"""Shared bodies of the glue / metric parity tests: the same checks run against the CPU interpreter backend
(-m "not gpu") and against the HIP library on the device (-m gpu)."""
import os

import numpy as np
import torch

from oracle import glue_oracle as GO
from pde_amd import boundary as PB
from pde_amd import glue as PG
from pde_amd import metrics as PM

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-5      # fp32 product path against fp64 golden values


def gold():
    return np.load(os.path.join(GOLD, "glue.npz"))


def _rel(a, b):
    """relative L2 distance; NaNs (empty Fourier bands of the reference) must sit at the same places"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _t(x, dev, grad=False):
    t = torch.tensor(np.asarray(x), dtype=torch.float32, device=dev)
    return t.requires_grad_(True) if grad else t


BOUNDARY = {
    "bc_strip_w1": ("pred", "target", lambda p, g, m: PB.compute_boundary_loss(p, g, m, 1)),
    "bc_strip_w2": ("pred", "target", lambda p, g, m: PB.compute_boundary_loss(p, g, m, 2)),
    "bc_edge": ("pred", "target", lambda p, g, m: PB.compute_boundary_loss(p, g, m, variant="edges")),
    "bc_1d": ("pred1d", "target1d", lambda p, g, m: PB.compute_boundary_loss(p, g, m)),
    "bc_3d_faces": ("pred3d", "target3d", lambda p, g, m: PB.compute_boundary_loss(p, g, m, variant="faces")),
    "bc_3d_shells": ("pred3d", "target3d", lambda p, g, m: PB.compute_boundary_loss(p, g, m)),
    "noslip": ("pred", None, lambda p, g, m: PB.compute_noslip_loss(p)),
    "patch": ("pred_patch", None, lambda p, g, m: PB.compute_patch_boundary_loss(p, m)),
}


def check_boundary(name, dev):
    d = gold()
    pk, gk, fn = BOUNDARY[name]
    p = _t(d[pk], dev, grad=True)
    g = _t(d[gk], dev) if gk else None
    m = torch.tensor(d["mask"], device=dev)
    val = fn(p, g, m)
    assert val.dim() == 0 and val.dtype == torch.float32
    val.backward()
    assert abs(float(val.detach()) - float(d[name])) <= RTOL * abs(float(d[name])), (float(val.detach()), float(d[name]))
    assert _rel(p.grad.cpu().numpy(), d[name + "_grad"]) <= RTOL
    # exactly the reference's sparsity: zero wherever the reference gradient is zero
    assert np.array_equal(p.grad.cpu().numpy() != 0, d[name + "_grad"] != 0)


def check_boundary_scaled_upstream(dev):
    """the loss is used as lambda_bc * bc_loss + ...: the upstream gradient is not 1"""
    d = gold()
    p = _t(d["pred"], dev, grad=True)
    val = PB.compute_boundary_loss(p, _t(d["target"], dev), torch.tensor(d["mask"], device=dev), 2)
    (0.25 * val * val).backward()
    expect = 0.5 * float(d["bc_strip_w2"]) * d["bc_strip_w2_grad"]
    assert _rel(p.grad.cpu().numpy(), expect) <= RTOL


def check_metrics(dev):
    d = gold()
    p, g, m = _t(d["pred"], dev), _t(d["target"], dev), torch.tensor(d["mask"], device=dev)
    v, r = PM.compute_vrmse(p, g, m)
    assert abs(float(v) - float(d["vrmse"])) <= RTOL * float(d["vrmse"])
    assert abs(float(r) - float(d["rmse"])) <= RTOL * float(d["rmse"])
    v0, r0 = PM.compute_vrmse(p, g, torch.zeros_like(m))          # no valid channel: means over nothing
    assert torch.isnan(v0) and torch.isnan(r0)
    s = PM.nrmse_per_sample(p[:, 0, ..., 3], g[:, 0, ..., 3])
    assert _rel(s.cpu().numpy(), d["nrmse_per_sample"]) <= RTOL
    pc = PM.nrmse_per_channel(p[:, 0], g[:, 0], [(0, "vx"), (1, "vy"), (15, "p")])
    assert list(pc) == ["vx", "vy", "p"]
    assert _rel(np.stack([pc[k].cpu().numpy() for k in pc]), d["nrmse_per_channel"]) <= RTOL
    assert _rel(PM.compute_rse(p[..., :3], g[..., :3]).cpu().numpy(), d["rse"]) <= 1e-4
    mr = PM.compute_mean_rse(p[..., :3], g[..., :3])
    assert isinstance(mr, float) and abs(mr - float(d["mean_rse"])) <= 1e-4 * float(d["mean_rse"])


def check_pdebench_and_lp(dev):
    d = gold()
    p, g = _t(d["pred"], dev), _t(d["target"], dev)
    avg, per = PM.compute_pdebench_nrmse(p[:, 0], g[:, 0])
    assert isinstance(avg, float) and isinstance(per, list) and len(per) == 18
    assert _rel(np.asarray(per), d["pdebench_nrmse"]) <= RTOL and abs(avg - float(d["pdebench_nrmse_avg"])) <= RTOL * avg
    x, y = p[:, 0, ..., 2].contiguous(), g[:, 0, ..., 2].contiguous()
    for pn in (1, 2, 3):
        for red in ("none", "mean", "sum"):
            got = PM.LpLoss(p=pn, reduction=red)(x, y)
            assert _rel(got.cpu().numpy(), d[f"lp{pn}_{red}"]) <= RTOL, (pn, red)
    assert _rel(PM.FftLpLoss(p=2, reduction="none")(x, y).cpu().numpy(), d["fftlp2_none"]) <= RTOL
    assert _rel(PM.FftLpLoss(p=2, reduction="mean")(x, y, flow=1, fhigh=5).cpu().numpy(), d["fftlp2_band_mean"]) <= 5 * RTOL
    assert _rel(PM.FftLpLoss(p=1, reduction="sum")(x, y).cpu().numpy(), d["fftlp1_sum"]) <= 5 * RTOL


def check_accumulator(dev):
    gen = torch.Generator().manual_seed(3)
    batches = [(torch.randn(2, 3, 6, 5, 18, generator=gen), torch.randn(2, 3, 6, 5, 18, generator=gen)) for _ in range(3)]
    mask = torch.zeros(2, 18)
    mask[:, [0, 1, 15]] = 1
    want = GO.eval_nrmse([(p.double(), g.double(), mask) for p, g in batches], per_timestep=True)
    acc = PM.NRMSEAccumulator(per_timestep=True)
    for p, g in batches:
        acc.update(p.to(dev), g.to(dev), mask.to(dev))
    got = acc.result()
    assert got["num_samples"] == want["num_samples"] == 6
    for key in ("per_channel_nrmse", "per_channel_rmse", "per_channel_gt_rms"):
        assert sorted(got[key]) == sorted(want[key]) == [0, 1, 15]
        for c in want[key]:
            assert abs(got[key][c] - want[key][c]) <= RTOL * want[key][c]
    assert abs(got["overall_nrmse"] - want["overall_nrmse"]) <= RTOL * want["overall_nrmse"]
    for c in want["per_timestep_nrmse"]:
        assert _rel(got["per_timestep_nrmse"][c], want["per_timestep_nrmse"][c]) <= RTOL


def check_metric_func(dev, tag):
    d = np.load(os.path.join(GOLD, "metric_func.npz"))
    p, g = _t(d[f"{tag}_pred"], dev), _t(d[f"{tag}_target"], dev)
    for mode, if_mean in (("mean", True), ("full", False)):
        res = PM.metric_func(p, g, if_mean=if_mean, Lx=1.0, Ly=1.0, Lz=1.0, iLow=2, iHigh=4)
        assert len(res) == 7
        for i, r in enumerate(res):
            want = d[f"{tag}_{mode}_{i}"]
            assert tuple(r.shape) == tuple(want.shape), (tag, mode, i, r.shape, want.shape)
            assert _rel(r.cpu().numpy(), want) <= 2e-5, (tag, mode, i, _rel(r.cpu().numpy(), want))


def check_band_power(dev):
    """The binning kernel alone on awkward shapes: frame counts that are no power of two or exceed one lane row, an
    empty bin, a bin with a single member, bins longer than a workgroup; against a float64 sum."""
    from pde_amd import reductions as R
    rng = np.random.default_rng(5)
    for G, Pn, nt, sizes in ((3, 50, 5, [0, 1, 7, 30]), (2, 2000, 70, [1500, 0, 300]), (1, 9, 1, [9])):
        F = (rng.standard_normal((G, Pn, nt)) + 1j * rng.standard_normal((G, Pn, nt))).astype(np.complex64)
        members = rng.permutation(Pn)[:sum(sizes)].astype(np.int32)
        start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        got = R.get_backend().band_power(torch.from_numpy(F).to(dev), torch.from_numpy(start).to(dev),
                                         torch.from_numpy(members).to(dev), len(sizes))
        want = np.stack([(np.abs(F[:, members[start[i]:start[i + 1]]].astype(np.complex128)) ** 2).sum(axis=1)
                         for i in range(len(sizes))], axis=1)
        assert tuple(got.shape) == (G, len(sizes), nt)
        assert np.allclose(got.cpu().numpy(), want, rtol=2e-6, atol=0), (G, Pn, nt)


def _toy_model(x):
    """A stand-in for the network: each output frame mixes the two newest input frames, with a mild nonlinearity."""
    nxt = 1.6 * x[:, -1] - 0.62 * x[:, -2] + 0.05 * torch.tanh(x[:, -1])
    return torch.cat([x[:, 1:], nxt.unsqueeze(1)], dim=1)


def check_rollout(dev):
    """evaluate_rollout / summarize_rollout against the loop of tools/eval_rollout.py:93-146, :212-234 restated in
    the oracle, on the same toy model and trajectories (float64 on the CPU there)."""
    from pde_amd import rollout as PR
    rng = np.random.default_rng(11)
    M, T, H, W, C, t_in = 3, 9, 6, 8, 18, 4
    base = np.cumsum(0.3 * rng.standard_normal((M, T, H, W, C)), axis=1) + 1.0
    data = torch.tensor(base, dtype=torch.float32)
    chans = [(0, "v0"), (1, "v1"), (5, "s2")]
    want_one, want_roll = GO.rollout_evaluate(_toy_model, data.double(), t_in, chans)
    one, roll = PR.evaluate_rollout(_toy_model, data, t_in, chans, device=dev)
    for got, want in ((one, want_one), (roll, want_roll)):
        assert list(got) == [n for _, n in chans]
        for n in want:
            assert len(got[n]) == T - t_in
            assert _rel(got[n], want[n]) <= RTOL, (n, got[n], want[n])
    s, ws = PR.summarize_rollout(one, roll), GO.rollout_summary(want_one, want_roll)
    for k in ("one_step_overall", "rollout_overall"):
        assert abs(s[k] - ws[k]) <= RTOL * abs(ws[k])
    for k in ("one_step_gmean", "rollout_gmean"):
        for n in ws[k]:
            assert abs(s[k][n] - ws[k][n]) <= RTOL * abs(ws[k][n])
    # the reference's calling convention of the model
    class Net:
        def __call__(self, x, return_normalized=False):
            return (_toy_model(x) - 0.5) / 2.0, torch.tensor(0.5, device=x.device), torch.tensor(2.0, device=x.device)
    one2, _ = PR.evaluate_rollout(PR.denormalised(Net()), data, t_in, chans, device=dev)
    assert _rel(one2["v0"], want_one["v0"]) <= RTOL
    try:
        PR.evaluate_rollout(_toy_model, data[:, :t_in], t_in, chans, device=dev)
        assert False, "no frame to predict must raise"
    except ValueError:
        pass


def check_scale_accumulator(dev):
    """ResidualScaleAccumulator against tools/compute_gt_scales_per_t.py:109-152 restated in the oracle, residuals
    from the float64 oracle: Gray-Scott with the spectral Laplacian (the tool's own case) in two chunks of unequal
    size, Taylor-Green with per-sample viscosity, and Rayleigh-Benard, whose lattice is cropped."""
    import pde_amd
    from pde_amd.scales import ResidualScaleAccumulator
    from oracle import pde_oracle as PO
    rng = np.random.default_rng(3)

    def smooth(shape, amp=1.0):
        x = rng.standard_normal(shape)
        for ax in range(1, len(shape)):
            x = x + np.roll(x, 1, ax) + np.roll(x, -1, ax)
        return amp * x / 3.0 ** (len(shape) - 1)

    # Gray-Scott, FFT Laplacian
    kw = dict(dx=2.0 / 16, dy=2.0 / 16, dt=0.5, F=0.098, k=0.057, D_A=1.81e-3, D_B=1.39e-3)
    A, B = 0.5 + 0.2 * smooth((5, 7, 16, 16)), 0.3 + 0.2 * smooth((5, 7, 16, 16))
    fn = pde_amd.GrayScottPDELoss(nx=16, ny=16, **kw).to(dev)
    acc = ResidualScaleAccumulator(fn)
    batches = []
    for sl in (slice(0, 3), slice(3, 5)):
        acc.update(_t(A[sl], dev), _t(B[sl], dev))
        batches.append(PO.residuals_gray_scott_fft(torch.tensor(A[sl]), torch.tensor(B[sl]), **kw))
    res = acc.result()
    want = GO.gt_scales_per_t(batches, 16 * 16)
    assert list(res["eq_scales_per_t"]) == ["A_equation", "B_equation"]
    for k, eq in enumerate(res["eq_scales_per_t"]):
        got = res["eq_scales_per_t"][eq]
        assert got.dtype == torch.float32 and tuple(got.shape) == (5,)
        assert _rel(got.numpy(), want[k].numpy()) <= RTOL, (eq, got, want[k])
        assert abs(res["eq_scales"][eq] - float(torch.sqrt((want[k].double() ** 2).mean()))) <= RTOL * res["eq_scales"][eq]

    # Taylor-Green, per-sample nu
    u, v, p = (smooth((2, 6, 8, 12)) for _ in range(3))
    nu = np.array([0.02, 0.07])
    kw = dict(dx=2 * np.pi / 12, dy=2 * np.pi / 8, dt=0.01)
    fn = pde_amd.TaylorGreen2DPDELoss(nx=12, ny=8, dt=0.01).to(dev)
    acc = ResidualScaleAccumulator(fn)
    acc.update(_t(u, dev), _t(v, dev), _t(p, dev), nu=_t(nu, dev))
    res = acc.result()
    want = GO.gt_scales_per_t([PO.residuals_tg2d(*(torch.tensor(x) for x in (u, v, p)), torch.tensor(nu), **kw)], 8 * 12)
    for k, eq in enumerate(["continuity", "x_momentum", "y_momentum"]):
        assert _rel(res["eq_scales_per_t"][eq].numpy(), want[k].numpy()) <= RTOL, eq

    # Rayleigh-Benard: residual lattice cropped by skip_bl on the wall axis
    fn = pde_amd.RayleighBenardFullPDELoss(nx=16, ny=12, skip_bl=2).to(dev)
    ub, vb, bb = (smooth((2, 5, 16, 12)) for _ in range(3))
    kap = np.array([1e-3, 4e-3])
    acc = ResidualScaleAccumulator(fn)
    acc.update(_t(ub, dev), _t(vb, dev), _t(bb, dev), kappa=_t(kap, dev))
    res = acc.result()
    pr = fn._params()
    R = PO.residuals_rayleigh_benard(*(torch.tensor(x) for x in (ub, vb, bb)), torch.tensor(kap), dx=pr["dx"], dy=pr["dy"], dt=pr["dt"], skip_bl=2)
    sites = int(np.prod(R[0].shape[2:]))
    want = GO.gt_scales_per_t([R], sites)
    for k, eq in enumerate(fn.EQS):
        assert _rel(res["eq_scales_per_t"][eq].numpy(), want[k].numpy()) <= RTOL, eq


def check_gather(dev, transpose, spatial=(6, 8), C=18, chans=None):
    gen = torch.Generator().manual_seed(11)
    out = torch.randn(2, 3, *spatial, C, generator=gen)
    inp = torch.randn(2, 4, *spatial, C, generator=gen)
    if chans is None:
        chans = (0, 1, 15) if not transpose else (0, 1, 3)
    o = out.clone().to(dev).requires_grad_(True)
    fields = PG.gather_fields(o, inp.to(dev), chans, transpose=transpose)
    o64 = out.double().requires_grad_(True)
    want = GO.gather(o64, inp.double(), chans, transpose)
    ws = [torch.randn(f.shape, generator=gen) for f in want]
    for f, w in zip(fields, want):
        assert f.is_contiguous() and f.dtype == torch.float32
        assert torch.equal(f.cpu(), w.detach().float())           # a copy: bit-exact
    sum((f * w.to(dev)).sum() for f, w in zip(fields, ws)).backward()
    sum((f * w.double()).sum() for f, w in zip(want, ws)).backward()
    assert torch.equal(o.grad.cpu(), o64.grad.float())


def check_compute_pde_loss(dev):
    """trainer glue end to end: gather -> Taylor-Green residual loss -> gradient w.r.t. the 18-channel output"""
    import pde_amd
    from oracle import classes as OC
    gen = torch.Generator().manual_seed(5)
    out = torch.randn(2, 4, 16, 16, 18, generator=gen)
    inp = torch.randn(2, 4, 16, 16, 18, generator=gen)
    nu = torch.tensor([0.03, 0.08])
    kw = dict(nx=16, ny=16, dt=0.01)
    o = out.clone().to(dev).requires_grad_(True)
    total, losses = PG.compute_pde_loss_taylor_green_2d(o, inp.to(dev), pde_amd.TaylorGreen2DPDELoss(**kw).to(dev), nu.to(dev))
    total.backward()
    o64 = out.double().requires_grad_(True)
    u, v, p = GO.gather(o64, inp.double(), (0, 1, 15))
    want, _ = OC.build("TaylorGreen2DPDELoss", **kw)(u.double(), v.double(), p.double(), nu=nu.double())
    want.backward()
    assert abs(float(total.detach()) - float(want.detach())) <= RTOL * float(want.detach())
    assert _rel(o.grad.cpu().numpy(), o64.grad.numpy()) <= RTOL
    cfg = {"pde_type": "taylor_green_2d", "physics": dict(kw, eq_scales=None, unknown_key=1)}
    fn = PG.create_pde_loss(cfg, dev)
    t2, _ = PG.compute_pde_loss_generic(o.detach(), inp.to(dev), fn, cfg, {"nu": nu})
    assert abs(float(t2) - float(want.detach())) <= RTOL * float(want.detach())


def check_denormalised(dev):
    """Denormalised(output_norm, mean, std) against the trainers' `output = output_norm * std + mean` followed by the same
    losses (finetune/train_taylor_green_2d_lora_v3.py:500-508): every loss value bit for bit (the kernels round the
    product as the materialised tensor would hold it), the gradient w.r.t. output_norm to fp32 accuracy against
    autograd through the materialised product, with and without the shared gradient buffer; Rayleigh-Benard's
    transposed gather; misuse raises."""
    import pde_amd
    gen = torch.Generator().manual_seed(9)
    B, T, H, W, C = 2, 4, 20, 16, 18
    on0 = torch.randn(B, T, H, W, C, generator=gen).to(dev)
    inp = torch.randn(B, 3, H, W, C, generator=gen).to(dev)
    tgt = torch.randn(B, T, H, W, C, generator=gen).to(dev)
    mean = (0.5 * torch.randn(B, 1, 1, 1, C, generator=gen)).to(dev)
    std = (0.5 + torch.rand(B, 1, 1, 1, C, generator=gen)).to(dev)
    mask = torch.zeros(B, C); mask[:, [0, 1, 15]] = 1
    mask = mask.to(dev)
    nu = torch.tensor([0.02, 0.07]).to(dev)
    fn = pde_amd.TaylorGreen2DPDELoss(nx=W, ny=H, dt=0.01).to(dev)
    rb = pde_amd.RayleighBenardFullPDELoss(nx=H, ny=W, skip_bl=2).to(dev)

    def losses(o):
        pde, _ = PG.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)
        rbl, _ = PG.compute_pde_loss_rayleigh_benard(o, inp, rb, nu)
        bc = PB.compute_boundary_loss(o, tgt, mask, 2)
        ns = PB.compute_noslip_loss(o)
        seam = PB.compute_patch_boundary_loss(o, mask, patch_size=4)
        return 1e-6 * pde + 1e-4 * rbl + 0.5 * bc + 0.25 * ns + 0.125 * seam, (pde, rbl, bc, ns, seam)

    a = on0.clone().requires_grad_(True)
    la, pa = losses(a * std + mean)
    la.backward()
    for share in (True, False):
        b = on0.clone().requires_grad_(True)
        den = pde_amd.Denormalised(b, mean, std, share=share)
        assert tuple(den.shape) == (B, T, H, W, C) and den.dim() == 5
        lb, pb = losses(den)
        lb.backward()
        for x, y in zip(pa, pb):
            assert float(x.detach()) == float(y.detach()), (share, float(x.detach()), float(y.detach()))
        assert _rel(b.grad.cpu().numpy(), a.grad.cpu().numpy()) <= 2e-6, share
        assert np.array_equal((b.grad == 0).cpu().numpy(), (a.grad == 0).cpu().numpy())
    # 1-D and 3-D layouts: strips / shells and the single-field gathers, values bit for bit, gradients to fp32 accuracy
    for shape, pde in (((2, 3, 24, C), "b1d"), ((2, 3, 8, 6, 8, C), "h3d")):
        nd = len(shape) - 3
        o0 = torch.randn(*shape, generator=gen).to(dev)
        i0 = torch.randn(*shape, generator=gen).to(dev)
        t0 = torch.randn(*shape, generator=gen).to(dev)
        bc = (shape[0],) + (1,) * (nd + 1) + (C,)
        mu, sd = (0.3 * torch.randn(*bc, generator=gen)).to(dev), (0.5 + torch.rand(*bc, generator=gen)).to(dev)
        m3 = torch.zeros(shape[0], C); m3[:, [0, 3, 5]] = 1
        m3 = m3.to(dev)
        par = torch.tensor([0.02, 0.05]).to(dev)
        if pde == "b1d":
            fn1 = pde_amd.Burgers1DPDELoss(dx=2 * np.pi / 24, dt=0.01).to(dev)
            f = lambda o: PB.compute_boundary_loss(o, t0, m3, 2) + 1e-6 * PG.compute_pde_loss_burgers_1d(o, i0, fn1, par)[0]
        else:
            fn1 = pde_amd.Heat3DPDELoss(nx=8, ny=6, nz=8).to(dev)
            f = lambda o: PB.compute_boundary_loss(o, t0, m3, 1) + 1e-3 * PG.compute_pde_loss_heat_3d(o, i0, fn1, par)[0]
        a1 = o0.clone().requires_grad_(True)
        l1 = f(a1 * sd + mu)
        l1.backward()
        b1 = o0.clone().requires_grad_(True)
        l2 = f(pde_amd.Denormalised(b1, mu, sd))
        l2.backward()
        assert float(l1.detach()) == float(l2.detach()), pde
        assert _rel(b1.grad.cpu().numpy(), a1.grad.cpu().numpy()) <= 2e-6, pde
    # compute_vrmse applies std and mean while reading (per-sample groups, so only the order of the fp64 sums differs);
    # consumers outside the package take the materialised product
    den = pde_amd.Denormalised(on0, mean, std)
    v1 = PM.compute_vrmse(den, tgt, mask)
    v2 = PM.compute_vrmse(on0 * std + mean, tgt, mask)
    assert abs(float(v1[0]) - float(v2[0])) <= 1e-6 * float(v2[0]) and abs(float(v1[1]) - float(v2[1])) <= 1e-6 * float(v2[1])
    assert torch.equal(den.materialise(), on0 * std + mean)
    # scalars and [C] vectors broadcast too; a per-frame std does not describe one value per sample and channel
    d2 = pde_amd.Denormalised(on0, 0.25, torch.full((C,), 2.0).to(dev))
    assert float(PB.compute_noslip_loss(d2)) == float(PB.compute_noslip_loss(on0 * 2.0 + 0.25))
    try:
        pde_amd.Denormalised(on0, mean, torch.ones(B, T, 1, 1, C).to(dev))
        assert False, "per-frame std must raise"
    except ValueError:
        pass
    assert pde_amd.shared_grad(den) is den
    try:
        den + 1
        assert False, "a Denormalised is not a tensor"
    except TypeError:
        pass


def check_shared_grad(dev):
    """every loss of one training step on one tagged output: same values, and the gradient written through the shared buffer
    equals the sum autograd builds from the separate full-size gradients (up to the order of a few fp32 additions)"""
    import pde_amd
    gen = torch.Generator().manual_seed(5)
    B, T, H, W, C = 2, 4, 20, 16, 18
    out0 = torch.randn(B, T, H, W, C, generator=gen).to(dev)
    inp = torch.randn(B, 3, H, W, C, generator=gen).to(dev)
    tgt = torch.randn(B, T, H, W, C, generator=gen).to(dev)
    mask = torch.zeros(B, C); mask[:, [0, 1, 15]] = 1
    nu = torch.tensor([0.02, 0.07]).to(dev)
    fn = pde_amd.TaylorGreen2DPDELoss(nx=W, ny=H, dt=0.01).to(dev)

    def losses(o):
        pde, parts = PG.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)
        bc = PB.compute_boundary_loss(o, tgt, mask.to(dev), 2)
        ns = PB.compute_noslip_loss(o)
        seam = PB.compute_patch_boundary_loss(o, mask.to(dev), patch_size=4)
        extra = (o[..., 3] ** 2).mean()          # an ordinary torch consumer of the same tensor
        return 1e-6 * pde + 0.5 * bc + 0.25 * ns + 0.125 * seam + extra, (pde, bc, ns, seam)

    a = out0.clone().requires_grad_(True)
    la, pa = losses(a)
    la.backward()
    b = out0.clone().requires_grad_(True)
    lb, pb = losses(pde_amd.shared_grad(b))
    lb.backward()
    for x, y in zip(pa, pb):
        assert torch.equal(x.detach().cpu(), y.detach().cpu())
    ga, gb = a.grad.cpu().double(), b.grad.cpu().double()
    assert float((ga - gb).abs().max()) <= 1e-6 * float(ga.abs().max())
    # a second step reuses nothing of the first; no-grad and non-fp32 inputs come back untagged
    with torch.no_grad():
        assert pde_amd.shared_grad(b) is b
    assert pde_amd.shared_grad(b.detach()) is not None
    c = out0.clone().requires_grad_(True)
    lc, _ = losses(pde_amd.shared_grad(c))
    lc.backward()
    assert torch.equal(c.grad.cpu(), b.grad.cpu())


def check_residuals_f64(dev):
    """Double-precision fields take the double-precision residual evaluation: the residual fields against the fp64 oracle
    to rounding (1e-12, where the fp32 evaluation stops near 1e-6), and the scale accumulator on them against
    tools/compute_gt_scales_per_t.py:109-152 restated in the oracle, to the same accuracy."""
    import pde_amd
    from pde_amd.scales import ResidualScaleAccumulator
    from oracle import pde_oracle as PO
    rng = np.random.default_rng(11)

    def smooth(shape):
        x = rng.standard_normal(shape)
        for ax in range(1, len(shape)):
            x = x + np.roll(x, 1, ax) + np.roll(x, -1, ax)
        return x / 3.0 ** (len(shape) - 1)

    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    tight = 1e-12
    # Taylor-Green, per-sample viscosity
    u, v, p = (smooth((2, 6, 8, 12)) for _ in range(3))
    nu = np.array([0.02, 0.07])
    fn = pde_amd.TaylorGreen2DPDELoss(nx=12, ny=8, dt=0.01).to(dev)
    got = fn.residuals(t64(u), t64(v), t64(p), nu=t64(nu))
    want = PO.residuals_tg2d(torch.tensor(u), torch.tensor(v), torch.tensor(p), torch.tensor(nu), dx=2 * np.pi / 12, dy=2 * np.pi / 8, dt=0.01)
    for eq, w in zip(fn.EQS, want):
        assert got[eq].dtype == torch.float64
        assert _rel(got[eq].cpu().numpy(), w.numpy()) <= tight, eq
    r32 = fn.residuals(_t(u, dev), _t(v, dev), _t(p, dev), nu=_t(nu, dev))
    assert r32["continuity"].dtype == torch.float32 and _rel(r32["x_momentum"].cpu().numpy(), want[1].numpy()) > 1e-9
    # Gray-Scott with the spectral Laplacian (the scale tool's own case), two chunks; the transforms run in double as well
    kw = dict(dx=2.0 / 16, dy=2.0 / 16, dt=0.5, F=0.098, k=0.057, D_A=1.81e-3, D_B=1.39e-3)
    A, B = 0.5 + 0.2 * smooth((5, 7, 16, 16)), 0.3 + 0.2 * smooth((5, 7, 16, 16))
    fn = pde_amd.GrayScottPDELoss(nx=16, ny=16, **kw).to(dev)
    acc = ResidualScaleAccumulator(fn)
    batches = []
    for sl in (slice(0, 3), slice(3, 5)):
        acc.update(t64(A[sl]), t64(B[sl]))
        batches.append(PO.residuals_gray_scott_fft(torch.tensor(A[sl]), torch.tensor(B[sl]), **kw))
    res = acc.result()
    want = GO.gt_scales_per_t(batches, 16 * 16)
    for k, eq in enumerate(res["eq_scales_per_t"]):
        assert _rel(res["eq_scales_per_t"][eq].double().numpy(), want[k].float().double().numpy()) <= 1e-7, eq      # stored as fp32, like the tool's
        assert abs(res["eq_scales"][eq] - float(torch.sqrt((want[k].double() ** 2).mean()))) <= 1e-7 * res["eq_scales"][eq]
    # Rayleigh-Benard: walls and a cropped lattice
    u, v, b = (smooth((2, 5, 16, 12)) for _ in range(3))
    kap = np.array([1e-3, 5e-3])
    kwr = dict(nx=16, ny=12, Lx=4.0, Ly=1.0, dt=0.25, skip_bl=3)
    fn = pde_amd.RayleighBenardFullPDELoss(**kwr).to(dev)
    got = fn.residuals(t64(u), t64(v), t64(b), kappa=t64(kap))
    want = PO.residuals_rayleigh_benard(torch.tensor(u), torch.tensor(v), torch.tensor(b), torch.tensor(kap), dx=4.0 / 16, dy=1.0 / 11,
                                        dt=0.25, skip_bl=3)
    for eq, w in zip(fn.EQS, want):
        g = got[eq].cpu().numpy()[..., 3:-3]
        assert g.shape == tuple(w.shape) and _rel(g, w.numpy()) <= tight, eq
        assert float(got[eq][..., :3].abs().max()) == 0.0 and float(got[eq][..., -3:].abs().max()) == 0.0


def check_validation_scalars(dev, C=18):
    """validation_scalars (one sweep over the output) against the separate functions validate() calls
    (finetune/train_taylor_green_2d_lora_v3.py:252-275), on a plain output and on a Denormalised one: the boundary and
    residual losses bit for bit, RMSE and VRMSE to the rounding of their fp64 sums.  C = 36: two LDS images of 256 points
    would exceed a workgroup's 64 KiB, so the sweep takes the kernel without the images."""
    import pde_amd
    gen = torch.Generator().manual_seed(17)
    B, T, H, W = 3, 4, 12, 20
    on = torch.randn(B, T, H, W, C, generator=gen).to(dev)
    inp = torch.randn(B, 2, H, W, C, generator=gen).to(dev)
    tgt = torch.randn(B, T, H, W, C, generator=gen).to(dev)
    mean = (0.3 * torch.randn(B, 1, 1, 1, C, generator=gen)).to(dev)
    std = (0.5 + torch.rand(B, 1, 1, 1, C, generator=gen)).to(dev)
    mask = torch.zeros(B, C)
    mask[:, [0, 1, 15]] = 1
    mask = mask.to(dev)
    nu = torch.tensor([0.02, 0.07, 0.04]).to(dev)
    fn = pde_amd.TaylorGreen2DPDELoss(nx=W, ny=H, dt=0.01).to(dev)
    for make in (lambda: on, lambda: pde_amd.Denormalised(on, mean, std)):
        with torch.no_grad():
            o = make()
            want = (PB.compute_boundary_loss(o, tgt, mask, 2), PG.compute_pde_loss_taylor_green_2d(o, inp, fn, nu)[0],
                    *reversed(PM.compute_vrmse(o, tgt, mask)))
        got = pde_amd.validation_scalars(make(), inp, tgt, mask, fn, (0, 1, 15), bc_width=2, nu=nu)
        for a, b, name in zip(got, want, ("bc", "pde", "rmse", "vrmse")):      # the statistics are added in another order: fp64 rounding
            assert float(a) == float(b) if name in ("bc", "pde") else abs(float(a) - float(b)) <= 1e-6 * abs(float(b)), (name, float(a), float(b))
