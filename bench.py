"""Benchmark: fused PDE-residual loss forward+backward, Taylor-Green 2-D, B=16 T=16 256x256 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one TaylorGreen2DPDELoss forward + backward over one batch of synthetic fields already
resident in HBM.  With N > 1 (launched through torch.distributed.run, one rank per GPU) every rank owns
its own 16 samples (weak scaling) and the step's loss scalars go through one packed RCCL all-reduce.
Rank 0 prints one JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import pde_amd  # noqa: E402
from pde_amd import dist as pdist  # noqa: E402
from pde_amd import functional as PF  # noqa: E402

B, T, H, W = 16, 16, 256, 256
HBM_PEAK_GBS = 8000.0
EQ_SCALES = {"continuity": 6.065e-07, "x_momentum": 5.960e-06, "y_momentum": 5.960e-06}


def synth_fields(b, t, h, w, seed, device):
    """Band-limited decaying cosines plus white noise (sigma 0.1), generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.arange(w, device=device, dtype=torch.float32) * (2 * math.pi / w)
    y = torch.arange(h, device=device, dtype=torch.float32) * (2 * math.pi / h)
    tt = torch.arange(t, device=device, dtype=torch.float32)
    out = []
    for f in range(3):
        acc = 0.1 * torch.randn((b, t, h, w), generator=g, device=device)
        for k in range(1, 9):
            a = (0.1 + 0.9 * torch.rand((b, 1, 1, 1), generator=g, device=device)) / k
            ph = 6.2831853 * torch.rand((b, 1, 1, 2), generator=g, device=device)
            gam = 0.02 * k
            acc = acc + a * torch.cos(k * x.view(1, 1, 1, w) + ph[..., 0:1]) * torch.cos(k * y.view(1, 1, h, 1) + ph[..., 1:2]) \
                * torch.exp(-gam * tt.view(1, t, 1, 1))
        out.append(acc.contiguous())
    return out


def cpu_baseline(seconds=12.0):
    """The CPU oracle (reference-equivalent eager PyTorch, fp32) on a bounded sample of the same workload."""
    from oracle import classes as OC
    bs = 2
    g = torch.Generator().manual_seed(1234)
    fields = [torch.randn(bs, T, H, W, generator=g).requires_grad_(True) for _ in range(3)]
    nu = 0.01 + 0.09 * torch.rand(bs, generator=g)
    fn = OC.build("TaylorGreen2DPDELoss", nx=W, ny=H, dt=0.01, eq_scales=EQ_SCALES)
    pts = bs * (T - 2) * H * W
    n, t0 = 0, None
    best = float("inf")
    start = time.perf_counter()
    while True:
        for f in fields:
            f.grad = None
        t0 = time.perf_counter()
        total, _ = fn(*fields, nu=nu)
        total.backward()
        dt = time.perf_counter() - t0
        n += 1
        if n > 1:
            best = min(best, dt)
        if time.perf_counter() - start > seconds and n >= 3:
            break
    # the same sample on one thread (a few passes only)
    threads = torch.get_num_threads()
    one = float("inf")
    try:
        torch.set_num_threads(1)
        for i in range(3):
            for f in fields:
                f.grad = None
            t0 = time.perf_counter()
            total, _ = fn(*fields, nu=nu)
            total.backward()
            if i:
                one = min(one, time.perf_counter() - t0)
    finally:
        torch.set_num_threads(threads)
    return {"value": pts / best, "unit": "grid-points/s", "cores": threads, "kind": "port",
            "sample": f"oracle TaylorGreen2D fwd+bwd fp32, B={bs} T={T} {H}x{W}, best of {n - 1} after warm-up",
            "one_thread_value": pts / one}


def measured_traffic(kernel_substr):
    """HBM-side bytes per launch of a kernel from the committed hardware-counter passes (tools/profile_round.sh):
    2 x FETCH_SIZE + WRITE_SIZE, each collected in its own rocprofv3 --pmc run.  None when no measurement is on file."""
    path = os.path.join(ROOT, "pde_amd", "measured", "tg2d_traffic.json")
    try:
        with open(path) as fh:
            data = json.load(fh)
    except OSError:
        return None
    for name, rec in data.items():
        if kernel_substr in name:
            return rec["total_bytes"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--mode", default="fused", choices=["fused", "auto"],
                    help="loss module backward_mode: 'fused' = loss and gradient in one pass of the adjoint kernel (it recomputes "
                         "every residual anyway), 'auto' = forward kernel, then adjoint kernel")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("PDE_AMD_BENCH_FORCE_DIST") == "1"     # the latter: rehearse the RCCL path with one rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    fields = [f.requires_grad_(True) for f in synth_fields(B, T, H, W, 1234 + 1000 * 2 + rank, dev)]
    gnu = torch.Generator(device=dev).manual_seed(99 + rank)
    nu = 0.01 + 0.09 * torch.rand(B, generator=gnu, device=dev)
    loss_fn = pde_amd.TaylorGreen2DPDELoss(nx=W, ny=H, dt=0.01, eq_scales=EQ_SCALES).to(dev)
    loss_fn.backward_mode = args.mode      # upstream_hint stays 1.0: the step backpropagates `total` itself

    backend = PF.get_backend()

    one = torch.ones((), device=dev)

    def loss_step():
        """forward + backward of the loss through the nn.Module API; gradients land in fields[i].grad"""
        for f in fields:
            f.grad = None
        total, losses = loss_fn(*fields, nu=nu)
        total.backward(gradient=one)      # the upstream gradient of the loss: a resident 1.0 instead of a fill launch per step
        # detached: nothing of the autograd graph (and its per-leaf accumulation nodes, which remember the stream they
        # were created on) outlives the step, so a later capture on another stream starts clean
        # the step's scalars as one vector, ready for the packed all-reduce
        return pdist.pack_scalars({"total": total, **losses}), ["total", *losses]

    for _ in range(args.warmup):
        buf, keys = loss_step()
        pdist.reduce_packed_async(keys, buf, force=distributed).result()

    # The step is launch-bound on the host (five small launches plus autograd for ~0.18 ms of GPU work), so the
    # steady-state loop replays it as a HIP graph; every `probe`-th step (10th by default) of the timed region runs eagerly instead with
    # HIP events recorded right around the forward and the adjoint launches, which is where the roofline numbers
    # come from.  --no-graph runs every step eagerly.
    # Two graphs of the same step that leave their packed scalars in two different buffers: the all-reduce of step i runs in
    # place on its buffer, beside step i + 1, and is joined before step i + 2 writes that buffer again.
    graphs, g_bufs = [], []
    if not args.no_graph:
        try:
            buf = None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                loss_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for _ in range(2):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) stay free to query events
                    gb, keys = loss_step()
                gr.replay()
                torch.cuda.synchronize()
                graphs.append(gr)
                g_bufs.append(gb)
        except Exception as e:      # capture unsupported here: eager loop
            print(f"bench: HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs, g_bufs = [], []
            torch.cuda.synchronize()
    graph = graphs[0] if graphs else None
    # an eager probe step costs about twice a replayed one (the host cannot keep up), so only every 10th step is one
    probe = max(2, min(10, args.steps // 3)) if graph is not None else 1
    n_probe = (args.steps + probe - 1) // probe
    pool = lambda: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_probe + 2)]
    bwd_events, fwd_events = [], []

    def step(i):
        if graph is not None and i % probe != probe - 1:
            graphs[i % 2].replay()
            buf = g_bufs[i % 2]
        else:
            backend.backward_events, backend.forward_events = bwd_events, fwd_events
            buf, names = loss_step()
            backend.backward_events = backend.forward_events = None
            return pdist.reduce_packed_async(names, buf, force=distributed)
        # the step's scalars: one packed all-reduce, started now and joined a step later so that it runs beside the next step
        return pdist.reduce_packed_async(keys, buf, force=distributed)

    backend.backward_event_pool, backend.forward_event_pool = pool(), pool()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for i in range(args.steps):
        nxt = step(i)
        if pending is not None:
            pending.join()
        pending = nxt
    red = pending.result()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    pts_per_rank = B * (T - 2) * H * W
    ms_step = elapsed / args.steps * 1e3
    value = pts_per_rank * world / (elapsed / args.steps)

    if rank == 0:
        k_ms = sorted(x.elapsed_time(y) for x, y in bwd_events)
        f_ms = sorted(x.elapsed_time(y) for x, y in fwd_events)      # empty in fused mode: there is no forward launch
        bwd_avg = sum(k_ms) / len(k_ms)                   # the adjoint kernel launch alone
        e_in = 3 * B * T * H * W
        bwd_bytes = 8 * e_in            # adjoint: read every field once, write every gradient once
        achieved = bwd_bytes / (bwd_avg * 1e-3) / 1e9
        out = {
            "metric": "PDE-residual loss fwd+bwd grid-points/sec",
            "value": value, "unit": "grid-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Taylor-Green 2D incompressible NS residual fwd+bwd, B=16 per GPU, T=16, 256x256, 3 fields",
                       "parallelism": f"dp{world}", "points_per_step_per_gpu": pts_per_rank,
                       "launch": f"hip-graph replay, every {probe}th step eager with kernel events" if graph is not None else "eager",
                       "backward_mode": args.mode},
            "roofline": {"bound": "hbm", "kernel": "bwd_march_q_kernel<Pde_tg2d> (the adjoint; one launch per step; HIP events recorded by "
                                                     "the library on the launch stream right before and after that launch" +
                                                     ("; in fused mode the kernel also sums the loss)" if args.mode == "fused" else ")"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("bwd_march_q_kernel<pde_amd::Pde_tg2d, 4, true>" if args.mode == "fused" else "bwd_march_q_kernel<pde_amd::Pde_tg2d, 4>"), "algorithmic_bytes": bwd_bytes, "avg_ms": bwd_avg,
                         "median_ms": k_ms[len(k_ms) // 2], "timed_launches": len(k_ms),
                         **({"fwd_median_ms": f_ms[len(f_ms) // 2], "fwd_algorithmic_bytes": 4 * e_in,
                             "fwd_traffic": measured_traffic("fwd_march_q_kernel")} if f_ms else {})},
            "loss": float(red["total"]),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
