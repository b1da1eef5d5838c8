"""Benchmark: fused PDE-residual loss forward+backward, Taylor-Green 2-D, B=16 T=16 256x256 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rotate K]
    python bench.py --gpus N --case gray_scott --strong      (64 Gray-Scott samples split over the ranks)

One step = one TaylorGreen2DPDELoss forward + backward over one batch of synthetic fields already
resident in HBM.  With N > 1 every rank (one process per GPU) owns its own 16 samples (weak scaling)
and the step's loss scalars go through one packed RCCL all-reduce.  `python bench.py --gpus N` starts
the N ranks itself (a torch.distributed.run child process, started before anything touches the GPU);
under an external launcher (WORLD_SIZE set) it is one of the ranks.  Rank 0 prints one JSON line.
"""
import argparse
import json
import math
import os
import socket
import subprocess
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


def _time_oracle(bs, seconds, min_passes, threads=None):
    """Best fwd+bwd time of the CPU oracle on a batch of `bs` samples of the benchmark's shape."""
    from oracle import classes as OC
    g = torch.Generator().manual_seed(1234)
    fields = [torch.randn(bs, T, H, W, generator=g).requires_grad_(True) for _ in range(3)]
    nu = 0.01 + 0.09 * torch.rand(bs, generator=g)
    fn = OC.build("TaylorGreen2DPDELoss", nx=W, ny=H, dt=0.01, eq_scales=EQ_SCALES)
    old = torch.get_num_threads()
    if threads is not None:
        torch.set_num_threads(threads)
    try:
        n, best = 0, float("inf")
        start = time.perf_counter()
        while True:
            for f in fields:
                f.grad = None
            t0 = time.perf_counter()
            total, _ = fn(*fields, nu=nu)
            total.backward()
            dt = time.perf_counter() - t0
            n += 1
            if n > 1:                      # the first pass warms the allocator up
                best = min(best, dt)
            if n >= min_passes + 1 and time.perf_counter() - start > seconds:
                break
    finally:
        torch.set_num_threads(old)
    return bs * (T - 2) * H * W / best, n - 1


def cpu_baseline():
    """The CPU oracle (reference-equivalent eager PyTorch, fp32) on the benchmark's own batch (B=16), all host threads
    torch uses; next to it a B=2 sample, which stays cache-resident and runs far faster per point, and the same on
    one thread."""
    threads = torch.get_num_threads()
    full, n_full = _time_oracle(B, 14.0, 2)
    small, _ = _time_oracle(2, 4.0, 3)
    one, _ = _time_oracle(2, 0.0, 2, threads=1)
    return {"value": full, "unit": "grid-points/s", "cores": threads, "kind": "port",
            "sample": f"oracle TaylorGreen2D fwd+bwd fp32 at the benchmark's batch, B={B} T={T} {H}x{W}, best of {n_full} "
                      f"passes after one warm-up",
            "small_sample_value": small, "small_sample": f"B=2 T={T} {H}x{W} (cache-resident)",
            "one_thread_value": one}


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` (its name as a profiler prints it, pde_amd_last_adjoint_kernel) from the committed
    hardware-counter passes (tools/profile_round.sh): 2 x FETCH_SIZE + WRITE_SIZE, each collected in its own rocprofv3 --pmc
    run.  A kernel without a measurement on file is an error: the figure must belong to the kernel that was timed."""
    path = os.path.join(ROOT, "pde_amd", "measured", "tg2d_traffic.json")
    with open(path) as fh:
        data = json.load(fh)
    if kernel not in data:
        raise KeyError(f"bench: no HBM-traffic measurement on file for {kernel!r} in {path}; run tools/profile_round.sh and copy "
                       f"its <tag>_traffic.json there (kernels on file: {sorted(data)})")
    return data[kernel]["total_bytes"]


def last_adjoint_kernel():
    """the adjoint kernel of the last launch that was timed (the backend asks the library on the launching thread)"""
    return PF.HipBackend.last_adjoint_kernel


# The reference's other benchmark configurations, run after the timed region (they are not part of `value`): class,
# constructor arguments, field shape, per-sample parameters.
def other_case_table(dev):
    U = lambda b, lo, hi: lo + (hi - lo) * torch.rand(b, device=dev)
    return [
        ("Burgers 1-D B=20, [20,9,256] x1", "Burgers1DPDELoss", {}, (20, 9, 256), lambda b: dict(nu=U(b, 0.005, 0.05))),
        ("Rayleigh-Benard B=64, [64,9,512,128] x3", "RayleighBenardFullPDELoss", {}, (64, 9, 512, 128), lambda b: dict(kappa=U(b, 1e-4, 1e-2))),
        ("shear flow n-PINN B=64, [64,9,256,512] x4", "ShearFlowPDELossNPINN", {}, (64, 9, 256, 512), lambda b: {}),
        ("Gray-Scott FD-c4, 8 samples per GPU, [8,9,128,128] x2", "GrayScottPDELoss", dict(laplacian="fd4"), (8, 9, 128, 128), lambda b: {}),
        ("Gray-Scott FFT Laplacian, 8 samples per GPU, [8,9,128,128] x2", "GrayScottPDELoss", {}, (8, 9, 128, 128), lambda b: {}),
        ("heat 3-D B=2, [2,9,128,128,128] x1", "Heat3DPDELoss", {}, (2, 9, 128, 128, 128), lambda b: dict(alpha=U(b, 0.005, 0.05))),
        ("APEBench Burgers 3-D B=1, [1,9,128,128,128] x3", "APEBenchBurgers3DPDELoss", dict(nx=128), (1, 9, 128, 128, 128), lambda b: {}),
    ]


def run_other_cases(dev, launches=16):
    """fwd+bwd of each other configuration through the nn.Module API in 'fused' mode (which resolves to the fastest measured
    kernels of each PDE): step time from events around whole steps, the adjoint kernel's launch time from events the library
    records around that launch, its name from the library, frac = 8 E_in / time / 8 TB/s (read every field once, write
    every gradient once)."""
    import inspect
    backend = PF.get_backend()
    out = []
    for tag, cls_name, kw, shape, params in other_case_table(dev):
        mod = getattr(pde_amd, cls_name)(**kw).to(dev)
        mod.backward_mode = "fused"
        names = set(params(1))
        nf = sum(1 for q in inspect.signature(mod.forward).parameters.values() if q.default is inspect.Parameter.empty and q.name not in names)
        gen = torch.Generator(device=dev).manual_seed(4321)
        fields = [(0.5 * torch.randn(shape, device=dev, generator=gen)).requires_grad_(True) for _ in range(nf)]
        pk = params(shape[0])

        def step():
            for f in fields:
                f.grad = None
            total, _ = mod(*fields, **pk)
            total.backward()

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < 100.0:      # working clocks
            for _ in range(8):
                step()
            torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for a, b in evs:
            a.record(); step(); b.record()
        torch.cuda.synchronize()
        step_ms = sorted(a.elapsed_time(b) for a, b in evs)[launches // 2]
        rep = {}
        # the same step replayed as a HIP graph (loss_fn.capture): what a training loop pays once the host's launch work
        # is out of the way, which is most of an eager step of the small configurations
        try:
            cap = mod.capture(*fields, **pk)
            for _ in range(8):
                cap()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4 * launches):
                cap()
            b.record()
            torch.cuda.synchronize()
            rep["replayed_ms_per_step"] = a.elapsed_time(b) / (4 * launches)
            del cap
        except Exception as e:      # capture unsupported here
            rep["replayed_ms_per_step"] = None
            rep["replay_error"] = f"{type(e).__name__}: {e}"[:200]
        # The adjoint launch by the library's events.  The host needs longer to enqueue an eager step of the small cases than
        # the GPU to run it, and on an idle GPU the start event is stamped when it is enqueued, i.e. a launch latency before
        # the kernel starts: a few hundred microseconds of fill work queued ahead of every step keep the GPU busy until the
        # step's launches are all enqueued (and leave the caches cold, like the headline's rotating input sets).  Only where
        # the replayed step shows the eager one to be host-bound.
        kev = []
        host_bound = rep.get("replayed_ms_per_step") is not None and rep["replayed_ms_per_step"] < 0.9 * step_ms
        busy = torch.empty(1 << 27 if host_bound else 1, dtype=torch.float32, device=dev)
        backend.backward_events, backend.forward_events = kev, None
        try:
            for _ in range(launches):
                for _ in range(3 if host_bound else 0):
                    busy.zero_()
                step()
        finally:
            backend.backward_events = backend.forward_events = None
        torch.cuda.synchronize()
        k_ms = sorted(a.elapsed_time(b) for a, b in kev)
        e_in = nf
        for v in shape:
            e_in *= v
        rec = {"workload": tag, "ms_per_step": step_ms, "backward_mode": PF.resolve_mode("fused", mod.SPEC, fields[0].numel()),
               "kernel": last_adjoint_kernel(), "algorithmic_bytes": 8 * e_in, "timed_launches": len(k_ms)}
        if k_ms:
            avg = sum(k_ms) / len(k_ms)
            rec.update(avg_ms=avg, frac=8 * e_in / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS)
        rec.update(rep, gpu_kept_busy_for_kernel_timing=host_bound)
        out.append(rec)
        del fields, mod, busy
        torch.cuda.empty_cache()
    return out


GS_TOTAL, GS_T, GS_N = 64, 9, 128      # Gray-Scott as the reference runs it: 64 samples of 128 x 128, split over the ranks


def gray_scott_strong(args, world, rank, dev, distributed):
    """`--case gray_scott --strong`: the Gray-Scott residual loss fwd+bwd on 64 samples in total, 64 / N per rank (strong
    scaling), with the reference's spectral Laplacian ('fft') and with the 4th-order stencil ('fd4').  A step is the
    module call + backward replayed as one HIP graph (loss_fn.capture()); its loss scalars go through one packed
    all-reduce per window of steps.  Prints one JSON line: `value` is the spectral variant's global grid-points/s."""
    import torch.distributed as dist
    from pde_amd.stencil.specs import SPECS
    if GS_TOTAL % world:
        print(f"bench: {GS_TOTAL} samples do not split evenly over {world} ranks", file=sys.stderr)
        return 2
    b = GS_TOTAL // world
    backend = PF.get_backend()
    window = max(1, args.reduce_window)
    variants = {}
    for lap in ("fft", "fd4"):
        mod = pde_amd.GrayScottPDELoss(laplacian=lap).to(dev)
        mod.backward_mode = "fused"
        spec = SPECS[mod.SPEC]
        gen = torch.Generator(device=dev).manual_seed(77 + rank)
        fields = [(0.5 + 0.25 * torch.rand((b, GS_T, GS_N, GS_N), device=dev, generator=gen)).requires_grad_(True) for _ in range(2)]
        names = [*mod.EQS, "total"]
        win = pdist.ScalarWindow(names, window, dev, force=distributed)
        staging = torch.zeros(len(names), device=dev)
        mod.scalars_out = staging
        try:
            step_fn = mod.capture(*fields)
            launch = "hip-graph replay (loss_fn.capture)"
        except Exception as e:      # capture unsupported here: eager steps
            print(f"bench: HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            launch = "eager"

            def step_fn():
                for f in fields:
                    f.grad = None
                total, losses = mod(*fields)
                total.backward(gradient=mod.upstream_scalar(dev))
                return total, losses

        def run(k_steps):
            for i in range(k_steps):
                win.before(i)
                step_fn()
                win.row(i).copy_(staging)
                win.after(i)
            win.flush(k_steps)
            return win.drain()

        run(max(args.warmup, 2 * window))
        if distributed and lap == "fft":
            pdist.reduce_packed_async(["x"], torch.zeros(1, device=dev), force=True).result()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < args.settle_ms:      # working clocks
            for _ in range(16):
                step_fn()
            torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        red = {k: v[-1] for k, v in run(args.steps)[-1][1].items()}
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if distributed:
            tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax)
        pts = GS_TOTAL * (GS_T - spec.tlo - spec.thi) * GS_N * GS_N
        rec = {"value": pts / (elapsed / args.steps), "ms_per_step": elapsed / args.steps * 1e3, "launch": launch,
               "loss": float(red["total"])}
        # the stencil adjoint's launch time, from a few eager steps with the library's events around that launch
        kev = []
        mod.scalars_out = None
        backend.backward_events, backend.forward_events = kev, None
        try:
            for _ in range(16):
                for f in fields:
                    f.grad = None
                total, _ = mod(*fields)
                total.backward(gradient=mod.upstream_scalar(dev))
        finally:
            backend.backward_events = backend.forward_events = None
        torch.cuda.synchronize()
        k_ms = sorted(x.elapsed_time(y) for x, y in kev)
        if k_ms:
            e_in = (4 if lap == "fft" else 2) * b * GS_T * GS_N * GS_N      # fft: the two Laplacian fields are inputs of the kernel too
            avg = sum(k_ms) / len(k_ms)
            rec["roofline"] = {"bound": "hbm", "kernel": last_adjoint_kernel(), "achieved": 8 * e_in / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": 8 * e_in / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": 8 * e_in,
                               "avg_ms": avg, "timed_launches": len(k_ms),
                               "note": "the pointwise/stencil adjoint launch of one rank's share; the spectral variant's transforms are rocFFT launches beside it"}
        variants[lap] = rec
        del fields, mod, step_fn
        torch.cuda.empty_cache()
    if rank == 0:
        main_v = variants["fft"]
        out = {"metric": "PDE-residual loss fwd+bwd grid-points/sec", "value": main_v["value"], "unit": "grid-points/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_v["ms_per_step"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"Gray-Scott 2D reaction-diffusion residual fwd+bwd, B={GS_TOTAL} in total ({b} per GPU), T={GS_T}, "
                                      f"{GS_N}x{GS_N}, 2 fields, spectral Laplacian (value) and 4th-order stencil (variants.fd4)",
                          "parallelism": f"dp{world}", "launch": main_v["launch"], "backward_mode": "fused"},
               "roofline": main_v.get("roofline"), "variants": variants, "loss": main_v["loss"]}
        if world > 1 or distributed:
            out["rccl_world_size"] = dist.get_world_size()
        print(json.dumps(out))
    return 0


def launch_command(gpus, argv, port, python=None):
    """The child command that starts `gpus` ranks of this script on one node (one process per GPU)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed steps (on top of --warmup) until this much wall time has passed, so that the timed steps "
                         "run at the clocks of a busy GPU; 0 = none")
    ap.add_argument("--rotate", type=int, default=4,
                    help="number of independent input sets the steps cycle through; 4 sets (805 MB of fields, as much again of "
                         "gradients) exceed the 256 MiB Infinity Cache, so every step reads its fields from HBM ('cold'); "
                         "1 = the same 201 MB every step ('warm')")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="leave roofline.traffic null instead of looking the kernel's counter measurement up (the profiling runs "
                         "that produce that measurement use this)")
    ap.add_argument("--no-other-cases", action="store_true",
                    help="skip the sweep over the reference's other benchmark configurations after the timed region")
    ap.add_argument("--no-graph", action="store_true", help="run every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--mode", default="fused", choices=["fused", "auto"],
                    help="loss module backward_mode: 'fused' = loss and gradient in one pass of the adjoint kernel (it recomputes "
                         "every residual anyway), 'auto' = forward kernel, then adjoint kernel")
    ap.add_argument("--reduce-window", type=int, default=8,
                    help="steps whose logging scalars share one all-reduce (dist.ScalarWindow); 1 = a collective every step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--case", default="taylor_green", choices=["taylor_green", "gray_scott"],
                    help="taylor_green: the headline configuration, B=16 per GPU (weak scaling); gray_scott (with --strong): the "
                         "reference's Gray-Scott configuration, 64 samples split over the ranks")
    ap.add_argument("--strong", action="store_true", help="strong scaling: the total batch is fixed and split over the ranks (--case gray_scott)")
    ap.add_argument("--dry", action="store_true",
                    help="rehearse the launcher, the rendezvous and the packed all-reduce without a GPU: no kernel runs, "
                         "the printed value is 0 and the line carries \"dry\": true (use with --backend gloo)")
    return ap.parse_args(argv)


def dry_run(args, world, rank):
    """The bench's distributed skeleton on CPU tensors: rendezvous, warm-up, timed loop of packed all-reduces, one JSON line."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(args.backend)
    win = pdist.ScalarWindow(["continuity", "x_momentum", "y_momentum", "total"], max(1, args.reduce_window), "cpu", dtype=torch.float64)
    n = args.warmup + args.steps
    for i in range(n):
        win.before(i)
        win.row(i).copy_(torch.tensor([2.0, 3.0, 4.0, 1.0 + rank + i], dtype=torch.float64))
        win.after(i)
    win.flush(n)
    first, last = win.drain()[-1]
    red = {k: v[-1] for k, v in last.items()}
    assert first + len(last["total"]) == n and abs(float(red["total"]) - (n + (world - 1) / 2)) < 1e-9
    if world > 1:
        dist.barrier()
        n = torch.ones(1)
        dist.all_reduce(n)
        assert int(n) == world
    if rank == 0:
        print(json.dumps({"metric": "PDE-residual loss fwd+bwd grid-points/sec", "value": 0.0, "unit": "grid-points/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry": True,
                          "config": {"parallelism": f"dp{world}"}, "loss": float(red["total"])}))
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Not under a launcher: start the ranks as a child process group.  Nothing in this process has touched the GPU yet
        # (importing torch and pde_amd does not), and it never will: it only relays the child's exit status; rank 0 of
        # the child prints the JSON line on the shared stdout.
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1" if args.dry else os.environ.get("OMP_NUM_THREADS", "4"))
        return subprocess.run(launch_command(args.gpus, argv, free_port()), env=env).returncode

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus must be the world size", file=sys.stderr)
        return 2
    if (args.case == "gray_scott") != args.strong:
        print("bench: --case gray_scott is the strong-scaling configuration and --strong belongs to it: pass both or neither", file=sys.stderr)
        return 2
    if args.dry:
        dry_run(args, world, rank)
        return 0
    distributed = world > 1 or os.environ.get("PDE_AMD_BENCH_FORCE_DIST") == "1"     # the latter: rehearse the RCCL path with one rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(args.backend, device_id=dev)
        assert dist.get_world_size() == world

    if args.case == "gray_scott":
        rc = gray_scott_strong(args, world, rank, dev, distributed)
        if distributed:
            dist.destroy_process_group()
        return rc

    n_sets = max(1, args.rotate)
    sets = []
    for k in range(n_sets):
        fs = [f.requires_grad_(True) for f in synth_fields(B, T, H, W, 1234 + 1000 * 2 + 17 * k + rank, dev)]
        gnu = torch.Generator(device=dev).manual_seed(99 + 31 * k + rank)
        sets.append((fs, 0.01 + 0.09 * torch.rand(B, generator=gnu, device=dev)))
    loss_fn = pde_amd.TaylorGreen2DPDELoss(nx=W, ny=H, dt=0.01, eq_scales=EQ_SCALES).to(dev)
    loss_fn.backward_mode = args.mode      # upstream_hint stays 1.0: the step backpropagates `total` itself

    backend = PF.get_backend()

    one = torch.ones((), device=dev)

    # The step's logging scalars (the per-equation losses and the total, which the reference reduces over the ranks every
    # step) go straight from the reduction kernel into a row of a window buffer; one all-reduce per window of steps.
    window = max(1, args.reduce_window)
    win = pdist.ScalarWindow([*loss_fn.EQS, "total"], window, dev, force=distributed)

    def loss_step(k, row=None):
        """forward + backward of the loss on input set k through the nn.Module API; gradients land in fields[i].grad and
        the loss scalars in `row`"""
        fields, nu = sets[k]
        for f in fields:
            f.grad = None
        loss_fn.scalars_out = row
        total, losses = loss_fn(*fields, nu=nu)
        # the upstream gradient of the loss: the module's own resident scalar (upstream_hint = 1.0) instead of a fill launch
        # per step; handed the very tensor it assumed, the fused step knows without looking that no rescaling is due
        total.backward(gradient=loss_fn.upstream_scalar(dev) if args.mode == "fused" else one)

    for i in range(max(args.warmup, n_sets)):
        loss_step(i % n_sets)
    if distributed:      # the first collective sets RCCL up: not inside the timed region
        pdist.reduce_packed_async(["x"], torch.zeros(1, device=dev), force=True).result()

    # The step is launch-bound on the host (a few small launches plus autograd for ~0.15 ms of GPU work), so the
    # steady-state loop replays it as a HIP graph; every `probe`-th step of the timed region runs eagerly instead with
    # HIP events recorded right around the adjoint launch, which is where the roofline numbers come from.
    # --no-graph runs every step eagerly.  A graph is tied to its input set and to the window row its scalars land in: one
    # graph per residue of the step index modulo lcm(2 * window, input sets).
    n_graphs = math.lcm(2 * window, n_sets)
    graphs = []
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                loss_step(0)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # one memory pool for all graphs: they replay one after the other, in capture order, and the gradients of a
            # capture are dropped when its input set is captured again, so the graphs share a few gradient buffers
            # instead of holding one set each
            pool = torch.cuda.graph_pool_handle()
            for j in range(n_graphs):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, pool=pool, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) stay free to query events
                    loss_step(j % n_sets, win.row(j))
                gr.replay()
                torch.cuda.synchronize()
                graphs.append(gr)
        except Exception as e:      # capture unsupported here: eager loop
            print(f"bench: HIP graph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphs = []
            torch.cuda.synchronize()
    graph = graphs[0] if graphs else None
    # an eager probe step costs more than a replayed one (the host cannot keep up), so only every 10th step is one
    probe = 10 if args.steps >= 10 else max(1, args.steps // 2)
    if n_sets > 1 and probe % n_sets == 0:
        probe += 1                       # the probes visit every input set in turn
    probe = max(1, min(probe, args.steps))      # at least one probe, however few steps
    n_probe = (args.steps + probe - 1) // probe
    pool = lambda n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + 2)]
    bwd_events, fwd_events = [], []

    def step(i):
        win.before(i)      # opening a window: joins the all-reduce that still reads the rows this window overwrites
        if i % probe != probe - 1:
            if graph is None:      # --no-graph: an ordinary eager step through the C++ operator
                loss_step(i % n_sets, win.row(i))
            else:
                graphs[i % n_graphs].replay()
        else:
            # fused mode has no forward launch to time, and with only the adjoint timer armed the step keeps the C++ operator
            backend.backward_events, backend.forward_events = bwd_events, (None if args.mode == "fused" else fwd_events)
            loss_step(i % n_sets, win.row(i))
            backend.backward_events = backend.forward_events = None
        win.after(i)       # closing a window: starts its all-reduce, which runs beside the next window's steps

    backend.backward_event_pool, backend.forward_event_pool = pool(n_probe + 40), pool(n_probe + 40)
    # Nothing of the probes' set-up belongs in the timed region: an event's handle only exists after a first record, and the
    # first timed step walks code that no step before it has run (the library's timer entry point, the event hand-over).
    for a, b in backend.backward_event_pool + backend.forward_event_pool:
        a.record(); b.record()

    def run(k_steps):
        """what the timed region runs: the steps, then the scalars of the last, unfinished window"""
        for i in range(k_steps):
            step(i)
        win.flush(k_steps)
        return win.drain()

    # A rehearsal of the timed region, untimed: in a fresh process the first pass through any code path pays for what is
    # loaded on first use (kernels of the few torch operations of the window hand-over, the probes' event path), which in a
    # region of a few milliseconds is not noise (measured on a fresh box: 0.131-0.140 ms per step at 20 steps without it,
    # 0.117 with).
    run(min(args.steps, probe + 2 * window))
    torch.cuda.synchronize()
    bwd_events.clear(); fwd_events.clear()
    # A few dozen 0.15 ms steps do not bring an idle GPU up to its working clocks (the same kernel measures 10 % slower in
    # the first milliseconds of activity): more untimed steps of the very same kind, until settle-ms have passed.
    settle_steps = 0
    if args.settle_ms > 0:
        torch.cuda.synchronize()
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < args.settle_ms:
            for j in range(16):
                if graph is not None:
                    graphs[(settle_steps + j) % n_graphs].replay()
                else:
                    loss_step((settle_steps + j) % n_sets)
            settle_steps += 16
            torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    red = {k: v[-1] for k, v in run(args.steps)[-1][1].items()}      # the last step's scalars, mean over the ranks
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
        # more cold launches for the kernel figure than the few probes inside the timed region: eager steps that keep cycling
        # through the input sets (every launch reads fields the caches have since lost), events around the adjoint launch only
        if n_sets > 1:
            backend.backward_events, backend.forward_events = bwd_events, (None if args.mode == "fused" else fwd_events)
            for j in range(max(0, 24 - len(bwd_events))):
                loss_step((args.steps + j) % n_sets)
            backend.backward_events = backend.forward_events = None
            torch.cuda.synchronize()
        kernel_name = last_adjoint_kernel()
        k_ms = sorted(x.elapsed_time(y) for x, y in bwd_events)
        f_ms = sorted(x.elapsed_time(y) for x, y in fwd_events)      # empty in fused mode: there is no forward launch
        bwd_avg = sum(k_ms) / len(k_ms)                   # the adjoint kernel launch alone
        e_in = 3 * B * T * H * W
        bwd_bytes = 8 * e_in            # adjoint: read every field once, write every gradient once
        achieved = bwd_bytes / (bwd_avg * 1e-3) / 1e9
        # after the timed region: the same kernel on ONE input set, step after step, so that its 201 MB of fields stay in
        # the Infinity Cache between launches (the gradient stores are nontemporal and do not displace them)
        other = {}
        if n_sets > 1:
            warm = []
            backend.backward_events, backend.forward_events = warm, (None if args.mode == "fused" else [])
            for _ in range(24):
                loss_step(0)
            backend.backward_events = backend.forward_events = None
            torch.cuda.synchronize()
            w_ms = sorted(x.elapsed_time(y) for x, y in warm)[: 16]      # the first launches still find another set cached
            w_avg = sum(w_ms) / len(w_ms)
            other = {"warm_avg_ms": w_avg, "warm_achieved": bwd_bytes / (w_avg * 1e-3) / 1e9,
                     "warm_frac": bwd_bytes / (w_avg * 1e-3) / 1e9 / HBM_PEAK_GBS}
        cache = (f"cold: the steps cycle through {n_sets} input sets ({n_sets * 4 * e_in / 1e6:.0f} MB of fields), more than the 256 MiB "
                 f"Infinity Cache holds; warm_*: one set, step after step") if n_sets > 1 else "warm: the same input set every step"
        out = {
            "metric": "PDE-residual loss fwd+bwd grid-points/sec",
            "value": value, "unit": "grid-points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Taylor-Green 2D incompressible NS residual fwd+bwd, B=16 per GPU, T=16, 256x256, 3 fields",
                       "parallelism": f"dp{world}", "points_per_step_per_gpu": pts_per_rank,
                       "launch": f"hip-graph replay, every {probe}th step eager with kernel events" if graph is not None else f"eager (C++ operator), every {probe}th step with kernel events",
                       "backward_mode": args.mode, "input_sets": n_sets,
                       "logging": f"the step's loss scalars of {window} steps share one all-reduce (mean over the ranks), started "
                                  f"when the window closes and joined a window later" if window > 1 else
                                  "one all-reduce of the step's loss scalars per step, joined a step later"},
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "kernel_note": "the adjoint; one launch per step; HIP events recorded by the library on the launch stream right "
                                        "before and after that launch" + ("; in fused mode the kernel also sums the loss" if args.mode == "fused" else ""),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if args.no_traffic else measured_traffic(kernel_name), "algorithmic_bytes": bwd_bytes, "avg_ms": bwd_avg,
                         "median_ms": k_ms[len(k_ms) // 2], "timed_launches": len(k_ms), "cache": cache, **other,
                         **({"fwd_median_ms": f_ms[len(f_ms) // 2], "fwd_algorithmic_bytes": 4 * e_in,
                             } if f_ms else {})},
            "loss": float(red["total"]),
        }
        if world > 1 or distributed:
            out["rccl_world_size"] = dist.get_world_size()
        if world == 1 and not args.no_other_cases:
            out["other_cases"] = run_other_cases(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
