"""Every backward mode of every tools/bench_all.py case in one process, next to what 'auto' and 'fused' resolve to:
    python tools/mode_sweep.py [case filters...] [--assert]
Prints one line per case (times in ms of the step replayed as a HIP graph, loss_fn.capture(): the small cases are bound by
the host's launch work when run eagerly, whatever the kernels) and, with --assert, exits non-zero when some case's 'fused'
step is more than 3 % slower than its 'auto' step, when 'auto' is more than 3 % slower than the better of the two modes it
chooses between (stored, recompute), or when 'fused' is more than 3 % slower than the best forced mode: the per-spec flags
of pde_amd/stencil/specs.py (auto_stored, fused_slower) and the small-problem rule are measured choices, and this is the
check."""
import inspect
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pde_amd
import tools.bench_all as BA
from pde_amd import functional as PF


def main():
    filt = [a for a in sys.argv[1:] if not a.startswith("--")]
    bad = []
    print(f"{'case':28s} {'stored':>8s} {'recomp':>8s} {'fused':>8s} | {'auto':>8s} -> {'kernels':9s} {'fused':>8s} -> {'kernels':9s}")
    for tag, cls_name, kw, shape, params in BA.CASES:
        if filt and not any(o.lower() in tag.lower() for o in filt):
            continue
        mod = getattr(pde_amd, cls_name)(**kw).to(BA.dev)
        fwd_params = inspect.signature(mod.forward).parameters
        nf = sum(1 for q in fwd_params.values() if q.default is inspect.Parameter.empty and q.name not in BA.pk_names(params))
        fields = [(1.0 + 0.1 * torch.randn(shape, device=BA.dev)).requires_grad_(True) for _ in range(nf)]
        pk = params(shape[0])

        def timed():
            """median of 5 timings of 40 replays each, clocks settled"""
            step = mod.capture(*fields, **pk)
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            import time
            ts = time.perf_counter()
            while (time.perf_counter() - ts) * 1e3 < 100.0:
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
            out = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(40):
                    step()
                b.record()
                torch.cuda.synchronize()
                out.append(a.elapsed_time(b) / 40)
            return sorted(out)[2]

        t = {}
        spec = getattr(mod, "SPEC", None)
        can = PF.get_backend().can_recompute(spec) if spec else False
        for mode in ("stored", "recompute", "fused"):
            if mode != "stored" and not can:
                t[mode] = float("nan")
                continue
            mod.backward_mode = mode
            os.environ["PDE_AMD_FORCE_FUSED"] = "1"      # measure the one-pass step even where the spec prefers other kernels
            try:
                t[mode] = timed()
            except Exception as e:      # a mode the class does not support
                t[mode] = float("nan")
            finally:
                os.environ.pop("PDE_AMD_FORCE_FUSED", None)
        res = {}
        for mode in ("auto", "fused"):
            mod.backward_mode = mode
            res[mode] = (timed(), PF.resolve_mode(mode, spec, fields[0].numel()) if spec else "?")
        best = min(v for v in t.values() if v == v)
        best_auto = min(v for k, v in t.items() if v == v and k != "fused")
        slack = 0.003      # ms: what two timings of the same replayed step differ by
        off = (res["fused"][0] > 1.03 * res["auto"][0] + slack or res["auto"][0] > 1.03 * best_auto + slack
               or res["fused"][0] > 1.03 * best + slack)
        print(f"{tag:28s} {t['stored']:8.3f} {t['recompute']:8.3f} {t['fused']:8.3f} | {res['auto'][0]:8.3f} -> {res['auto'][1]:9s} "
              f"{res['fused'][0]:8.3f} -> {res['fused'][1]:9s}" + ("   <-- not the best mode" if off else ""), flush=True)
        if off:
            bad.append(tag)
    if "--assert" in sys.argv and bad:
        print("mode choice off by more than 3 %:", ", ".join(bad))
        sys.exit(1)


if __name__ == "__main__":
    main()
