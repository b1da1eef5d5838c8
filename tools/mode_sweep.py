"""Every backward mode of every tools/bench_all.py case in one process, next to what 'auto' and 'fused' resolve to:
    python tools/mode_sweep.py [case filters...] [--assert]
Prints one line per case (times in ms, the settled median of tools/bench_all.timeit) and, with --assert, exits non-zero when
some case's 'fused' step is more than 3 % slower than its 'auto' step, or either is more than 3 % slower than the best forced
mode: the per-spec flags of pde_amd/stencil/specs.py (auto_stored, fused_slower) are measured choices, and this is the check."""
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

        def step():
            for f in fields:
                f.grad = None
            mod(*fields, **pk)[0].backward()

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
                t[mode] = BA.timeit(step, n=21, settle_ms=100.0)
            except Exception as e:      # a mode the class does not support
                t[mode] = float("nan")
            finally:
                os.environ.pop("PDE_AMD_FORCE_FUSED", None)
        res = {}
        for mode in ("auto", "fused"):
            mod.backward_mode = mode
            res[mode] = (BA.timeit(step, n=21, settle_ms=100.0), PF.resolve_mode(mode, spec) if spec else "?")
        best = min(v for v in t.values() if v == v)
        print(f"{tag:28s} {t['stored']:8.3f} {t['recompute']:8.3f} {t['fused']:8.3f} | {res['auto'][0]:8.3f} -> {res['auto'][1]:9s} "
              f"{res['fused'][0]:8.3f} -> {res['fused'][1]:9s}" + ("" if max(res['auto'][0], res['fused'][0]) <= 1.03 * best + 0.002 else "   <-- not the best mode"))
        if res["fused"][0] > 1.03 * res["auto"][0] + 0.002 or max(res["auto"][0], res["fused"][0]) > 1.03 * best + 0.002:
            bad.append(tag)
    if "--assert" in sys.argv and bad:
        print("mode choice off by more than 3 %:", ", ".join(bad))
        sys.exit(1)


if __name__ == "__main__":
    main()
