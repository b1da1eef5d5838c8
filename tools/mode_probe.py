"""Diagnostic: one bench_all case in a given backward mode:  tools/mode_probe.py <case filter> <mode>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import inspect, torch, pde_amd
import tools.bench_all as BA
for tag, cls_name, kw, shape, params in BA.CASES:
    if sys.argv[1].lower() not in tag.lower():
        continue
    mod = getattr(pde_amd, cls_name)(**kw).to(BA.dev)
    mod.backward_mode = sys.argv[2]
    fwd_params = inspect.signature(mod.forward).parameters
    nf = sum(1 for q in fwd_params.values() if q.default is inspect.Parameter.empty and q.name not in BA.pk_names(params))
    fields = [(1.0 + 0.1 * torch.randn(shape, device=BA.dev)).requires_grad_(True) for _ in range(nf)]
    pk = params(shape[0])
    def step():
        for f in fields:
            f.grad = None
        mod(*fields, **pk)[0].backward()
    print(tag, sys.argv[2], f"{BA.timeit(step):.3f} ms")
