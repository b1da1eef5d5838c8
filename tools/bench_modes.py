"""Compare backward modes (recompute / stored) of one loss class at a given shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pde_amd
from tools.bench_all import CASES, timeit, dev
from pde_amd.stencil.specs import SPECS

for tag, cls_name, kw, shape, params in CASES:
    if not any(o.lower() in tag.lower() for o in sys.argv[1:]):
        continue
    for mode in ("recompute", "stored"):
        mod = getattr(pde_amd, cls_name)(**kw).to(dev)
        mod.backward_mode = mode
        nf = len(SPECS[mod.SPEC].fields)
        fields = [(1.0 + 0.1 * torch.randn(shape, device=dev)).requires_grad_(True) for _ in range(nf)]
        pk = params(shape[0])
        def step():
            for f in fields:
                f.grad = None
            mod(*fields, **pk)[0].backward()
        try:
            ms = timeit(step)
            print(f"{tag:26s} {mode:10s} {ms:8.3f} ms")
        except Exception as e:
            print(f"{tag:26s} {mode:10s} failed: {e}")
