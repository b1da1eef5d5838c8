"""Diagnostic helper: rewrite the extra keyword arguments of the Taylor-Green spec and regenerate (tools/tg2d_cfg.py "quad_width_bwd=2, ...")."""
import re, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(root, "pde_amd/stencil/specs.py")
s = open(p).read()
extra = sys.argv[1] if len(sys.argv) > 1 else ""
s = re.sub(r"return b.done\(\[cont, xm, ym\], threads_march=512, waves_per_simd=4.*\)",
           "return b.done([cont, xm, ym], threads_march=512, waves_per_simd=4" + (", " + extra if extra else "") + ")", s)
open(p, "w").write(s)
sys.path.insert(0, root)
from pde_amd.stencil import codegen
codegen.generate()
