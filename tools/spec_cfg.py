"""Diagnostic helper: regenerate the device code with some tile/thread attributes of one spec overridden, e.g.
tools/spec_cfg.py ape_burgers3d "quad_fwd=(4, 8, 32), quad_bwd=(4, 8, 32)".  Without arguments: regenerate as committed."""
import os
import sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from pde_amd.stencil import codegen
from pde_amd.stencil.specs import SPECS
if len(sys.argv) > 2:
    for k, v in eval(f"dict({sys.argv[2]})").items():
        assert hasattr(SPECS[sys.argv[1]], k), k
        setattr(SPECS[sys.argv[1]], k, v)
codegen.generate()
