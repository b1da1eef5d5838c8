"""The shared library loads without a GPU and exports exactly what include/pde_amd.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from pde_amd import _lib
from pde_amd.stencil.specs import PDE_IDS, SPECS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "pde_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pde_amd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/pde_amd.h but not exported"


def test_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_struct_layouts_match_header_constants():
    txt = open(os.path.join(ROOT, "include", "pde_amd.h")).read()
    consts = {k: int(v) for k, v in re.findall(r"#define (PDE_AMD_[A-Z_]+) (\d+)", txt)}
    assert consts["PDE_AMD_MAX_CONSTS"] == _lib.MAX_CONSTS and consts["PDE_AMD_MAX_FIELDS"] == _lib.MAX_FIELDS
    assert consts["PDE_AMD_NSTAT"] == _lib.NSTAT and consts["PDE_AMD_MAX_TERMS"] == _lib.MAX_TERMS
    assert C.sizeof(_lib.Problem) == 4 * (3 + 9 + 1) + 4 * _lib.MAX_CONSTS
    assert C.sizeof(_lib.Term) == 4 * (4 + 4 + 2 + 3 + 3 + 1)


def test_registry_queries_work_without_a_gpu():
    l = _lib.lib()
    assert l.pde_amd_num_pdes() == len(SPECS)
    for name, pid in PDE_IDS.items():
        assert l.pde_amd_pde_id(name.encode()) == pid
        info = _lib.pde_info(name)
        s = SPECS[name]
        assert (info.ndim, info.n_fields, info.n_eqs, info.n_sparams) == (s.ndim, len(s.fields), len(s.eqs), len(s.sparams))
        assert info.name.decode() == name
    assert l.pde_amd_pde_id(b"no_such_pde") < 0
    assert l.pde_amd_status_string(-3).decode() != ""


def test_argument_validation_without_a_gpu():
    """the entry points reject bad arguments before touching the device"""
    l = _lib.lib()
    assert l.pde_amd_pair_stats(None, None, 1, 4, 18, 1e-8, None, None, 0, None) == -1          # EINVAL: null pointers
    assert l.pde_amd_pair_stats(None, None, 1, 4, 2000, 1e-8, None, None, 0, None) == -2        # ESHAPE: too many channels
    assert l.pde_amd_pair_stats_workspace_bytes(4, 1000, 18) > 0
    assert l.pde_amd_terms_workspace_bytes(3) > 0
    p = _lib.Problem()
    p.pde_id = 999
    assert l.pde_amd_residual_workspace_bytes(C.byref(p)) == 0


def test_torch_operator_library_loads_and_rejects_cpu_tensors():
    """The C++ operator library registers pde_amd::residual_losses and binds the kernel library; without a HIP tensor
    it refuses to compute (no CPU path behind it either)."""
    import pytest
    import torch
    from pde_amd import _lib
    op = _lib.torch_op()
    assert "Tensor[] fields" in str(op._schema) and str(op._schema).endswith("-> Tensor[]")
    with pytest.raises(RuntimeError, match="HIP device"):
        op([torch.zeros(1, 3, 8)], [], [], torch.zeros(1, 2), 0, [1.0], [0, 0, 0], [0, 0, 0], 1, 0, 1, 0, None)
