"""ctypes binding of libpde_amd.so (the C ABI declared in include/pde_amd.h).

There is no CPU implementation behind this module: if the shared library is missing the import
of any compute entry point raises.  `torch` is imported first so that the HIP runtime the library
binds to is the one PyTorch already loaded (same SONAME), which makes torch streams and device
pointers valid inside the library.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: shares libamdhip64 with the extension)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDE_AMD_LIB", os.path.join(_HERE, "libpde_amd.so"))   # override: A/B builds while tuning

MAX_FIELDS, MAX_EQS, MAX_AUX, MAX_SPARAMS, MAX_CONSTS = 4, 4, 2, 3, 16


class Problem(C.Structure):
    _fields_ = [("pde_id", C.c_int32), ("batch", C.c_int32), ("frames", C.c_int32),
                ("n", C.c_int32 * 3), ("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3),
                ("n_consts", C.c_int32), ("consts", C.c_float * MAX_CONSTS)]


class PdeInfo(C.Structure):
    _fields_ = [("id", C.c_int32), ("ndim", C.c_int32), ("n_fields", C.c_int32), ("n_eqs", C.c_int32),
                ("n_aux", C.c_int32), ("n_consts", C.c_int32), ("n_sparams", C.c_int32),
                ("t_lo", C.c_int32), ("t_hi", C.c_int32), ("halo", C.c_int32 * 3), ("can_recompute", C.c_int32),
                ("name", C.c_char_p)]


class Term(C.Structure):
    _fields_ = [("n", C.c_int32 * 4), ("idx_off", C.c_int32 * 4), ("axis", C.c_int32), ("n_taps", C.c_int32),
                ("shift", C.c_int32 * 3), ("coef", C.c_float * 3), ("use_target", C.c_int32)]


NSTAT, MAX_TERMS = 6, 12
STATIC_SCRATCH_BYTES = 4 << 20

_lib = None

# every exported symbol with its signature; tests check that the library exports exactly these
SIGNATURES = {
    "pde_amd_num_pdes": (C.c_int, []),
    "pde_amd_pde_id": (C.c_int, [C.c_char_p]),
    "pde_amd_get_pde_info": (C.c_int, [C.c_int, C.POINTER(PdeInfo)]),
    "pde_amd_status_string": (C.c_char_p, [C.c_int]),
    "pde_amd_residual_workspace_bytes": (C.c_size_t, [C.POINTER(Problem)]),
    "pde_amd_residual_forward": (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p]),
    "pde_amd_residual_backward": (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p,
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    "pde_amd_residual_forward_backward": (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                    C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p]),
    "pde_amd_residual_forward_backward_static": (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                           C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                           C.POINTER(C.c_void_p), C.c_void_p]),
    "pde_amd_residual_forward_f64": (C.c_int, [C.POINTER(Problem), C.POINTER(C.c_double), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "pde_amd_time_next_adjoint": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pde_amd_last_adjoint_kernel": (C.c_char_p, []),
    "pde_amd_diag_poison_lds": (C.c_int, [C.c_void_p]),
    "pde_amd_rescale_gradients": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pde_amd_pair_stats_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "pde_amd_pair_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "pde_amd_vrmse_from_stats": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pde_amd_band_power": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pde_amd_terms_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "pde_amd_terms_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(Term), C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pde_amd_terms_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(Term), C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pde_amd_pair_stats_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pde_amd_pair_stats_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p), C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    "pde_amd_terms_forward_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(Term), C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pde_amd_terms_backward_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(Term), C.c_int32,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pde_amd_gather_fields_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                               C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.c_void_p]),
    "pde_amd_scatter_field_grads_affine": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                                     C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                                     C.c_void_p, C.c_void_p]),
    "pde_amd_gather_fields": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                        C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.POINTER(C.c_void_p), C.c_void_p]),
    "pde_amd_scatter_field_grads": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                              C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p]),
    "pde_amd_spectral_pair_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "pde_amd_spectral_laplacian_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                                  C.c_int32, C.c_void_p]),
}


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"pde_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


TORCH_LIB_PATH = os.path.join(_HERE, "libpde_amd_torch.so")
_torch_op = None


def torch_op():
    """The C++ operator pde_amd::residual_losses (autograd node in C++, bound to the loaded kernel library); raises
    when the operator library has not been built."""
    global _torch_op
    if _torch_op is None:
        lib()
        if not os.path.exists(TORCH_LIB_PATH):
            raise RuntimeError(
                f"pde_amd: {TORCH_LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'`, "
                "or set PDE_AMD_HOST=ctypes to drive the kernels from the Python autograd function instead.")
        torch.ops.load_library(TORCH_LIB_PATH)
        binder = C.CDLL(TORCH_LIB_PATH)
        binder.pde_amd_torch_bind.restype, binder.pde_amd_torch_bind.argtypes = C.c_int, [C.c_char_p]
        if binder.pde_amd_torch_bind(LIB_PATH.encode()) != 0:
            raise RuntimeError(f"pde_amd: could not bind {LIB_PATH} into the torch operator library")
        _torch_op = torch.ops.pde_amd.residual_losses.default
    return _torch_op


# Diagnostic (PDE_AMD_POISON_LDS=1): every buffer the wrappers hand to a kernel starts as NaNs (all-ones bytes), so an
# element the kernel was supposed to write and did not shows up in the parity tests.
POISON = os.environ.get("PDE_AMD_POISON_LDS") == "1"


def empty(*size, **kw):
    t = torch.empty(*size, **kw)
    if POISON and t.numel():
        t.view(torch.uint8).fill_(255) if t.dtype is not torch.uint8 else t.fill_(255)
    return t


def empty_like(x):
    t = torch.empty_like(x)
    if POISON and t.numel():
        t.view(torch.uint8).fill_(255)
    return t


def check(status: int, what: str):
    if status != 0:
        msg = lib().pde_amd_status_string(status).decode()
        raise RuntimeError(f"pde_amd: {what} failed: {msg} ({status})")


def ptr_array(tensors, n):
    """Host array of device pointers; None entries become NULL."""
    arr = (C.c_void_p * max(n, 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


_info_cache = {}


def pde_info(name: str) -> PdeInfo:
    if name not in _info_cache:
        l = lib()
        i = PdeInfo()
        check(l.pde_amd_get_pde_info(l.pde_amd_pde_id(name.encode()), C.byref(i)), f"pde_info({name})")
        _info_cache[name] = i
    return _info_cache[name]
