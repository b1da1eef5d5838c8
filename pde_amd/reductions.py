"""Raw device primitives behind the metrics, boundary losses and glue: thin wrappers over the C ABI.

  pair_stats(pred, target, groups, rows, channels, rse_eps) -> fp64 [groups, channels, NSTAT]
  terms_forward(pred, target, outer, dims, terms, indices)  -> fp64 [n_terms]
  terms_backward(pred, target, outer, dims, terms, indices, scales, grad)   (accumulates into grad)
  gather_fields(output, input, channels, transpose)         -> list of [B, T_out + 1, *spatial] tensors
  scatter_field_grads(grads, out_shape, channels, transpose) -> [B, T_out, *spatial, C]

Everything runs on the current torch HIP stream; there is no CPU path in this module.  The CPU tests
replace the backend object with an interpreter that lives under tests/.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import torch

from . import _lib

NSTAT = _lib.NSTAT
S_D2, S_G2, S_G, S_P, S_RSE, S_MAXD = range(6)


_POISON_LDS = os.environ.get("PDE_AMD_POISON_LDS") == "1"      # diagnostic: NaNs into every CU's LDS before each launch


def _stream(t: torch.Tensor):
    """the launch stream as the C ABI takes it (every launch below asks for it right before the call)"""
    st = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    if _POISON_LDS:
        with torch.cuda.device(t.device):
            _lib.check(_lib.lib().pde_amd_diag_poison_lds(st), "diag_poison_lds")
    return st


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"pde_amd: {what} must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _affine_args(affine, outer: int):
    """(scale pointer, shift pointer, outer indices per sample) of an optional (scale[B, C], shift[B, C]) pair"""
    if affine is None:
        return None, None, 1
    scale, shift = affine
    B = int(scale.shape[0])
    if scale.dtype != torch.float32 or shift.dtype != torch.float32 or scale.shape != shift.shape or not scale.is_contiguous() \
            or not shift.is_contiguous() or B < 1 or outer % B:
        raise ValueError("affine map must be two contiguous fp32 [B, C] tensors with B dividing the outer extent")
    return scale.data_ptr(), shift.data_ptr(), outer // B


class HipReductions:
    name = "hip"

    def pair_stats(self, pred, target, groups: int, rows: int, channels: int, rse_eps=None, affine=None):
        """affine = (scale, shift), each [groups, channels]: pred is read as pred * scale + shift.  rse_eps=None leaves the
        S_RSE statistic out (zeros); only compute_mean_rse needs it."""
        l = _lib.lib()
        pred, target = _f32c(pred, "pred"), _f32c(target, "target")
        if pred.numel() != groups * rows * channels or target.numel() != pred.numel():
            raise ValueError(f"pair_stats: {tuple(pred.shape)} / {tuple(target.shape)} is not [{groups}, {rows}, {channels}]")
        stats = _lib.empty((groups, channels, NSTAT), dtype=torch.float64, device=pred.device)
        nbytes = l.pde_amd_pair_stats_workspace_bytes(groups, rows, channels)
        ws = _lib.empty(max(nbytes, 8), dtype=torch.uint8, device=pred.device)
        sc, sh, _ = _affine_args(affine, groups)
        if affine is not None and tuple(affine[0].shape) != (groups, channels):
            raise ValueError(f"pair_stats: the affine map must be [{groups}, {channels}], got {tuple(affine[0].shape)}")
        _lib.check(l.pde_amd_pair_stats_affine(pred.data_ptr(), target.data_ptr(), groups, rows, channels, -1.0 if rse_eps is None else float(rse_eps), sc, sh,
                                               stats.data_ptr(), ws.data_ptr(), nbytes, _stream(pred)), "pair_stats")
        return stats

    def pair_stats_gather(self, pred, target, input_data, channels: Sequence[int], affine=None):
        """One sweep over pred [B, T, *spatial, C]: the per-sample statistics [B, C, NSTAT] and, as a by-product, the fields
        gather_fields(pred, input_data, channels) would produce ([B, T + 1, *spatial] each, frame 0 from input_data[:, 0])."""
        l = _lib.lib()
        pred, target = _f32c(pred, "pred"), _f32c(target, "target")
        B, T, Cn = int(pred.shape[0]), int(pred.shape[1]), int(pred.shape[-1])
        spatial = tuple(pred.shape[2:-1])
        rows = pred.numel() // max(B * Cn, 1)
        if target.shape != pred.shape or input_data.shape[0] != B or tuple(input_data.shape[2:]) != spatial + (Cn,):
            raise ValueError("pair_stats_gather: pred, target [B, T, *spatial, C] and input_data [B, T_in, *spatial, C] do not match")
        stats = _lib.empty((B, Cn, NSTAT), dtype=torch.float64, device=pred.device)
        fields = [_lib.empty((B, T + 1) + spatial, dtype=torch.float32, device=pred.device) for _ in channels]
        for f, c in zip(fields, channels):      # frame 0: a strip of the input (the gather of the reference's preamble)
            f[:, 0].copy_(input_data[:, 0, ..., int(c)])
        nbytes = l.pde_amd_pair_stats_workspace_bytes(B, rows, Cn)
        ws = _lib.empty(max(nbytes, 8), dtype=torch.uint8, device=pred.device)
        sc, sh, _ = _affine_args(affine, B)
        ch = (C.c_int32 * max(len(channels), 1))(*[int(c) for c in channels])
        _lib.check(l.pde_amd_pair_stats_gather(pred.data_ptr(), target.data_ptr(), B, rows, Cn, -1.0, sc, sh, stats.data_ptr(), T, ch,
                                               len(channels), _lib.ptr_array(fields, len(fields)), ws.data_ptr(), nbytes, _stream(pred)),
                   "pair_stats_gather")
        return stats, fields

    def vrmse_from_stats(self, stats, valid: Sequence[int], rows: int):
        """[2] fp32 (vrmse, rmse) from one group's statistics [C, NSTAT]; at most 64 valid channels."""
        out = _lib.empty(2, dtype=torch.float32, device=stats.device)
        arr = (C.c_int32 * max(len(valid), 1))(*[int(v) for v in valid])
        _lib.check(_lib.lib().pde_amd_vrmse_from_stats(stats.data_ptr(), int(stats.shape[0]), arr, len(valid), int(rows),
                                                       out.data_ptr(), _stream(stats)), "vrmse_from_stats")
        return out

    def band_power(self, spectrum, bin_start, point_index, nbins: int):
        """spectrum complex64 [groups, points, nt]; CSR bins (int32 device tensors) -> [groups, nbins, nt] fp32 sums of |.|^2."""
        if spectrum.dtype != torch.complex64 or not spectrum.is_cuda:
            raise RuntimeError("band_power: spectrum must be a complex64 tensor on a HIP device (there is no CPU fallback)")
        spectrum = spectrum.contiguous()
        G, Pn, nt = (int(v) for v in spectrum.shape)
        out = torch.zeros((G, nbins, nt), dtype=torch.float32, device=spectrum.device)
        _lib.check(_lib.lib().pde_amd_band_power(torch.view_as_real(spectrum).data_ptr(), G, Pn, nt, bin_start.data_ptr(),
                                                 point_index.data_ptr(), int(nbins), out.data_ptr(), _stream(spectrum)), "band_power")
        return out

    @staticmethod
    def _term_array(terms: Sequence["_lib.Term"]):
        arr = (_lib.Term * len(terms))()
        for i, t in enumerate(terms):
            arr[i] = t
        return arr

    def terms_forward(self, pred, target, outer: int, dims: Sequence[int], terms, indices, affine=None):
        """affine = (scale[B, C], shift[B, C]) reads pred as pred * scale + shift per sample and channel."""
        l = _lib.lib()
        sums = _lib.empty(len(terms), dtype=torch.float64, device=pred.device)
        nbytes = l.pde_amd_terms_workspace_bytes(len(terms))
        ws = _lib.empty(max(nbytes, 8), dtype=torch.uint8, device=pred.device)
        d = (C.c_int32 * 4)(*[int(v) for v in dims])
        sc, sh, ops = _affine_args(affine, outer)
        _lib.check(l.pde_amd_terms_forward_affine(pred.data_ptr(), target.data_ptr() if target is not None else None, outer, d,
                                                  self._term_array(terms), len(terms), indices.data_ptr(), sc, sh, ops,
                                                  sums.data_ptr(), ws.data_ptr(), nbytes, _stream(pred)), "terms_forward")
        return sums

    def terms_backward(self, pred, target, outer: int, dims: Sequence[int], terms, indices, scales, affine, disjoint, grad):
        """grad += d(sum_k scales[k] * sums[k]) / d pred; disjoint=True promises that no two terms touch a common element."""
        l = _lib.lib()
        d = (C.c_int32 * 4)(*[int(v) for v in dims])
        sc, sh, ops = _affine_args(affine, outer)
        _lib.check(l.pde_amd_terms_backward_affine(pred.data_ptr(), target.data_ptr() if target is not None else None, outer, d,
                                                   self._term_array(terms), len(terms), indices.data_ptr(), sc, sh, ops,
                                                   int(bool(disjoint)), scales.data_ptr(), grad.data_ptr(), _stream(pred)), "terms_backward")
        return grad

    def gather_fields(self, output, input_data, channels: Sequence[int], transpose: bool, affine=None) -> List[torch.Tensor]:
        l = _lib.lib()
        B, T_out, T_in = output.shape[0], output.shape[1], input_data.shape[1]
        spatial = list(output.shape[2:-1])
        Cn = output.shape[-1]
        S = 1
        for v in spatial:
            S *= v
        rows, cols = (spatial[-2], spatial[-1]) if len(spatial) >= 2 else (1, 1)
        fshape = spatial[:-2] + [cols, rows] if transpose else spatial
        fields = [_lib.empty([B, T_out + 1] + fshape, dtype=torch.float32, device=output.device) for _ in channels]
        ch = (C.c_int32 * len(channels))(*[int(c) for c in channels])
        sc, sh, _ = _affine_args(affine, B)
        _lib.check(l.pde_amd_gather_fields_affine(output.data_ptr(), input_data.data_ptr(), B, T_out, T_in, S, Cn, ch, len(channels),
                                                  rows, cols, int(bool(transpose)), sc, sh, _lib.ptr_array(fields, len(fields)),
                                                  _stream(output)), "gather_fields")
        return fields

    def scatter_field_grads(self, grads, out_shape, channels: Sequence[int], transpose: bool, scale=None):
        l = _lib.lib()
        B, T_out = out_shape[0], out_shape[1]
        spatial = list(out_shape[2:-1])
        Cn = out_shape[-1]
        S = 1
        for v in spatial:
            S *= v
        rows, cols = (spatial[-2], spatial[-1]) if len(spatial) >= 2 else (1, 1)
        gout = _lib.empty(list(out_shape), dtype=torch.float32, device=grads[0].device)
        ch = (C.c_int32 * len(channels))(*[int(c) for c in channels])
        _lib.check(l.pde_amd_scatter_field_grads_affine(_lib.ptr_array(grads, len(grads)), B, T_out, S, Cn, ch, len(channels),
                                                        rows, cols, int(bool(transpose)), scale.data_ptr() if scale is not None else None,
                                                        gout.data_ptr(), _stream(gout)), "scatter_field_grads")
        return gout


_backend = HipReductions()


def set_backend(b):
    """Swap the primitive provider (tests install a CPU interpreter); returns the previous one."""
    global _backend
    old, _backend = _backend, b
    return old


def get_backend():
    return _backend
