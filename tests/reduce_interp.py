"""CPU stand-in for pde_amd.reductions.HipReductions (TEST INFRASTRUCTURE ONLY).

Implements the same five primitives with NumPy in fp64 so that the host logic of pde_amd.metrics, pde_amd.boundary
and pde_amd.glue (term construction, counts, epsilons, autograd wiring) is testable without a GPU.  Installed through
pde_amd.reductions.set_backend by the CPU tests; never used by the product.
"""
import numpy as np
import torch

from pde_amd import reductions as R


def _np(t):
    return t.detach().cpu().double().numpy()


class InterpReductions:
    name = "interp"

    def vrmse_from_stats(self, stats, valid, rows):
        st = _np(stats)
        if not len(valid):
            return torch.full((2,), float("nan"))
        sel = st[list(valid)]
        n = float(rows)
        var = sel[:, R.S_G2] / n - (sel[:, R.S_G] / n) ** 2
        v = np.sqrt((sel[:, R.S_D2] / n) / (var + 1e-8)).mean()
        r = np.sqrt(sel[:, R.S_D2].sum() / (n * len(valid)) + 1e-8)
        return torch.tensor([v, r], dtype=torch.float32)

    def band_power(self, spectrum, bin_start, point_index, nbins):
        F = spectrum.detach().cpu().numpy().astype(np.complex128)
        bs, pi = bin_start.cpu().numpy().astype(np.int64), point_index.cpu().numpy().astype(np.int64)
        out = np.zeros((F.shape[0], nbins, F.shape[2]))
        for i in range(nbins):
            members = pi[bs[i]:bs[i + 1]]
            out[:, i] = (np.abs(F[:, members]) ** 2).sum(axis=1)
        return torch.from_numpy(out.astype(np.float32)).to(spectrum.device)

    def pair_stats(self, pred, target, groups, rows, channels, rse_eps=None, affine=None):
        if affine is not None:      # fp32 mul, then add, as torch does
            pred = pred.detach().float().reshape(groups, rows, channels) * affine[0][:, None, :] + affine[1][:, None, :]
        p = _np(pred).reshape(groups, rows, channels)
        g = _np(target).reshape(groups, rows, channels)
        d = p - g
        st = np.zeros((groups, channels, R.NSTAT))
        st[..., R.S_D2] = (d * d).sum(1)
        st[..., R.S_G2] = (g * g).sum(1)
        st[..., R.S_G] = g.sum(1)
        st[..., R.S_P] = p.sum(1)
        if rse_eps is not None:
            st[..., R.S_RSE] = (d * d / (g * g + rse_eps)).sum(1)
        st[..., R.S_MAXD] = np.abs(d).max(1) if rows else 0.0
        return torch.tensor(st, dtype=torch.float64, device=pred.device)

    def pair_stats_gather(self, pred, target, input_data, channels, affine=None):
        B, Cn = int(pred.shape[0]), int(pred.shape[-1])
        rows = pred.numel() // (B * Cn)
        stats = self.pair_stats(pred, target, B, rows, Cn, affine=affine)
        return stats, self.gather_fields(pred.detach().float(), input_data.detach().float(), channels, False, affine)

    @staticmethod
    def _term_offsets(dims, t, pool):
        lists = [pool[t.idx_off[k]: t.idx_off[k] + t.n[k]] for k in range(4)]
        strides = [dims[1] * dims[2] * dims[3], dims[2] * dims[3], dims[3], 1]
        off = np.zeros((1,), dtype=np.int64)
        for k in range(4):
            off = (off[:, None] + np.asarray(lists[k], dtype=np.int64)[None, :] * strides[k]).reshape(-1)
        return off, strides[t.axis]

    @staticmethod
    def _denorm(pred, outer, dims, affine):
        """fp32 pred * scale + shift per (sample, channel), mul then add as torch does; [outer, n] float64"""
        n = dims[0] * dims[1] * dims[2] * dims[3]
        x = pred.detach().cpu().float().reshape(outer, n // dims[3], dims[3])
        if affine is not None:
            sc, sh = (a.detach().cpu().float() for a in affine)
            per = outer // sc.shape[0]
            x = x * sc.repeat_interleave(per, 0)[:, None, :] + sh.repeat_interleave(per, 0)[:, None, :]
        return x.double().numpy().reshape(outer, n)

    def _values(self, pred, target, outer, dims, t, pool, affine=None):
        n = dims[0] * dims[1] * dims[2] * dims[3]
        x = self._denorm(pred, outer, dims, affine)
        if t.use_target:
            x = x - _np(target).reshape(outer, n)
        off, ts = self._term_offsets(dims, t, pool)
        v = np.zeros((outer, off.size))
        for j in range(t.n_taps):
            v += t.coef[j] * x[:, off + t.shift[j] * ts]
        return v, off, ts

    def terms_forward(self, pred, target, outer, dims, terms, indices, affine=None):
        pool = indices.cpu().numpy()
        return torch.tensor([float((self._values(pred, target, outer, dims, t, pool, affine)[0] ** 2).sum()) for t in terms],
                            dtype=torch.float64, device=pred.device)

    def terms_backward(self, pred, target, outer, dims, terms, indices, scales, affine, disjoint, grad):
        pool = indices.cpu().numpy()
        n = dims[0] * dims[1] * dims[2] * dims[3]
        g = np.zeros((outer, n))
        sc = scales.detach().cpu().double().numpy()
        for k, t in enumerate(terms):
            v, off, ts = self._values(pred, target, outer, dims, t, pool, affine)
            for j in range(t.n_taps):
                np.add.at(g, (slice(None), off + t.shift[j] * ts), 2.0 * sc[k] * t.coef[j] * v)
        if affine is not None:
            per = outer // affine[0].shape[0]
            g = (g.reshape(outer, n // dims[3], dims[3]) * _np(affine[0]).repeat(per, 0)[:, None, :]).reshape(outer, n)
        grad += torch.tensor(g.reshape(grad.shape), dtype=grad.dtype)
        return grad

    @staticmethod
    def _bc(x, ndim):
        return x.reshape((x.shape[0],) + (1,) * (ndim - 2) + (x.shape[1],))

    def gather_fields(self, output, input_data, channels, transpose, affine=None):
        if affine is not None:
            output = output.float() * self._bc(affine[0], output.dim()) + self._bc(affine[1], output.dim())
        out = []
        for c in channels:
            f = torch.cat([input_data[:, 0:1, ..., c], output[..., c]], dim=1).float()
            out.append((f.transpose(-1, -2) if transpose else f).contiguous())
        return out

    def scatter_field_grads(self, grads, out_shape, channels, transpose, scale=None):
        g = torch.zeros(out_shape, dtype=torch.float32)
        for c, gf in zip(channels, grads):
            gf = gf.transpose(-1, -2) if transpose else gf
            g[..., c] += gf[:, 1:]
        if scale is not None:
            g = g * self._bc(scale, g.dim())
        return g
