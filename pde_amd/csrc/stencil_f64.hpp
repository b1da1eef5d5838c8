// Double-precision residual forward: the same generated residual code (P::residual is generic in the accessor's scalar
// type) evaluated in fp64 on fp64 fields, for the tooling that measures residual scales of ground-truth data, where the
// residual is a small difference of large terms.  One workgroup per (sample, residual frame) strides over the sites of
// its frame and reads every tap straight from global memory (wrapped on periodic axes, zero beyond a wall, as the tile
// loaders of the training kernels do).  Throughput is not a goal here.
#pragma once
#include "stencil_kernels.hpp"

namespace pde_amd {

struct Args64 {
  const double* f[MAXF];
  const double* aux[MAXA];
  const double* sp[MAXSP];
  double* rout[MAXE];      // residual output [B, TR, S] per equation (optional)
  const double* wt;        // weights [NE][TR]
  double* partial;         // per-workgroup partial sums [B * TR][NE]
  double c[MAXC];
  Geom geo;
};

template <class P>
struct DirectAcc64 {
  using real = double;
  double cv[MAXC], spv[MAXSP], uv[P::NU > 0 ? P::NU : 1];
  const double* fp[MAXF];      // field pointers at (b, frame t, 0)
  const double* xp[MAXA];      // auxiliary-field pointers at (b, 0)
  long S;
  int i0, i1, i2, N0, N1, N2;
  __device__ __forceinline__ double c(int i) const { return cv[i]; }
  __device__ __forceinline__ double k(double x) const { return x; }
  __device__ __forceinline__ double sp(int i) const { return spv[i]; }
  __device__ __forceinline__ double u(int i) const { return uv[i]; }
  static __device__ __forceinline__ double sel(double c, double x, double y) { return c >= 0.0 ? x : y; }
  static __device__ __forceinline__ double pick(bool p, double x, double y) { return p ? x : y; }
  __device__ __forceinline__ bool at(int axis, int k) const { return (axis == 0 ? i0 : axis == 1 ? i1 : i2) == k; }
  __device__ __forceinline__ bool ath(int axis, int k) const {
    return axis == 0 ? (i0 == N0 - 1 - k) : axis == 1 ? (i1 == N1 - 1 - k) : (i2 == N2 - 1 - k);
  }
  template <bool PER>
  static __device__ __forceinline__ int coord(int c, int N, bool& ok) {
    if (PER) {
      c %= N;
      return c < 0 ? c + N : c;
    }
    if (c < 0 || c >= N) { ok = false; return 0; }
    return c;
  }
  __device__ __forceinline__ long site(int o0, int o1, int o2, bool& ok) const {
    const int c0 = coord<P::PER0>(i0 + o0, N0, ok), c1 = coord<P::PER1>(i1 + o1, N1, ok), c2 = coord<P::PER2>(i2 + o2, N2, ok);
    return ((long)c0 * N1 + c1) * N2 + c2;
  }
  __device__ __forceinline__ double F(int f, int ot, int o0, int o1, int o2) const {
    bool ok = true;
    const long s = site(o0, o1, o2, ok);
    return ok ? fp[f][(long)ot * S + s] : 0.0;
  }
  __device__ __forceinline__ double X(int x, int o0, int o1, int o2) const {
    bool ok = true;
    const long s = site(o0, o1, o2, ok);
    return ok ? xp[x][s] : 0.0;
  }
  __device__ __forceinline__ double G(int, int, int, int, int) const { return 0.0; }
};

template <class P>
__global__ __launch_bounds__(256) void fwd_direct_f64_kernel(const Args64 a) {
  __shared__ double s_red[4 * P::NE];
  const Geom& g = a.geo;
  const int w = blockIdx.x;
  const int tr = w % g.TR, b = w / g.TR;
  const int t = tr + P::TLO;
  DirectAcc64<P> acc;
#pragma unroll
  for (int i = 0; i < P::NC; ++i) acc.cv[i] = a.c[i];
#pragma unroll
  for (int i = 0; i < P::NSP; ++i) acc.spv[i] = a.sp[i][b];
  P::uniforms(acc, acc.uv);
#pragma unroll
  for (int f = 0; f < P::NF; ++f) acc.fp[f] = a.f[f] + ((long)b * g.T + t) * g.S;
#pragma unroll
  for (int x = 0; x < P::NA; ++x) acc.xp[x] = a.aux[x] + (long)b * g.S;
  acc.S = g.S; acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  double wk[P::NE], sum[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) { wk[k] = a.wt[k * g.TR + tr]; sum[k] = 0.0; }
  for (long s = threadIdx.x; s < g.S; s += 256) {
    acc.i2 = (int)(s % g.N2);
    acc.i1 = (int)((s / g.N2) % g.N1);
    acc.i0 = (int)(s / ((long)g.N2 * g.N1));
    const bool on = acc.i0 >= g.lo0 && acc.i0 < g.N0 - g.hi0 && acc.i1 >= g.lo1 && acc.i1 < g.N1 - g.hi1 &&
                    acc.i2 >= g.lo2 && acc.i2 < g.N2 - g.hi2;
    double R[P::NE];
#pragma unroll
    for (int k = 0; k < P::NE; ++k) R[k] = 0.0;
    if (on) P::residual(acc, R);
#pragma unroll
    for (int k = 0; k < P::NE; ++k) {
      sum[k] += wk[k] * R[k] * R[k];
      if (a.rout[k]) a.rout[k][((long)b * g.TR + tr) * g.S + s] = R[k];
    }
  }
  block_sum_store<P::NE, 256>(sum, s_red, a.partial + (long)w * P::NE);
}

// sums[B][NE] -> totals[k] = sum_b sums[b][k] and totals[NE] = their sum (left to right), in double
template <int NE>
__global__ __launch_bounds__(64) void totals_f64_kernel(const double* __restrict__ sums, double* __restrict__ totals, int B) {
  double acc[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < B; b += 64) {
#pragma unroll
    for (int k = 0; k < NE; ++k) acc[k] += sums[(long)b * NE + k];
  }
  double tot = 0.0;
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const double s = wave_sum(acc[k]);
    tot += s;
    if (threadIdx.x == 0) totals[k] = s;
  }
  if (threadIdx.x == 0) totals[NE] = tot;
}

}  // namespace pde_amd
