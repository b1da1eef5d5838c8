// Hand-written gfx950 kernel skeletons for finite-difference residual losses and their adjoints.
//
// A PDE is a traits struct (csrc/gen/pde_*.inc) that supplies the residual of every equation at one
// lattice site and the gather-form adjoint at one grid point, both written against an accessor.
// The skeletons here own everything else: LDS halo tiles with periodic wrap or wall fill, the
// work -> (sample, frame, tile) map, fp64 wave/block reductions without atomics, and the stores.
//
// Layout: every field is a contiguous fp32 tensor [B, T, N0, N1, N2] (unused leading axes = 1).
// A workgroup owns one spatial tile of one frame; a wave covers 64 consecutive points along the
// fastest axis, so global loads/stores are coalesced 256-byte rows and LDS reads are conflict-free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pde_amd {

constexpr int MAXF = 4;    // fields
constexpr int MAXE = 4;    // equations
constexpr int MAXA = 2;    // time-independent auxiliary fields
constexpr int MAXSP = 3;   // per-sample parameters
constexpr int MAXC = 16;   // per-instance constants

struct Geom {
  int B, T, TR;                        // samples, frames, residual frames (T - TLO - THI)
  int N0, N1, N2;                      // spatial extents
  int lo0, lo1, lo2, hi0, hi1, hi2;    // residual lattice: lo <= index < N - hi
  int nt0, nt1, nt2;                   // tiles per axis
  long S;                              // N0 * N1 * N2
};

struct Args {
  const float* f[MAXF];
  const float* aux[MAXA];
  const float* sp[MAXSP];
  const float* r[MAXE];    // stored residuals [B, TR, S] per equation (backward, stored mode)
  float* rout[MAXE];       // residual output (forward, optional)
  float* g[MAXF];          // gradients
  const float* wt;         // weights [NE][TR]
  const float* ups[MAXE + 1];   // backward: upstream gradients of the per-equation losses and of their total (device scalars, nullable)
  const double* gsums;     // backward: upstream gradient of sums[B][NE] (nullable)
  double* partial;         // forward: per-workgroup partial sums [nwg][NE]
  int chunk;               // marching kernels: frames per workgroup
  float c[MAXC];
  Geom geo;
};

// Uniform read of a small read-only table (weights, per-sample parameters, upstream gradients) through the
// constant address space: the compiler may then use scalar loads, whose waits (lgkmcnt) do not get
// entangled with the vector-memory counter that the frame prefetch relies on.
typedef const float __attribute__((address_space(4))) * cfloat_ptr;
typedef const double __attribute__((address_space(4))) * cdouble_ptr;
__device__ __forceinline__ float ldc(const float* p, long i) { return ((cfloat_ptr)(uintptr_t)p)[i]; }
__device__ __forceinline__ double ldc(const double* p, long i) { return ((cdouble_ptr)(uintptr_t)p)[i]; }

// ------------------------------------------------------------------------------------------------
// tiles
// ------------------------------------------------------------------------------------------------
template <int T0_, int T1_, int T2_, int H0_, int H1_, int H2_>
struct Tile {
  static constexpr int T0 = T0_, T1 = T1_, T2 = T2_, H0 = H0_, H1 = H1_, H2 = H2_;
  static constexpr int D0 = T0 + 2 * H0, D1 = T1 + 2 * H1, D2 = T2 + 2 * H2;
  static constexpr int S1 = D2, S0 = D1 * D2, SIZE = D0 * D1 * D2;
  __device__ static __forceinline__ int at(int l0, int l1, int l2) {
    return (l0 + H0) * S0 + (l1 + H1) * S1 + (l2 + H2);
  }
};

template <bool PER>
__device__ __forceinline__ int wrap_coord(int c, int N, bool& ok) {
  if (PER) {
    if (c < 0) c += N;
    else if (c >= N) c -= N;
    c = min(max(c, 0), N - 1);   // beyond one period: only reachable from sites outside the grid
  } else {
    if (c < 0 || c >= N) { ok = false; c = 0; }
  }
  return c;
}

// Fill one LDS tile (with halo) from one frame of one field. Out-of-grid points of a wall axis read 0.
template <class TL, bool P0, bool P1, bool P2, int NT>
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const float* __restrict__ src,
                                          int o0, int o1, int o2, const Geom& g, float scale) {
  for (int i = threadIdx.x; i < TL::SIZE; i += NT) {
    const int l2 = i % TL::D2, rr = i / TL::D2, l1 = rr % TL::D1, l0 = rr / TL::D1;
    bool ok = true;
    const int c0 = wrap_coord<P0>(o0 + l0 - TL::H0, g.N0, ok);
    const int c1 = wrap_coord<P1>(o1 + l1 - TL::H1, g.N1, ok);
    const int c2 = wrap_coord<P2>(o2 + l2 - TL::H2, g.N2, ok);
    float v = 0.0f;
    if (ok) v = src[((long)c0 * g.N1 + c1) * g.N2 + c2] * scale;
    dst[i] = v;
  }
}

template <class TL, int NT>
__device__ __forceinline__ void zero_tile(float* dst) {
  for (int i = threadIdx.x; i < TL::SIZE; i += NT) dst[i] = 0.0f;
}

// Blocks b and b+8 share an XCD (and its L2): give each XCD a contiguous run of work items so that
// neighbouring tiles, which share halos, meet in one L2. Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int pid, int n) {
  const int q = n >> 3, r = n & 7, x = pid & 7, s = pid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// ------------------------------------------------------------------------------------------------
// accessors handed to the generated residual / adjoint code
// ------------------------------------------------------------------------------------------------
// Moves a wave-uniform value into a vector register and hides its uniformity from the compiler.
__device__ __forceinline__ float to_vgpr(float x) {
  float r;
  asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
  return r;
}

template <class P>
struct SiteBase {
  using real = float;        // the scalar type the generated code computes in
  float cv[MAXC];            // constants (only the first P::NC are ever set or read)
  float spv[MAXSP];          // per-sample parameters of this sample
  float uv[P::NU > 0 ? P::NU : 1];   // hoisted uniform subexpressions (P::uniforms); entries an accessor never reads are dead code
  int i0, i1, i2, N0, N1, N2;
  bool vg = false;           // set_uniforms<true>: uniform values live in vector registers
  __device__ __forceinline__ float c(int i) const { return cv[i]; }
  __device__ __forceinline__ float k(float x) const { return vg ? to_vgpr(x) : x; }   // numeric literal
  __device__ __forceinline__ float sp(int i) const { return spv[i]; }
  __device__ __forceinline__ float u(int i) const { return uv[i]; }
  static __device__ __forceinline__ float sel(float c, float x, float y) { return c >= 0.0f ? x : y; }   // upwind choice
  static __device__ __forceinline__ float pick(bool p, float x, float y) { return p ? x : y; }            // wall / ghost-cell choice
  // VG: keep the uniform values in vector registers.  A VALU instruction with a scalar-register operand
  // issues at half the rate of one whose operands are all vector registers (measured: 4 vs 2-2.5 cycles per
  // wave64 instruction), and these values are operands of most of the generated arithmetic.
  template <bool VG>
  __device__ __forceinline__ void set_uniforms(const float* c, const float* const* spp, int b) {
    vg = VG;
#pragma unroll
    for (int i = 0; i < P::NC; ++i) cv[i] = VG ? to_vgpr(c[i]) : c[i];
#pragma unroll
    for (int i = 0; i < P::NSP; ++i) spv[i] = VG ? to_vgpr(ldc(spp[i], b)) : ldc(spp[i], b);
    P::uniforms(*this, uv);
  }
  __device__ __forceinline__ bool at(int axis, int k) const { return (axis == 0 ? i0 : axis == 1 ? i1 : i2) == k; }
  __device__ __forceinline__ bool ath(int axis, int k) const {
    return axis == 0 ? (i0 == N0 - 1 - k) : axis == 1 ? (i1 == N1 - 1 - k) : (i2 == N2 - 1 - k);
  }
};

// Residual evaluation: ot == 0 reads the halo tile; ot != 0 reads a time tile when the PDE has spatial
// offsets at other frames, else the same grid point of the neighbouring frame straight from global memory.
template <class P, class TL, class TT, class TX>
struct ResAcc : SiteBase<P> {
  const float* lds;          // &tile_f0[site]
  const float* ldst;         // &time_tile_f0_slot0[site]
  const float* ldsx;         // &aux_tile[site]
  const float* fp[MAXF];     // field pointers at (b, frame, 0)
  long S;
  int goff;
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    if (ot == 0) return lds[f * TL::SIZE + o0 * TL::S0 + o1 * TL::S1 + o2];
    if ((P::RT0 | P::RT1 | P::RT2) != 0) {
      const int slot = ot < 0 ? ot + P::TLO : ot + P::TLO - 1;
      return ldst[(slot * P::NF + f) * TT::SIZE + o0 * TT::S0 + o1 * TT::S1 + o2];
    }
    return fp[f][goff + (long)ot * S];
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    return ldsx[x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + o2];
  }
  __device__ __forceinline__ float G(int, int, int, int, int) const { return 0.0f; }
};

// Adjoint evaluation at an output point of frame tq.
template <class P, class TG, class TGT, class TA, class TAT, class TX>
struct AdjAcc : SiteBase<P> {
  static constexpr int NG = P::GTLO + P::GTHI + 1;
  const float* ldsg;         // G tile of frame tq
  const float* ldsgt;        // G tiles of the other frames (when they need a halo)
  const float* ldsa;         // F tile of frame tq
  const float* ldsat;        // F tiles of the other frames (when they need a halo)
  const float* ldsx;
  const float* fp[MAXF];     // field pointers at (b, tq, 0)
  const float* rp[MAXE];     // stored-residual pointers at (b, 0, 0)
  float cf[NG][MAXE];        // coefficient of each needed residual frame (0 when that frame is outside the lattice)
  int tr0;                   // residual-frame index of tq (tq - TLO)
  int tq, T;
  long S;
  int goff;
  __device__ __forceinline__ float G(int k, int ot, int o0, int o1, int o2) const {
    if (ot == 0) return ldsg[k * TG::SIZE + o0 * TG::S0 + o1 * TG::S1 + o2];
    if ((P::GT0 | P::GT1 | P::GT2) != 0) {
      const int slot = ot < 0 ? ot + P::GTLO : ot + P::GTLO - 1;
      return ldsgt[(slot * P::NE + k) * TGT::SIZE + o0 * TGT::S0 + o1 * TGT::S1 + o2];
    }
    const float cfk = cf[ot + P::GTLO][k];
    return cfk != 0.0f ? cfk * rp[k][(long)(tr0 + ot) * S + goff] : 0.0f;
  }
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    if (ot == 0) return ldsa[f * TA::SIZE + o0 * TA::S0 + o1 * TA::S1 + o2];
    if ((P::AT0 | P::AT1 | P::AT2) != 0) {
      const int slot = ot < 0 ? ot + P::ATLO : ot + P::ATLO - 1;
      return ldsat[(slot * P::NF + f) * TAT::SIZE + o0 * TAT::S0 + o1 * TAT::S1 + o2];
    }
    return ((unsigned)(tq + ot) < (unsigned)T) ? fp[f][goff + (long)ot * S] : 0.0f;
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    return ldsx[x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + o2];
  }
};

// ------------------------------------------------------------------------------------------------
// reductions: 64-lane shuffle tree, then one LDS slot per wave, no atomics, fixed order
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <int NE, int NT>
__device__ __forceinline__ void block_sum_store(const double (&acc)[NE], double* red, double* out) {
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[wv * NE + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NE) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * NE + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

template <class P, class TL, int NT = P::NT>
__device__ __forceinline__ void point_coords(int k, int& l0, int& l1, int& l2) {
  // a thread owns PPT consecutive rows of one tile column
  constexpr int PPT = (TL::T0 * TL::T1 * TL::T2) / NT;
  const int row = (threadIdx.x / TL::T2) * PPT + k;
  l2 = threadIdx.x % TL::T2;
  l1 = row % TL::T1;
  l0 = row / TL::T1;
}

template <class P>
__device__ __forceinline__ bool in_lattice(const Geom& g, int i0, int i1, int i2) {
  return i0 >= g.lo0 && i0 < g.N0 - g.hi0 && i1 >= g.lo1 && i1 < g.N1 - g.hi1 && i2 >= g.lo2 && i2 < g.N2 - g.hi2;
}

// G_k = coef * R_k with coef = 2 * weight[k][frame] * (dL/dloss_k + dL/dtotal) + 2 * weight * dL/dsums[b][k]
template <class P>
__device__ __forceinline__ void upstream_of(const Args& a, int b, float (&up)[P::NE]) {
  const float gt = a.ups[P::NE] ? ldc(a.ups[P::NE], 0) : 0.0f;
#pragma unroll
  for (int k = 0; k < P::NE; ++k) {
    float u = gt + (a.ups[k] ? ldc(a.ups[k], 0) : 0.0f);
    if (a.gsums) u += (float)ldc(a.gsums, (long)b * P::NE + k);
    up[k] = 2.0f * u;
  }
}

// ------------------------------------------------------------------------------------------------
// forward: one workgroup = (sample, residual frame, tile)
//   sums of wt[k][frame] * R_k^2 over the tile -> partial[work][k];   optional store of R_k
// ------------------------------------------------------------------------------------------------
template <class P, bool WRITE_R>
__global__ __launch_bounds__(P::NT) void fwd_kernel(const Args a) {
  using TL = Tile<P::FT0, P::FT1, P::FT2, P::RH0, P::RH1, P::RH2>;
  using TT = Tile<P::FT0, P::FT1, P::FT2, P::RT0, P::RT1, P::RT2>;
  using TX = Tile<P::FT0, P::FT1, P::FT2, P::XH0, P::XH1, P::XH2>;
  constexpr bool TT_ON = (P::RT0 | P::RT1 | P::RT2) != 0;
  constexpr bool AUX_ON = P::NA > 0;
  constexpr int NT = P::NT;
  constexpr int PPT = (TL::T0 * TL::T1 * TL::T2) / NT;
  static_assert(PPT * NT == TL::T0 * TL::T1 * TL::T2, "tile must be a multiple of the workgroup");
  static_assert(NT % TL::T2 == 0, "a wave must lie along the fastest axis");

  __shared__ float s_f[P::NF * TL::SIZE];
  __shared__ float s_t[TT_ON ? (P::TLO + P::THI) * P::NF * TT::SIZE : 1];
  __shared__ float s_x[AUX_ON ? P::NA * TX::SIZE : 1];
  __shared__ double s_red[(NT / 64) * P::NE];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int tile = w % ntile, rest = w / ntile;
  const int tr = rest % g.TR, b = rest / g.TR;
  const int tau = tr + P::TLO;
  const int o2 = (tile % g.nt2) * TL::T2, o1 = ((tile / g.nt2) % g.nt1) * TL::T1, o0 = (tile / (g.nt2 * g.nt1)) * TL::T0;
  const long frame = ((long)b * g.T + tau) * g.S;

#pragma unroll
  for (int f = 0; f < P::NF; ++f)
    load_tile<TL, P::PER0, P::PER1, P::PER2, NT>(s_f + f * TL::SIZE, a.f[f] + frame, o0, o1, o2, g, 1.0f);
  if (TT_ON) {
#pragma unroll
    for (int s = 0; s < P::TLO + P::THI; ++s) {
      const int ot = s < P::TLO ? s - P::TLO : s - P::TLO + 1;
#pragma unroll
      for (int f = 0; f < P::NF; ++f)
        load_tile<TT, P::PER0, P::PER1, P::PER2, NT>(s_t + (s * P::NF + f) * TT::SIZE, a.f[f] + frame + ot * g.S, o0, o1, o2, g, 1.0f);
    }
  }
  if (AUX_ON) {
#pragma unroll
    for (int x = 0; x < P::NA; ++x)
      load_tile<TX, P::PER0, P::PER1, P::PER2, NT>(s_x + x * TX::SIZE, a.aux[x] + (long)b * g.S, o0, o1, o2, g, 1.0f);
  }
  __syncthreads();

  ResAcc<P, TL, TT, TX> acc;
  acc.template set_uniforms<false>(a.c, a.sp, b);
#pragma unroll
  for (int f = 0; f < P::NF; ++f) acc.fp[f] = a.f[f] + frame;
  acc.S = g.S; acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;

  float wk[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) wk[k] = ldc(a.wt, k * g.TR + tr);

  double sum[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) sum[k] = 0.0;

#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    int l0, l1, l2;
    point_coords<P, TL>(p, l0, l1, l2);
    const int i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
    const bool inside = i0 < g.N0 && i1 < g.N1 && i2 < g.N2;
    const bool active = inside && in_lattice<P>(g, i0, i1, i2);
    const int goff = (i0 * g.N1 + i1) * g.N2 + i2;
    float R[P::NE];
#pragma unroll
    for (int k = 0; k < P::NE; ++k) R[k] = 0.0f;
    if (active) {
      acc.i0 = i0; acc.i1 = i1; acc.i2 = i2; acc.goff = goff;
      acc.lds = s_f + TL::at(l0, l1, l2);
      acc.ldst = s_t + TT::at(l0, l1, l2);
      acc.ldsx = s_x + TX::at(l0, l1, l2);
      P::residual(acc, R);
#pragma unroll
      for (int k = 0; k < P::NE; ++k) sum[k] += (double)R[k] * (double)R[k] * (double)wk[k];
    }
    if (WRITE_R && inside) {
      const long ro = ((long)b * g.TR + tr) * g.S + goff;
#pragma unroll
      for (int k = 0; k < P::NE; ++k) a.rout[k][ro] = R[k];
    }
  }
  block_sum_store<P::NE, NT>(sum, s_red, a.partial + (long)w * P::NE);
}

// partial[b][items][NE] -> sums[b][NE], fixed summation order
template <int NE>
__global__ __launch_bounds__(256) void finalize_kernel(const double* __restrict__ partial, double* __restrict__ sums, int items) {
  __shared__ double red[4 * NE];
  const int b = blockIdx.x;
  double acc[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) acc[k] = 0.0;
  const double* p = partial + (long)b * items * NE;
  for (int i = threadIdx.x; i < items; i += 256) {
#pragma unroll
    for (int k = 0; k < NE; ++k) acc[k] += p[(long)i * NE + k];
  }
  block_sum_store<NE, 256>(acc, red, sums + (long)b * NE);
}

// sums[B][NE] -> totals[k] = fp32(sum_b sums[b][k]) and totals[NE] = their fp32 sum (left to right)
template <int NE>
__global__ __launch_bounds__(64) void totals_kernel(const double* __restrict__ sums, float* __restrict__ totals, int B) {
  double acc[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) acc[k] = 0.0;
  for (int b = threadIdx.x; b < B; b += 64) {
#pragma unroll
    for (int k = 0; k < NE; ++k) acc[k] += sums[(long)b * NE + k];
  }
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const double s = wave_sum(acc[k]);
    const float f = (float)s;
    tot += f;
    if (threadIdx.x == 0) totals[k] = f;
  }
  if (threadIdx.x == 0) totals[NE] = tot;
}

// Both of the above in one launch for small problems: one workgroup of 16 waves.  With 16 or more samples wave w sums
// the partials of samples w, w+16, ... (lanes stride over a sample's items); with fewer, 16/B waves share a sample's
// items in contiguous segments that one thread then adds in segment order.  Thread 0 adds the per-slot totals in slot
// order.  Fixed order, so repeatable; it saves a launch and the gap between two tiny kernels.
template <int NE>
__global__ __launch_bounds__(1024) void finalize_totals_kernel(const double* __restrict__ partial, double* __restrict__ sums,
                                                               float* __restrict__ totals, int items, int B) {
  __shared__ double s_tot[16 * NE];
  __shared__ double s_seg[16 * NE];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nseg = B >= 16 ? 1 : 16 / B;      // waves per sample
  if (nseg == 1) {
    double tot[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) tot[k] = 0.0;
    for (int b = wv; b < B; b += 16) {
      double acc[NE];
#pragma unroll
      for (int k = 0; k < NE; ++k) acc[k] = 0.0;
      const double* p = partial + (long)b * items * NE;
      for (int i = lane; i < items; i += 64) {
#pragma unroll
        for (int k = 0; k < NE; ++k) acc[k] += p[(long)i * NE + k];
      }
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        const double s = wave_sum(acc[k]);      // complete in lane 0
        if (lane == 0) { sums[(long)b * NE + k] = s; tot[k] += s; }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NE; ++k) s_tot[wv * NE + k] = tot[k];
    }
  } else {
    const int b = wv / nseg, sg = wv % nseg;
    double acc[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) acc[k] = 0.0;
    if (b < B) {
      const int len = (items + nseg - 1) / nseg;
      const int lo = sg * len, hi = lo + len < items ? lo + len : items;
      const double* p = partial + (long)b * items * NE;
      for (int i = lo + lane; i < hi; i += 64) {
#pragma unroll
        for (int k = 0; k < NE; ++k) acc[k] += p[(long)i * NE + k];
      }
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double s = wave_sum(acc[k]);
      if (lane == 0) s_seg[wv * NE + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 16) {      // thread b: its sample's segments in order; slots past the last sample hold zero
      const int bb = threadIdx.x;
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        double s = 0.0;
        if (bb < B) {
          for (int g = 0; g < nseg; ++g) s += s_seg[(bb * nseg + g) * NE + k];
          sums[(long)bb * NE + k] = s;
        }
        s_tot[bb * NE + k] = s;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      double s = 0.0;
      for (int w = 0; w < 16; ++w) s += s_tot[w * NE + k];
      const float f = (float)s;
      totals[k] = f;
      t += f;
    }
    totals[NE] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// backward, stored-residual mode: one workgroup = (sample, output frame tq, tile)
//   G_k = coef[b][k][frame] * R_k is staged in LDS (with the adjoint's halo) and gathered
// ------------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(P::NT) void bwd_stored_kernel(const Args a) {
  using TG = Tile<P::BT0, P::BT1, P::BT2, P::GH0, P::GH1, P::GH2>;
  using TGT = Tile<P::BT0, P::BT1, P::BT2, P::GT0, P::GT1, P::GT2>;
  using TA = Tile<P::BT0, P::BT1, P::BT2, P::AH0, P::AH1, P::AH2>;
  using TAT = Tile<P::BT0, P::BT1, P::BT2, P::AT0, P::AT1, P::AT2>;
  using TX = Tile<P::BT0, P::BT1, P::BT2, P::XH0, P::XH1, P::XH2>;
  constexpr bool GT_ON = (P::GT0 | P::GT1 | P::GT2) != 0;
  constexpr bool AT_ON = (P::AT0 | P::AT1 | P::AT2) != 0;
  constexpr bool AF_ON = P::ANY_AF != 0;
  constexpr bool AUX_ON = P::NA > 0;
  constexpr int NT = P::NT;
  constexpr int PPT = (TG::T0 * TG::T1 * TG::T2) / NT;
  constexpr int NGT = P::GTLO + P::GTHI, NAT = P::ATLO + P::ATHI;
  static_assert(PPT * NT == TG::T0 * TG::T1 * TG::T2, "tile must be a multiple of the workgroup");
  static_assert(NT % TG::T2 == 0, "a wave must lie along the fastest axis");

  __shared__ float s_g[P::NE * TG::SIZE];
  __shared__ float s_gt[GT_ON ? NGT * P::NE * TGT::SIZE : 1];
  __shared__ float s_a[AF_ON ? P::NF * TA::SIZE : 1];
  __shared__ float s_at[AT_ON ? (NAT > 0 ? NAT : 1) * P::NF * TAT::SIZE : 1];
  __shared__ float s_x[AUX_ON ? P::NA * TX::SIZE : 1];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int tile = w % ntile, rest = w / ntile;
  const int tq = rest % g.T, b = rest / g.T;
  const int o2 = (tile % g.nt2) * TG::T2, o1 = ((tile / g.nt2) % g.nt1) * TG::T1, o0 = (tile / (g.nt2 * g.nt1)) * TG::T0;
  const long frame = ((long)b * g.T + tq) * g.S;

  AdjAcc<P, TG, TGT, TA, TAT, TX> acc;
  acc.template set_uniforms<false>(a.c, a.sp, b);
#pragma unroll
  for (int f = 0; f < P::NF; ++f) acc.fp[f] = a.f[f] + frame;
#pragma unroll
  for (int k = 0; k < P::NE; ++k) acc.rp[k] = a.r[k] + (long)b * g.TR * g.S;
  acc.S = g.S; acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  acc.tq = tq; acc.T = g.T; acc.tr0 = tq - P::TLO;

  // coefficients of the residual frames this output frame touches
  float up[P::NE];
  upstream_of<P>(a, b, up);
#pragma unroll
  for (int s = 0; s < P::GTLO + P::GTHI + 1; ++s) {
    const int trs = tq + (s - P::GTLO) - P::TLO;
    const bool ok = trs >= 0 && trs < g.TR;
#pragma unroll
    for (int k = 0; k < P::NE; ++k) acc.cf[s][k] = ok ? up[k] * ldc(a.wt, k * g.TR + trs) : 0.0f;
  }

#pragma unroll
  for (int k = 0; k < P::NE; ++k) {
    const float cfk = acc.cf[P::GTLO][k];
    if (cfk != 0.0f)
      load_tile<TG, P::PER0, P::PER1, P::PER2, NT>(s_g + k * TG::SIZE, acc.rp[k] + (long)acc.tr0 * g.S, o0, o1, o2, g, cfk);
    else
      zero_tile<TG, NT>(s_g + k * TG::SIZE);
  }
  if (GT_ON) {
#pragma unroll
    for (int s = 0; s < NGT; ++s) {
      const int ot = s < P::GTLO ? s - P::GTLO : s - P::GTLO + 1;
#pragma unroll
      for (int k = 0; k < P::NE; ++k) {
        const float cfk = acc.cf[ot + P::GTLO][k];
        float* dst = s_gt + (s * P::NE + k) * TGT::SIZE;
        if (cfk != 0.0f)
          load_tile<TGT, P::PER0, P::PER1, P::PER2, NT>(dst, acc.rp[k] + (long)(acc.tr0 + ot) * g.S, o0, o1, o2, g, cfk);
        else
          zero_tile<TGT, NT>(dst);
      }
    }
  }
  if (AF_ON) {
#pragma unroll
    for (int f = 0; f < P::NF; ++f)
      load_tile<TA, P::PER0, P::PER1, P::PER2, NT>(s_a + f * TA::SIZE, a.f[f] + frame, o0, o1, o2, g, 1.0f);
  }
  if (AT_ON) {
#pragma unroll
    for (int s = 0; s < NAT; ++s) {
      const int ot = s < P::ATLO ? s - P::ATLO : s - P::ATLO + 1;
      const bool ok = (unsigned)(tq + ot) < (unsigned)g.T;
#pragma unroll
      for (int f = 0; f < P::NF; ++f) {
        float* dst = s_at + (s * P::NF + f) * TAT::SIZE;
        if (ok) load_tile<TAT, P::PER0, P::PER1, P::PER2, NT>(dst, a.f[f] + frame + (long)ot * g.S, o0, o1, o2, g, 1.0f);
        else zero_tile<TAT, NT>(dst);
      }
    }
  }
  if (AUX_ON) {
#pragma unroll
    for (int x = 0; x < P::NA; ++x)
      load_tile<TX, P::PER0, P::PER1, P::PER2, NT>(s_x + x * TX::SIZE, a.aux[x] + (long)b * g.S, o0, o1, o2, g, 1.0f);
  }
  __syncthreads();

#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    int l0, l1, l2;
    point_coords<P, TG>(p, l0, l1, l2);
    const int i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
    if (i0 < g.N0 && i1 < g.N1 && i2 < g.N2) {
      acc.i0 = i0; acc.i1 = i1; acc.i2 = i2;
      acc.goff = (i0 * g.N1 + i1) * g.N2 + i2;
      acc.ldsg = s_g + TG::at(l0, l1, l2);
      acc.ldsgt = s_gt + TGT::at(l0, l1, l2);
      acc.ldsa = s_a + TA::at(l0, l1, l2);
      acc.ldsat = s_at + TAT::at(l0, l1, l2);
      acc.ldsx = s_x + TX::at(l0, l1, l2);
      float gr[P::NF];
      P::adjoint(acc, gr);
#pragma unroll
      for (int f = 0; f < P::NF; ++f) __builtin_nontemporal_store(gr[f], &a.g[f][frame + acc.goff]);   // written once, never re-read here
    }
  }
}


// ================================================================================================
// Time-marching kernels: one workgroup owns a spatial tile of one sample and walks through the
// frames, so every frame of every field is fetched from HBM once (plus halo) instead of once per
// frame that references it.  Frames live in an LDS ring; the next frame is fetched into registers
// while the current one is being used.
// ================================================================================================
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Per-thread share of one halo tile: element e = tid + j * NT of the tile, its in-frame global offset
// (or -1 for a wall-side zero) computed once per workgroup; the same offsets serve every field and frame.
template <class TL, bool P0, bool P1, bool P2, int NT>
struct Stager {
  static constexpr int EPT = (TL::SIZE + NT - 1) / NT;
  int off[EPT];
  __device__ __forceinline__ void init(int o0, int o1, int o2, const Geom& g) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = threadIdx.x + j * NT;
      const int l2 = i % TL::D2, rr = i / TL::D2, l1 = rr % TL::D1, l0 = rr / TL::D1;
      bool ok = i < TL::SIZE;
      const int c0 = wrap_coord<P0>(o0 + l0 - TL::H0, g.N0, ok);
      const int c1 = wrap_coord<P1>(o1 + l1 - TL::H1, g.N1, ok);
      const int c2 = wrap_coord<P2>(o2 + l2 - TL::H2, g.N2, ok);
      off[j] = ok ? (c0 * g.N1 + c1) * g.N2 + c2 : -1;
    }
  }
  __device__ __forceinline__ void fetch(float (&v)[EPT], const float* __restrict__ frame) const {
#pragma unroll
    // always a valid address, and no use of the result here: the loads stay in flight until store()
    for (int j = 0; j < EPT; ++j) v[j] = frame[max(off[j], 0)];
  }
  __device__ __forceinline__ void store(const float (&v)[EPT], float* __restrict__ tile) const {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = threadIdx.x + j * NT;
      if (EPT * NT == TL::SIZE || i < TL::SIZE) tile[i] = off[j] >= 0 ? v[j] : 0.0f;   // wall-side points read 0
    }
  }
};

// Residual accessor over the LDS ring: frame (tau + ot) sits in slot (tau + ot) mod NS.
template <class P, class TL, class TX, int NS>
struct RingAcc : SiteBase<P> {
  const float* fr[NS];       // &slot_of(tau - TLO + i)[field 0][site]
  const float* ldsx;
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    return fr[ot + P::TLO][f * TL::SIZE + o0 * TL::S0 + o1 * TL::S1 + o2];
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    return ldsx[x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + o2];
  }
  __device__ __forceinline__ float G(int, int, int, int, int) const { return 0.0f; }
};

// ------------------------------------------------------------------------------------------------
// forward, marching: workgroup = (sample, chunk of residual frames, tile)
// ------------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(P::NTM) void fwd_march_kernel(const Args a) {
  using TL = Tile<P::FT0, P::FT1, P::FT2, cmax(P::RH0, P::RT0), cmax(P::RH1, P::RT1), cmax(P::RH2, P::RT2)>;
  using TX = Tile<P::FT0, P::FT1, P::FT2, P::XH0, P::XH1, P::XH2>;
  constexpr int NS = P::TLO + P::THI + 1;
  constexpr bool AUX_ON = P::NA > 0;
  constexpr int NT = P::NTM;
  constexpr int PPT = (TL::T0 * TL::T1 * TL::T2) / NT;
  using ST = Stager<TL, P::PER0, P::PER1, P::PER2, NT>;
  static_assert(PPT * NT == TL::T0 * TL::T1 * TL::T2, "tile must be a multiple of the workgroup");
  static_assert(NT % TL::T2 == 0, "a wave must lie along the fastest axis");

  __shared__ float s_f[NS * P::NF * TL::SIZE];
  __shared__ float s_x[AUX_ON ? P::NA * TX::SIZE : 1];
  __shared__ double s_red[(NT / 64) * P::NE];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int nchunk = (g.TR + a.chunk - 1) / a.chunk;
  const int tile = w % ntile, rest = w / ntile;
  const int ch = rest % nchunk, b = rest / nchunk;
  const int ra = P::TLO + ch * a.chunk;                    // first residual frame of this chunk
  const int rb = min(ra + a.chunk, g.T - P::THI);          // one past the last
  const int o2 = (tile % g.nt2) * TL::T2, o1 = ((tile / g.nt2) % g.nt1) * TL::T1, o0 = (tile / (g.nt2 * g.nt1)) * TL::T0;

  ST st;
  st.init(o0, o1, o2, g);
  const float* fb[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) fb[f] = a.f[f] + (long)b * g.T * g.S;

  // prologue: frames ra-TLO .. ra+THI straight into their slots
  for (int fr = ra - P::TLO; fr <= ra + P::THI; ++fr) {
#pragma unroll
    for (int f = 0; f < P::NF; ++f) {
      float v[ST::EPT];
      st.fetch(v, fb[f] + (long)fr * g.S);
      st.store(v, s_f + ((fr % NS) * P::NF + f) * TL::SIZE);
    }
  }
  if (AUX_ON) {
#pragma unroll
    for (int x = 0; x < P::NA; ++x)
      load_tile<TX, P::PER0, P::PER1, P::PER2, NT>(s_x + x * TX::SIZE, a.aux[x] + (long)b * g.S, o0, o1, o2, g, 1.0f);
  }
  __syncthreads();

  RingAcc<P, TL, TX, NS> acc;
  acc.template set_uniforms<false>(a.c, a.sp, b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;

  // the points this thread owns
  int site[PPT], tsite[PPT], pi0[PPT], pi1[PPT], pi2[PPT];
  bool active[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    int l0, l1, l2;
    point_coords<P, TL, NT>(p, l0, l1, l2);
    pi0[p] = o0 + l0; pi1[p] = o1 + l1; pi2[p] = o2 + l2;
    site[p] = TL::at(l0, l1, l2);
    tsite[p] = TX::at(l0, l1, l2);
    active[p] = pi0[p] < g.N0 && pi1[p] < g.N1 && pi2[p] < g.N2 && in_lattice<P>(g, pi0[p], pi1[p], pi2[p]);
  }

  double sum[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) sum[k] = 0.0;

  float wnext[P::NE];      // weights are fetched one step ahead so that their latency hides behind a step
#pragma unroll
  for (int k = 0; k < P::NE; ++k) wnext[k] = ldc(a.wt, k * g.TR + (ra - P::TLO));

  for (int tau = ra; tau < rb; ++tau) {
    // fetch the frame the NEXT step adds while this step computes; it replaces the oldest frame afterwards
    const bool more = tau + 1 < rb;
    const int fnew = tau + P::THI + 1;
    float stage[P::NF][ST::EPT];
    if (more) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) st.fetch(stage[f], fb[f] + (long)fnew * g.S);
    }

    float wk[P::NE];
#pragma unroll
    for (int k = 0; k < P::NE; ++k) {
      wk[k] = wnext[k];
      wnext[k] = ldc(a.wt, k * g.TR + min(tau + 1 - P::TLO, g.TR - 1));
    }
    // every owned point is evaluated (all LDS addresses are valid); points outside the lattice add zero
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      acc.i0 = pi0[p]; acc.i1 = pi1[p]; acc.i2 = pi2[p];
#pragma unroll
      for (int i = 0; i < NS; ++i) acc.fr[i] = s_f + (((tau - P::TLO + i) % NS) * P::NF) * TL::SIZE + site[p];
      acc.ldsx = s_x + tsite[p];
      float R[P::NE];
      P::residual(acc, R);
#pragma unroll
      for (int k = 0; k < P::NE; ++k) {
        const float r = active[p] ? R[k] : 0.0f;
        sum[k] += (double)(r * r) * (double)wk[k];
      }
    }
    if (more) {
      __syncthreads();   // every wave is done with the oldest frame
#pragma unroll
      for (int f = 0; f < P::NF; ++f) st.store(stage[f], s_f + ((fnew % NS) * P::NF + f) * TL::SIZE);
      __syncthreads();
    }
  }
  __syncthreads();
  block_sum_store<P::NE, NT>(sum, s_red, a.partial + (long)w * P::NE);
}


// ------------------------------------------------------------------------------------------------
// backward, marching with recomputation: workgroup = (sample, chunk of output frames, tile)
//
// For central-in-time residuals whose other-frame references are pointwise.  At step n the ring
// holds F(n-1), F(n), F(n+1) with a halo wide enough to evaluate the residual on the tile plus the
// adjoint's reach; G(n) = coef * R(n) is recomputed there into LDS, the spatial part of the adjoint of
// frame n is gathered from it, and the gradient of frame n-1 is completed with the time part, which
// needs G(n-2) and G(n) at the point itself (kept in registers).  Nothing but the fields is read and
// every gradient element is written once.
// ------------------------------------------------------------------------------------------------
template <class P, class TF, class TG, class TX>
struct MarchAdjAcc : SiteBase<P> {
  const float* ldsg;         // &G tile[site] of the frame being gathered
  const float* ldsf;         // &F tile[site] of the same frame
  const float* ldsx;
  float gm[MAXE], gp[MAXE];  // G at this point on the previous / next residual frame
  __device__ __forceinline__ float G(int k, int ot, int o0, int o1, int o2) const {
    if (ot == 0) return ldsg[k * TG::SIZE + o0 * TG::S0 + o1 * TG::S1 + o2];
    return ot < 0 ? gm[k] : gp[k];
  }
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    return ldsf[f * TF::SIZE + o0 * TF::S0 + o1 * TF::S1 + o2];
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    return ldsx[x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + o2];
  }
};

template <class P>
struct MarchOK {
  static constexpr bool value = P::TLO == 1 && P::THI == 1 && (P::RT0 | P::RT1 | P::RT2) == 0 &&
                                (P::GT0 | P::GT1 | P::GT2) == 0 && P::ATLO == 0 && P::ATHI == 0 &&
                                P::GTLO <= 1 && P::GTHI <= 1;
};

#ifndef PDE_AMD_LDS_BUDGET
#define PDE_AMD_LDS_BUDGET (96 * 1024)
#endif
constexpr int LDS_BUDGET = PDE_AMD_LDS_BUDGET;   // keep at least one other workgroup's worth of LDS free per CU

template <class P>
struct BwdMarchFits {
  static constexpr int HB0 = cmax(P::GH0 + P::RH0, P::AH0), HB1 = cmax(P::GH1 + P::RH1, P::AH1), HB2 = cmax(P::GH2 + P::RH2, P::AH2);
  using TF = Tile<P::BT0, P::BT1, P::BT2, HB0, HB1, HB2>;
  using TG = Tile<P::BT0, P::BT1, P::BT2, P::GH0, P::GH1, P::GH2>;
  static constexpr long BYTES = 4L * (3 * P::NF * TF::SIZE + P::NE * TG::SIZE);
  static constexpr bool value = MarchOK<P>::value && BYTES <= LDS_BUDGET;
};

template <class P>
struct FwdMarchFits {
  using TL = Tile<P::FT0, P::FT1, P::FT2, cmax(P::RH0, P::RT0), cmax(P::RH1, P::RT1), cmax(P::RH2, P::RT2)>;
  static constexpr long BYTES = 4L * (P::TLO + P::THI + 1) * P::NF * TL::SIZE;
  static constexpr bool value = BYTES <= 64 * 1024;
};

template <class P>
__global__ __launch_bounds__(P::NTM, P::WPS) void bwd_march_kernel(const Args a) {
  constexpr int HB0 = cmax(P::GH0 + P::RH0, P::AH0), HB1 = cmax(P::GH1 + P::RH1, P::AH1), HB2 = cmax(P::GH2 + P::RH2, P::AH2);
  using TF = Tile<P::BT0, P::BT1, P::BT2, HB0, HB1, HB2>;
  using TG = Tile<P::BT0, P::BT1, P::BT2, P::GH0, P::GH1, P::GH2>;
  using TX = Tile<P::BT0, P::BT1, P::BT2, cmax(P::XH0, P::GH0), cmax(P::XH1, P::GH1), cmax(P::XH2, P::GH2)>;
  constexpr bool AUX_ON = P::NA > 0;
  constexpr int NT = P::NTM;
  constexpr int PPT = (TG::T0 * TG::T1 * TG::T2) / NT;
  constexpr int RPT = (TG::SIZE + NT - 1) / NT;      // region sites (tile + adjoint reach) per thread
  using ST = Stager<TF, P::PER0, P::PER1, P::PER2, NT>;
  static_assert(MarchOK<P>::value, "PDE not eligible for the marching adjoint");
  static_assert(PPT * NT == TG::T0 * TG::T1 * TG::T2, "tile must be a multiple of the workgroup");
  static_assert(NT % TG::T2 == 0, "a wave must lie along the fastest axis");

  __shared__ float s_f[3 * P::NF * TF::SIZE];
  __shared__ float s_g[P::NE * TG::SIZE];
  __shared__ float s_x[AUX_ON ? P::NA * TX::SIZE : 1];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int nchunk = (g.T + a.chunk - 1) / a.chunk;
  const int tile = w % ntile, rest = w / ntile;
  const int ch = rest % nchunk, b = rest / nchunk;
  const int qa = ch * a.chunk, qb = min(qa + a.chunk, g.T);     // output frames [qa, qb)
  const int o2 = (tile % g.nt2) * TG::T2, o1 = ((tile / g.nt2) % g.nt1) * TG::T1, o0 = (tile / (g.nt2 * g.nt1)) * TG::T0;

  ST st;
  st.init(o0, o1, o2, g);
  const float* fb[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) fb[f] = a.f[f] + (long)b * g.T * g.S;
  float up[P::NE];
  upstream_of<P>(a, b, up);

  // residual frames whose G this chunk needs: n in [qa-1, qb] clipped to the lattice [1, T-1)
  const int na = max(qa - 1, 1), nb = min(qb, g.T - 2);       // inclusive range of frames with non-zero G
  // frames to have in the ring before the first step (n = qa-1): n-1, n, n+1  -> clipped to [0, T)
  const int n0 = qa - 1;
  for (int fr = max(n0 - 1, 0); fr <= min(n0 + 1, g.T - 1); ++fr) {
#pragma unroll
    for (int f = 0; f < P::NF; ++f) {
      float v[ST::EPT];
      st.fetch(v, fb[f] + (long)fr * g.S);
      st.store(v, s_f + ((fr % 3) * P::NF + f) * TF::SIZE);
    }
  }
  if (AUX_ON) {
#pragma unroll
    for (int x = 0; x < P::NA; ++x)
      load_tile<TX, P::PER0, P::PER1, P::PER2, NT>(s_x + x * TX::SIZE, a.aux[x] + (long)b * g.S, o0, o1, o2, g, 1.0f);
  }

  // region sites of this thread: LDS offsets into the F tile / G tile, lattice flag, grid coordinates
  int rF[RPT], rG[RPT], rX[RPT], r0[RPT], r1[RPT], r2[RPT];
  bool rok[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int iraw = threadIdx.x + j * NT;
    const int i = iraw < TG::SIZE ? iraw : 0;     // surplus entries of the last round alias site 0 and store nothing
    const int l2 = i % TG::D2 - TG::H2, rr = i / TG::D2, l1 = rr % TG::D1 - TG::H1, l0 = rr / TG::D1 - TG::H0;
    bool ok = iraw < TG::SIZE;
    const int c0 = wrap_coord<P::PER0>(o0 + l0, g.N0, ok);
    const int c1 = wrap_coord<P::PER1>(o1 + l1, g.N1, ok);
    const int c2 = wrap_coord<P::PER2>(o2 + l2, g.N2, ok);
    r0[j] = c0; r1[j] = c1; r2[j] = c2;
    rok[j] = ok && in_lattice<P>(g, c0, c1, c2);
    rG[j] = i;
    rF[j] = TF::at(l0, l1, l2);
    rX[j] = TX::at(l0, l1, l2);
  }
  // output points of this thread
  int sF[PPT], sG[PPT], sX[PPT], goff[PPT], pi0[PPT], pi1[PPT], pi2[PPT];
  bool inside[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    int l0, l1, l2;
    point_coords<P, TG, NT>(p, l0, l1, l2);
    pi0[p] = o0 + l0; pi1[p] = o1 + l1; pi2[p] = o2 + l2;
    inside[p] = pi0[p] < g.N0 && pi1[p] < g.N1 && pi2[p] < g.N2;
    sF[p] = TF::at(l0, l1, l2); sG[p] = TG::at(l0, l1, l2); sX[p] = TX::at(l0, l1, l2);
    goff[p] = (pi0[p] * g.N1 + pi1[p]) * g.N2 + pi2[p];
  }

  RingAcc<P, TF, TX, 3> racc;
  racc.template set_uniforms<false>(a.c, a.sp, b);
  MarchAdjAcc<P, TF, TG, TX> aacc;
  aacc.template set_uniforms<false>(a.c, a.sp, b);
  racc.N0 = aacc.N0 = g.N0; racc.N1 = aacc.N1 = g.N1; racc.N2 = aacc.N2 = g.N2;

  float Sprev[PPT][P::NF];     // spatial adjoint of frame n-1
  float Gm2[PPT][P::NE];       // G at the point, frame n-2
  float Gm1[PPT][P::NE];       // G at the point, frame n-1
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
#pragma unroll
    for (int f = 0; f < P::NF; ++f) Sprev[p][f] = 0.0f;
#pragma unroll
    for (int k = 0; k < P::NE; ++k) { Gm2[p][k] = 0.0f; Gm1[p][k] = 0.0f; }
  }
  __syncthreads();

  float wnext[P::NE];      // weights are fetched one step ahead so that their latency hides behind a step
#pragma unroll
  for (int k = 0; k < P::NE; ++k) wnext[k] = ldc(a.wt, k * g.TR + min(max(n0 - 1, 0), g.TR - 1));

  for (int n = n0; n <= qb; ++n) {
    // prefetch F(n+2) for the next step
    const int fnew = n + 2;
    const bool more = n < qb && fnew < g.T;
    float stage[P::NF][ST::EPT];
    if (more) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) st.fetch(stage[f], fb[f] + (long)fnew * g.S);
    }

    // G(n) on the tile plus the adjoint's reach
    const bool gvalid = n >= na && n <= nb;
    float cf[P::NE];
#pragma unroll
    for (int k = 0; k < P::NE; ++k) {
      cf[k] = gvalid ? up[k] * wnext[k] : 0.0f;
      wnext[k] = ldc(a.wt, k * g.TR + min(max(n, 0), g.TR - 1));
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      float R[P::NE];
#pragma unroll
      for (int k = 0; k < P::NE; ++k) R[k] = 0.0f;
      // uniform in gvalid; in the last round only the waves that still own sites take part
      if (gvalid && (RPT * NT == TG::SIZE || j < RPT - 1 || threadIdx.x + j * NT < TG::SIZE)) {
        racc.i0 = r0[j]; racc.i1 = r1[j]; racc.i2 = r2[j];
#pragma unroll
        for (int i = 0; i < 3; ++i) racc.fr[i] = s_f + (((n + 2 + i) % 3) * P::NF) * TF::SIZE + rF[j];
        racc.ldsx = s_x + rX[j];
        P::residual(racc, R);
      }
      if (RPT * NT == TG::SIZE || threadIdx.x + j * NT < TG::SIZE) {
#pragma unroll
        for (int k = 0; k < P::NE; ++k) s_g[k * TG::SIZE + rG[j]] = rok[j] ? cf[k] * R[k] : 0.0f;
      }
    }
    __syncthreads();

    const bool do_space = n >= qa && n < qb;        // frame n is an output frame of this chunk
    const bool do_out = n - 1 >= qa && n - 1 < qb;  // frame n-1 gets completed now
    const float* fcur = s_f + (((n + 3) % 3) * P::NF) * TF::SIZE;
    const float* fprev = s_f + (((n + 2) % 3) * P::NF) * TF::SIZE;   // slot of frame n-1
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      float Gc[P::NE];
#pragma unroll
      for (int k = 0; k < P::NE; ++k) Gc[k] = s_g[k * TG::SIZE + sG[p]];
      aacc.i0 = pi0[p]; aacc.i1 = pi1[p]; aacc.i2 = pi2[p];
      aacc.ldsg = s_g + sG[p]; aacc.ldsx = s_x + sX[p];
      if (do_out) {      // uniform
        aacc.ldsf = fprev + sF[p];
#pragma unroll
        for (int k = 0; k < P::NE; ++k) { aacc.gm[k] = Gm2[p][k]; aacc.gp[k] = Gc[k]; }
        float gt[P::NF];
        P::adjoint_time(aacc, gt);
        if (inside[p]) {
          const long o = (long)(n - 1) * g.S + goff[p];
#pragma unroll
          for (int f = 0; f < P::NF; ++f) __builtin_nontemporal_store(Sprev[p][f] + gt[f], &a.g[f][(long)b * g.T * g.S + o]);
        }
      }
      float gs[P::NF];
#pragma unroll
      for (int f = 0; f < P::NF; ++f) gs[f] = 0.0f;
      if (do_space) {    // uniform
        aacc.ldsf = fcur + sF[p];
        P::adjoint_space(aacc, gs);
      }
#pragma unroll
      for (int f = 0; f < P::NF; ++f) Sprev[p][f] = gs[f];
#pragma unroll
      for (int k = 0; k < P::NE; ++k) { Gm2[p][k] = Gm1[p][k]; Gm1[p][k] = Gc[k]; }
    }
    __syncthreads();     // gathers are done with s_g and with the slot of frame n-1
    if (more) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) st.store(stage[f], s_f + ((fnew % 3) * P::NF + f) * TF::SIZE);
      __syncthreads();
    }
  }
}

}  // namespace pde_amd
