// Plane-marching kernels for 3-D PDEs whose workgroup owns SEVERAL FRAMES of a (y, x) tile ("z columns").
//
// The plane-marching kernels of stencil_zmarch.hpp give every frame a workgroup of its own, so a frame is read by the
// workgroups of its two neighbours as well (the time taps), and those reads miss the XCD's L2 more often than not: a
// frame is fetched up to three times.  Here a workgroup marches a (y, x) tile through a chunk of z planes for a whole
// group of residual frames at once: a set of waves per frame, all at the same plane, one barrier per plane.
//   * the time taps of a frame are the plane images of its neighbours in LDS (the group's two outer frames, which have
//     no waves of their own, are staged by the waves of the first and last frame);
//   * the taps along z come out of the lane's own registers (a column of 2 * reach + 1 planes of its cell), so LDS holds
//     ONE plane per frame and field (twice: the image of plane k + 1 is written while plane k is read) instead of a ring;
//   * every plane of every frame is fetched once per (y, x) tile.
// Requirements (ZColCfg): a 3-D PDE, periodic on every axis, without auxiliary fields, whose z taps are pure (no (y, x)
// offset on a tap with a z offset), whose time taps are pointwise one frame back and ahead, and (adjoint) whose field
// reads are at the output point.  Everything else keeps the kernels of stencil_zmarch.hpp / stencil_quad.hpp.
#pragma once
#include "stencil_zmarch.hpp"

namespace pde_amd {

template <class P, int W>
struct ZColCfg {
  static constexpr bool ON = P::ND == 3 && P::ZCOL != 0 && P::ZPURE != 0 && P::NA == 0 && P::TLO == 1 && P::THI == 1 &&
                             (P::RT0 | P::RT1 | P::RT2) == 0 && P::PER0 && P::PER1 && P::PER2;
  static constexpr int T1 = ON ? P::ZCT1 : 8, T2 = ON ? P::ZCT2 : 32;
  static constexpr int QR = T2 / W;              // cells (W points) per tile row
  static constexpr int LPF = T1 * QR;            // lanes per frame: one cell each
#ifndef PDE_AMD_ZCOL_MAXTF
#define PDE_AMD_ZCOL_MAXTF 512
#endif
#ifndef PDE_AMD_ZCOL_MAXTB
#define PDE_AMD_ZCOL_MAXTB 512
#endif
  // lanes per workgroup at most: it bounds the registers a lane may use (512 lanes: 256, 768: 168, 1024: 128)
  static constexpr int MAXTF = PDE_AMD_ZCOL_MAXTF, MAXTB = PDE_AMD_ZCOL_MAXTB;
  static constexpr int FPWF = MAXTF / LPF, FPWB = MAXTB / LPF;      // residual frames per workgroup at most
#ifndef PDE_AMD_ZCOL_WPSF
#define PDE_AMD_ZCOL_WPSF 1
#endif
#ifndef PDE_AMD_ZCOL_WPSB
#define PDE_AMD_ZCOL_WPSB 1
#endif
  static constexpr int WPSF = PDE_AMD_ZCOL_WPSF, WPSB = PDE_AMD_ZCOL_WPSB;      // waves per SIMD to compile for
  using TP = Tile<1, T1, T2, 0, P::RH1, padw<W>(P::RH2)>;      // plane image of a field: the residual's (y, x) halo
  using TG = Tile<1, T1, T2, 0, P::GH1, padw<W>(P::GH2)>;      // plane image of a scaled residual: the adjoint's halo
  static constexpr bool SHAPE_OK = T2 % W == 0 && LPF % 64 == 0 && LPF <= 512;
  static constexpr bool FWD_OK = ON && SHAPE_OK;
#ifdef PDE_AMD_ZCOL_NOBWD      // diagnostic builds: the adjoint keeps the other kernels
  static constexpr bool BWD_OK = false && (P::GT0 | P::GT1 | P::GT2) == 0 && P::GTLO == 1 && P::GTHI == 1 &&
                                 P::ATLO == 0 && P::ATHI == 0 && (P::AH0 | P::AH1 | P::AH2) == 0 &&
                                 (P::AT0 | P::AT1 | P::AT2) == 0 && P::TIME_AF == 0;
#else
  static constexpr bool BWD_OK = ON && SHAPE_OK && (P::GT0 | P::GT1 | P::GT2) == 0 && P::GTLO == 1 && P::GTHI == 1 &&
                                 P::ATLO == 0 && P::ATHI == 0 && (P::AH0 | P::AH1 | P::AH2) == 0 &&
                                 (P::AT0 | P::AT1 | P::AT2) == 0 && P::TIME_AF == 0;
#endif
  // dynamic LDS of a workgroup of nfw frames: the block reduction's slots, then 2 buffers x (nfw + 2) frame slots
  static constexpr size_t RED_BYTES = 16 * MAXE * sizeof(double);
  static size_t fwd_lds(int nfw) { return RED_BYTES + 2ul * (nfw + 2) * P::NF * TP::SIZE * sizeof(float); }
  static size_t bwd_lds(int nfw) { return RED_BYTES + 2ul * (nfw + 2) * P::NE * TG::SIZE * sizeof(float); }
};

// The cells of a plane image that are not the tile's own: rows above and below (whole image rows), then the side cells
// of the tile's rows.  Dealt to the lanes of a frame; offsets computed once per workgroup, periodic wrap applied.
template <class TL, int W, int LPF>
struct HaloCells {
  static constexpr int QD = TL::D2 / W, HQ = TL::H2 / W, QT = TL::T2 / W;
  static constexpr int NH = 2 * TL::H1 * QD + TL::T1 * 2 * HQ;
  static constexpr int EPT = NH > 0 ? (NH + LPF - 1) / LPF : 1;
  unsigned goff[EPT];      // element offset inside a plane
  int lds[EPT];            // float offset inside the plane image; < 0: this lane has no cell j
  __device__ __forceinline__ void init(int lid, int o1, int o2, const Geom& g) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = lid + j * LPF;
      int l1, lq;
      if (i < TL::H1 * QD) { l1 = i / QD; lq = i % QD; }
      else if (i < 2 * TL::H1 * QD) { const int r = i - TL::H1 * QD; l1 = TL::H1 + TL::T1 + r / QD; lq = r % QD; }
      else { const int r = i - 2 * TL::H1 * QD, c = HQ > 0 ? r % (2 * HQ) : 0; l1 = TL::H1 + (HQ > 0 ? r / (2 * HQ) : 0); lq = c < HQ ? c : QT + c; }
      bool ok = true;
      const int c1 = wrap_coord<true>(o1 + l1 - TL::H1, g.N1, ok);
      const int c2 = wrap_quad<true, W>(o2 + W * lq - TL::H2, g.N2, ok);
      const bool have = NH > 0 && i < NH;
      goff[j] = have ? (unsigned)(c1 * g.N2 + c2) : 0u;
      lds[j] = have ? l1 * TL::S1 + W * lq : -1;
    }
  }
};

// fp64 block sum over a workgroup whose wave count is only known at run time (at most 16 waves)
template <int NE>
__device__ __forceinline__ void block_sum_store_rt(const double (&acc)[NE], double* red, double* out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[wv * NE + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NE) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[w * NE + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// Work item -> (sample, group of residual frames, chunk of planes, (y, x) tile); tiles fastest, so that the workgroups an
// XCD runs side by side are neighbouring tiles of one chunk, whose halos meet in its L2.
struct ZColItem {
  int b, grp, zc, tile;
  __device__ __forceinline__ ZColItem(int w, const Geom& g, int ngroups) {
    const int ntile = g.nt1 * g.nt2;
    tile = w % ntile;
    int r = w / ntile;
    zc = r % g.nt0; r /= g.nt0;
    grp = r % ngroups; b = r / ngroups;
  }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <class P, class TP, int W>
struct ZColResAcc : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  static constexpr int NZ = 2 * P::RH0 + 1;
  const float* site;         // &image[this frame's slot][field 0][own row][own cell] in the buffer being read
  fv col[P::NF][NZ];         // the lane's cell at planes k - RH0 .. k + RH0
  int px;
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    if (ot != 0) {           // the neighbouring frame's image of the same plane, at the own cell
      const fv v = *reinterpret_cast<const fv*>(site + (ot * P::NF + f) * TP::SIZE);
      return v[px];
    }
    if (o0 != 0) return col[f][o0 + P::RH0][px];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    if (o1 == 0 && qd == 0) return col[f][P::RH0][e];
    const fv v = *reinterpret_cast<const fv*>(site + f * TP::SIZE + o1 * TP::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float X(int, int, int, int) const { return 0.0f; }
  __device__ __forceinline__ float G(int, int, int, int, int) const { return 0.0f; }
};

// launched with LPF * nfw threads and ZColCfg::fwd_lds(nfw) bytes of dynamic LDS
template <class P, int W, bool WRITE_R>
__global__ __launch_bounds__((ZColCfg<P, W>::MAXTF), (ZColCfg<P, W>::WPSF)) void fwd_zcol_kernel(const Args a, const int nfw) {
  using C = ZColCfg<P, W>;
  using TP = typename C::TP;
  using fv = typename VecOf<W>::type;
  using HC = HaloCells<TP, W, C::LPF>;
  constexpr int LPF = C::LPF, RH0 = P::RH0, NZ = 2 * RH0 + 1, NF = P::NF, NE = P::NE, QR = C::QR, EPT = HC::EPT;
  static_assert(C::FWD_OK, "PDE not eligible for the z-column forward");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  double* s_red = reinterpret_cast<double*>(s_raw);
  float* s_img = reinterpret_cast<float*>(s_raw + C::RED_BYTES);

  const Geom& g = a.geo;      // nt1, nt2: (y, x) tiles; nt0: chunks of planes; a.chunk: planes per chunk
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ngroups = (g.TR + nfw - 1) / nfw;
  const ZColItem it(w, g, ngroups);
  const int fw = threadIdx.x / LPF, lid = threadIdx.x % LPF;      // fw is wave-uniform: LPF is a multiple of 64
  const int nlive = min(nfw, g.TR - it.grp * nfw);                // frames of this group (the last group may be short)
  const bool live = fw < nlive;
  const int tr = it.grp * nfw + min(fw, nlive - 1), t = tr + 1;
  const int o2 = (it.tile % g.nt2) * TP::T2, o1 = (it.tile / g.nt2) * TP::T1;
  const int za = it.zc * a.chunk, zb = min(za + a.chunk, g.N0);
  const int plane = g.N1 * g.N2;
  const int bufsz = (nfw + 2) * NF * TP::SIZE;
  const bool first = live && fw == 0, last = live && fw == nlive - 1;      // these waves also stage the group's outer frames

  // own cell
  const int row = lid / QR, l2 = W * (lid % QR);
  const int pi1 = o1 + row, pi2 = o2 + l2;
  const bool inside = pi1 < g.N1 && pi2 < g.N2;
  const bool rowok = inside && pi1 >= g.lo1 && pi1 < g.N1 - g.hi1;
  unsigned ooff;      // element offset of the cell inside a plane; cells past the grid's end hold the periodic image
  {
    bool ok = true;
    const int c1 = wrap_coord<true>(pi1, g.N1, ok), c2 = wrap_quad<true, W>(pi2, g.N2, ok);
    ooff = (unsigned)(c1 * g.N2 + c2);
  }
  const int olds = TP::at(0, row, l2);
  HC hc;
  hc.init(lid, o1, o2, g);

  const float* fb[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) fb[f] = a.f[f] + ((long)it.b * g.T + t) * g.S;

  ZColResAcc<P, TP, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, it.b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  acc.i1 = pi1;
  float wk[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) wk[k] = ldc(a.wt, k * g.TR + tr);
  double sum[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) sum[k] = 0.0;

  // the column of the first plane, its halo cells and the outer frames' cells: all in flight together
  fv hal[NF][EPT], elo[NF], ehi[NF], nxt[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    elo[f] = ehi[f] = nxt[f] = vec_zero<W>();
#pragma unroll
    for (int j = 0; j < EPT; ++j) hal[f][j] = vec_zero<W>();
#pragma unroll
    for (int i = 0; i < NZ; ++i) acc.col[f][i] = vec_zero<W>();
  }
  PlaneCursor<true> cur;      // the plane that enters the column next
  cur.init(za + RH0 + 1, g.N0, plane);
  if (live) {
    bool on;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const int pz = z_plane<true>(za - RH0 + i, g.N0, on);
#pragma unroll
      for (int f = 0; f < NF; ++f) acc.col[f][i] = *reinterpret_cast<const fv*>(fb[f] + (long)pz * plane + ooff);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) hal[f][j] = *reinterpret_cast<const fv*>(fb[f] + (long)za * plane + hc.goff[j]);
    }
    if (first) {
#pragma unroll
      for (int f = 0; f < NF; ++f) elo[f] = *reinterpret_cast<const fv*>(fb[f] - g.S + (long)za * plane + ooff);
    }
    if (last) {
#pragma unroll
      for (int f = 0; f < NF; ++f) ehi[f] = *reinterpret_cast<const fv*>(fb[f] + g.S + (long)za * plane + ooff);
    }
  }
  long ko = (long)za * plane;          // element offset of plane k inside a frame
  const long rframe = ((long)it.b * g.TR + tr) * g.S;

  // The residuals of plane k are stored at the top of step k + 1, ahead of that step's loads: the memory counter is in
  // order, so stores issued after a step's loads would have to complete before the next step may use what it loaded.
  fv rst[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) rst[e] = vec_zero<W>();
  const long rown = rframe + (long)pi1 * g.N2 + pi2;

  for (int k = za; k < zb; ++k) {
    float* img = s_img + ((k - za) & 1) * bufsz;
    float* mine = img + (fw + 1) * NF * TP::SIZE;
    if (live) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        *reinterpret_cast<fv*>(mine + f * TP::SIZE + olds) = acc.col[f][RH0];
#pragma unroll
        for (int j = 0; j < EPT; ++j)
          if (hc.lds[j] >= 0) *reinterpret_cast<fv*>(mine + f * TP::SIZE + hc.lds[j]) = hal[f][j];
      }
      if (first) {
#pragma unroll
        for (int f = 0; f < NF; ++f) *reinterpret_cast<fv*>(img + f * TP::SIZE + olds) = elo[f];
      }
      if (last) {
#pragma unroll
        for (int f = 0; f < NF; ++f) *reinterpret_cast<fv*>(mine + (NF + f) * TP::SIZE + olds) = ehi[f];
      }
      if (WRITE_R && inside && k > za) {
#pragma unroll
        for (int e = 0; e < NE; ++e) *reinterpret_cast<fv*>(a.rout[e] + rown + ko - plane) = rst[e];
      }
    }
    __syncthreads();
    const bool more = k + 1 < zb;
    if (live && more) {      // plane k + 1: the column's next entry, the halo cells, the outer frames; in flight during the step
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        nxt[f] = *reinterpret_cast<const fv*>(fb[f] + cur.off + ooff);
#pragma unroll
        for (int j = 0; j < EPT; ++j) hal[f][j] = *reinterpret_cast<const fv*>(fb[f] + ko + plane + hc.goff[j]);
      }
      if (first) {
#pragma unroll
        for (int f = 0; f < NF; ++f) elo[f] = *reinterpret_cast<const fv*>(fb[f] - g.S + ko + plane + ooff);
      }
      if (last) {
#pragma unroll
        for (int f = 0; f < NF; ++f) ehi[f] = *reinterpret_cast<const fv*>(fb[f] + g.S + ko + plane + ooff);
      }
    }
    cur.advance(g.N0, plane);
    if (live) {
      acc.i0 = k;
      acc.site = mine + olds;
      const bool zok = k >= g.lo0 && k < g.N0 - g.hi0;
      float Rq[W][NE];
#pragma unroll
      for (int px = 0; px < W; ++px) {
        acc.px = px; acc.i2 = pi2 + px;
        P::residual(acc, Rq[px]);
        const bool ok = zok && rowok && acc.i2 >= g.lo2 && acc.i2 < g.N2 - g.hi2;
#pragma unroll
        for (int e = 0; e < NE; ++e) Rq[px][e] = ok ? Rq[px][e] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        float part = 0.0f;
#pragma unroll
        for (int px = 0; px < W; ++px) part += Rq[px][e] * Rq[px][e];
        sum[e] += (double)part * (double)wk[e];      // fp32 over the cell, fp64 across planes
      }
      if (WRITE_R) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
          for (int px = 0; px < W; ++px) rst[e][px] = Rq[px][e];
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int i = 0; i + 1 < NZ; ++i) acc.col[f][i] = acc.col[f][i + 1];
      acc.col[f][NZ - 1] = nxt[f];
    }
    ko += plane;
  }
  if (WRITE_R && live && inside && zb > za) {
#pragma unroll
    for (int e = 0; e < NE; ++e) *reinterpret_cast<fv*>(a.rout[e] + rown + ko - plane) = rst[e];
  }
  if (!live) {
#pragma unroll
    for (int k = 0; k < NE; ++k) sum[k] = 0.0;
  }
  __syncthreads();
  block_sum_store_rt<NE>(sum, s_red, a.partial + (long)w * NE);
}

// ------------------------------------------------------------------------------------------------
// adjoint from stored residuals
// ------------------------------------------------------------------------------------------------
template <class P, class TG, int W>
struct ZColAdjAcc : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  static constexpr int NZ = 2 * P::GH0 + 1;
  const float* site;         // &image[this frame's slot][equation 0][own row][own cell]
  fv gcol[P::NE][NZ];        // the cell's scaled residuals at planes k - GH0 .. k + GH0
  fv fq[P::NF];              // the cell's fields at plane k
  int px;
  __device__ __forceinline__ float G(int k, int ot, int o0, int o1, int o2) const {
    if (ot != 0) {
      const fv v = *reinterpret_cast<const fv*>(site + (ot * P::NE + k) * TG::SIZE);
      return v[px];
    }
    if (o0 != 0) return gcol[k][o0 + P::GH0][px];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    if (o1 == 0 && qd == 0) return gcol[k][P::GH0][e];
    const fv v = *reinterpret_cast<const fv*>(site + k * TG::SIZE + o1 * TG::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float F(int f, int, int, int, int) const { return fq[f][px]; }
  __device__ __forceinline__ float X(int, int, int, int) const { return 0.0f; }
};

// The clip's first and last frame carry no residual: their gradient is the time share of the one residual frame next to
// them, which is the cell the lane holds in registers.
template <class P, int W>
struct ZColEdgeAcc : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  fv gq[P::NE];
  int side;                  // +1: the residual frame lies one frame ahead of the output frame; -1: one frame back
  int px;
  __device__ __forceinline__ float G(int k, int ot, int, int, int) const { return ot == side ? gq[k][px] : 0.0f; }
  __device__ __forceinline__ float F(int, int, int, int, int) const { return 0.0f; }
  __device__ __forceinline__ float X(int, int, int, int) const { return 0.0f; }
};

// launched with LPF * nfw threads and ZColCfg::bwd_lds(nfw) bytes of dynamic LDS; waves <-> residual frames, the waves of
// residual frame tr write the gradient of frame tr + 1 (and of the clip's first / last frame where tr is the first / last)
// MULTI: the clip is split into several frame groups (a group then also stages the scaled residuals of the residual
// frames next to it, which belong to other groups); the usual clips are one group and spare those registers
template <class P, int W, bool MULTI>
__global__ __launch_bounds__((ZColCfg<P, W>::MAXTB), (ZColCfg<P, W>::WPSB)) void bwd_stored_zcol_kernel(const Args a, const int nfw) {
  using C = ZColCfg<P, W>;
  using TG = typename C::TG;
  using fv = typename VecOf<W>::type;
  using HC = HaloCells<TG, W, C::LPF>;
  constexpr int LPF = C::LPF, GH0 = P::GH0, NZ = 2 * GH0 + 1, NF = P::NF, NE = P::NE, QR = C::QR, EPT = HC::EPT;
  constexpr bool AF = P::ANY_AF != 0;
  static_assert(C::BWD_OK, "PDE not eligible for the z-column stored-residual adjoint");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  float* s_img = reinterpret_cast<float*>(s_raw + C::RED_BYTES);

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ngroups = (g.TR + nfw - 1) / nfw;
  const ZColItem it(w, g, ngroups);
  const int fw = threadIdx.x / LPF, lid = threadIdx.x % LPF;
  const int nlive = min(nfw, g.TR - it.grp * nfw);
  const bool live = fw < nlive;
  const int tr = it.grp * nfw + min(fw, nlive - 1), t = tr + 1;
  const int o2 = (it.tile % g.nt2) * TG::T2, o1 = (it.tile / g.nt2) * TG::T1;
  const int za = it.zc * a.chunk, zb = min(za + a.chunk, g.N0);
  const int plane = g.N1 * g.N2;
  const int bufsz = (nfw + 2) * NE * TG::SIZE;
  const bool first = live && fw == 0, last = live && fw == nlive - 1;
  const bool has_lo = MULTI && first && tr > 0, has_hi = MULTI && last && tr + 1 < g.TR;      // a residual frame of another group next to this one

  const int row = lid / QR, l2 = W * (lid % QR);
  const int pi1 = o1 + row, pi2 = o2 + l2;
  const bool inside = pi1 < g.N1 && pi2 < g.N2;
  unsigned ooff;
  {
    bool ok = true;
    const int c1 = wrap_coord<true>(pi1, g.N1, ok), c2 = wrap_quad<true, W>(pi2, g.N2, ok);
    ooff = (unsigned)(c1 * g.N2 + c2);
  }
  const int olds = TG::at(0, row, l2);
  HC hc;
  hc.init(lid, o1, o2, g);

  const float* rp[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) rp[k] = a.r[k] + ((long)it.b * g.TR + tr) * g.S;
  const float* fb[NF];
  float* gb[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) { fb[f] = a.f[f] + ((long)it.b * g.T + t) * g.S; gb[f] = a.g[f] + ((long)it.b * g.T + t) * g.S; }

  float up[NE];
  upstream_of<P>(a, it.b, up);
  float cf[NE], cflo[NE], cfhi[NE];      // coefficients of this residual frame and of the group's outer residual frames
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    cf[k] = up[k] * ldc(a.wt, k * g.TR + tr);
    cflo[k] = up[k] * ldc(a.wt, k * g.TR + max(tr - 1, 0));
    cfhi[k] = up[k] * ldc(a.wt, k * g.TR + min(tr + 1, g.TR - 1));
  }

  ZColAdjAcc<P, TG, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, it.b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  acc.i1 = pi1;
  ZColEdgeAcc<P, W> eacc;
  eacc.template set_uniforms<true>(a.c, a.sp, it.b);
  eacc.N0 = g.N0; eacc.N1 = g.N1; eacc.N2 = g.N2;
  eacc.i1 = pi1;

  fv hal[NE][EPT], elo[NE], ehi[NE], nxt[NE], fnx[NF];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    elo[e] = ehi[e] = nxt[e] = vec_zero<W>();
#pragma unroll
    for (int j = 0; j < EPT; ++j) hal[e][j] = vec_zero<W>();
#pragma unroll
    for (int i = 0; i < NZ; ++i) acc.gcol[e][i] = vec_zero<W>();
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) fnx[f] = acc.fq[f] = vec_zero<W>();
  PlaneCursor<true> cur;
  cur.init(za + GH0 + 1, g.N0, plane);
  if (live) {
    bool on;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const int pz = z_plane<true>(za - GH0 + i, g.N0, on);
#pragma unroll
      for (int e = 0; e < NE; ++e) acc.gcol[e][i] = *reinterpret_cast<const fv*>(rp[e] + (long)pz * plane + ooff);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) hal[e][j] = *reinterpret_cast<const fv*>(rp[e] + (long)za * plane + hc.goff[j]);
    }
    if (has_lo) {
#pragma unroll
      for (int e = 0; e < NE; ++e) elo[e] = *reinterpret_cast<const fv*>(rp[e] - g.S + (long)za * plane + ooff);
    }
    if (has_hi) {
#pragma unroll
      for (int e = 0; e < NE; ++e) ehi[e] = *reinterpret_cast<const fv*>(rp[e] + g.S + (long)za * plane + ooff);
    }
    if (AF) {
#pragma unroll
      for (int f = 0; f < NF; ++f) fnx[f] = *reinterpret_cast<const fv*>(fb[f] + (long)za * plane + ooff);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
      for (int i = 0; i < NZ; ++i) acc.gcol[e][i] = acc.gcol[e][i] * cf[e];
    }
  }
  long ko = (long)za * plane;
  // gradients of plane k leave at the top of step k + 1, ahead of that step's loads (in-order memory counter: see the
  // forward kernel); the gradients of the clip's first / last frame (two of the clip's waves) leave at once
  fv gst[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) gst[f] = vec_zero<W>();
  const long gown = (long)pi1 * g.N2 + pi2;

  for (int k = za; k < zb; ++k) {
    float* img = s_img + ((k - za) & 1) * bufsz;
    float* mine = img + (fw + 1) * NE * TG::SIZE;
    if (live) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        *reinterpret_cast<fv*>(mine + e * TG::SIZE + olds) = acc.gcol[e][GH0];
#pragma unroll
        for (int j = 0; j < EPT; ++j)
          if (hc.lds[j] >= 0) *reinterpret_cast<fv*>(mine + e * TG::SIZE + hc.lds[j]) = hal[e][j] * cf[e];
      }
      if (first) {      // the slot before the group's first frame: a residual frame of the previous group, or nothing
#pragma unroll
        for (int e = 0; e < NE; ++e) *reinterpret_cast<fv*>(img + e * TG::SIZE + olds) = has_lo ? elo[e] * cflo[e] : vec_zero<W>();
      }
      if (last) {
#pragma unroll
        for (int e = 0; e < NE; ++e) *reinterpret_cast<fv*>(mine + (NE + e) * TG::SIZE + olds) = has_hi ? ehi[e] * cfhi[e] : vec_zero<W>();
      }
      if (inside && k > za) {
#pragma unroll
        for (int f = 0; f < NF; ++f) __builtin_nontemporal_store(gst[f], reinterpret_cast<fv*>(gb[f] + gown + ko - plane));
      }
    }
    if (AF) {
#pragma unroll
      for (int f = 0; f < NF; ++f) acc.fq[f] = fnx[f];
    }
    __syncthreads();
    const bool more = k + 1 < zb;
    if (live && more) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        nxt[e] = *reinterpret_cast<const fv*>(rp[e] + cur.off + ooff);
#pragma unroll
        for (int j = 0; j < EPT; ++j) hal[e][j] = *reinterpret_cast<const fv*>(rp[e] + ko + plane + hc.goff[j]);
      }
      if (has_lo) {
#pragma unroll
        for (int e = 0; e < NE; ++e) elo[e] = *reinterpret_cast<const fv*>(rp[e] - g.S + ko + plane + ooff);
      }
      if (has_hi) {
#pragma unroll
        for (int e = 0; e < NE; ++e) ehi[e] = *reinterpret_cast<const fv*>(rp[e] + g.S + ko + plane + ooff);
      }
      if (AF) {
#pragma unroll
        for (int f = 0; f < NF; ++f) fnx[f] = *reinterpret_cast<const fv*>(fb[f] + ko + plane + ooff);
      }
    }
    cur.advance(g.N0, plane);
    if (live && inside) {
      acc.i0 = k;
      acc.site = mine + olds;
#pragma unroll
      for (int px = 0; px < W; ++px) {
        acc.px = px; acc.i2 = pi2 + px;
        float gr[NF];
        P::adjoint(acc, gr);
#pragma unroll
        for (int f = 0; f < NF; ++f) gst[f][px] = gr[f];
      }
      // the clip's first / last frame: the time share of this residual frame alone
      if (tr == 0 || tr == g.TR - 1) {
        eacc.i0 = k;
#pragma unroll
        for (int e = 0; e < NE; ++e) eacc.gq[e] = acc.gcol[e][GH0];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s == 0 ? tr != 0 : tr != g.TR - 1) continue;
          eacc.side = s == 0 ? 1 : -1;
          fv est[NF];
#pragma unroll
          for (int px = 0; px < W; ++px) {
            eacc.px = px; eacc.i2 = pi2 + px;
            float gr[NF];
            P::adjoint_time(eacc, gr);
#pragma unroll
            for (int f = 0; f < NF; ++f) est[f][px] = gr[f];
          }
          const long shift = s == 0 ? -g.S : g.S;
#pragma unroll
          for (int f = 0; f < NF; ++f) __builtin_nontemporal_store(est[f], reinterpret_cast<fv*>(gb[f] + shift + gown + ko));
        }
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
      for (int i = 0; i + 1 < NZ; ++i) acc.gcol[e][i] = acc.gcol[e][i + 1];
      acc.gcol[e][NZ - 1] = nxt[e] * cf[e];
    }
    ko += plane;
  }
  if (live && inside && zb > za) {
#pragma unroll
    for (int f = 0; f < NF; ++f) __builtin_nontemporal_store(gst[f], reinterpret_cast<fv*>(gb[f] + gown + ko - plane));
  }
}

}  // namespace pde_amd
