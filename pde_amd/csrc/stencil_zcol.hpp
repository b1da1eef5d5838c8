// Plane-marching kernels for 3-D PDEs whose workgroup owns SEVERAL FRAMES of a (y, x) tile ("z columns").
//
// The plane-marching kernels of stencil_zmarch.hpp give every frame a workgroup of its own, so a frame is read by the
// workgroups of its two neighbours as well (the time taps), and those reads miss the XCD's L2 more often than not: a
// frame is fetched up to three times.  Here a workgroup marches a (y, x) tile through a chunk of z planes for a whole
// group of residual frames at once: a set of waves per frame, all at the same plane, one barrier per plane.
//   * the time taps of a frame are the plane images of its neighbours in LDS (the group's two outer frames, which have
//     no waves of their own, are staged by the waves of the group's first and last frame);
//   * the taps along z come out of the lane's own registers (a column of 2 * reach + 1 planes of its cell), so LDS holds
//     ONE plane per frame and field (twice: the image of plane k + 1 is written while plane k is read) instead of a ring;
//   * every plane of every frame is fetched once per (y, x) tile.
// The loop body holds no vector-memory instruction under a branch: lanes that have nothing to load or store pass an
// offset beyond the buffer's size (the load returns zeros, the store is dropped).  The memory counter of a wave is in
// order, and where two paths with different numbers of loads and stores meet, the compiler has to wait for the longer
// path's count: the stores of one step would then have to complete before the next step may use what it loaded.  With
// one path, every wait names exactly the loads it needs and the stores drain behind the next step's arithmetic.
// Requirements (ZColCfg): a 3-D PDE, periodic on every axis, without auxiliary fields, whose z taps are pure (no (y, x)
// offset on a tap with a z offset), whose time taps are pointwise one frame back and ahead, and (adjoint) whose field
// reads are at the output point.  Everything else keeps the kernels of stencil_zmarch.hpp / stencil_quad.hpp.
#pragma once
#include "stencil_zmarch.hpp"

namespace pde_amd {

#ifndef PDE_AMD_ZCOL_MAXTF
#define PDE_AMD_ZCOL_MAXTF 512
#endif
#ifndef PDE_AMD_ZCOL_MAXTB
#define PDE_AMD_ZCOL_MAXTB 512
#endif
#ifndef PDE_AMD_ZCOL_WPSF
#define PDE_AMD_ZCOL_WPSF 1
#endif
#ifndef PDE_AMD_ZCOL_WPSB
#define PDE_AMD_ZCOL_WPSB 1
#endif

// BWD: the adjoint's configuration (its own tile shape; the forward and the adjoint are separate launches)
template <class P, int W, bool BWD = false>
struct ZColCfg {
  static constexpr bool ON = P::ND == 3 && P::ZCOL != 0 && P::ZPURE != 0 && P::NA == 0 && P::TLO == 1 && P::THI == 1 &&
                             (P::RT0 | P::RT1 | P::RT2) == 0 && P::PER0 && P::PER1 && P::PER2 && (W == 2 || W == 4);
  static constexpr int T1 = ON ? (BWD ? P::ZCB1 : P::ZCT1) : 8, T2 = ON ? (BWD ? P::ZCB2 : P::ZCT2) : 32;
  static constexpr int QR = T2 / W;              // cells (W points) per tile row
  static constexpr int LPF = T1 * QR;            // lanes per frame: one cell each
  // lanes per workgroup at most: it bounds the registers a lane may use (512 lanes: 256, 768: 168, 1024: 128)
  static constexpr int MAXTF = PDE_AMD_ZCOL_MAXTF, MAXTB = PDE_AMD_ZCOL_MAXTB;
  static constexpr int FPWF = MAXTF / LPF, FPWB = MAXTB / LPF;      // residual frames per workgroup at most
  static constexpr int WPSF = PDE_AMD_ZCOL_WPSF, WPSB = PDE_AMD_ZCOL_WPSB;      // waves per SIMD to compile for
  using TP = Tile<1, T1, T2, 0, P::RH1, padw<W>(P::RH2)>;      // plane image of a field: the residual's (y, x) halo
  using TG = Tile<1, T1, T2, 0, P::GH1, padw<W>(P::GH2)>;      // plane image of a scaled residual: the adjoint's halo
  static constexpr bool SHAPE_OK = T2 % W == 0 && LPF % 64 == 0 && LPF <= 512;
  static constexpr bool FWD_OK = ON && SHAPE_OK;
#ifdef PDE_AMD_ZCOL_NOBWD      // diagnostic builds: the adjoint keeps the other kernels
  static constexpr bool BWD_OK = false;
#else
  static constexpr bool BWD_OK = ON && SHAPE_OK && (P::GT0 | P::GT1 | P::GT2) == 0 && P::GTLO == 1 && P::GTHI == 1 &&
                                 P::ATLO == 0 && P::ATHI == 0 && P::AFPURE != 0 &&
                                 (P::AT0 | P::AT1 | P::AT2) == 0 && P::TIME_AF == 0;
#endif
  // the adjoint's field reads: NFL fields at the own cell (a z column of 2 AH0 + 1 planes in registers); when it also
  // reads them beside the site in (y, x), NFA of them as plane images of the own frame
  static constexpr int NFL = P::ANY_AF ? P::AFN : 0;
  static constexpr int NFA = (P::AH1 | P::AH2) != 0 ? NFL : 0;
  using TA = Tile<1, T1, T2, 0, P::AH1, padw<W>(P::AH2)>;
  // dynamic LDS of a workgroup of nfw frames: the block reduction's slots, then 2 buffers x (nfw + 2) frame slots
  static constexpr size_t RED_BYTES = 16 * MAXE * sizeof(double);
  static size_t fwd_lds(int nfw) { return RED_BYTES + 2ul * (nfw + 2) * P::NF * TP::SIZE * sizeof(float); }
  static size_t bwd_lds(int nfw) {
    return RED_BYTES + 2ul * ((nfw + 2) * P::NE * TG::SIZE + nfw * NFA * TA::SIZE) * sizeof(float);
  }
};

// A lane offset no buffer of these kernels reaches (they are launched for clips of at most 2 GiB per field): loads through
// it return zeros, stores through it are dropped.  Far from 2^32, so that offset + access width cannot wrap into range.
constexpr unsigned ZC_NOWHERE = 0x80000000u;

// The cells of a plane image that are not the tile's own: rows above and below (whole image rows), then the side cells
// of the tile's rows.  Dealt to the lanes of a frame; offsets computed once per workgroup, periodic wrap applied.
template <class TL, int W, int LPF>
struct HaloCells {
  static constexpr int QD = TL::D2 / W, HQ = TL::H2 / W, QT = TL::T2 / W;
  static constexpr int NH = 2 * TL::H1 * QD + TL::T1 * 2 * HQ;
  static constexpr int EPT = NH > 0 ? (NH + LPF - 1) / LPF : 1;
  unsigned goff[EPT];      // byte offset inside a plane; ZC_NOWHERE: this lane has no cell j
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
      goff[j] = have ? 4u * (unsigned)(c1 * g.N2 + c2) : ZC_NOWHERE;
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

// What a launch of the z-column kernels covers besides Args: the clip's residual frames in `ngroups` groups of `nfw`
// frames each; the last group may be shorter, and the waves of its missing frames run along with every lane offset
// "nowhere" (they load zeros, store nothing, write no image), so that the loop stays free of branches around memory.
struct ZColLaunch {
  int nfw, ngroups;
};

// Wave priorities: a wave runs at priority 2 from the point where it starts writing the next plane's images, through the
// barrier and the issue of the loads for the plane after, and at 0 while it evaluates the stencil.  The images are what
// every wave of the workgroup waits for at the barrier and the loads are what the step after next waits for, so the
// arbiter should prefer them to the arithmetic of the other workgroup on the CU (Burgers-3D adjoint 185 -> 169 us,
// forward writing residuals 120 -> 112 us, advection-3D adjoint 69 -> 53 us; dropping the priority right after the
// barrier instead: 175 us).
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

// launched with LPF * nfw threads and ZColCfg::fwd_lds(nfw) bytes of dynamic LDS.  SOLO: one frame per workgroup, whose
// waves stage both outer frames; otherwise the waves of the group's first frame stage the frame before it and those of
// its last frame the one after.
template <class P, int W, bool WRITE_R, bool SOLO>
__global__ __launch_bounds__((ZColCfg<P, W>::MAXTF), (ZColCfg<P, W>::WPSF)) void fwd_zcol_kernel(const Args a, const ZColLaunch zl) {
  using C = ZColCfg<P, W>;
  using TP = typename C::TP;
  using fv = typename VecOf<W>::type;
  using HC = HaloCells<TP, W, C::LPF>;
  constexpr int LPF = C::LPF, RH0 = P::RH0, NZ = 2 * RH0 + 1, NF = P::NF, NE = P::NE, QR = C::QR, EPT = HC::EPT;
  constexpr int NED = SOLO ? 2 : 1;      // outer frames a wave may have to stage
  static_assert(C::FWD_OK, "PDE not eligible for the z-column forward");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  double* s_red = reinterpret_cast<double*>(s_raw);
  float* s_img = reinterpret_cast<float*>(s_raw + C::RED_BYTES);

  const Geom& g = a.geo;      // nt1, nt2: (y, x) tiles; nt0: chunks of planes; a.chunk: planes per chunk
  const int nfw = zl.nfw;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const ZColItem it(w, g, zl.ngroups);
  // fw is wave-uniform (LPF is a multiple of 64); said so explicitly, everything derived from it (frame offsets, image
  // slots) stays in scalar registers and the buffer instructions take it as their scalar offset
  const int fw = __builtin_amdgcn_readfirstlane((int)threadIdx.x / LPF), lid = threadIdx.x % LPF;
  const int nlive = min(nfw, g.TR - it.grp * nfw);      // frames of this group
  const bool live = fw < nlive;
  const int tr = it.grp * nfw + min(fw, nlive - 1), t = tr + 1;
  const int o2 = (it.tile % g.nt2) * TP::T2, o1 = (it.tile / g.nt2) * TP::T1;
  const int za = it.zc * a.chunk, zb = min(za + a.chunk, g.N0);
  const unsigned plane4 = 4u * (unsigned)(g.N1 * g.N2), frame4 = 4u * (unsigned)g.S;
  const int bufsz = (nfw + 2) * NF * TP::SIZE;

  // own cell
  const int row = lid / QR, l2 = W * (lid % QR);
  const int pi1 = o1 + row, pi2 = o2 + l2;
  const bool inside = pi1 < g.N1 && pi2 < g.N2;
  const bool rowok = inside && pi1 >= g.lo1 && pi1 < g.N1 - g.hi1;
  unsigned ooff;      // byte offset of the cell inside a plane; cells past the grid's end hold the periodic image
  {
    bool ok = true;
    const int c1 = wrap_coord<true>(pi1, g.N1, ok), c2 = wrap_quad<true, W>(pi2, g.N2, ok);
    ooff = live ? 4u * (unsigned)(c1 * g.N2 + c2) : ZC_NOWHERE;
  }
  const unsigned soff = live && inside ? 4u * (unsigned)(pi1 * g.N2 + pi2) : ZC_NOWHERE;      // where the cell's results go
  const int olds = TP::at(0, row, l2);
  HC hc;
  hc.init(lid, o1, o2, g);
  if (!live) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) hc.goff[j] = ZC_NOWHERE;
  }

  __amdgpu_buffer_rsrc_t fbuf[NF], rbuf[NE];
#pragma unroll
  for (int f = 0; f < NF; ++f) fbuf[f] = buffer_of(a.f[f] + (long)it.b * g.T * g.S, 4u * (unsigned)(g.T * g.S));
#pragma unroll
  for (int e = 0; e < NE; ++e) rbuf[e] = buffer_of(WRITE_R ? a.rout[e] + (long)it.b * g.TR * g.S : a.f[0], WRITE_R ? 4u * (unsigned)(g.TR * g.S) : 0u);
  const unsigned fr = (unsigned)t * frame4;      // this frame inside the sample's fields
  // the outer frame(s) this wave stages: byte offset of the frame, image slot, lane offset (nowhere: none)
  unsigned efr[NED], eoff[NED];
  int eslot[NED];
  if (SOLO) {
    efr[0] = fr - frame4; eslot[0] = 0; eoff[0] = ooff;
    efr[NED - 1] = fr + frame4; eslot[NED - 1] = 2; eoff[NED - 1] = ooff;
  } else {
    const bool lo = fw == 0, hi = fw == nlive - 1;
    efr[0] = lo ? fr - frame4 : fr + frame4;
    eslot[0] = lo ? 0 : nlive + 1;
    eoff[0] = lo || hi ? ooff : ZC_NOWHERE;
  }

  ZColResAcc<P, TP, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, it.b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  acc.i1 = pi1;
  float wk[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) wk[k] = live ? ldc(a.wt, k * g.TR + tr) : 0.0f;
  double sum[NE];
#pragma unroll
  for (int k = 0; k < NE; ++k) sum[k] = 0.0;

  // the column of the first plane, its halo cells and the outer frames' cells, then the first image
  fv hal[NF][EPT], edg[NED][NF], nxt[NF];
  PlaneCursor<true> cur;      // the plane that enters the column next
  cur.init(za + RH0 + 1, g.N0, 1);
  {
    bool on;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const unsigned pz = (unsigned)z_plane<true>(za - RH0 + i, g.N0, on) * plane4;
#pragma unroll
      for (int f = 0; f < NF; ++f) acc.col[f][i] = BufIO<W>::load(fbuf[f], ooff, fr + pz);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) hal[f][j] = BufIO<W>::load(fbuf[f], hc.goff[j], fr + (unsigned)za * plane4);
#pragma unroll
      for (int s = 0; s < NED; ++s) edg[s][f] = BufIO<W>::load(fbuf[f], eoff[s], efr[s] + (unsigned)za * plane4);
    }
  }
  auto write_image = [&](float* img) __attribute__((always_inline)) {
    if (!live) return;
    float* mine = img + (fw + 1) * NF * TP::SIZE;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      *reinterpret_cast<fv*>(mine + f * TP::SIZE + olds) = acc.col[f][RH0];
#pragma unroll
      for (int j = 0; j < EPT; ++j)
        if (hc.lds[j] >= 0) *reinterpret_cast<fv*>(mine + f * TP::SIZE + hc.lds[j]) = hal[f][j];
#pragma unroll
      for (int s = 0; s < NED; ++s)
        if (eoff[s] != ZC_NOWHERE) *reinterpret_cast<fv*>(img + (eslot[s] * NF + f) * TP::SIZE + olds) = edg[s][f];
    }
  };
  write_image(s_img);
  unsigned ko = (unsigned)za * plane4;          // byte offset of plane k inside a frame
  const unsigned rfr = (unsigned)tr * frame4;   // this residual frame inside the sample's residuals

  for (int k = za; k < zb; ++k) {
    float* img = s_img + ((k - za) & 1) * bufsz;
    __syncthreads();      // the image of plane k is complete, and nobody reads the other buffer any more
    // plane k + 1 (the last step fetches its own plane again, and what it writes below is never read): the column's
    // next entry, the halo cells, the outer frames; in flight during the step
    const unsigned kn = k + 1 < zb ? ko + plane4 : ko;
#ifndef PDE_AMD_ZC_NOLOAD
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      nxt[f] = BufIO<W>::load(fbuf[f], ooff, fr + (unsigned)cur.p * plane4);
#pragma unroll
      for (int j = 0; j < EPT; ++j) hal[f][j] = BufIO<W>::load(fbuf[f], hc.goff[j], fr + kn);
#pragma unroll
      for (int s = 0; s < NED; ++s) edg[s][f] = BufIO<W>::load(fbuf[f], eoff[s], efr[s] + kn);
    }
#else      // ablation (wrong results): no global loads inside the loop
#pragma unroll
    for (int f = 0; f < NF; ++f) nxt[f] = acc.col[f][0];
#endif
    cur.advance(g.N0, 1);
    __builtin_amdgcn_s_setprio(0);
    {
      acc.i0 = k;
      acc.site = img + (fw + 1) * NF * TP::SIZE + olds;
      const bool zok = k >= g.lo0 && k < g.N0 - g.hi0;
      float Rq[W][NE];
#pragma unroll
      for (int px = 0; px < W; ++px) {
        acc.px = px; acc.i2 = pi2 + px;
#ifdef PDE_AMD_ZC_NOCOMP      // ablation (wrong results): a few taps instead of the residual
#pragma unroll
        for (int e = 0; e < NE; ++e) Rq[px][e] = acc.F(e % NF, 0, -RH0, 0, 0) + acc.F(e % NF, 0, RH0, 0, 0) + acc.F(e % NF, 0, 0, 1, 0) + acc.F(e % NF, 1, 0, 0, 0);
#else
        P::residual(acc, Rq[px]);
#endif
        const bool ok = zok && rowok && acc.i2 >= g.lo2 && acc.i2 < g.N2 - g.hi2;
#pragma unroll
        for (int e = 0; e < NE; ++e) Rq[px][e] = ok ? Rq[px][e] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        float part = 0.0f;
#pragma unroll
        for (int px = 0; px < W; ++px) part += Rq[px][e] * Rq[px][e];
        // fp32 over the cell, fp64 across planes.  The waves of a short last group's missing frames read image slots nobody
        // wrote: what they computed is dropped by selection, never multiplied by a zero weight.
        sum[e] += live ? (double)part * (double)wk[e] : 0.0;
      }
      if (WRITE_R) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          fv r;
#pragma unroll
          for (int px = 0; px < W; ++px) r[px] = Rq[px][e];
          BufIO<W>::store(r, rbuf[e], soff, rfr + ko);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int i = 0; i + 1 < NZ; ++i) acc.col[f][i] = acc.col[f][i + 1];
      acc.col[f][NZ - 1] = nxt[f];
    }
    ko += plane4;
    __builtin_amdgcn_s_setprio(2);      // see the note on wave priorities above the kernels
    write_image(s_img + ((k + 1 - za) & 1) * bufsz);      // plane k + 1 into the buffer nobody reads during this step
  }
  __syncthreads();
  block_sum_store_rt<NE>(sum, s_red, a.partial + (long)w * NE);
}

// ------------------------------------------------------------------------------------------------
// adjoint from stored residuals
// ------------------------------------------------------------------------------------------------
template <class P, class TG, class TA, int W>
struct ZColAdjAcc : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  static constexpr int NZ = 2 * P::GH0 + 1, NZA = 2 * P::AH0 + 1;
  const float* site;         // &image[this frame's slot][equation 0][own row][own cell]
  const float* fsite;        // &field image[this frame][field 0][own row][own cell] (PDEs whose adjoint reads fields beside the site)
  fv gcol[P::NE][NZ];        // the cell's scaled residuals at planes k - GH0 .. k + GH0
  fv fcol[P::NF][NZA];       // the cell's fields at planes k - AH0 .. k + AH0
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
  __device__ __forceinline__ float F(int f, int, int o0, int o1, int o2) const {
    if (o0 != 0) return fcol[f][o0 + P::AH0][px];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    if (o1 == 0 && qd == 0) return fcol[f][P::AH0][e];
    const fv v = *reinterpret_cast<const fv*>(fsite + f * TA::SIZE + o1 * TA::S1 + W * qd);
    return v[e];
  }
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
// residual frame tr write the gradient of frame tr + 1 (and of the clip's first / last frame where tr is the first / last).
// SOLO as in the forward kernel; the outer slots hold the scaled residuals of the residual frames next to the group, which
// belong to other groups (zeros where the clip ends).
template <class P, int W, bool SOLO>
__global__ __launch_bounds__((ZColCfg<P, W, true>::MAXTB), (ZColCfg<P, W, true>::WPSB)) void bwd_stored_zcol_kernel(const Args a, const ZColLaunch zl) {
  using C = ZColCfg<P, W, true>;
  using TG = typename C::TG;
  using fv = typename VecOf<W>::type;
  using TA = typename C::TA;
  using HC = HaloCells<TG, W, C::LPF>;
  using HA = HaloCells<TA, W, C::LPF>;
  constexpr int LPF = C::LPF, GH0 = P::GH0, NZ = 2 * GH0 + 1, NF = P::NF, NE = P::NE, QR = C::QR, EPT = HC::EPT;
  constexpr int NFL = C::NFL, NFA = C::NFA, AH0 = P::AH0, NZA = 2 * AH0 + 1, EPA = HA::EPT;
  constexpr int NED = SOLO ? 2 : 1;
  static_assert(C::BWD_OK, "PDE not eligible for the z-column stored-residual adjoint");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  float* s_img = reinterpret_cast<float*>(s_raw + C::RED_BYTES);

  const Geom& g = a.geo;
  const int nfw = zl.nfw;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const ZColItem it(w, g, zl.ngroups);
  const int fw = __builtin_amdgcn_readfirstlane((int)threadIdx.x / LPF), lid = threadIdx.x % LPF;
  const int nlive = min(nfw, g.TR - it.grp * nfw);
  const bool live = fw < nlive;
  const int tr = it.grp * nfw + min(fw, nlive - 1), t = tr + 1;
  const int o2 = (it.tile % g.nt2) * TG::T2, o1 = (it.tile / g.nt2) * TG::T1;
  const int za = it.zc * a.chunk, zb = min(za + a.chunk, g.N0);
  const unsigned plane4 = 4u * (unsigned)(g.N1 * g.N2), frame4 = 4u * (unsigned)g.S;
  const int bufsz = (nfw + 2) * NE * TG::SIZE, fbufsz = nfw * NFA * TA::SIZE;
  float* s_fimg = s_img + 2 * bufsz;      // field images of the frames of the group, two buffers

  const int row = lid / QR, l2 = W * (lid % QR);
  const int pi1 = o1 + row, pi2 = o2 + l2;
  const bool inside = pi1 < g.N1 && pi2 < g.N2;
  unsigned ooff;
  {
    bool ok = true;
    const int c1 = wrap_coord<true>(pi1, g.N1, ok), c2 = wrap_quad<true, W>(pi2, g.N2, ok);
    ooff = live ? 4u * (unsigned)(c1 * g.N2 + c2) : ZC_NOWHERE;
  }
  const unsigned soff = live && inside ? 4u * (unsigned)(pi1 * g.N2 + pi2) : ZC_NOWHERE;
  // the clip's first / last frame get their gradient from the waves of the first / last residual frame
  const unsigned soff_first = tr == 0 ? soff : ZC_NOWHERE, soff_last = tr == g.TR - 1 ? soff : ZC_NOWHERE;
  const int olds = TG::at(0, row, l2), oldsa = TA::at(0, row, l2);
  HC hc;
  hc.init(lid, o1, o2, g);
  HA ha;
  ha.init(lid, o1, o2, g);
  if (!live) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) hc.goff[j] = ZC_NOWHERE;
#pragma unroll
    for (int j = 0; j < EPA; ++j) ha.goff[j] = ZC_NOWHERE;
  }

  __amdgpu_buffer_rsrc_t rbuf[NE], fbuf[NF], gbuf[NF];
#pragma unroll
  for (int e = 0; e < NE; ++e) rbuf[e] = buffer_of(a.r[e] + (long)it.b * g.TR * g.S, 4u * (unsigned)(g.TR * g.S));
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    fbuf[f] = buffer_of(a.f[f] + (long)it.b * g.T * g.S, 4u * (unsigned)(g.T * g.S));
    gbuf[f] = buffer_of(a.g[f] + (long)it.b * g.T * g.S, 4u * (unsigned)(g.T * g.S));
  }
  const unsigned rfr = (unsigned)tr * frame4, fr = (unsigned)t * frame4;

  float up[NE];
  upstream_of<P>(a, it.b, up);
  float cf[NE];      // coefficient of this residual frame
#pragma unroll
  for (int k = 0; k < NE; ++k) cf[k] = up[k] * ldc(a.wt, k * g.TR + tr);
  // the outer residual frame(s) this wave stages: byte offset, image slot, lane offset (nowhere: zeros), coefficients,
  // and whether this wave writes the slot at all
  unsigned efr[NED], eoff[NED];
  int eslot[NED];
  float ecf[NED][NE];
  bool ewr[NED];
  {
    const bool lo = fw == 0, hi = fw == nlive - 1;
    if (SOLO) {
      efr[0] = rfr - frame4; eslot[0] = 0; eoff[0] = tr > 0 ? ooff : ZC_NOWHERE; ewr[0] = true;
      efr[NED - 1] = rfr + frame4; eslot[NED - 1] = 2; eoff[NED - 1] = tr + 1 < g.TR ? ooff : ZC_NOWHERE; ewr[NED - 1] = true;
    } else {
      efr[0] = lo ? rfr - frame4 : rfr + frame4;
      eslot[0] = lo ? 0 : nlive + 1;
      eoff[0] = (lo && tr > 0) || (hi && !lo && tr + 1 < g.TR) ? ooff : ZC_NOWHERE;
      ewr[0] = lo || hi;
    }
#pragma unroll
    for (int s = 0; s < NED; ++s) {
      const int tre = min(max(eslot[s] == 0 ? tr - 1 : tr + 1, 0), g.TR - 1);
#pragma unroll
      for (int k = 0; k < NE; ++k) ecf[s][k] = up[k] * ldc(a.wt, k * g.TR + tre);
    }
  }

  ZColAdjAcc<P, TG, TA, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, it.b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  acc.i1 = pi1;
  ZColEdgeAcc<P, W> eacc;
  eacc.template set_uniforms<true>(a.c, a.sp, it.b);
  eacc.N0 = g.N0; eacc.N1 = g.N1; eacc.N2 = g.N2;
  eacc.i1 = pi1;

  fv hal[NE][EPT], edg[NED][NE], nxt[NE], fnx[NFL > 0 ? NFL : 1], fhal[NFA > 0 ? NFA : 1][EPA];
  PlaneCursor<true> cur, curf;      // the planes the columns take in next: k + 1 + GH0 (residuals), k + 1 + AH0 (fields)
  cur.init(za + GH0 + 1, g.N0, 1);
  curf.init(za + AH0 + 1, g.N0, 1);
  {
    bool on;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const unsigned pz = (unsigned)z_plane<true>(za - GH0 + i, g.N0, on) * plane4;
#pragma unroll
      for (int e = 0; e < NE; ++e) acc.gcol[e][i] = BufIO<W>::load(rbuf[e], ooff, rfr + pz);
    }
#pragma unroll
    for (int i = 0; i < NZA; ++i) {
      const unsigned pz = (unsigned)z_plane<true>(za - AH0 + i, g.N0, on) * plane4;
#pragma unroll
      for (int f = 0; f < NF; ++f) acc.fcol[f][i] = f < NFL ? BufIO<W>::load(fbuf[f], ooff, fr + pz) : vec_zero<W>();
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
      for (int j = 0; j < EPT; ++j) hal[e][j] = BufIO<W>::load(rbuf[e], hc.goff[j], rfr + (unsigned)za * plane4);
#pragma unroll
      for (int s = 0; s < NED; ++s) edg[s][e] = BufIO<W>::load(rbuf[e], eoff[s], efr[s] + (unsigned)za * plane4);
    }
#pragma unroll
    for (int f = 0; f < NFA; ++f) {
#pragma unroll
      for (int j = 0; j < EPA; ++j) fhal[f][j] = BufIO<W>::load(fbuf[f], ha.goff[j], fr + (unsigned)za * plane4);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
      for (int i = 0; i < NZ; ++i) acc.gcol[e][i] = acc.gcol[e][i] * cf[e];
    }
  }
  auto write_image = [&](float* img, float* fimg) __attribute__((always_inline)) {
    if (!live) return;
    float* mine = img + (fw + 1) * NE * TG::SIZE;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      *reinterpret_cast<fv*>(mine + e * TG::SIZE + olds) = acc.gcol[e][GH0];
#pragma unroll
      for (int j = 0; j < EPT; ++j)
        if (hc.lds[j] >= 0) *reinterpret_cast<fv*>(mine + e * TG::SIZE + hc.lds[j]) = hal[e][j] * cf[e];
#pragma unroll
      for (int s = 0; s < NED; ++s)
        if (ewr[s]) *reinterpret_cast<fv*>(img + (eslot[s] * NE + e) * TG::SIZE + olds) = edg[s][e] * ecf[s][e];
    }
    float* fmine = fimg + fw * NFA * TA::SIZE;
#pragma unroll
    for (int f = 0; f < NFA; ++f) {
      *reinterpret_cast<fv*>(fmine + f * TA::SIZE + oldsa) = acc.fcol[f][AH0];
#pragma unroll
      for (int j = 0; j < EPA; ++j)
        if (ha.lds[j] >= 0) *reinterpret_cast<fv*>(fmine + f * TA::SIZE + ha.lds[j]) = fhal[f][j];
    }
  };
  write_image(s_img, s_fimg);
  unsigned ko = (unsigned)za * plane4;

  for (int k = za; k < zb; ++k) {
    float* img = s_img + ((k - za) & 1) * bufsz;
    float* fimg = s_fimg + ((k - za) & 1) * fbufsz;
    __syncthreads();
    const unsigned kn = k + 1 < zb ? ko + plane4 : ko;
#ifndef PDE_AMD_ZC_NOLOAD
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      nxt[e] = BufIO<W>::load(rbuf[e], ooff, rfr + (unsigned)cur.p * plane4);
#pragma unroll
      for (int j = 0; j < EPT; ++j) hal[e][j] = BufIO<W>::load(rbuf[e], hc.goff[j], rfr + kn);
#pragma unroll
      for (int s = 0; s < NED; ++s) edg[s][e] = BufIO<W>::load(rbuf[e], eoff[s], efr[s] + kn);
    }
#pragma unroll
    for (int f = 0; f < NFL; ++f) fnx[f] = BufIO<W>::load(fbuf[f], ooff, fr + (unsigned)curf.p * plane4);
#pragma unroll
    for (int f = 0; f < NFA; ++f) {
#pragma unroll
      for (int j = 0; j < EPA; ++j) fhal[f][j] = BufIO<W>::load(fbuf[f], ha.goff[j], fr + kn);
    }
#else      // ablation (wrong results): no global loads inside the loop
#pragma unroll
    for (int e = 0; e < NE; ++e) nxt[e] = acc.gcol[e][0];
#endif
    cur.advance(g.N0, 1);
    curf.advance(g.N0, 1);
    __builtin_amdgcn_s_setprio(0);
    {
      acc.i0 = k;
      acc.site = img + (fw + 1) * NE * TG::SIZE + olds;
      acc.fsite = fimg + fw * NFA * TA::SIZE + oldsa;
      fv out[NF];
#pragma unroll
      for (int px = 0; px < W; ++px) {
        acc.px = px; acc.i2 = pi2 + px;
        float gr[NF];
#ifdef PDE_AMD_ZC_NOCOMP      // ablation (wrong results): a few taps instead of the adjoint
#pragma unroll
        for (int f = 0; f < NF; ++f) gr[f] = acc.G(f % NE, 0, -GH0, 0, 0) + acc.G(f % NE, 0, GH0, 0, 0) + acc.G(f % NE, 0, 0, 1, 0) + acc.G(f % NE, 1, 0, 0, 0) + acc.F(f, 0, 0, 0, 0);
#else
        P::adjoint(acc, gr);
#endif
#pragma unroll
        for (int f = 0; f < NF; ++f) out[f][px] = gr[f];
      }
#ifndef PDE_AMD_ZC_NOSTORE
#pragma unroll
      for (int f = 0; f < NF; ++f) BufIO<W>::store_nt(out[f], gbuf[f], soff, fr + ko);
#endif
      // the clip's first / last frame: the time share of this residual frame alone; every wave evaluates the few terms,
      // the stores of all but the first / last residual frame's waves go nowhere
      eacc.i0 = k;
#pragma unroll
      for (int e = 0; e < NE; ++e) eacc.gq[e] = acc.gcol[e][GH0];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
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
#ifndef PDE_AMD_ZC_NOSTORE
#pragma unroll
        for (int f = 0; f < NF; ++f) BufIO<W>::store_nt(est[f], gbuf[f], s == 0 ? soff_first : soff_last, (s == 0 ? fr - frame4 : fr + frame4) + ko);
#endif
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
      for (int i = 0; i + 1 < NZ; ++i) acc.gcol[e][i] = acc.gcol[e][i + 1];
      acc.gcol[e][NZ - 1] = nxt[e] * cf[e];
    }
#pragma unroll
    for (int f = 0; f < NFL; ++f) {
#pragma unroll
      for (int i = 0; i + 1 < NZA; ++i) acc.fcol[f][i] = acc.fcol[f][i + 1];
      acc.fcol[f][NZA - 1] = fnx[f];
    }
    ko += plane4;
    __builtin_amdgcn_s_setprio(2);      // see the note on wave priorities above the kernels
    write_image(s_img + ((k + 1 - za) & 1) * bufsz, s_fimg + ((k + 1 - za) & 1) * fbufsz);
  }
}

}  // namespace pde_amd
