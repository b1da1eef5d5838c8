// Plane-marching kernels for 3-D PDEs ("2.5-D blocking").
//
// A workgroup owns a (y, x) tile and marches along the slowest spatial axis (tensor dim -3, "z" below) through a chunk of
// planes of one frame.  The planes the stencil reaches along z live in an LDS ring (2 * reach + 1 planes per field, each
// with the (y, x) halo), so every plane of a field is fetched once per (y, x) tile instead of once per 3-D tile that
// touches it, the halo is two-dimensional (rows of the full tile width: whole cache lines), and the set-up of a workgroup
// (staging offsets, lattice tests, uniform coefficients) is paid once per chunk of planes instead of once per tile.
// References to other frames must be pointwise (no spatial offset): they are the lane's own quads, read from global
// memory.  Used for the 3-D PDEs whose time ring of 3-D halo tiles does not fit LDS (the per-frame kernels of
// stencil_quad.hpp remain the fallback).
#pragma once
#include "stencil_quad.hpp"

namespace pde_amd {

template <class P, int W>
struct ZCfg {
  static constexpr int NT = P::NTZ;
  static constexpr bool ON = P::ND == 3 && P::ZT1 > 0 && P::ZT2 > 0 && P::NA == 0;
  static constexpr int T1 = ON ? P::ZT1 : 8, T2 = ON ? P::ZT2 : 64;
  static constexpr bool SHAPE_OK = T2 % W == 0 && (T1 * (T2 / W)) % NT == 0 && NT % (T2 / W) == 0 && NT % 64 == 0 && NT <= 1024;
  // forward: ring of field planes with the residual's (y, x) halo
  using TP = Tile<1, T1, T2, 0, P::RH1, padw<W>(P::RH2)>;
  static constexpr int NZ = 2 * P::RH0 + 2;      // the planes a step reads plus a spare slot: the next plane lands there while they are read
  static constexpr bool TT_ON = (P::RT0 | P::RT1 | P::RT2) != 0;
  static constexpr long FWD_BYTES = 4L * P::NF * NZ * TP::SIZE;
  static constexpr bool FWD_OK = ON && SHAPE_OK && !TT_ON && FWD_BYTES <= 64 * 1024;
  // adjoint from stored residuals: ring of scaled residual planes with the halo the adjoint reaches; the fields it reads
  // are a second ring, or (read at the output point only) the lane's own quads
  using TGP = Tile<1, T1, T2, 0, P::GH1, padw<W>(P::GH2)>;
  using TAP = Tile<1, T1, T2, 0, P::AH1, padw<W>(P::AH2)>;
  static constexpr int NZG = 2 * P::GH0 + 2, NZA = 2 * P::AH0 + 2;      // with the spare slot, as NZ
  static constexpr bool AF_TILE = P::ANY_AF != 0 && (P::AH0 | P::AH1 | P::AH2) != 0;
  static constexpr bool GT_ON = (P::GT0 | P::GT1 | P::GT2) != 0;
  static constexpr long BWD_BYTES = 4L * (P::NE * NZG * TGP::SIZE + (AF_TILE ? P::NF * NZA * TAP::SIZE : 0));
  static constexpr bool BWD_OK = ON && SHAPE_OK && !GT_ON && P::ATLO == 0 && P::ATHI == 0 && (P::AT0 | P::AT1 | P::AT2) == 0 &&
                                 BWD_BYTES <= 64 * 1024;
};

// One plane of a field staged through registers into a ring slot; a plane outside a non-periodic z range is zeros.
template <class TP, bool P1, bool P2, int NT, int W>
struct PlaneStager {
  using St = StagerQ<TP, true, P1, P2, NT, W>;
  using fv = typename VecOf<W>::type;
  St st;
  __device__ __forceinline__ void init(int o1, int o2, const Geom& g) { st.init(0, o1, o2, g); }
  __device__ __forceinline__ void fetch(fv (&v)[St::EPT], const float* __restrict__ plane) const { st.fetch(v, plane); }
  __device__ __forceinline__ void store(const fv (&v)[St::EPT], float* __restrict__ slot, bool on) const {
    if (on) {
      st.store(v, slot);
    } else {      // wave-uniform: the whole plane lies outside the grid
      st.touch(v);
#pragma unroll
      for (int j = 0; j < St::EPT; ++j) {
        const int i = threadIdx.x + j * NT;
        if (St::EPT * NT == St::NQ || i < St::NQ) *reinterpret_cast<fv*>(slot + W * i) = vec_zero<W>();
      }
    }
  }
};

// plane index along z seen from plane k at offset o: wrapped when periodic; `ok` false when it leaves a wall-bounded grid
template <bool PER>
__device__ __forceinline__ int z_plane(int p, int N, bool& ok) {
  ok = true;
  if (PER) {
    p %= N;
    return p < 0 ? p + N : p;
  }
  if (p < 0 || p >= N) { ok = false; return 0; }
  return p;
}

// The plane a ring fetches next, advanced one plane per step without a division or a 64-bit multiply when periodic.
template <bool PER>
struct PlaneCursor {
  int p;        // plane index (unwrapped when wall-bounded)
  long off;     // element offset of the plane inside a frame (0 when it lies outside the grid)
  bool on;      // false when the plane lies outside a wall-bounded grid: the ring slot is filled with zeros
  __device__ __forceinline__ void init(int first, int N, long plane) {
    if (PER) { p = z_plane<true>(first, N, on); }
    else { p = first; on = first >= 0 && first < N; }
    off = on ? p * plane : 0;
  }
  __device__ __forceinline__ void advance(int N, long plane) {
    p += 1;
    if (PER) {
      off += plane;
      if (p == N) { p = 0; off = 0; }
    } else {
      on = p >= 0 && p < N;
      off = on ? p * plane : 0;
    }
  }
};

template <class P, class TP, int NZ, int W>
struct ZRingAccQ : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  static constexpr int NTS = P::TLO + P::THI;
  const float* site;         // &ring[field 0][slot 0][row][quad start]
  int ring[NZ - 1];          // element offset of the slot that holds plane k + i - RH0
  fv fc[NTS > 0 ? NTS : 1][MAXF];     // this quad at the other frames
  int px;
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    if (ot != 0) return fc[ot < 0 ? ot + P::TLO : ot + P::TLO - 1][f][px];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(site + f * NZ * TP::SIZE + ring[o0 + P::RH0] + o1 * TP::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float X(int, int, int, int) const { return 0.0f; }
  __device__ __forceinline__ float G(int, int, int, int, int) const { return 0.0f; }
};

// ------------------------------------------------------------------------------------------------
// forward: workgroup = (sample, residual frame, (y, x) tile, chunk of planes)
// ------------------------------------------------------------------------------------------------
template <class P, int W, bool WRITE_R>
__global__ __launch_bounds__((ZCfg<P, W>::NT)) void fwd_zmarch_kernel(const Args a) {
  using C = ZCfg<P, W>;
  using TP = typename C::TP;
  using fv = typename VecOf<W>::type;
  constexpr int NT = C::NT, NZ = C::NZ, NTS = P::TLO + P::THI, RH0 = P::RH0;
  constexpr int QR = TP::T2 / W;
  constexpr int QPT = (TP::T1 * QR) / NT;
  using PS = PlaneStager<TP, P::PER1, P::PER2, NT, W>;
  constexpr int EPT = PS::St::EPT;
  static_assert(C::FWD_OK, "PDE not eligible for the plane-marching forward");

  __shared__ __attribute__((aligned(16))) float s_f[P::NF * NZ * TP::SIZE];
  __shared__ double s_red[(NT / 64) * P::NE];

  const Geom& g = a.geo;      // nt1, nt2: (y, x) tiles; nt0: chunks of planes; a.chunk: planes per chunk
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt1 * g.nt2;
  // frames fastest: the work-groups of one tile and chunk that differ in the frame read the same field planes
  // (time taps), run at the same time on one XCD and meet in its L2
  const int tr = w % g.TR, r1 = w / g.TR;
  const int tile = r1 % ntile, r2 = r1 / ntile;
  const int zc = r2 % g.nt0, b = r2 / g.nt0;
  const int t = tr + P::TLO;
  const int o2 = (tile % g.nt2) * TP::T2, o1 = (tile / g.nt2) * TP::T1;
  const int za = zc * a.chunk, zb = min(za + a.chunk, g.N0);
  const long plane = (long)g.N1 * g.N2;
  const float* fb[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) fb[f] = a.f[f] + ((long)b * g.T + t) * g.S;

  PS ps;
  ps.init(o1, o2, g);
  // the planes the first step reads, all in flight before the first is stored
  {
    fv v[NZ - 1][P::NF][EPT];
    bool on[NZ - 1];
    int pz[NZ - 1];
#pragma unroll
    for (int i = 0; i < NZ - 1; ++i) {
      pz[i] = z_plane<P::PER0>(za - RH0 + i, g.N0, on[i]);
#pragma unroll
      for (int f = 0; f < P::NF; ++f) ps.fetch(v[i][f], fb[f] + pz[i] * plane);
    }
#pragma unroll
    for (int i = 0; i < NZ - 1; ++i)
#pragma unroll
      for (int f = 0; f < P::NF; ++f) ps.store(v[i][f], s_f + (f * NZ + i) * TP::SIZE, on[i]);
  }

  int site[QPT], pi1[QPT], pi2[QPT];
  long goff[QPT];
  bool inside[QPT], rowok[QPT];
#pragma unroll
  for (int q = 0; q < QPT; ++q) {
    const int row = (threadIdx.x / QR) * QPT + q;
    const int l2 = W * owned_quad<QR, TP::S1, W>(), l1 = row;
    pi1[q] = o1 + l1; pi2[q] = o2 + l2;
    inside[q] = pi1[q] < g.N1 && pi2[q] < g.N2;
    rowok[q] = inside[q] && pi1[q] >= g.lo1 && pi1[q] < g.N1 - g.hi1;
    goff[q] = inside[q] ? (long)pi1[q] * g.N2 + pi2[q] : 0;
    site[q] = TP::at(0, l1, l2);
  }

  ZRingAccQ<P, TP, NZ, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  float wk[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) wk[k] = ldc(a.wt, k * g.TR + tr);
  double sum[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) sum[k] = 0.0;
  const long rframe = ((long)b * g.TR + tr) * g.S;
  __syncthreads();

  // Loop state kept incrementally (no division, no 64-bit multiply per plane): the slot offsets of the ring, the plane to
  // fetch next and the base of the current plane.
  int ro[NZ - 1], spare = (NZ - 1) * TP::SIZE;
#pragma unroll
  for (int i = 0; i < NZ - 1; ++i) ro[i] = i * TP::SIZE;
  PlaneCursor<P::PER0> cur;      // the plane that enters the ring next
  cur.init(za + RH0 + 1, g.N0, plane);
  long ko = za * plane;          // element offset of plane k inside a frame
  // Loads are issued at the top of the step that uses them: issuing them a step earlier (two register stages) costs
  // 24 registers, one wave per SIMD less on the residual-writing variant, and gained nothing when measured.
  constexpr int NTSA = NTS > 0 ? NTS : 1;
  for (int k = za; k < zb; ++k) {
    const bool more = k + 1 < zb;
    fv stage[P::NF][EPT];      // plane k + 1 + RH0: in flight during the step
    fv fcq[QPT][NTSA][P::NF];  // this lane's quads of the other frames
#ifdef PDE_AMD_ZDIAG_NOLOAD      // ablation (wrong results): no global loads inside the loop
#pragma unroll
    for (int f = 0; f < P::NF; ++f)
#pragma unroll
      for (int j = 0; j < EPT; ++j) stage[f][j] = vec_zero<W>() + (float)k;
#pragma unroll
    for (int q = 0; q < QPT; ++q)
#pragma unroll
      for (int s_ = 0; s_ < NTS; ++s_)
#pragma unroll
        for (int f = 0; f < P::NF; ++f) fcq[q][s_][f] = vec_zero<W>() + (float)(k + f);
#else
    if (more) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) ps.fetch(stage[f], fb[f] + cur.off);
    }
#pragma unroll
    for (int q = 0; q < QPT; ++q)
#pragma unroll
      for (int s = 0; s < NTS; ++s) {
        const int ot = s < P::TLO ? s - P::TLO : s - P::TLO + 1;
#pragma unroll
        for (int f = 0; f < P::NF; ++f) fcq[q][s][f] = *reinterpret_cast<const fv*>(fb[f] + (long)ot * g.S + ko + goff[q]);
      }
#endif
    const bool son = cur.on;
    cur.advance(g.N0, plane);
    const bool zok = k >= g.lo0 && k < g.N0 - g.hi0;
    float part[P::NE];
#pragma unroll
    for (int e = 0; e < P::NE; ++e) part[e] = 0.0f;
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      acc.i0 = k; acc.i1 = pi1[q];
#pragma unroll
      for (int i = 0; i < NZ - 1; ++i) acc.ring[i] = ro[i];
      acc.site = s_f + site[q];
#pragma unroll
      for (int s = 0; s < NTS; ++s)
#pragma unroll
        for (int f = 0; f < P::NF; ++f) acc.fc[s][f] = fcq[q][s][f];
      float Rq[W][P::NE];
#pragma unroll
      for (int px = 0; px < W; ++px) {
        acc.px = px; acc.i2 = pi2[q] + px;
#ifdef PDE_AMD_ZDIAG_NOCOMP      // ablation (wrong results): a few taps instead of the residual
#pragma unroll
        for (int e = 0; e < P::NE; ++e) Rq[px][e] = acc.F(e % P::NF, 0, -P::RH0, 0, 0) + acc.F(e % P::NF, 0, P::RH0, 0, 0) + acc.F(e % P::NF, NTS ? -1 : 0, 0, 0, 0);
#else
        P::residual(acc, Rq[px]);
#endif
        const bool ok = zok && rowok[q] && acc.i2 >= g.lo2 && acc.i2 < g.N2 - g.hi2;
#pragma unroll
        for (int e = 0; e < P::NE; ++e) Rq[px][e] = ok ? Rq[px][e] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < P::NE; ++e)
#pragma unroll
        for (int px = 0; px < W; ++px) part[e] += Rq[px][e] * Rq[px][e];
      if (WRITE_R && inside[q]) {
        const long ro_ = rframe + ko + goff[q];
#pragma unroll
        for (int e = 0; e < P::NE; ++e) {
          fv r;
#pragma unroll
          for (int px = 0; px < W; ++px) r[px] = Rq[px][e];
          *reinterpret_cast<fv*>(a.rout[e] + ro_) = r;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < P::NE; ++e) sum[e] += (double)part[e] * (double)wk[e];      // fp32 over a lane's few points of a plane, fp64 across planes
    if (more) {      // into the spare slot, which nobody reads during this step; one barrier per plane
#pragma unroll
      for (int f = 0; f < P::NF; ++f) ps.store(stage[f], s_f + f * NZ * TP::SIZE + spare, son);
#ifndef PDE_AMD_ZDIAG_NOBAR      // ablation (wrong results): no barrier inside the loop
      __syncthreads();
#endif
    }
    // advance: the oldest plane's slot becomes the spare one, the plane just stored the newest of the window
    const int oldest = ro[0];
#pragma unroll
    for (int i = 0; i + 1 < NZ - 1; ++i) ro[i] = ro[i + 1];
    ro[NZ - 2] = spare;
    spare = oldest;
    ko += plane;
  }
  __syncthreads();
  block_sum_store<P::NE, NT>(sum, s_red, a.partial + (long)w * P::NE);
}

// ------------------------------------------------------------------------------------------------
// adjoint from stored residuals: workgroup = (sample, output frame, (y, x) tile, chunk of planes)
// ------------------------------------------------------------------------------------------------
template <class P, class TGP, class TAP, int NZG, int NZA, int W>
struct ZStoredAdjAccQ : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  static constexpr int NG = P::GTLO + P::GTHI + 1;
  const float* siteg;        // &G ring[equation 0][slot 0][row][quad start]
  const float* sitea;        // &field ring[field 0][slot 0][row][quad start]
  int ringg[NZG - 1], ringa[NZA - 1];      // element offsets of the slots
  fv gc[NG][MAXE];           // this quad's scaled residuals at the other frames
  fv fq[MAXF];               // this quad's fields when the adjoint reads them at the output point only
  int px;
  __device__ __forceinline__ float G(int k, int ot, int o0, int o1, int o2) const {
    if (ot != 0) return gc[ot + P::GTLO][k][px];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(siteg + k * NZG * TGP::SIZE + ringg[o0 + P::GH0] + o1 * TGP::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float F(int f, int, int o0, int o1, int o2) const {
    if ((P::AH0 | P::AH1 | P::AH2) == 0) return fq[f][px];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(sitea + f * NZA * TAP::SIZE + ringa[o0 + P::AH0] + o1 * TAP::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float X(int, int, int, int) const { return 0.0f; }
};

template <class P, int W>
__global__ __launch_bounds__((ZCfg<P, W>::NT)) void bwd_stored_zmarch_kernel(const Args a) {
  using C = ZCfg<P, W>;
  using TGP = typename C::TGP;
  using TAP = typename C::TAP;
  using fv = typename VecOf<W>::type;
  constexpr int NT = C::NT, NZG = C::NZG, NZA = C::NZA, GH0 = P::GH0, AH0 = P::AH0, NG = P::GTLO + P::GTHI + 1;
  constexpr bool AF_TILE = C::AF_TILE, AF_REG = P::ANY_AF != 0 && !C::AF_TILE;
  constexpr int QR = TGP::T2 / W;
  constexpr int QPT = (TGP::T1 * QR) / NT;
  using PSG = PlaneStager<TGP, P::PER1, P::PER2, NT, W>;
  using PSA = PlaneStager<TAP, P::PER1, P::PER2, NT, W>;
  constexpr int EG = PSG::St::EPT, EA = PSA::St::EPT;
  static_assert(C::BWD_OK, "PDE not eligible for the plane-marching stored-residual adjoint");

  __shared__ __attribute__((aligned(16))) float s_g[P::NE * NZG * TGP::SIZE];
  __shared__ __attribute__((aligned(16))) float s_a[AF_TILE ? P::NF * NZA * TAP::SIZE : 4];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt1 * g.nt2;
  const int tq = w % g.T, r1 = w / g.T;      // frames fastest, as in the forward kernel
  const int tile = r1 % ntile, r2 = r1 / ntile;
  const int zc = r2 % g.nt0, b = r2 / g.nt0;
  const int o2 = (tile % g.nt2) * TGP::T2, o1 = (tile / g.nt2) * TGP::T1;
  const int za = zc * a.chunk, zb = min(za + a.chunk, g.N0);
  const long plane = (long)g.N1 * g.N2;
  const int tr0 = tq - P::TLO;
  const int trc = max(min(tr0, g.TR - 1), 0);
  const float* rp[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) rp[k] = a.r[k] + (long)b * g.TR * g.S;
  const float* fb[P::NF];
  float* gb[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) { fb[f] = a.f[f] + ((long)b * g.T + tq) * g.S; gb[f] = a.g[f] + ((long)b * g.T + tq) * g.S; }

  float up[P::NE];
  upstream_of<P>(a, b, up);
  float cf[NG][P::NE];      // coefficient of the residual frames tr0 + s - GTLO (zero outside the range: such values are dropped)
#pragma unroll
  for (int s = 0; s < NG; ++s) {
    const int trs = tr0 + (s - P::GTLO);
    const bool ok = trs >= 0 && trs < g.TR;
#pragma unroll
    for (int k = 0; k < P::NE; ++k) cf[s][k] = ok ? up[k] * ldc(a.wt, k * g.TR + max(min(trs, g.TR - 1), 0)) : 0.0f;
  }
  bool fon[NG];          // whether the residual frame tr0 + s - GTLO contributes at all (wave-uniform)
#pragma unroll
  for (int s = 0; s < NG; ++s) {
    fon[s] = false;
#pragma unroll
    for (int k = 0; k < P::NE; ++k) fon[s] = fon[s] || cf[s][k] != 0.0f;
  }
  const bool gon = fon[P::GTLO];      // this output frame has a residual frame of its own

  PSG psg;
  psg.init(o1, o2, g);
  PSA psa;
  if (AF_TILE) psa.init(o1, o2, g);
  {
    fv v[NZG - 1][P::NE][EG];
    bool on[NZG - 1];
#pragma unroll
    for (int i = 0; i < NZG - 1; ++i) {
      const int pz = z_plane<P::PER0>(za - GH0 + i, g.N0, on[i]);
      on[i] = on[i] && gon;
      if (gon) {
#pragma unroll
        for (int k = 0; k < P::NE; ++k) psg.fetch(v[i][k], rp[k] + (long)trc * g.S + pz * plane);
      }
    }
#pragma unroll
    for (int i = 0; i < NZG - 1; ++i)
#pragma unroll
      for (int k = 0; k < P::NE; ++k) {
        const float cfk = cf[P::GTLO][k];
#pragma unroll
        for (int j = 0; j < EG; ++j) v[i][k][j] = v[i][k][j] * cfk;
        psg.store(v[i][k], s_g + (k * NZG + i) * TGP::SIZE, on[i] && cfk != 0.0f);
      }
  }
  if (AF_TILE) {
    fv v[NZA - 1][P::NF][EA];
    bool on[NZA - 1];
#pragma unroll
    for (int i = 0; i < NZA - 1; ++i) {
      const int pz = z_plane<P::PER0>(za - AH0 + i, g.N0, on[i]);
#pragma unroll
      for (int f = 0; f < P::NF; ++f) psa.fetch(v[i][f], fb[f] + pz * plane);
    }
#pragma unroll
    for (int i = 0; i < NZA - 1; ++i)
#pragma unroll
      for (int f = 0; f < P::NF; ++f) psa.store(v[i][f], s_a + (f * NZA + i) * TAP::SIZE, on[i]);
  }

  int sG[QPT], sA[QPT], pi1[QPT], pi2[QPT];
  long goff[QPT];
  bool inside[QPT];
#pragma unroll
  for (int q = 0; q < QPT; ++q) {
    const int row = (threadIdx.x / QR) * QPT + q;
    const int l2 = W * owned_quad<QR, TGP::S1, W>(), l1 = row;
    pi1[q] = o1 + l1; pi2[q] = o2 + l2;
    inside[q] = pi1[q] < g.N1 && pi2[q] < g.N2;
    goff[q] = inside[q] ? (long)pi1[q] * g.N2 + pi2[q] : 0;
    sG[q] = TGP::at(0, l1, l2); sA[q] = TAP::at(0, l1, l2);
  }

  ZStoredAdjAccQ<P, TGP, TAP, NZG, NZA, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  __syncthreads();

  // loop state kept incrementally, as in the forward kernel
  int rog[NZG - 1], spareg = (NZG - 1) * TGP::SIZE, roa[NZA - 1], sparea = (NZA - 1) * TAP::SIZE;
#pragma unroll
  for (int i = 0; i < NZG - 1; ++i) rog[i] = i * TGP::SIZE;
#pragma unroll
  for (int i = 0; i < NZA - 1; ++i) roa[i] = i * TAP::SIZE;
  PlaneCursor<P::PER0> cg, ca;      // the planes that enter the rings next
  cg.init(za + GH0 + 1, g.N0, plane);
  ca.init(za + AH0 + 1, g.N0, plane);
  long ko = za * plane;
  const long rcur = (long)trc * g.S;

  constexpr int NFA = AF_TILE ? P::NF : 1;
  for (int k = za; k < zb; ++k) {
    const bool more = k + 1 < zb;
    // loads at the top of the step that uses them (a step earlier: 27 registers more, one wave per SIMD less, slower)
    fv stg[P::NE][EG], sta[NFA][EA];      // planes k + 1 + halo of the rings: in flight during the step
#ifdef PDE_AMD_ZDIAG_NOLOAD
#pragma unroll
    for (int e = 0; e < P::NE; ++e)
#pragma unroll
      for (int j = 0; j < EG; ++j) stg[e][j] = vec_zero<W>() + (float)k;
    if (false) {
#else
    if (more) {
#endif
      if (gon) {
#pragma unroll
        for (int e = 0; e < P::NE; ++e) psg.fetch(stg[e], rp[e] + rcur + cg.off);
      }
      if (AF_TILE) {
#pragma unroll
        for (int f = 0; f < P::NF; ++f) psa.fetch(sta[f], fb[f] + ca.off);
      }
    }
    const bool gnow = cg.on, anow = ca.on;
    cg.advance(g.N0, plane);
    if (AF_TILE) ca.advance(g.N0, plane);
    // this lane's quads at plane k: the fields (when read at the output point only) and the residuals of the other frames
    fv fqq[QPT][P::NF], gcq[QPT][NG][P::NE];
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      if (AF_REG) {
#pragma unroll
        for (int f = 0; f < P::NF; ++f) fqq[q][f] = *reinterpret_cast<const fv*>(fb[f] + ko + goff[q]);
      }
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        if (s == P::GTLO) continue;
        if (!fon[s]) continue;      // no such residual frame (or all its weights are zero): nothing to read
        const int trs = tr0 + s - P::GTLO;
#pragma unroll
        for (int e = 0; e < P::NE; ++e) gcq[q][s][e] = *reinterpret_cast<const fv*>(rp[e] + (long)trs * g.S + ko + goff[q]);
      }
    }
#ifdef PDE_AMD_ZDIAG_NOLOAD      // ablation (wrong results): no global loads inside the loop
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) fqq[q][f] = vec_zero<W>() + (float)(k + f);
#pragma unroll
      for (int s = 0; s < NG; ++s)
#pragma unroll
        for (int e = 0; e < P::NE; ++e) gcq[q][s][e] = vec_zero<W>() + (float)(k + e + s);
    }
#endif
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      if (inside[q]) {
        acc.i0 = k; acc.i1 = pi1[q];
#pragma unroll
        for (int i = 0; i < NZG - 1; ++i) acc.ringg[i] = rog[i];
#pragma unroll
        for (int i = 0; i < NZA - 1; ++i) acc.ringa[i] = roa[i];
        acc.siteg = s_g + sG[q];
        acc.sitea = s_a + sA[q];
#pragma unroll
        for (int s = 0; s < NG; ++s) {
          if (s == P::GTLO) continue;
#pragma unroll
          for (int e = 0; e < P::NE; ++e) {
            const float cfk = cf[s][e];
            acc.gc[s][e] = fon[s] && cfk != 0.0f ? gcq[q][s][e] * cfk : vec_zero<W>();
          }
        }
        if (AF_REG) {
#pragma unroll
          for (int f = 0; f < P::NF; ++f) acc.fq[f] = fqq[q][f];
        }
        fv out[P::NF];
#pragma unroll
        for (int px = 0; px < W; ++px) {
          acc.px = px; acc.i2 = pi2[q] + px;
          float gr[P::NF];
#ifdef PDE_AMD_ZDIAG_NOCOMP      // ablation (wrong results): a few taps instead of the adjoint
#pragma unroll
          for (int f = 0; f < P::NF; ++f) gr[f] = acc.G(f % P::NE, 0, -P::GH0, 0, 0) + acc.G(f % P::NE, 0, P::GH0, 0, 0) + acc.G(f % P::NE, 1, 0, 0, 0) + acc.G(f % P::NE, -1, 0, 0, 0) + acc.F(f, 0, 0, 0, 0);
#else
          P::adjoint(acc, gr);
#endif
#pragma unroll
          for (int f = 0; f < P::NF; ++f) out[f][px] = gr[f];
        }
#pragma unroll
#ifdef PDE_AMD_ZDIAG_NOSTORE     // ablation (wrong results): one store per work-group instead of one per quad
        for (int f = 0; f < P::NF; ++f) if (out[f][0] == 1.2345f) __builtin_nontemporal_store(out[f], reinterpret_cast<fv*>(gb[f] + ko + goff[q]));
#else
        for (int f = 0; f < P::NF; ++f) __builtin_nontemporal_store(out[f], reinterpret_cast<fv*>(gb[f] + ko + goff[q]));
#endif
      }
    }
    if (more) {      // into the spare slots; one barrier per plane
#pragma unroll
      for (int e = 0; e < P::NE; ++e) {
        const float cfk = cf[P::GTLO][e];
#pragma unroll
        for (int j = 0; j < EG; ++j) stg[e][j] = stg[e][j] * cfk;
        psg.store(stg[e], s_g + e * NZG * TGP::SIZE + spareg, gon && gnow && cfk != 0.0f);
      }
      if (AF_TILE) {
#pragma unroll
        for (int f = 0; f < P::NF; ++f) psa.store(sta[f], s_a + f * NZA * TAP::SIZE + sparea, anow);
      }
      __syncthreads();
    }
    {
      const int oldest = rog[0];
#pragma unroll
      for (int i = 0; i + 1 < NZG - 1; ++i) rog[i] = rog[i + 1];
      rog[NZG - 2] = spareg;
      spareg = oldest;
    }
    if (AF_TILE) {
      const int oldest = roa[0];
#pragma unroll
      for (int i = 0; i + 1 < NZA - 1; ++i) roa[i] = roa[i + 1];
      roa[NZA - 2] = sparea;
      sparea = oldest;
    }
    ko += plane;
  }
}

}  // namespace pde_amd
