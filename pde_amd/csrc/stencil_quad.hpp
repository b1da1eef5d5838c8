// Time-marching kernels, register-blocked along the fastest axis ("quad" kernels).
//
// A lane owns W = 4 or 2 consecutive points of a row (a "quad"; W is a template parameter).  Global loads
// and stores are 4W bytes per lane, LDS tiles are read as aligned W-vectors (ds_read_b128 / ds_read_b64:
// twice the bytes per LDS cycle of dword reads and 1/W of the instructions), and the neighbours along the
// row come out of the lane's own registers.  W = 4 needs half the lanes of W = 2 for a tile but about twice
// the registers per lane, so it runs at half the waves per SIMD: which one wins depends on the PDE.
// The generated residual / adjoint code is unchanged: it is evaluated once per point with the point's
// position inside the quad (`px`) known at compile time, so every stencil reference folds to
// "element e of the vector at compile-time offset d", and identical vector loads are merged.
//
// Requirements (checked on the host, else the dword kernels of stencil_kernels.hpp run): the fastest
// extent is a multiple of W and every field pointer is 4W-byte aligned.
#pragma once
#include "stencil_kernels.hpp"

namespace pde_amd {

template <int W> struct VecOf;
template <> struct VecOf<1> { typedef float type __attribute__((ext_vector_type(1))); };
template <> struct VecOf<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int W>
__device__ __forceinline__ typename VecOf<W>::type vec_zero() {
  typename VecOf<W>::type z;
#pragma unroll
  for (int i = 0; i < W; ++i) z[i] = 0.0f;
  return z;
}

// true in every lane iff the predicate holds in all active lanes of the wave (a scalar-register result)
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0; }

template <int W> constexpr int padw(int h) { return (h + W - 1) / W * W; }

template <bool PER, int W>
__device__ __forceinline__ int wrap_quad(int c, int N, bool& ok) {
  if (PER) {
    if (c < 0) c += N;
    else if (c >= N) c -= N;
    c = min(max(c, 0), N - W);
  } else {
    if (c < 0 || c >= N) { ok = false; c = 0; }
  }
  return c;
}

// Global memory through buffer instructions: a scalar descriptor (base pointer, size) plus a scalar byte offset
// (the frame) plus one 32-bit lane offset.  Plain pointer arithmetic makes the compiler keep a 64-bit address per
// lane and field, two registers each, because it folds the loop-invariant part of "base + frame + lane" into the
// lane term; the buffer form keeps everything uniform in scalar registers.  `bytes` (< 4 GiB, checked by the
// host) bounds the lane offset; reads past it return zero and writes are dropped.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
template <int W> struct BufIO;
template <> struct BufIO<2> {
  typedef unsigned uv __attribute__((ext_vector_type(2)));
  typedef VecOf<2>::type fv;
  static __device__ __forceinline__ fv load(__amdgpu_buffer_rsrc_t r, unsigned lane, unsigned frame) {
    return __builtin_bit_cast(fv, __builtin_amdgcn_raw_buffer_load_b64(r, lane, frame, 0));
  }
  static __device__ __forceinline__ void store_nt(fv v, __amdgpu_buffer_rsrc_t r, unsigned lane, unsigned frame) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uv, v), r, lane, frame, 2);
  }
  static __device__ __forceinline__ void store(fv v, __amdgpu_buffer_rsrc_t r, unsigned lane, unsigned frame) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uv, v), r, lane, frame, 0);
  }
};
// The 16-byte stores carry the frame offset in the lane offset, never in the scalar offset register: a store of more
// than 8 bytes per lane reads its data registers over several cycles, and a vector instruction that overwrites them
// right behind it corrupts what is stored.  The compiler spaces the two apart on its own only when the store has no
// scalar offset register; with one it assumes the hardware interlocks, which gfx950 was measured not to do once two
// waves share a SIMD (first data register of the last lanes of each 16 written with the next value).
template <> struct BufIO<4> {
  typedef unsigned uv __attribute__((ext_vector_type(4)));
  typedef VecOf<4>::type fv;
  static __device__ __forceinline__ fv load(__amdgpu_buffer_rsrc_t r, unsigned lane, unsigned frame) {
    return __builtin_bit_cast(fv, __builtin_amdgcn_raw_buffer_load_b128(r, lane, frame, 0));
  }
  static __device__ __forceinline__ void store_nt(fv v, __amdgpu_buffer_rsrc_t r, unsigned lane, unsigned frame) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uv, v), r, lane + frame, 0, 2);
  }
  static __device__ __forceinline__ void store(fv v, __amdgpu_buffer_rsrc_t r, unsigned lane, unsigned frame) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uv, v), r, lane + frame, 0, 0);
  }
};

// Per-thread vector elements of one halo tile; offsets are computed once per workgroup.
template <class TL, bool P0, bool P1, bool P2, int NT, int W>
struct StagerQ {
  using fv = typename VecOf<W>::type;
  static_assert(TL::D2 % W == 0 && TL::H2 % W == 0, "quad tiles need x extents in multiples of the lane width");
  static constexpr int QROW = TL::D2 / W;
  static constexpr int NQ = TL::D0 * TL::D1 * QROW;
  static constexpr int EPT = (NQ + NT - 1) / NT;
  // Byte offset of each element inside a frame, as 32 bits (the host checks that a frame is smaller than 2 GiB):
  // with a wave-uniform frame pointer the load then takes "scalar base + 32-bit lane offset" and no 64-bit
  // per-lane address has to be kept.  Bit 31 marks an element that lies outside a non-periodic domain.
  unsigned off[EPT];
  static constexpr unsigned OUTSIDE = 0x80000000u;
  bool plain[EPT];       // wave-uniform: no lane of the wave has to substitute zeros for element j
  __device__ __forceinline__ void init(int o0, int o1, int o2, const Geom& g) {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = threadIdx.x + j * NT;
      const int lq = i % QROW, rr = i / QROW, l1 = rr % TL::D1, l0 = rr / TL::D1;
      bool ok = i < NQ;
      const int c0 = wrap_coord<P0>(o0 + l0 - TL::H0, g.N0, ok);
      const int c1 = wrap_coord<P1>(o1 + l1 - TL::H1, g.N1, ok);
      const int c2 = wrap_quad<P2, W>(o2 + W * lq - TL::H2, g.N2, ok);
      off[j] = ok ? 4u * (unsigned)((c0 * g.N1 + c1) * g.N2 + c2) : OUTSIDE;
      plain[j] = wave_all(ok || i >= NQ);
    }
  }
  __device__ __forceinline__ void fetch(fv (&v)[EPT], const float* __restrict__ frame) const {
    // always a valid address, and no use of the result here: the loads stay in flight until store()
#pragma unroll
    for (int j = 0; j < EPT; ++j) v[j] = *reinterpret_cast<const fv*>(reinterpret_cast<const char*>(frame) + (off[j] & ~OUTSIDE));
  }
  // the same through a buffer descriptor of the sample's frames; `frame` is the byte offset of the frame
  __device__ __forceinline__ void fetch(fv (&v)[EPT], __amdgpu_buffer_rsrc_t r, unsigned frame) const {
#pragma unroll
    for (int j = 0; j < EPT; ++j) v[j] = BufIO<W>::load(r, off[j] & ~OUTSIDE, frame);
  }
  // An unconditional use of the fetched registers.  The compiler places its wait for a load at the first
  // use; when every use sits under a branch, the path around the branch leaves the load "pending" in the
  // compiler's bookkeeping and the next iteration drains the whole memory queue before it may reuse the
  // registers.  Call this on every path before the conditional store().
  __device__ __forceinline__ void touch(const fv (&v)[EPT]) const {
#pragma unroll
    for (int j = 0; j < EPT; ++j) asm volatile("" ::"v"(v[j]));
  }
  // ALLPLAIN: the caller knows that no element of any lane lies outside the domain (periodic axes)
  template <bool ALLPLAIN = false>
  __device__ __forceinline__ void store(const fv (&v)[EPT], float* __restrict__ tile) const {
    touch(v);      // elements past the tile end are fetched but not stored
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int i = threadIdx.x + j * NT;
      const fv z = vec_zero<W>();
      if (EPT * NT == NQ || i < NQ) {
        // wall-side quads read 0; the select is skipped (a real branch) where no lane of the wave needs it
        if (ALLPLAIN || plain[j]) *reinterpret_cast<fv*>(tile + W * i) = v[j];
        else *reinterpret_cast<fv*>(tile + W * i) = (off[j] & OUTSIDE) ? z : v[j];
      }
    }
  }
};

// Which quad of its row a lane owns.  With 16 float4 quads per row a wave spans 4 rows whose LDS images
// are S1 floats apart; ds_read_b128 is serviced in fixed 16-lane groups that mix lanes of two rows, and the
// groups are conflict-free only if lane l sits in 16-byte slot (l mod 16).  Rotating the ownership by the
// row skew restores that (the row still maps to one contiguous 256-byte segment of global memory).
// ds_read_b64 is serviced in 32-lane halves, which are whole rows when a row has 32 pairs: no rotation.
template <int QR, int S1, int W>
__device__ __forceinline__ int owned_quad() {
  const int k = threadIdx.x % QR;
  if (W != 4 || QR != 16) return k;
  const int rw = (threadIdx.x % 64) / QR;
  return (k + rw * (QR - S1 / 4)) & (QR - 1);
}

// element (px + o2) of the row, as "vector number qd, lane e"
#define PDE_AMD_QUAD_SPLIT(col, qd, e) \
  const int qd = (col) >= 0 ? (col) / W : -((W - 1 - (col)) / W); \
  const int e = (col) - W * qd;

// LDS image of the time ring: field f owns NS consecutive tile slots when the residuals read it at other frames,
// else a single slot (it is only ever needed at the frame being evaluated).
template <class P, int NS, bool COMPACT>
struct RingLayout {
  static constexpr int slots(int f) { return (!COMPACT || P::tuse(f)) ? NS : 1; }
  static constexpr int base(int f) { return f == 0 ? 0 : base(f - 1) + slots(f - 1); }     // in tiles
  static constexpr int TILES = COMPACT ? base(P::NF) : NS * P::NF;
  // tile index of field f at ring position r (r = frame % NS); the plain layout is position-major
  static __device__ __forceinline__ int tile(int f, int r) { return COMPACT ? base(f) + (slots(f) == 1 ? 0 : r) : r * P::NF + f; }
  // the frame a single-slot field holds while the ring fields hold `frame` as their newest: THI behind it
  static constexpr bool single(int f) { return COMPACT && !P::tuse(f); }
};

template <class P, class TL, class TX, int NS, int W, bool COMPACT = false>
struct RingAccQ : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  using RL = RingLayout<P, NS, COMPACT>;
  const float* site;         // &ring[tile 0][row][quad start]
  int ring[NS];              // ring position of frame tau - TLO + i
  const float* ldsx;
  int px;                    // which of the lane's 4 points
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(site + RL::tile(f, ring[ot + P::TLO]) * TL::SIZE + o0 * TL::S0 + o1 * TL::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(ldsx + x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float G(int, int, int, int, int) const { return 0.0f; }
};

template <class P, int W>
struct QuadCfg {
  static_assert(W == 2 || W == 4, "lane width");
  // the same tile is covered by 4/W times the lanes
  static constexpr int NT = P::NTQ * 4 / W;
  static constexpr int NTB = P::NTQB * 4 / W;  // the adjoint may use another workgroup size than the forward kernel
  static constexpr int WPS = P::WPSQ;          // waves per SIMD the adjoint is compiled for
  // forward tile
  using TL = Tile<P::QF0, P::QF1, P::QF2, cmax(P::RH0, P::RT0), cmax(P::RH1, P::RT1), padw<W>(cmax(P::RH2, P::RT2))>;
  using TXF = Tile<P::QF0, P::QF1, P::QF2, P::XH0, P::XH1, padw<W>(P::XH2)>;
  static constexpr int NS = P::TLO + P::THI + 1;
  static constexpr long FWD_BYTES = 4L * (RingLayout<P, NS, true>::TILES * TL::SIZE + (P::NA ? P::NA * TXF::SIZE : 0));
  static constexpr bool FWD_OK = FWD_BYTES <= 64 * 1024 && P::QF2 % W == 0 && (P::QF0 * P::QF1 * (P::QF2 / W)) % NT == 0 &&
                                 NT % (P::QF2 / W) == 0 && NT <= 1024;
  // backward tiles
  // The region is evaluated on whole quads, so its outermost padding columns reach up to
  // padw<W>(GH2) + RH2 - HB2 < 16 floats past a row end; those reads stay inside the LDS array (front and
  // back pads of LDS_PAD floats) and only feed G columns that the gather never uses.
  static constexpr int HB0 = cmax(P::GH0 + P::RH0, P::AH0), HB1 = cmax(P::GH1 + P::RH1, P::AH1),
                       HB2 = padw<W>(cmax(P::GH2 + P::RH2, P::AH2));
  static constexpr int LDS_PAD = 16;
  static_assert(padw<W>(P::GH2) + P::RH2 - HB2 + W <= LDS_PAD, "region overreach exceeds the LDS pad");
  using TF = Tile<P::QB0, P::QB1, P::QB2, HB0, HB1, HB2>;
  using TG = Tile<P::QB0, P::QB1, P::QB2, P::GH0, P::GH1, padw<W>(P::GH2)>;
  using TXB = Tile<P::QB0, P::QB1, P::QB2, cmax(P::XH0, P::GH0), cmax(P::XH1, P::GH1), padw<W>(cmax(P::XH2, padw<W>(P::GH2)))>;
  static constexpr long BWD_BYTES = 4L * (3 * P::NF * TF::SIZE + 2 * LDS_PAD + P::NE * TG::SIZE + (P::NA ? P::NA * TXB::SIZE : 0));
  static constexpr bool BWD_OK = MarchOK<P>::value && BWD_BYTES <= LDS_BUDGET && P::QB2 % W == 0 &&
                                 (P::QB0 * P::QB1 * (P::QB2 / W)) % NTB == 0 && NTB % (P::QB2 / W) == 0 && NTB <= 1024;
};

// ------------------------------------------------------------------------------------------------
// forward, marching, quads: workgroup = (sample, chunk of residual frames, tile)
// ------------------------------------------------------------------------------------------------
// FULL: as in the adjoint below, the host has checked that no quad of any lane can leave the lattice.
template <class P, int W, bool WRITE_R, bool FULL = false>
__global__ __launch_bounds__((QuadCfg<P, W>::NT), P::WPSF) void fwd_march_q_kernel(const Args a) {
  using C = QuadCfg<P, W>;
  using fv = typename VecOf<W>::type;
  using TL = typename C::TL;
  using TX = typename C::TXF;
  constexpr int NS = C::NS;
  constexpr bool AUX_ON = P::NA > 0;
  constexpr int NT = C::NT;
  constexpr int QR = TL::T2 / W;                        // quads per tile row
  constexpr int QPT = (TL::T0 * TL::T1 * QR) / NT;      // quads (consecutive rows) per thread
  using ST = StagerQ<TL, P::PER0, P::PER1, P::PER2, NT, W>;

  using RL = RingLayout<P, NS, true>;
  __shared__ __attribute__((aligned(16))) float s_f[RL::TILES * TL::SIZE];
  __shared__ __attribute__((aligned(16))) float s_x[AUX_ON ? P::NA * TX::SIZE : 4];
  __shared__ double s_red[(NT / 64) * P::NE];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int nchunk = (g.TR + a.chunk - 1) / a.chunk;
  const int tile = w % ntile, rest = w / ntile;
  const int ch = rest % nchunk, b = rest / nchunk;
  const int ra = P::TLO + ch * a.chunk;
  const int rb = min(ra + a.chunk, g.T - P::THI);
  const int o2 = (tile % g.nt2) * TL::T2, o1 = ((tile / g.nt2) % g.nt1) * TL::T1, o0 = (tile / (g.nt2 * g.nt1)) * TL::T0;

  ST st;
  st.init(o0, o1, o2, g);
  const float* fb[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) fb[f] = a.f[f] + (long)b * g.T * g.S;
  // buffer descriptors of this sample's frames; a frame is addressed by its byte offset
  const unsigned sample_bytes = 4u * (unsigned)(g.T * g.S), frame_bytes = 4u * (unsigned)g.S;
  __amdgpu_buffer_rsrc_t fbuf[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) fbuf[f] = buffer_of(fb[f], sample_bytes);

  {   // all first frames are fetched before any is stored: one memory round trip instead of one per frame
    fv v0[NS][P::NF][ST::EPT];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int f = 0; f < P::NF; ++f)
        if (!RL::single(f) || i == P::TLO) st.fetch(v0[i][f], fbuf[f], (unsigned)(ra - P::TLO + i) * frame_bytes);   // single-slot fields: frame ra only
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
      for (int f = 0; f < P::NF; ++f)
        if (!RL::single(f) || i == P::TLO) st.store(v0[i][f], s_f + RL::tile(f, (ra - P::TLO + i) % NS) * TL::SIZE);
  }
  if (AUX_ON) {
    using SX = StagerQ<TX, P::PER0, P::PER1, P::PER2, NT, W>;
    SX sx;
    sx.init(o0, o1, o2, g);
#pragma unroll
    for (int x = 0; x < P::NA; ++x) {
      fv v[SX::EPT];
      sx.fetch(v, a.aux[x] + (long)b * g.S);
      sx.store(v, s_x + x * TX::SIZE);
    }
  }
  __syncthreads();

  RingAccQ<P, TL, TX, NS, W, true> acc;
  acc.template set_uniforms<true>(a.c, a.sp, b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;

  int site[QPT], tsite[QPT], pi0[QPT], pi1[QPT], pi2[QPT];
  bool rowok[QPT], wfull[QPT];      // wfull: wave-uniform, every point of every lane's quad lies on the lattice
#pragma unroll
  for (int q = 0; q < QPT; ++q) {
    const int row = (threadIdx.x / QR) * QPT + q;
    const int l2 = W * owned_quad<QR, TL::S1, W>(), l1 = row % TL::T1, l0 = row / TL::T1;
    pi0[q] = o0 + l0; pi1[q] = o1 + l1; pi2[q] = o2 + l2;
    site[q] = TL::at(l0, l1, l2);
    tsite[q] = TX::at(l0, l1, l2);
    rowok[q] = pi0[q] < g.N0 && pi1[q] < g.N1 && pi0[q] >= g.lo0 && pi0[q] < g.N0 - g.hi0 && pi1[q] >= g.lo1 && pi1[q] < g.N1 - g.hi1;
    wfull[q] = wave_all(rowok[q] && pi2[q] >= g.lo2 && pi2[q] + W - 1 < g.N2 - g.hi2);
  }

  double sum[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) sum[k] = 0.0;

  float wnext[P::NE];      // weights are fetched one step ahead so that their latency hides behind a step
#pragma unroll
  for (int k = 0; k < P::NE; ++k) wnext[k] = ldc(a.wt, k * g.TR + (ra - P::TLO));

  for (int tau = ra; tau < rb; ++tau) {
    const bool more = tau + 1 < rb;
    const int fnew = tau + P::THI + 1;
    fv stage[P::NF][ST::EPT];
    if (more) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) st.fetch(stage[f], fbuf[f], (unsigned)(RL::single(f) ? tau + 1 : fnew) * frame_bytes);
    }
    float wk[P::NE];
#pragma unroll
    for (int k = 0; k < P::NE; ++k) {
      wk[k] = wnext[k];
      wnext[k] = ldc(a.wt, k * g.TR + min(tau + 1 - P::TLO, g.TR - 1));
    }
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      acc.i0 = pi0[q]; acc.i1 = pi1[q];
#pragma unroll
      for (int i = 0; i < NS; ++i) acc.ring[i] = (tau - P::TLO + i) % NS;
      acc.site = s_f + site[q];
      acc.ldsx = s_x + tsite[q];
      float part[P::NE];
#pragma unroll
      for (int k = 0; k < P::NE; ++k) part[k] = 0.0f;
      float Rq[W][P::NE];
#pragma unroll
      for (int px = 0; px < W; ++px) {
        acc.px = px; acc.i2 = pi2[q] + px;
        P::residual(acc, Rq[px]);
      }
      if (!FULL && !wfull[q]) {      // off-lattice points contribute nothing
#pragma unroll
        for (int px = 0; px < W; ++px) {
          const int i2 = pi2[q] + px;
          const bool ok = rowok[q] && i2 >= g.lo2 && i2 < g.N2 - g.hi2;
#pragma unroll
          for (int k = 0; k < P::NE; ++k) Rq[px][k] = ok ? Rq[px][k] : 0.0f;
        }
      }
#pragma unroll
      for (int px = 0; px < W; ++px)
#pragma unroll
        for (int k = 0; k < P::NE; ++k) part[k] += Rq[px][k] * Rq[px][k];
      if (WRITE_R && (FULL || (pi0[q] < g.N0 && pi1[q] < g.N1 && pi2[q] < g.N2))) {      // stored mode: residuals on the full grid, zeros off the lattice
        const long ro = ((long)b * g.TR + (tau - P::TLO)) * g.S + ((long)pi0[q] * g.N1 + pi1[q]) * g.N2 + pi2[q];
#pragma unroll
        for (int k = 0; k < P::NE; ++k) {
          fv r;
#pragma unroll
          for (int px = 0; px < W; ++px) r[px] = Rq[px][k];
          *reinterpret_cast<fv*>(a.rout[k] + ro) = r;
        }
      }
#pragma unroll
      for (int k = 0; k < P::NE; ++k) sum[k] += (double)part[k] * (double)wk[k];
    }
#pragma unroll
    for (int f = 0; f < P::NF; ++f) st.touch(stage[f]);
    if (more) {
      __syncthreads();
#pragma unroll
      for (int f = 0; f < P::NF; ++f) st.template store<FULL>(stage[f], s_f + RL::tile(f, fnew % NS) * TL::SIZE);
      __syncthreads();
    }
  }
  __syncthreads();
  block_sum_store<P::NE, NT>(sum, s_red, a.partial + (long)w * P::NE);
}

// ------------------------------------------------------------------------------------------------
// backward, marching with recomputation, quads (see bwd_march_kernel for the schedule)
// ------------------------------------------------------------------------------------------------
template <class P, class TF, class TG, class TX, int W>
struct MarchAdjAccQ : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  const float* ldsg;
  const float* ldsf;
  const float* ldsx;
  float gm[MAXE], gp[MAXE];
  int px;
  __device__ __forceinline__ float G(int k, int ot, int o0, int o1, int o2) const {
    if (ot != 0) return ot < 0 ? gm[k] : gp[k];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(ldsg + k * TG::SIZE + o0 * TG::S0 + o1 * TG::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(ldsf + f * TF::SIZE + o0 * TF::S0 + o1 * TF::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(ldsx + x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + W * qd);
    return v[e];
  }
};

// Diagnostic build only (-DPDE_AMD_STAMPS): per-phase shader-clock stamps of the first workgroups go to the
// workspace pointer, which no other code of the kernel reads.  Never defined in the shipped library.
#ifdef PDE_AMD_STAMPS
#define PDE_AMD_STAMP(slot) \
  do { if (a.partial && blockIdx.x < 16 && (threadIdx.x & 63) == 0 && (n - n0) < 20) { \
    __builtin_amdgcn_sched_barrier(0); \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0); \
    reinterpret_cast<unsigned long long*>(a.partial)[(((blockIdx.x * 8 + (threadIdx.x >> 6)) * 20) + (n - n0)) * 8 + (slot)] = t_; } } while (0)
#else
#define PDE_AMD_STAMP(slot) do {} while (0)
#endif

// WITH_LOSS: the kernel also leaves the forward's partial sums (sum of weight * R^2 over the tile, frames of this
// chunk) in a.partial: the residuals are recomputed here anyway, so a training step needs no separate forward launch.
// FULL: the host has checked that the grid is periodic along every axis with a halo, a whole number of tiles, and not
// cropped: every quad of every lane lies on the lattice, so the kernel carries no masks and no branches around them.
template <class P, int W, bool WITH_LOSS = false, bool FULL = false>
__global__ __launch_bounds__((QuadCfg<P, W>::NTB), (QuadCfg<P, W>::WPS)) void bwd_march_q_kernel(const Args a) {
  using C = QuadCfg<P, W>;
  using fv = typename VecOf<W>::type;
  using TF = typename C::TF;
  using TG = typename C::TG;
  using TX = typename C::TXB;
  constexpr bool AUX_ON = P::NA > 0;
  constexpr int NT = C::NTB;
  constexpr int QR = TG::T2 / W;                         // quads per tile row
  constexpr int QPT = (TG::T0 * TG::T1 * QR) / NT;       // output quads per thread
  constexpr int GQR = TG::D2 / W;                        // quads per region row
  constexpr int NRQ = TG::D0 * TG::D1 * GQR;             // region quads
  constexpr int RPT = (NRQ + NT - 1) / NT;
  // The last round of the residual stage covers what is left of the region after RPT-1 full rounds.  When that
  // is at most half a workgroup, the lower and the upper half of the workgroup take it on alternate steps, so
  // the extra work lands on all SIMDs in turn instead of always on the ones that run the first waves.
  // Other-frame adjoint terms that read no fields are linear in the scaled residuals with uniform coefficients:
  // the earlier frame's share is added as soon as it is known, and only one past frame of residuals is kept.
  constexpr bool SPLIT_T = P::TIME_AF == 0;
#ifdef PDE_AMD_ABL_NO_TIME      // diagnostic: drop the other-frame terms (wrong gradients, two fewer registers sets)
  constexpr bool NO_TIME = true;
#else
  constexpr bool NO_TIME = false;
#endif
#ifdef PDE_AMD_VGU
  constexpr bool VGU = PDE_AMD_VGU;
#else
  constexpr bool VGU = true;
#endif
  constexpr bool ALT = RPT >= 2 && NT % 128 == 0 && NRQ - (RPT - 1) * NT <= NT / 2;
  // Own quad first: when every lane owns exactly one output quad and the rest of the region (the rim) fits one more
  // round, round 0 evaluates the residuals at the lane's own output quad and the last round covers the rim.  The lane's
  // own scaled residuals then never come back from LDS, and only rim quads are excluded from the loss sums.
#ifdef PDE_AMD_NO_OWN_FIRST
  constexpr bool OWN = false;
#else
  constexpr bool OWN = QPT == 1 && RPT == 2 && TG::T0 == 1 && TG::D0 == 1 && TG::H2 % W == 0;
#endif
  // The residual code of the own quad hands subexpressions to the gather of the same step (a derivative at the site
  // costs registers across one barrier instead of LDS reads and arithmetic twice).
#ifdef PDE_AMD_NO_CARRY
  constexpr bool CARRY = false;
#else
  constexpr bool CARRY = OWN && P::NCAR > 0;
#endif
  constexpr int HQ = TG::H2 / W;                          // rim quads at either end of a row
  constexpr int NRIM = NRQ - NT;                          // with OWN: quads of the rim
  using ST = StagerQ<TF, P::PER0, P::PER1, P::PER2, NT, W>;
  static_assert(C::BWD_OK, "PDE not eligible for the quad marching adjoint");

  __shared__ __attribute__((aligned(16))) float s_fpad[3 * P::NF * TF::SIZE + 2 * C::LDS_PAD];
  __shared__ __attribute__((aligned(16))) float s_g[P::NE * TG::SIZE];
  float* const s_f = s_fpad + C::LDS_PAD;
  __shared__ __attribute__((aligned(16))) float s_x[AUX_ON ? P::NA * TX::SIZE : 4];
  __shared__ double s_red[WITH_LOSS ? (NT / 64) * P::NE : 1];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int nchunk = (g.T + a.chunk - 1) / a.chunk;
  const int tile = w % ntile, rest = w / ntile;
  const int ch = rest % nchunk, b = rest / nchunk;
  const int qa = ch * a.chunk, qb = min(qa + a.chunk, g.T);
  const int o2 = (tile % g.nt2) * TG::T2, o1 = ((tile / g.nt2) % g.nt1) * TG::T1, o0 = (tile / (g.nt2 * g.nt1)) * TG::T0;

  ST st;
  st.init(o0, o1, o2, g);
  const float* fb[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) fb[f] = a.f[f] + (long)b * g.T * g.S;
  // buffer descriptors of this sample's frames (fields) and gradient frames; frames are addressed by byte offset
  const unsigned sample_bytes = 4u * (unsigned)(g.T * g.S), frame_bytes = 4u * (unsigned)g.S;
  __amdgpu_buffer_rsrc_t fbuf[P::NF], gbuf[P::NF];
#pragma unroll
  for (int f = 0; f < P::NF; ++f) {
    fbuf[f] = buffer_of(fb[f], sample_bytes);
    gbuf[f] = buffer_of(a.g[f] + (long)b * g.T * g.S, sample_bytes);
  }
  float up[P::NE];
  upstream_of<P>(a, b, up);

  const int na = max(qa - 1, 1), nb = min(qb, g.T - 2);
  const int n0 = qa - 1;
  // one frame at a time: fetching all three at once measured slower here (the workgroups of a CU fall into step)
  for (int fr = max(n0 - 1, 0); fr <= min(n0 + 1, g.T - 1); ++fr) {
#pragma unroll
    for (int f = 0; f < P::NF; ++f) {
      fv v[ST::EPT];
      st.fetch(v, fbuf[f], (unsigned)fr * frame_bytes);
      st.store(v, s_f + ((fr % 3) * P::NF + f) * TF::SIZE);
    }
  }
  if (AUX_ON) {
    using SX = StagerQ<TX, P::PER0, P::PER1, P::PER2, NT, W>;
    SX sx;
    sx.init(o0, o1, o2, g);
#pragma unroll
    for (int x = 0; x < P::NA; ++x) {
      fv v[SX::EPT];
      sx.fetch(v, a.aux[x] + (long)b * g.S);
      sx.store(v, s_x + x * TX::SIZE);
    }
  }

  // region quads of this thread (tile plus the adjoint's reach, x reach rounded up to whole quads)
  int rF[RPT], rG[RPT], rX[RPT], r0[RPT], r1[RPT], r2[RPT];
  bool rrow[RPT], rfull[RPT];      // rfull: wave-uniform, every point of every active lane's quad lies on the lattice
  bool rint[RPT];                  // the quad belongs to the tile itself and to the grid: its residuals count for the loss
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int iraw = (ALT && j == RPT - 1 ? (int)threadIdx.x % (NT / 2) : (int)threadIdx.x) + j * NT;
    int i = iraw < NRQ ? iraw : 0;
    if (OWN) {      // region-linear index of the lane's own quad (round 0) or of rim quad number iraw - NT (round 1)
      if (j == 0) {
        i = ((int)threadIdx.x / QR + TG::H1) * GQR + HQ + owned_quad<QR, TG::S1, W>();
      } else {
        const int r = iraw < NRQ ? iraw - NT : 0;
        constexpr int NROWS = 2 * TG::H1 * GQR;           // quads of the full rows above and below the tile
        if (r < NROWS) {
          const int rr = r / GQR;
          i = (rr < TG::H1 ? rr : TG::T1 + rr) * GQR + r % GQR;
        } else {
          const int r2 = r - NROWS, c = HQ > 0 ? r2 % (2 * (HQ > 0 ? HQ : 1)) : 0, row = HQ > 0 ? r2 / (2 * (HQ > 0 ? HQ : 1)) : 0;
          i = (TG::H1 + row) * GQR + (c < HQ ? c : QR + c);
        }
      }
    }
    const int lq = i % GQR, rr = i / GQR;
    const int l2 = W * lq - TG::H2, l1 = rr % TG::D1 - TG::H1, l0 = rr / TG::D1 - TG::H0;
    bool ok = iraw < NRQ;
    const int c0 = wrap_coord<P::PER0>(o0 + l0, g.N0, ok);
    const int c1 = wrap_coord<P::PER1>(o1 + l1, g.N1, ok);
    const int c2 = wrap_quad<P::PER2, W>(o2 + l2, g.N2, ok);
    r0[j] = c0; r1[j] = c1; r2[j] = c2;
    rrow[j] = ok && c0 >= g.lo0 && c0 < g.N0 - g.hi0 && c1 >= g.lo1 && c1 < g.N1 - g.hi1;
    rfull[j] = wave_all(iraw >= NRQ || (rrow[j] && c2 >= g.lo2 && c2 + W - 1 < g.N2 - g.hi2));
    rint[j] = iraw < NRQ && l0 >= 0 && l0 < TG::T0 && l1 >= 0 && l1 < TG::T1 && l2 >= 0 && l2 < TG::T2 &&
              o0 + l0 < g.N0 && o1 + l1 < g.N1 && o2 + l2 < g.N2;
    rG[j] = W * i;
    rF[j] = TF::at(l0, l1, l2);
    rX[j] = TX::at(l0, l1, l2);
  }
  int sF[QPT], sG[QPT], sX[QPT], pi0[QPT], pi1[QPT], pi2[QPT];
  unsigned goff[QPT];      // byte offset inside a frame (32 bits: scalar base + lane offset addressing, see StagerQ)
  bool inside[QPT];
#pragma unroll
  for (int q = 0; q < QPT; ++q) {
    const int row = (threadIdx.x / QR) * QPT + q;
    const int l2 = W * owned_quad<QR, TG::S1, W>(), l1 = row % TG::T1, l0 = row / TG::T1;
    pi0[q] = o0 + l0; pi1[q] = o1 + l1; pi2[q] = o2 + l2;
    inside[q] = pi0[q] < g.N0 && pi1[q] < g.N1 && pi2[q] < g.N2;
    sF[q] = TF::at(l0, l1, l2); sG[q] = TG::at(l0, l1, l2); sX[q] = TX::at(l0, l1, l2);
    goff[q] = 4u * (unsigned)((pi0[q] * g.N1 + pi1[q]) * g.N2 + pi2[q]);
  }

  RingAccQ<P, TF, TX, 3, W> racc;
  racc.template set_uniforms<VGU>(a.c, a.sp, b);
  MarchAdjAccQ<P, TF, TG, TX, W> aacc;
  aacc.vg = VGU;
#pragma unroll
  for (int i = 0; i < P::NC; ++i) aacc.cv[i] = racc.cv[i];
#pragma unroll
  for (int i = 0; i < P::NSP; ++i) aacc.spv[i] = racc.spv[i];
#pragma unroll
  for (int i = 0; i < P::NU; ++i) aacc.uv[i] = racc.uv[i];
  racc.N0 = aacc.N0 = g.N0; racc.N1 = aacc.N1 = g.N1; racc.N2 = aacc.N2 = g.N2;

  float Sprev[QPT][W][P::NF];
  float Gm2[QPT][W][P::NE];
  float Gm1[QPT][W][P::NE];
#pragma unroll
  for (int q = 0; q < QPT; ++q)
#pragma unroll
    for (int px = 0; px < W; ++px) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) Sprev[q][px][f] = 0.0f;
#pragma unroll
      for (int k = 0; k < P::NE; ++k) { Gm2[q][px][k] = 0.0f; Gm1[q][px][k] = 0.0f; }
    }
  __syncthreads();

  float wnext[P::NE];      // weights are fetched one step ahead so that their latency hides behind a step
#pragma unroll
  for (int k = 0; k < P::NE; ++k) wnext[k] = ldc(a.wt, k * g.TR + min(max(n0 - 1, 0), g.TR - 1));
  // Loss sums of this lane: fp32 over runs of 16 steps (non-negative terms, a few points per step), handed to an fp64
  // sum at the end of every run, so that the error does not grow with the length of a clip; the workgroup's and every
  // later reduction are fp64.
  float lsum[P::NE];
  double lsum64[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) { lsum[k] = 0.0f; lsum64[k] = 0.0; }

  // Frames are fetched two steps before the step that needs them in LDS: the loads of frame n+3 are issued
  // at the top of step n into one of two register sets and written to the ring at the end of step n+1, so a
  // load has two steps of compute to hide behind.  The loop is unrolled by two so that both sets are
  // statically named registers.
  // The fetches are unconditional (past the end they re-read the last frame, an L2 hit, and the result is
  // dropped): a fetch under a branch would make the compiler count, on the path around it, fewer loads in
  // flight than there are, and its wait for the older set would then drain the younger set as well.
#ifdef PDE_AMD_SKEW
  if ((blockIdx.x / 256) & 1) { for (int i = 0; i < PDE_AMD_SKEW; ++i) __builtin_amdgcn_s_sleep(16); }
#endif
  // DEEP = false (PDEs whose adjoint already fills the register file): one set, fetched at the top of the step
  // that stores it.
  constexpr bool DEEP = P::QPD >= 2;
  fv stage_a[P::NF][ST::EPT], stage_b[P::NF][ST::EPT];
  if (DEEP) {
#pragma unroll
    for (int f = 0; f < P::NF; ++f) st.fetch(stage_b[f], fbuf[f], (unsigned)min(n0 + 2, g.T - 1) * frame_bytes);
  }

  auto step = [&](const int n, const int half, fv (&stage_new)[P::NF][ST::EPT], fv (&stage)[P::NF][ST::EPT]) __attribute__((always_inline)) {
    const int fnew = n + 2;
    const bool more = n < qb && fnew < g.T;
    PDE_AMD_STAMP(0);
#pragma unroll
    for (int f = 0; f < P::NF; ++f) {
      if (DEEP) st.fetch(stage_new[f], fbuf[f], (unsigned)min(n + 3, g.T - 1) * frame_bytes);
      else st.fetch(stage[f], fbuf[f], (unsigned)min(n + 2, g.T - 1) * frame_bytes);
    }
    PDE_AMD_STAMP(1);

    // G(n) = coef * R(n) on the tile plus the adjoint's reach
    const bool gvalid = n >= na && n <= nb;
    float cf[P::NE], wl[P::NE], lpart[P::NE];
#pragma unroll
    for (int k = 0; k < P::NE; ++k) lpart[k] = 0.0f;
    const bool lvalid = WITH_LOSS && gvalid && n >= qa && n < qb;      // every residual frame counts in exactly one chunk
#pragma unroll
    for (int k = 0; k < P::NE; ++k) {
      wl[k] = lvalid ? wnext[k] : 0.0f;
      cf[k] = to_vgpr(gvalid ? up[k] * wnext[k] : 0.0f);
      wnext[k] = ldc(a.wt, k * g.TR + min(max(n, 0), g.TR - 1));
    }
    fv Gown[P::NE];      // OWN: the scaled residuals of the lane's own quad
    float Cown[W][CARRY ? P::NCAR : 1];      // and the values its residual code hands to the same-frame adjoint
#pragma unroll
    for (int jj = 0; jj < RPT; ++jj) {
      const int j = OWN ? RPT - 1 - jj : jj;      // OWN: the rim first, so that the own quad's values are born last
      const bool mine = ALT ? ((int)threadIdx.x / (NT / 2) == half && (int)threadIdx.x % (NT / 2) + j * NT < NRQ)
                            : ((int)threadIdx.x + j * NT < NRQ);
      if (RPT * NT == NRQ || j < RPT - 1 || mine) {   // whole waves drop out of the last round
        fv Gq[P::NE];
#pragma unroll
        for (int k = 0; k < P::NE; ++k) Gq[k] = vec_zero<W>();
#ifndef PDE_AMD_ABL_NO_G
        if (gvalid)
#else
        if (gvalid && a.chunk < 0)
#endif
        {
          racc.i0 = r0[j]; racc.i1 = r1[j];
#pragma unroll
          for (int i = 0; i < 3; ++i) racc.ring[i] = (n + 2 + i) % 3;
          racc.site = s_f + rF[j];
          racc.ldsx = s_x + rX[j];
#pragma unroll
          for (int px = 0; px < W; ++px) {
            racc.px = px; racc.i2 = r2[j] + px;
            float R[P::NE];
            if (CARRY && j == 0) P::residual_c(racc, R, Cown[px]);
            else P::residual(racc, R);
#pragma unroll
            for (int k = 0; k < P::NE; ++k) Gq[k][px] = WITH_LOSS ? R[k] : cf[k] * R[k];
          }
          if (!FULL && !rfull[j]) {      // off-lattice points carry no residual
#pragma unroll
            for (int px = 0; px < W; ++px) {
              const int i2 = r2[j] + px;
              const bool ok = rrow[j] && i2 >= g.lo2 && i2 < g.N2 - g.hi2;
#pragma unroll
              for (int k = 0; k < P::NE; ++k) Gq[k][px] = ok ? Gq[k][px] : 0.0f;
            }
          }
          if (WITH_LOSS) {      // Gq holds the residuals: their squares first, then the scaled residuals
            if (OWN ? (j == 0 && rint[0]) : rint[j]) {      // fp32 over the few points a lane sees in one step, fp64 across steps
#pragma unroll
              for (int k = 0; k < P::NE; ++k)
#pragma unroll
                for (int px = 0; px < W; ++px) lpart[k] += Gq[k][px] * Gq[k][px];
            }
#pragma unroll
            for (int k = 0; k < P::NE; ++k) Gq[k] = Gq[k] * cf[k];
          }
        }
#pragma unroll
        for (int k = 0; k < P::NE; ++k) *reinterpret_cast<fv*>(s_g + k * TG::SIZE + rG[j]) = Gq[k];
        if (OWN && j == 0) {
#pragma unroll
          for (int k = 0; k < P::NE; ++k) Gown[k] = Gq[k];
        }
      }
    }
    if (WITH_LOSS) {
#pragma unroll
      for (int k = 0; k < P::NE; ++k) lsum[k] = fmaf(lpart[k], wl[k], lsum[k]);
      if ((n & 15) == 15) {      // uniform over the workgroup
#pragma unroll
        for (int k = 0; k < P::NE; ++k) { lsum64[k] += (double)lsum[k]; lsum[k] = 0.0f; }
      }
    }
    PDE_AMD_STAMP(2);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    PDE_AMD_STAMP(3);

    const bool do_space = n >= qa && n < qb;
    const bool do_out = n - 1 >= qa && n - 1 < qb;
    fv out[QPT][P::NF];      // gradient of frame n-1; stored at the end of the step (see below)
    const float* fcur = s_f + (((n + 3) % 3) * P::NF) * TF::SIZE;
    const float* fprev = s_f + (((n + 2) % 3) * P::NF) * TF::SIZE;
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      aacc.i0 = pi0[q]; aacc.i1 = pi1[q];
      aacc.ldsg = s_g + sG[q]; aacc.ldsx = s_x + sX[q];
      fv Gc[P::NE];
#pragma unroll
      for (int k = 0; k < P::NE; ++k) Gc[k] = OWN ? Gown[k] : *reinterpret_cast<const fv*>(s_g + k * TG::SIZE + sG[q]);
      if (do_out) {
        aacc.ldsf = fprev + sF[q];
#pragma unroll
        for (int px = 0; px < W; ++px) {
          aacc.px = px; aacc.i2 = pi2[q] + px;
#pragma unroll
          for (int k = 0; k < P::NE; ++k) { aacc.gm[k] = SPLIT_T ? 0.0f : Gm2[q][px][k]; aacc.gp[k] = Gc[k][px]; }
          float gt[P::NF];
          if (NO_TIME) { for (int f = 0; f < P::NF; ++f) gt[f] = 0.0f; }
          else if (SPLIT_T) P::adjoint_time_p(aacc, gt);      // the earlier frame's share is already in Sprev
          else P::adjoint_time(aacc, gt);
#pragma unroll
          for (int f = 0; f < P::NF; ++f) out[q][f][px] = Sprev[q][px][f] + gt[f];
        }
      }
      // One branch around all four points: the points share most of their LDS quads, and the loads are
      // only merged when the four evaluations sit in one basic block.
      float gs[W][P::NF];
#pragma unroll
      for (int px = 0; px < W; ++px)
#pragma unroll
        for (int f = 0; f < P::NF; ++f) gs[px][f] = 0.0f;
#ifndef PDE_AMD_ABL_NO_GATHER
      if (do_space && gvalid)      // G(n) is identically zero on frames without a residual: nothing to gather
#else
      if (do_space && a.chunk < 0)
#endif
      {
        aacc.ldsf = fcur + sF[q];
#pragma unroll
        for (int px = 0; px < W; ++px) {
          aacc.px = px; aacc.i2 = pi2[q] + px;
          if (CARRY) P::adjoint_space_c(aacc, Cown[px], gs[px]);
          else P::adjoint_space(aacc, gs[px]);
        }
      }
#pragma unroll
      for (int px = 0; px < W; ++px) {
        if (SPLIT_T && !NO_TIME) {      // frame n's gradient so far: its own frame's terms plus the share of frame n-1's residuals
          aacc.px = px; aacc.i2 = pi2[q] + px;      // the coefficient may sit in an auxiliary field (1 / c(x)^2 of the Gaussian-speed wave equation)
#pragma unroll
          for (int k = 0; k < P::NE; ++k) aacc.gm[k] = Gm1[q][px][k];
          float gt[P::NF];
          P::adjoint_time_m(aacc, gt);
#pragma unroll
          for (int f = 0; f < P::NF; ++f) Sprev[q][px][f] = gs[px][f] + gt[f];
        } else {
#pragma unroll
          for (int f = 0; f < P::NF; ++f) Sprev[q][px][f] = gs[px][f];
        }
#pragma unroll
        for (int k = 0; k < P::NE; ++k) { Gm2[q][px][k] = Gm1[q][px][k]; Gm1[q][px][k] = NO_TIME ? 0.0f : Gc[k][px]; }
      }
    }
    PDE_AMD_STAMP(4);
    // the slot of frame n-1 is overwritten next; only the other-frame adjoint terms may still be reading it
    if (P::TIME_AF) __syncthreads();
    PDE_AMD_STAMP(5);
#pragma unroll
    for (int f = 0; f < P::NF; ++f) st.touch(stage[f]);
    if (more) {
#pragma unroll
      for (int f = 0; f < P::NF; ++f) st.template store<FULL>(stage[f], s_f + ((fnew % 3) * P::NF + f) * TF::SIZE);
    }
    PDE_AMD_STAMP(6);
    // The gradient stores go out last: the wait for the prefetched frame above then covers loads only, and
    // the stores have the whole next step to drain before anything waits on the vector-memory counter again.
    if (do_out) {
#pragma unroll
      for (int q = 0; q < QPT; ++q) {
        if (FULL || inside[q]) {
#pragma unroll
          for (int f = 0; f < P::NF; ++f)      // streamed: keeps the fields in cache
            BufIO<W>::store_nt(out[q][f], gbuf[f], goff[q], (unsigned)(n - 1) * frame_bytes);
        }
      }
    }
    if (more) __syncthreads();
    // Wave priority: the fetch and the residual stage of the next step outrank the gather and the stores of the other
    // workgroup on the same SIMD.  The residual stage ends at a barrier the whole workgroup waits at, the gather's
    // results are only needed at the end of the step: with the arbiter preferring the former the barrier wait
    // shrinks (adjoint 136 -> 131 us on Taylor-Green; the opposite preference gains nothing).
    __builtin_amdgcn_s_setprio(2);
    PDE_AMD_STAMP(7);
  };
  // Both halves always run (a step past qb finds nothing to do: no residual frame, no output, no store).
  // A branch around the second half would be a path on which, in the compiler's bookkeeping, the first
  // half's loads are still in flight at the loop head, and it would drain the memory queue there.
  for (int n = n0; n <= qb; n += 2) {
    step(n, 0, stage_a, stage_b);
    step(n + 1, 1, stage_b, stage_a);      // without DEEP the sets are independent scratch and one of them is optimised away
  }
  if (WITH_LOSS) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P::NE; ++k) lsum64[k] += (double)lsum[k];
    block_sum_store<P::NE, NT>(lsum64, s_red, a.partial + (long)w * P::NE);
  }
}

// whole-tile staging helpers of the per-frame kernels (synchronous: fetch, scale, store)
template <class TL, bool P0, bool P1, bool P2, int NT, int W>
__device__ __forceinline__ void stage_tile(float* tile, const float* __restrict__ src, int o0, int o1, int o2, const Geom& g, float scale) {
  using fv = typename VecOf<W>::type;
  StagerQ<TL, P0, P1, P2, NT, W> st;
  st.init(o0, o1, o2, g);
  fv v[StagerQ<TL, P0, P1, P2, NT, W>::EPT];
  st.fetch(v, src);
#pragma unroll
  for (int j = 0; j < StagerQ<TL, P0, P1, P2, NT, W>::EPT; ++j) v[j] = v[j] * scale;
  st.store(v, tile);
}

// Staging of N tiles of one shape with every load in flight before the first LDS write: the workgroup is exposed to
// the memory latency once, not once per tile, and the offsets are computed once.
template <class TL, bool P0, bool P1, bool P2, int NT, int W, int N>
struct MultiStage {
  using St = StagerQ<TL, P0, P1, P2, NT, W>;
  using fv = typename VecOf<W>::type;
  St st;
  fv v[N][St::EPT];
  __device__ __forceinline__ void init(int o0, int o1, int o2, const Geom& g) { st.init(o0, o1, o2, g); }
  __device__ __forceinline__ void fetch(int n, const float* __restrict__ src) { st.fetch(v[n], src); }
  // tile = scale * fetched values; `on` false (wave-uniform) stores zeros whatever was fetched
  __device__ __forceinline__ void store(int n, float* __restrict__ tile, float scale, bool on = true) {
#pragma unroll
    for (int j = 0; j < St::EPT; ++j) v[n][j] = on ? v[n][j] * scale : vec_zero<W>();
    st.store(v[n], tile);
  }
};

template <class TL, int NT, int W>
__device__ __forceinline__ void zero_tile_q(float* tile) {
  using fv = typename VecOf<W>::type;
  static_assert(TL::SIZE % W == 0, "tile size");
  for (int i = threadIdx.x; i < TL::SIZE / W; i += NT) *reinterpret_cast<fv*>(tile + W * i) = vec_zero<W>();
}

// ------------------------------------------------------------------------------------------------
// forward, one residual frame per workgroup, quads: for PDEs whose time ring of halo tiles does not fit LDS
// (3-D halos).  The frame's tile + halo is staged in LDS; the neighbouring frames are staged with the (smaller)
// halo they need, or, when they need none, read as one vector per field straight from global memory.
// ------------------------------------------------------------------------------------------------
template <class P, int W>
struct FrameCfg {
  static constexpr int NT = P::NTQ * 4 / W;
  static constexpr int NTB = P::NTQB * 4 / W;  // the adjoint may use another workgroup size than the forward kernel
  using TL = Tile<P::QF0, P::QF1, P::QF2, P::RH0, P::RH1, padw<W>(P::RH2)>;
  using TT = Tile<P::QF0, P::QF1, P::QF2, P::RT0, P::RT1, padw<W>(P::RT2)>;
  using TX = Tile<P::QF0, P::QF1, P::QF2, P::XH0, P::XH1, padw<W>(P::XH2)>;
  static constexpr bool TT_ON = (P::RT0 | P::RT1 | P::RT2) != 0;
  static constexpr int NTS = P::TLO + P::THI;
  static constexpr long BYTES = 4L * (P::NF * TL::SIZE + (TT_ON ? NTS * P::NF * TT::SIZE : 0) + (P::NA ? P::NA * TX::SIZE : 0));
  static constexpr bool OK = BYTES <= LDS_BUDGET && P::QF2 % W == 0 && (P::QF0 * P::QF1 * (P::QF2 / W)) % NT == 0 &&
                             NT % (P::QF2 / W) == 0 && NT <= 1024;
};

template <class P, class TL, class TT, class TX, int W>
struct FrameAccQ : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  static constexpr int NTS = P::TLO + P::THI;
  const float* lds;
  const float* ldst;
  const float* ldsx;
  fv fc[NTS > 0 ? NTS : 1][MAXF];     // this quad at the other frames when they need no halo
  int px;
  __device__ __forceinline__ float F(int f, int ot, int o0, int o1, int o2) const {
    if (ot == 0) {
      PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
      const fv v = *reinterpret_cast<const fv*>(lds + f * TL::SIZE + o0 * TL::S0 + o1 * TL::S1 + W * qd);
      return v[e];
    }
    const int slot = ot < 0 ? ot + P::TLO : ot + P::TLO - 1;
    if ((P::RT0 | P::RT1 | P::RT2) != 0) {
      PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
      const fv v = *reinterpret_cast<const fv*>(ldst + (slot * P::NF + f) * TT::SIZE + o0 * TT::S0 + o1 * TT::S1 + W * qd);
      return v[e];
    }
    return fc[slot][f][px];
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(ldsx + x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float G(int, int, int, int, int) const { return 0.0f; }
};

template <class P, int W, bool WRITE_R>
__global__ __launch_bounds__((FrameCfg<P, W>::NT)) void fwd_frame_q_kernel(const Args a) {
  using C = FrameCfg<P, W>;
  using TL = typename C::TL;
  using TT = typename C::TT;
  using TX = typename C::TX;
  using fv = typename VecOf<W>::type;
  constexpr bool TT_ON = C::TT_ON, AUX_ON = P::NA > 0;
  constexpr int NT = C::NT, NTS = C::NTS;
  constexpr int QR = TL::T2 / W;
  constexpr int QPT = (TL::T0 * TL::T1 * QR) / NT;
  static_assert(C::OK, "PDE not eligible for the per-frame quad forward");

  __shared__ __attribute__((aligned(16))) float s_f[P::NF * TL::SIZE];
  __shared__ __attribute__((aligned(16))) float s_t[TT_ON ? NTS * P::NF * TT::SIZE : 4];
  __shared__ __attribute__((aligned(16))) float s_x[AUX_ON ? P::NA * TX::SIZE : 4];
  __shared__ double s_red[((NT + 63) / 64) * P::NE];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int tile = w % ntile, rest = w / ntile;
  const int tr = rest % g.TR, b = rest / g.TR;
  const int t = tr + P::TLO;
  const int o2 = (tile % g.nt2) * TL::T2, o1 = ((tile / g.nt2) % g.nt1) * TL::T1, o0 = (tile / (g.nt2 * g.nt1)) * TL::T0;
  const long frame = ((long)b * g.T + t) * g.S;

  // every global load of the workgroup is issued before the first wait: the frame's halo tiles and (when the other
  // frames need no halo) this lane's own quads of those frames
  MultiStage<TL, P::PER0, P::PER1, P::PER2, NT, W, P::NF> ms;
  ms.init(o0, o1, o2, g);
#pragma unroll
  for (int f = 0; f < P::NF; ++f) ms.fetch(f, a.f[f] + frame);
  fv fcq[QPT][NTS > 0 ? NTS : 1][P::NF];
  if (!TT_ON) {
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      const int row = (threadIdx.x / QR) * QPT + q;
      const int l2 = W * owned_quad<QR, TL::S1, W>(), l1 = row % TL::T1, l0 = row / TL::T1;
      const int i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
      const bool inside = i0 < g.N0 && i1 < g.N1 && i2 < g.N2;
      const long goff = inside ? ((long)i0 * g.N1 + i1) * g.N2 + i2 : 0;
#pragma unroll
      for (int s = 0; s < NTS; ++s) {
        const int ot = s < P::TLO ? s - P::TLO : s - P::TLO + 1;
#pragma unroll
        for (int f = 0; f < P::NF; ++f) fcq[q][s][f] = *reinterpret_cast<const fv*>(a.f[f] + frame + (long)ot * g.S + goff);
      }
    }
  }
#pragma unroll
  for (int f = 0; f < P::NF; ++f) ms.store(f, s_f + f * TL::SIZE, 1.0f);
  if (TT_ON) {
#pragma unroll
    for (int s = 0; s < NTS; ++s) {
      const int ot = s < P::TLO ? s - P::TLO : s - P::TLO + 1;
#pragma unroll
      for (int f = 0; f < P::NF; ++f)
        stage_tile<TT, P::PER0, P::PER1, P::PER2, NT, W>(s_t + (s * P::NF + f) * TT::SIZE, a.f[f] + frame + (long)ot * g.S, o0, o1, o2, g, 1.0f);
    }
  }
  if (AUX_ON) {
#pragma unroll
    for (int x = 0; x < P::NA; ++x) stage_tile<TX, P::PER0, P::PER1, P::PER2, NT, W>(s_x + x * TX::SIZE, a.aux[x] + (long)b * g.S, o0, o1, o2, g, 1.0f);
  }
  __syncthreads();

  FrameAccQ<P, TL, TT, TX, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;
  float wk[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) wk[k] = ldc(a.wt, k * g.TR + tr);
  double sum[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) sum[k] = 0.0;

#pragma unroll
  for (int q = 0; q < QPT; ++q) {
    const int row = (threadIdx.x / QR) * QPT + q;
    const int l2 = W * owned_quad<QR, TL::S1, W>(), l1 = row % TL::T1, l0 = row / TL::T1;
    const int i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
    const bool inside = i0 < g.N0 && i1 < g.N1 && i2 < g.N2;
    const bool rowok = inside && i0 >= g.lo0 && i0 < g.N0 - g.hi0 && i1 >= g.lo1 && i1 < g.N1 - g.hi1;
    const long goff = inside ? ((long)i0 * g.N1 + i1) * g.N2 + i2 : 0;
    acc.i0 = i0; acc.i1 = i1;
    acc.lds = s_f + TL::at(l0, l1, l2);
    acc.ldst = s_t + TT::at(l0, l1, l2);
    acc.ldsx = s_x + TX::at(l0, l1, l2);
    if (!TT_ON) {
#pragma unroll
      for (int s = 0; s < NTS; ++s) {
#pragma unroll
        for (int f = 0; f < P::NF; ++f) acc.fc[s][f] = fcq[q][s][f];
      }
    }
    float Rq[W][P::NE];
#pragma unroll
    for (int px = 0; px < W; ++px) {
      acc.px = px; acc.i2 = i2 + px;
      P::residual(acc, Rq[px]);
      const bool ok = rowok && acc.i2 >= g.lo2 && acc.i2 < g.N2 - g.hi2;
#pragma unroll
      for (int k = 0; k < P::NE; ++k) Rq[px][k] = ok ? Rq[px][k] : 0.0f;
    }
    float part[P::NE];
#pragma unroll
    for (int k = 0; k < P::NE; ++k) {
      part[k] = 0.0f;
#pragma unroll
      for (int px = 0; px < W; ++px) part[k] += Rq[px][k] * Rq[px][k];
      sum[k] += (double)part[k] * (double)wk[k];
    }
    if (WRITE_R && inside) {
      const long ro = ((long)b * g.TR + tr) * g.S + goff;
#pragma unroll
      for (int k = 0; k < P::NE; ++k) {
        fv r;
#pragma unroll
        for (int px = 0; px < W; ++px) r[px] = Rq[px][k];
        *reinterpret_cast<fv*>(a.rout[k] + ro) = r;
      }
    }
  }
  block_sum_store<P::NE, NT>(sum, s_red, a.partial + (long)w * P::NE);
}

// ------------------------------------------------------------------------------------------------
// backward from stored residuals, quads: workgroup = (sample, output frame, tile).  G = coef * R tiles of the frames
// the adjoint reaches are staged in LDS with vector loads; other-frame terms that need no halo read their quad
// straight from global memory.
// ------------------------------------------------------------------------------------------------
template <class P, int W>
struct StoredCfg {
  static constexpr int NT = P::NTQ * 4 / W;
  static constexpr int NTB = P::NTQB * 4 / W;  // the adjoint may use another workgroup size than the forward kernel
  using TG = Tile<P::QB0, P::QB1, P::QB2, P::GH0, P::GH1, padw<W>(P::GH2)>;
  using TGT = Tile<P::QB0, P::QB1, P::QB2, P::GT0, P::GT1, padw<W>(P::GT2)>;
  using TA = Tile<P::QB0, P::QB1, P::QB2, P::AH0, P::AH1, padw<W>(P::AH2)>;
  using TX = Tile<P::QB0, P::QB1, P::QB2, P::XH0, P::XH1, padw<W>(P::XH2)>;
  static constexpr bool GT_ON = (P::GT0 | P::GT1 | P::GT2) != 0;
  static constexpr int NGT = P::GTLO + P::GTHI;
  // fields that the adjoint reads at the output point only (no spatial reach) stay in registers: no tile
  static constexpr bool AF_TILE = P::ANY_AF != 0 && (P::AH0 | P::AH1 | P::AH2) != 0;
  static constexpr long BYTES = 4L * (P::NE * TG::SIZE + (GT_ON ? NGT * P::NE * TGT::SIZE : 0) + (AF_TILE ? P::NF * TA::SIZE : 0) +
                                      (P::NA ? P::NA * TX::SIZE : 0));
  static constexpr bool OK = P::ATLO == 0 && P::ATHI == 0 && (P::AT0 | P::AT1 | P::AT2) == 0 && BYTES <= LDS_BUDGET && P::QB2 % W == 0 &&
                             (P::QB0 * P::QB1 * (P::QB2 / W)) % NT == 0 && NT % (P::QB2 / W) == 0 && NT <= 1024;
};

template <class P, class TG, class TGT, class TA, class TX, int W>
struct StoredAdjAccQ : SiteBase<P> {
  using fv = typename VecOf<W>::type;
  static constexpr int NG = P::GTLO + P::GTHI + 1;
  const float* ldsg;
  const float* ldsgt;
  const float* ldsa;
  const float* ldsx;
  fv gc[NG][MAXE];           // this quad's G at the other frames when those terms need no halo
  fv fq[MAXF];               // this quad's fields when the adjoint reads them at the output point only
  int px;
  __device__ __forceinline__ float G(int k, int ot, int o0, int o1, int o2) const {
    if (ot == 0) {
      PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
      const fv v = *reinterpret_cast<const fv*>(ldsg + k * TG::SIZE + o0 * TG::S0 + o1 * TG::S1 + W * qd);
      return v[e];
    }
    if ((P::GT0 | P::GT1 | P::GT2) != 0) {
      const int slot = ot < 0 ? ot + P::GTLO : ot + P::GTLO - 1;
      PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
      const fv v = *reinterpret_cast<const fv*>(ldsgt + (slot * P::NE + k) * TGT::SIZE + o0 * TGT::S0 + o1 * TGT::S1 + W * qd);
      return v[e];
    }
    return gc[ot + P::GTLO][k][px];
  }
  __device__ __forceinline__ float F(int f, int, int o0, int o1, int o2) const {
    if ((P::AH0 | P::AH1 | P::AH2) == 0) return fq[f][px];
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(ldsa + f * TA::SIZE + o0 * TA::S0 + o1 * TA::S1 + W * qd);
    return v[e];
  }
  __device__ __forceinline__ float X(int x, int o0, int o1, int o2) const {
    PDE_AMD_QUAD_SPLIT(px + o2, qd, e)
    const fv v = *reinterpret_cast<const fv*>(ldsx + x * TX::SIZE + o0 * TX::S0 + o1 * TX::S1 + W * qd);
    return v[e];
  }
};

template <class P, int W>
__global__ __launch_bounds__((StoredCfg<P, W>::NT)) void bwd_stored_q_kernel(const Args a) {
  using C = StoredCfg<P, W>;
  using TG = typename C::TG;
  using TGT = typename C::TGT;
  using TA = typename C::TA;
  using TX = typename C::TX;
  using fv = typename VecOf<W>::type;
  constexpr bool GT_ON = C::GT_ON, AF_ON = C::AF_TILE, AF_REG = P::ANY_AF != 0 && !C::AF_TILE, AUX_ON = P::NA > 0;
  constexpr int NT = C::NT, NGT = C::NGT, NG = P::GTLO + P::GTHI + 1;
  constexpr int QR = TG::T2 / W;
  constexpr int QPT = (TG::T0 * TG::T1 * QR) / NT;
  static_assert(C::OK, "PDE not eligible for the quad stored-residual adjoint");

  __shared__ __attribute__((aligned(16))) float s_g[P::NE * TG::SIZE];
  __shared__ __attribute__((aligned(16))) float s_gt[GT_ON ? NGT * P::NE * TGT::SIZE : 4];
  __shared__ __attribute__((aligned(16))) float s_a[AF_ON ? P::NF * TA::SIZE : 4];
  __shared__ __attribute__((aligned(16))) float s_x[AUX_ON ? P::NA * TX::SIZE : 4];

  const Geom& g = a.geo;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = g.nt0 * g.nt1 * g.nt2;
  const int tile = w % ntile, rest = w / ntile;
  const int tq = rest % g.T, b = rest / g.T;
  const int o2 = (tile % g.nt2) * TG::T2, o1 = ((tile / g.nt2) % g.nt1) * TG::T1, o0 = (tile / (g.nt2 * g.nt1)) * TG::T0;
  const long frame = ((long)b * g.T + tq) * g.S;
  const int tr0 = tq - P::TLO;
  const float* rp[P::NE];
#pragma unroll
  for (int k = 0; k < P::NE; ++k) rp[k] = a.r[k] + (long)b * g.TR * g.S;

  StoredAdjAccQ<P, TG, TGT, TA, TX, W> acc;
  acc.template set_uniforms<true>(a.c, a.sp, b);
  acc.N0 = g.N0; acc.N1 = g.N1; acc.N2 = g.N2;

  float up[P::NE];
  upstream_of<P>(a, b, up);
  float cf[NG][P::NE];
#pragma unroll
  for (int s = 0; s < NG; ++s) {
    const int trs = tr0 + (s - P::GTLO);
    const bool ok = trs >= 0 && trs < g.TR;
#pragma unroll
    for (int k = 0; k < P::NE; ++k) cf[s][k] = ok ? up[k] * ldc(a.wt, k * g.TR + max(min(trs, g.TR - 1), 0)) : 0.0f;
  }

  // Every global load of the workgroup is issued before the first wait.  A residual frame outside the range has a
  // zero coefficient: its load is aimed at the nearest frame inside and the value is dropped.
  const int trc = max(min(tr0, g.TR - 1), 0);
  MultiStage<TG, P::PER0, P::PER1, P::PER2, NT, W, P::NE> msg;
  msg.init(o0, o1, o2, g);
#pragma unroll
  for (int k = 0; k < P::NE; ++k) msg.fetch(k, rp[k] + (long)trc * g.S);
  MultiStage<TGT, P::PER0, P::PER1, P::PER2, NT, W, GT_ON ? NGT * P::NE : 1> msgt;
  if (GT_ON) {
    msgt.init(o0, o1, o2, g);
#pragma unroll
    for (int s = 0; s < NGT; ++s) {
      const int ot = s < P::GTLO ? s - P::GTLO : s - P::GTLO + 1;
      const int trs = max(min(tr0 + ot, g.TR - 1), 0);
#pragma unroll
      for (int k = 0; k < P::NE; ++k) msgt.fetch(s * P::NE + k, rp[k] + (long)trs * g.S);
    }
  }
  MultiStage<TA, P::PER0, P::PER1, P::PER2, NT, W, AF_ON ? P::NF : 1> msa;
  if (AF_ON) {
    msa.init(o0, o1, o2, g);
#pragma unroll
    for (int f = 0; f < P::NF; ++f) msa.fetch(f, a.f[f] + frame);
  }
  fv gcq[QPT][NG][P::NE];
  fv fqq[QPT][P::NF];
  if (!GT_ON || AF_REG) {
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
      const int row = (threadIdx.x / QR) * QPT + q;
      const int l2 = W * owned_quad<QR, TG::S1, W>(), l1 = row % TG::T1, l0 = row / TG::T1;
      const int i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
      const bool inside = i0 < g.N0 && i1 < g.N1 && i2 < g.N2;
      const long goff = inside ? ((long)i0 * g.N1 + i1) * g.N2 + i2 : 0;
      if (AF_REG) {
#pragma unroll
        for (int f = 0; f < P::NF; ++f) fqq[q][f] = *reinterpret_cast<const fv*>(a.f[f] + frame + goff);
      }
#pragma unroll
      for (int s = 0; s < NG; ++s) {
        if (GT_ON || s == P::GTLO) continue;
        const int trs = max(min(tr0 + s - P::GTLO, g.TR - 1), 0);
#pragma unroll
        for (int k = 0; k < P::NE; ++k) gcq[q][s][k] = *reinterpret_cast<const fv*>(rp[k] + (long)trs * g.S + goff);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < P::NE; ++k) {
    const float cfk = cf[P::GTLO][k];      // wave-uniform
    msg.store(k, s_g + k * TG::SIZE, cfk, cfk != 0.0f);
  }
  if (AF_ON) {
#pragma unroll
    for (int f = 0; f < P::NF; ++f) msa.store(f, s_a + f * TA::SIZE, 1.0f);
  }
  if (GT_ON) {
#pragma unroll
    for (int s = 0; s < NGT; ++s) {
      const int ot = s < P::GTLO ? s - P::GTLO : s - P::GTLO + 1;
#pragma unroll
      for (int k = 0; k < P::NE; ++k) {
        const float cfk = cf[ot + P::GTLO][k];
        msgt.store(s * P::NE + k, s_gt + (s * P::NE + k) * TGT::SIZE, cfk, cfk != 0.0f);
      }
    }
  }
  if (AUX_ON) {
#pragma unroll
    for (int x = 0; x < P::NA; ++x) stage_tile<TX, P::PER0, P::PER1, P::PER2, NT, W>(s_x + x * TX::SIZE, a.aux[x] + (long)b * g.S, o0, o1, o2, g, 1.0f);
  }
  __syncthreads();

#pragma unroll
  for (int q = 0; q < QPT; ++q) {
    const int row = (threadIdx.x / QR) * QPT + q;
    const int l2 = W * owned_quad<QR, TG::S1, W>(), l1 = row % TG::T1, l0 = row / TG::T1;
    const int i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
    if (i0 < g.N0 && i1 < g.N1 && i2 < g.N2) {
      const long goff = ((long)i0 * g.N1 + i1) * g.N2 + i2;
      acc.i0 = i0; acc.i1 = i1;
      acc.ldsg = s_g + TG::at(l0, l1, l2);
      acc.ldsgt = s_gt + TGT::at(l0, l1, l2);
      acc.ldsa = s_a + TA::at(l0, l1, l2);
      acc.ldsx = s_x + TX::at(l0, l1, l2);
      if (!GT_ON) {
#pragma unroll
        for (int s = 0; s < NG; ++s) {
          if (s == P::GTLO) continue;
#pragma unroll
          for (int k = 0; k < P::NE; ++k) {
            const float cfk = cf[s][k];
            acc.gc[s][k] = cfk != 0.0f ? gcq[q][s][k] * cfk : vec_zero<W>();
          }
        }
      }
      if (AF_REG) {
#pragma unroll
        for (int f = 0; f < P::NF; ++f) acc.fq[f] = fqq[q][f];
      }
      fv out[P::NF];
#pragma unroll
      for (int px = 0; px < W; ++px) {
        acc.px = px; acc.i2 = i2 + px;
        float gr[P::NF];
        P::adjoint(acc, gr);
#pragma unroll
        for (int f = 0; f < P::NF; ++f) out[f][px] = gr[f];
      }
#pragma unroll
      for (int f = 0; f < P::NF; ++f) __builtin_nontemporal_store(out[f], reinterpret_cast<fv*>(a.g[f] + frame + goff));
    }
  }
}

}  // namespace pde_amd