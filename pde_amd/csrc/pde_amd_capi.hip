// C ABI (include/pde_amd.h) over the stencil kernels: validation, geometry, kernel choice, launches.
#include "../../include/pde_amd.h"
#include "stencil_kernels.hpp"
#include "stencil_quad.hpp"
#include "stencil_zmarch.hpp"
#include "stencil_zcol.hpp"
#include "stencil_f64.hpp"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>

// Partial-sum scratch of pde_amd_residual_forward_backward_static: one copy per GPU, held until the process ends.
// In host code this name is only a handle for hipGetSymbolAddress; the kernels get the per-device address it returns.
static __device__ __align__(256) unsigned char scratch[PDE_AMD_STATIC_SCRATCH_BYTES];

namespace pde_amd {
#include "gen/registry.inc"

static_assert(MAXF == PDE_AMD_MAX_FIELDS && MAXE == PDE_AMD_MAX_EQS && MAXA == PDE_AMD_MAX_AUX &&
              MAXSP == PDE_AMD_MAX_SPARAMS && MAXC == PDE_AMD_MAX_CONSTS, "header and kernel limits differ");

struct PdeVTable {
  pde_amd_pde_info info;
  size_t (*ws_bytes)(const pde_amd_problem*);
  int (*fwd)(const pde_amd_problem*, const float* const*, const float* const*, const float* const*, const float*,
             double*, float*, float* const*, void*, size_t, hipStream_t);
  int (*bwd)(const pde_amd_problem*, const float* const*, const float* const*, const float* const*, const float*,
             const float* const*, const double*, const float* const*, float* const*, void*, size_t, hipStream_t);
  int (*fwd_bwd)(const pde_amd_problem*, const float* const*, const float* const*, const float* const*, const float*,
                 const float*, double*, float*, float* const*, void*, size_t, hipStream_t);
  int (*fwd_f64)(const pde_amd_problem*, const double*, const double* const*, const double* const*, const double* const*,
                 const double*, double*, double*, double* const*, void*, size_t, hipStream_t);
};

// grads[f] *= g / hint, skipped by every workgroup when the two are equal (both are device scalars)
struct ScaleArgs {
  float* p[PDE_AMD_MAX_FIELDS];
  int n;
  long n4;      // float4s (or, unaligned, floats) per field
  const float* g;
  const float* hint;
};
template <int VEC>
__global__ __launch_bounds__(256) void scale_kernel(const ScaleArgs a) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const float r = ldc(a.g, 0) / ldc(a.hint, 0);
  if (r == 1.0f) return;
  for (int f = 0; f < a.n; ++f) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (long)gridDim.x * 256) {
      if (VEC == 4) { f4* q = reinterpret_cast<f4*>(a.p[f]); q[i] = q[i] * r; }
      else a.p[f][i] *= r;
    }
  }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// One-shot timing hook (pde_amd_time_next_adjoint): the next launch of the vector adjoint kernel on this host thread is
// bracketed by these two events on its own stream, nothing else of the call is.
static thread_local hipEvent_t t_ev0 = nullptr, t_ev1 = nullptr;
static thread_local bool t_ev_failed = false;      // a record of the pair failed: reported as PDE_AMD_EEVENT, not as a launch failure
struct AdjointTimer {
  hipStream_t st;
  hipEvent_t e1;
  explicit AdjointTimer(hipStream_t s) : st(s), e1(t_ev1) {
    t_ev_failed = false;
    if (t_ev0 && hipEventRecord(t_ev0, st) != hipSuccess) { t_ev_failed = true; (void)hipGetLastError(); }
    t_ev0 = t_ev1 = nullptr;
  }
  ~AdjointTimer() {
    if (e1 && hipEventRecord(e1, st) != hipSuccess) { t_ev_failed = true; (void)hipGetLastError(); }
  }
};
// Name of the adjoint kernel the last backward / one-pass call on this host thread launched, as a profiler prints it.
static thread_local char t_kernel[192] = "";
static void note_kernel(const char* base, const char* pde, const char* targs) {
  snprintf(t_kernel, sizeof(t_kernel), "%s<pde_amd::Pde_%s%s>", base, pde, targs);
}
// status of a call whose launches are done: a failed launch first, then a failed timing-event record
static inline int launch_status() {
  const bool ev = t_ev_failed;
  t_ev_failed = false;
  if (hipGetLastError() != hipSuccess) return PDE_AMD_EHIP;
  return ev ? PDE_AMD_EEVENT : PDE_AMD_OK;
}

// The register-blocked kernels move 16 bytes per lane: every pointer must be 16-byte aligned and the
// fastest extent a multiple of 4.
template <int N, int W>
static inline bool aligned_vec(const float* const* p) {
  for (int i = 0; i < N; ++i)
    if (reinterpret_cast<uintptr_t>(p[i]) & (4 * W - 1)) return false;
  return true;
}
// The vector kernels address a sample's frames of a field through one buffer descriptor and mark "no cell here" by the
// lane offset 2^31, which must lie beyond the descriptor's range: 2 GiB per sample and field at most.  Larger problems
// take the dword kernels, which use 64-bit addresses.
static inline bool quad_addressable(const pde_amd_problem* p) {
  const long S = (long)p->n[0] * p->n[1] * p->n[2];
  return (long)p->frames * S <= (1L << 29);
}
static inline int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

// Frames per marching workgroup: whole clips when samples x tiles already fill the chip, shorter chunks
// (each re-reads a few frames at its ends) when they do not.
static inline int pick_chunk(long blocks_per_chunking, int frames) {
  const long want = 768;   // about three workgroups per CU
#ifdef PDE_AMD_DIAG_ONLY
  if (getenv("PDE_AMD_CHUNK")) return atoi(getenv("PDE_AMD_CHUNK"));
#endif
  int nchunk = (int)((want + blocks_per_chunking - 1) / blocks_per_chunking);
  if (nchunk < 1) nchunk = 1;
  int chunk = cdiv(frames, nchunk);
  if (chunk < 4) chunk = frames < 4 ? frames : 4;
  return chunk;
}

// Planes per workgroup of the plane-marching 3-D kernels: a chunk re-reads the 2 * reach planes before its first one, so
// chunks are as long as still leaves the chip a few rounds of workgroups, and never shorter than 8 planes.
static inline int pick_plane_chunk(long blocks_per_chunking, int planes) {
#ifdef PDE_AMD_DIAG_ONLY
  if (getenv("PDE_AMD_ZCHUNK")) return atoi(getenv("PDE_AMD_ZCHUNK"));
#endif
  const long want = 4 * 768;
  int nchunk = (int)((want + blocks_per_chunking - 1) / blocks_per_chunking);
  if (nchunk < 1) nchunk = 1;
  int chunk = cdiv(planes, nchunk);
  if (chunk < 8) chunk = planes < 8 ? planes : 8;
  return chunk;
}

template <class P>
struct Ops {
  static void fill_geom(const pde_amd_problem* p, Geom& g, int t0, int t1, int t2) {
    g.B = p->batch; g.T = p->frames; g.TR = p->frames - P::TLO - P::THI;
    g.N0 = p->n[0]; g.N1 = p->n[1]; g.N2 = p->n[2];
    g.lo0 = p->lo[0]; g.lo1 = p->lo[1]; g.lo2 = p->lo[2];
    g.hi0 = p->hi[0]; g.hi1 = p->hi[1]; g.hi2 = p->hi[2];
    g.nt0 = cdiv(g.N0, t0); g.nt1 = cdiv(g.N1, t1); g.nt2 = cdiv(g.N2, t2);
    g.S = (long)g.N0 * g.N1 * g.N2;
  }

  static int check(const pde_amd_problem* p) {
    if (p->n_consts != P::NC) return PDE_AMD_EINVAL;
    if (p->batch < 1 || p->frames < 1 || p->n[0] < 1 || p->n[1] < 1 || p->n[2] < 1) return PDE_AMD_ESHAPE;
    if (P::ND < 3 && p->n[0] != 1) return PDE_AMD_ESHAPE;
    if (P::ND < 2 && p->n[1] != 1) return PDE_AMD_ESHAPE;
    if ((long)p->n[0] * p->n[1] * p->n[2] >= (1L << 31)) return PDE_AMD_ESHAPE;
    // a periodic axis must be at least as long as the widest halo any tile loads along it
    const int h0 = max3(P::RH0 + P::GH0, P::RT0, P::AH0), h1 = max3(P::RH1 + P::GH1, P::RT1, P::AH1),
              h2 = max3(P::RH2 + P::GH2, P::RT2, P::AH2);
    if (P::PER0 && p->n[0] < h0) return PDE_AMD_ESHAPE;
    if (P::PER1 && p->n[1] < h1) return PDE_AMD_ESHAPE;
    if (P::PER2 && p->n[2] < h2) return PDE_AMD_ESHAPE;
    for (int a = 0; a < 3; ++a)
      if (p->lo[a] < 0 || p->hi[a] < 0) return PDE_AMD_EINVAL;
    return PDE_AMD_OK;
  }

  // The marching adjoint's mask-free instantiation applies when no quad of any workgroup can leave the lattice: no crop,
  // a whole number of tiles along every axis, and periodic wrap-around wherever a tile has a halo.
  static bool full_grid(const pde_amd_problem* p) {
    using C = QuadCfg<P, P::QWB>;
    for (int a = 0; a < 3; ++a)
      if (p->lo[a] != 0 || p->hi[a] != 0) return false;
    if (p->n[0] % P::QB0 != 0 || p->n[1] % P::QB1 != 0 || p->n[2] % P::QB2 != 0) return false;
    return (P::PER0 || C::HB0 == 0) && (P::PER1 || C::HB1 == 0) && (P::PER2 || C::HB2 == 0) && P::NA == 0;
  }

  // the same for the marching forward kernel (its tiles and halos)
  static bool full_grid_fwd(const pde_amd_problem* p) {
    using TL = typename QuadCfg<P, P::QWF>::TL;
    for (int a = 0; a < 3; ++a)
      if (p->lo[a] != 0 || p->hi[a] != 0) return false;
    if (p->n[0] % P::QF0 != 0 || p->n[1] % P::QF1 != 0 || p->n[2] % P::QF2 != 0) return false;
    return (P::PER0 || TL::H0 == 0) && (P::PER1 || TL::H1 == 0) && (P::PER2 || TL::H2 == 0) && P::NA == 0;
  }

  static long fwd_tiles(const pde_amd_problem* p) {
    return (long)cdiv(p->n[0], P::FT0) * cdiv(p->n[1], P::FT1) * cdiv(p->n[2], P::FT2);
  }

  // Most partial sums any forward tiling leaves per sample and residual frame: the 3-D tiles, or the plane-marching
  // kernel's (y, x) tiles times its chunks of planes (never shorter than 8 planes: pick_plane_chunk), which outnumber
  // the 3-D tiles on thin grids.
  static long max_items_per_frame(const pde_amd_problem* p) {
    long m = fwd_tiles(p);
    if constexpr (ZCfg<P, P::ZW>::FWD_OK) {
      using C = ZCfg<P, P::ZW>;
      const long zm = (long)cdiv(p->n[0], p->n[0] < 8 ? p->n[0] : 8) * cdiv(p->n[1], C::T1) * cdiv(p->n[2], C::T2);
      if (zm > m) m = zm;
    }
    if constexpr (ZColCfg<P, P::ZCW>::FWD_OK) {      // at most one frame group per residual frame
      using C = ZColCfg<P, P::ZCW>;
      const long zm = (long)cdiv(p->n[0], p->n[0] < 8 ? p->n[0] : 8) * cdiv(p->n[1], C::T1) * cdiv(p->n[2], C::T2);
      if (zm > m) m = zm;
    }
    return m;
  }

  // z-column kernels: frames per workgroup, planes per chunk (a chunk re-reads 2 * reach planes: as long as still gives
  // every CU a workgroup, never shorter than 8 planes), geometry; `items`: workgroups (partial sums) per sample
  struct ZPlan { int nfw, ngroups; long items; };
  template <int LPF_, int FPW_, int T1_, int T2_>
  static ZPlan plan_zcol(const pde_amd_problem* p, Args& a, long tr) {
    ZPlan z;
    z.ngroups = cdiv((int)tr, FPW_);
    z.nfw = cdiv((int)tr, z.ngroups);      // groups of equal length, up to one frame
    z.ngroups = cdiv((int)tr, z.nfw);
    const long yx = (long)cdiv(p->n[1], T1_) * cdiv(p->n[2], T2_);
    const long blocks = (long)p->batch * z.ngroups * yx;
    int nchunk = (int)((256 + blocks - 1) / blocks);
#ifdef PDE_AMD_DIAG_ONLY
    if (getenv("PDE_AMD_ZCHUNK")) nchunk = cdiv(p->n[0], atoi(getenv("PDE_AMD_ZCHUNK")));
#endif
    if (nchunk < 1) nchunk = 1;
    int chunk = cdiv(p->n[0], nchunk);
    if (chunk < 8) chunk = p->n[0] < 8 ? p->n[0] : 8;
    a.chunk = chunk;
    fill_geom(p, a.geo, chunk, T1_, T2_);
    z.items = z.ngroups * a.geo.nt0 * yx;
    return z;
  }

  // KF: the kernel for groups of several frames, KS: for single frames
  template <class KF, class KS>
  static bool launch_zcol(const pde_amd_problem* p, const Args& a, const ZPlan& z, int lpf, KF kfull, KS ksolo, size_t lds, hipStream_t stream) {
    const void* fn = z.nfw == 1 ? reinterpret_cast<const void*>(ksolo) : reinterpret_cast<const void*>(kfull);
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    ZColLaunch zl;
    zl.nfw = z.nfw; zl.ngroups = z.ngroups;
    const unsigned grid = (unsigned)((long)p->batch * z.items);
    if (z.nfw == 1) hipLaunchKernelGGL(ksolo, dim3(grid), dim3(lpf), lds, stream, a, zl);
    else hipLaunchKernelGGL(kfull, dim3(grid), dim3(lpf * z.nfw), lds, stream, a, zl);
    return true;
  }

  // -1: too many workgroups; -2: more partial sums than `cap`; -3: the kernel's LDS could not be granted
  template <bool WRITE_R>
  static long launch_zcol_forward(const pde_amd_problem* p, Args& a, long tr, long cap, hipStream_t stream) {
    if constexpr (ZColCfg<P, P::ZCW>::FWD_OK) {
      using C = ZColCfg<P, P::ZCW>;
      const ZPlan z = plan_zcol<C::LPF, C::FPWF, C::T1, C::T2>(p, a, tr);
      if (p->batch * z.items >= (1L << 31)) return -1;
      if (z.items > cap) return -2;
      if (!launch_zcol(p, a, z, C::LPF, fwd_zcol_kernel<P, P::ZCW, WRITE_R, false>, fwd_zcol_kernel<P, P::ZCW, WRITE_R, true>,
                       C::fwd_lds(z.nfw), stream)) return -3;
      return z.items;
    }
    return -3;
  }

  template <int W = P::ZCW>
  static bool zcol_usable(const pde_amd_problem* p, const float* const* fields) {
    return p->n[2] % W == 0 && aligned_vec<P::NF, W>(fields) && quad_addressable(p);
  }

  static size_t ws_bytes(const pde_amd_problem* p) {
    const long tr = p->frames - P::TLO - P::THI;
    if (tr <= 0) return 256;
    return (size_t)(p->batch * tr * max_items_per_frame(p)) * P::NE * sizeof(double) + 256;
  }

  static void fill_args(Args& a, const pde_amd_problem* p, const float* const* fields, const float* const* aux,
                        const float* const* sparams) {
    memset(&a, 0, sizeof(a));
    for (int f = 0; f < P::NF; ++f) a.f[f] = fields[f];
    for (int x = 0; x < P::NA; ++x) a.aux[x] = aux[x];
    for (int s = 0; s < P::NSP; ++s) a.sp[s] = sparams[s];
    for (int c = 0; c < P::NC; ++c) a.c[c] = p->consts[c];
  }

  // the plane-marching forward runs where the time-marching one does not fit LDS, or where the spec prefers it
  static constexpr bool ZFWD = ZCfg<P, P::ZW>::FWD_OK && (P::ZFIRST != 0 || !QuadCfg<P, P::QWF>::FWD_OK);

  static int fwd(const pde_amd_problem* p, const float* const* fields, const float* const* aux,
                 const float* const* sparams, const float* weights, double* sums, float* totals,
                 float* const* resid_out, void* ws, size_t ws_bytes_given, hipStream_t stream) {
    int st = check(p);
    if (st) return st;
    if (!sums) return PDE_AMD_EINVAL;
    const long tr = p->frames - P::TLO - P::THI;
    if (tr <= 0) {   // no residual frame: the sums are empty sums
      (void)hipMemsetAsync(sums, 0, sizeof(double) * p->batch * P::NE, stream);
      if (totals) (void)hipMemsetAsync(totals, 0, sizeof(float) * (P::NE + 1), stream);
      return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP;
    }
    if (!fields || !weights || (P::NA && !aux) || (P::NSP && !sparams)) return PDE_AMD_EINVAL;
    if (!ws || ws_bytes_given < ws_bytes(p)) return PDE_AMD_EWORKSPACE;
    Args a;
    fill_args(a, p, fields, aux, sparams);
    fill_geom(p, a.geo, P::FT0, P::FT1, P::FT2);
    a.wt = weights;
    a.partial = reinterpret_cast<double*>(ws);
    const long tiles = fwd_tiles(p);
    // partial sums per sample the workspace holds (the caller may have passed more than ws_bytes(p); the bound used here
    // is the one ws_bytes promises)
    const long cap = tr * max_items_per_frame(p);
    long items = 0;   // partial sums per sample
    bool zc_done = false;      // the z-column forward without residual output ran
    if constexpr (ZColCfg<P, P::ZCW>::FWD_OK) {
      if (!resid_out && P::ZCOLV != 0 && zcol_usable(p, fields)) {
        items = launch_zcol_forward<false>(p, a, tr, cap, stream);
        if (items == -2) return PDE_AMD_EWORKSPACE;
        if (items == -1) return PDE_AMD_ESHAPE;
        zc_done = items > 0;
      }
    }
    if (zc_done) {
    } else if (resid_out) {
      for (int k = 0; k < P::NE; ++k) {
        if (!resid_out[k]) return PDE_AMD_EINVAL;
        a.rout[k] = resid_out[k];
      }
      bool launched = false;
      if constexpr (ZColCfg<P, P::ZCW>::FWD_OK) {
        if (P::ZCOLF != 0 && zcol_usable(p, fields) && aligned_vec<P::NE, P::ZCW>(resid_out)) {
          items = launch_zcol_forward<true>(p, a, tr, cap, stream);
          if (items == -2) return PDE_AMD_EWORKSPACE;
          if (items == -1) return PDE_AMD_ESHAPE;
          launched = items > 0;
        }
      }
      if constexpr (ZFWD) {
        if (!launched && p->n[2] % P::ZW == 0 && aligned_vec<P::NF, P::ZW>(fields) && aligned_vec<P::NE, P::ZW>(resid_out)) {
          items = launch_plane_forward<true>(p, a, tr, cap, stream);
          if (items == -2) return PDE_AMD_EWORKSPACE;
          if (items < 0) return PDE_AMD_ESHAPE;
          launched = true;
        }
      }
      if constexpr (QuadCfg<P, P::QWF>::FWD_OK) {
        if (!launched && quad_addressable(p) && p->n[2] % P::QWF == 0 && aligned_vec<P::NF, P::QWF>(fields) && aligned_vec<P::NE, P::QWF>(resid_out) &&
            (!P::NA || aligned_vec<P::NA, P::QWF>(aux))) {
          fill_geom(p, a.geo, P::QF0, P::QF1, P::QF2);
          const long qtiles = (long)a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
          a.chunk = pick_chunk(p->batch * qtiles, (int)tr);
          items = cdiv((int)tr, a.chunk) * qtiles;
          if (items > tr * tiles) return PDE_AMD_EWORKSPACE;
          hipLaunchKernelGGL((fwd_march_q_kernel<P, P::QWF, true>), dim3((unsigned)(p->batch * items)), dim3(QuadCfg<P, P::QWF>::NT), 0, stream, a);
          launched = true;
        }
      }
      if constexpr (!QuadCfg<P, P::QWF>::FWD_OK && FrameCfg<P, P::QWF>::OK) {
        if (!launched && quad_addressable(p) && p->n[2] % P::QWF == 0 && aligned_vec<P::NF, P::QWF>(fields) && aligned_vec<P::NE, P::QWF>(resid_out) &&
            (!P::NA || aligned_vec<P::NA, P::QWF>(aux))) {
          fill_geom(p, a.geo, P::QF0, P::QF1, P::QF2);
          items = tr * a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
          if (items > tr * tiles) return PDE_AMD_EWORKSPACE;
          if (p->batch * items >= (1L << 31)) return PDE_AMD_ESHAPE;
          hipLaunchKernelGGL((fwd_frame_q_kernel<P, P::QWF, true>), dim3((unsigned)(p->batch * items)), dim3(FrameCfg<P, P::QWF>::NT), 0, stream, a);
          launched = true;
        }
      }
      if (!launched) {
        items = tr * tiles;
        if (p->batch * items >= (1L << 31)) return PDE_AMD_ESHAPE;
        hipLaunchKernelGGL((fwd_kernel<P, true>), dim3((unsigned)(p->batch * items)), dim3(P::NT), 0, stream, a);
      }
    } else if (ZFWD && p->n[2] % P::ZW == 0 && aligned_vec<P::NF, P::ZW>(fields)) {
      if constexpr (ZFWD) {
        items = launch_plane_forward<false>(p, a, tr, cap, stream);
        if (items == -2) return PDE_AMD_EWORKSPACE;
        if (items < 0) return PDE_AMD_ESHAPE;
      }
    } else if (QuadCfg<P, P::QWF>::FWD_OK && quad_addressable(p) && p->n[2] % P::QWF == 0 && aligned_vec<P::NF, P::QWF>(fields) &&
               (!P::NA || aligned_vec<P::NA, P::QWF>(aux))) {
      if constexpr (QuadCfg<P, P::QWF>::FWD_OK) {
        fill_geom(p, a.geo, P::QF0, P::QF1, P::QF2);
        const long qtiles = (long)a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
        a.chunk = pick_chunk(p->batch * qtiles, (int)tr);
        items = cdiv((int)tr, a.chunk) * qtiles;
        if (items > tr * tiles) return PDE_AMD_EWORKSPACE;
        if (full_grid_fwd(p))
          hipLaunchKernelGGL((fwd_march_q_kernel<P, P::QWF, false, true>), dim3((unsigned)(p->batch * items)), dim3(QuadCfg<P, P::QWF>::NT), 0, stream, a);
        else
          hipLaunchKernelGGL((fwd_march_q_kernel<P, P::QWF, false>), dim3((unsigned)(p->batch * items)), dim3(QuadCfg<P, P::QWF>::NT), 0, stream, a);
      }
    } else if (!QuadCfg<P, P::QWF>::FWD_OK && FrameCfg<P, P::QWF>::OK && quad_addressable(p) && p->n[2] % P::QWF == 0 && aligned_vec<P::NF, P::QWF>(fields) &&
               (!P::NA || aligned_vec<P::NA, P::QWF>(aux))) {
      if constexpr (!QuadCfg<P, P::QWF>::FWD_OK && FrameCfg<P, P::QWF>::OK) {
        fill_geom(p, a.geo, P::QF0, P::QF1, P::QF2);
        items = tr * a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
        if (items > tr * tiles) return PDE_AMD_EWORKSPACE;
        if (p->batch * items >= (1L << 31)) return PDE_AMD_ESHAPE;
        hipLaunchKernelGGL((fwd_frame_q_kernel<P, P::QWF, false>), dim3((unsigned)(p->batch * items)), dim3(FrameCfg<P, P::QWF>::NT), 0, stream, a);
      }
    } else if constexpr (FwdMarchFits<P>::value) {
      a.chunk = pick_chunk(p->batch * tiles, (int)tr);
      items = cdiv((int)tr, a.chunk) * tiles;
      if (p->batch * items >= (1L << 31)) return PDE_AMD_ESHAPE;
      hipLaunchKernelGGL((fwd_march_kernel<P>), dim3((unsigned)(p->batch * items)), dim3(P::NTM), 0, stream, a);
    } else {
      items = tr * tiles;
      if (p->batch * items >= (1L << 31)) return PDE_AMD_ESHAPE;
      hipLaunchKernelGGL((fwd_kernel<P, false>), dim3((unsigned)(p->batch * items)), dim3(P::NT), 0, stream, a);
    }
    reduce_partials(a.partial, sums, totals, items, p->batch, stream);
    return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP;
  }

  // Plane-marching forward of a 3-D PDE; returns the partial sums per sample (-1: too many workgroups; -2: more partial
  // sums than `cap`, the number the workspace holds; nothing is launched in either case).
  template <bool WRITE_R>
  static long launch_plane_forward(const pde_amd_problem* p, Args& a, long tr, long cap, hipStream_t stream) {
    if constexpr (ZCfg<P, P::ZW>::FWD_OK) {
      using C = ZCfg<P, P::ZW>;
      const long yx = (long)cdiv(p->n[1], C::T1) * cdiv(p->n[2], C::T2);
      a.chunk = pick_plane_chunk(p->batch * tr * yx, p->n[0]);
      fill_geom(p, a.geo, a.chunk, C::T1, C::T2);
      const long items = tr * a.geo.nt0 * yx;
      if (p->batch * items >= (1L << 31)) return -1;
      if (items > cap) return -2;
      hipLaunchKernelGGL((fwd_zmarch_kernel<P, P::ZW, WRITE_R>), dim3((unsigned)(p->batch * items)), dim3(C::NT), 0, stream, a);
      return items;
    }
    return -1;
  }

  static void reduce_partials(const double* partial, double* sums, float* totals, long items, int batch, hipStream_t stream) {
    if (totals && (long)batch * items <= 16384) {      // small: one launch does both reductions
      hipLaunchKernelGGL((finalize_totals_kernel<P::NE>), dim3(1), dim3(1024), 0, stream, partial, sums, totals, (int)items, batch);
    } else {
      hipLaunchKernelGGL((finalize_kernel<P::NE>), dim3(batch), dim3(256), 0, stream, partial, sums, (int)items);
      if (totals) hipLaunchKernelGGL((totals_kernel<P::NE>), dim3(1), dim3(64), 0, stream, sums, totals, batch);
    }
  }

  static int fwd_f64(const pde_amd_problem* p, const double* consts, const double* const* fields, const double* const* aux,
                     const double* const* sparams, const double* weights, double* sums, double* totals,
                     double* const* resid_out, void* ws, size_t ws_bytes_given, hipStream_t stream) {
    int st = check(p);
    if (st) return st;
    if (!sums || !fields || (P::NC && !consts) || (P::NA && !aux) || (P::NSP && !sparams)) return PDE_AMD_EINVAL;
    const long tr = p->frames - P::TLO - P::THI;
    if (tr <= 0) {
      (void)hipMemsetAsync(sums, 0, sizeof(double) * p->batch * P::NE, stream);
      if (totals) (void)hipMemsetAsync(totals, 0, sizeof(double) * (P::NE + 1), stream);
      return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP;
    }
    if (!weights || !ws) return PDE_AMD_EINVAL;
    const long nwg = (long)p->batch * tr;
    if (nwg >= (1L << 31)) return PDE_AMD_ESHAPE;
    if ((size_t)nwg * P::NE * sizeof(double) > ws_bytes_given) return PDE_AMD_EWORKSPACE;
    Args64 a;
    memset(&a, 0, sizeof(a));
    for (int f = 0; f < P::NF; ++f) { if (!fields[f]) return PDE_AMD_EINVAL; a.f[f] = fields[f]; }
    for (int x = 0; x < P::NA; ++x) a.aux[x] = aux[x];
    for (int s = 0; s < P::NSP; ++s) a.sp[s] = sparams[s];
    for (int c = 0; c < P::NC; ++c) a.c[c] = consts[c];
    for (int k = 0; k < P::NE; ++k) a.rout[k] = resid_out ? resid_out[k] : nullptr;
    a.wt = weights;
    a.partial = reinterpret_cast<double*>(ws);
    fill_geom(p, a.geo, 1, 1, 1);
    hipLaunchKernelGGL((fwd_direct_f64_kernel<P>), dim3((unsigned)nwg), dim3(256), 0, stream, a);
    hipLaunchKernelGGL((finalize_kernel<P::NE>), dim3(p->batch), dim3(256), 0, stream, a.partial, sums, (int)tr);
    if (totals) hipLaunchKernelGGL((totals_f64_kernel<P::NE>), dim3(1), dim3(64), 0, stream, sums, totals, p->batch);
    return launch_status();
  }

  // Loss and gradient of one training step in one pass where the vector adjoint applies (it recomputes every residual
  // anyway and adds up weight * R^2 on the way); otherwise the forward and the backward launch one after the other.
  static int fwd_bwd(const pde_amd_problem* p, const float* const* fields, const float* const* aux,
                     const float* const* sparams, const float* weights, const float* upstream_total, double* sums,
                     float* totals, float* const* grads, void* ws, size_t ws_bytes_given, hipStream_t stream) {
    int st = check(p);
    if (st) return st;
    if (!sums || !grads || !upstream_total) return PDE_AMD_EINVAL;
    const long tr = p->frames - P::TLO - P::THI;
    bool fused = false;
    if constexpr (QuadCfg<P, P::QWB>::BWD_OK) {
      if (tr > 0 && fields && weights && (!P::NA || aux) && (!P::NSP || sparams) && ws && quad_addressable(p) &&
          p->n[2] % P::QWB == 0 && aligned_vec<P::NF, P::QWB>(fields) && aligned_vec<P::NF, P::QWB>(grads) &&
          (!P::NA || aligned_vec<P::NA, P::QWB>(aux))) {
        for (int f = 0; f < P::NF; ++f)
          if (!grads[f]) return PDE_AMD_EINVAL;
        Args a;
        fill_args(a, p, fields, aux, sparams);
        fill_geom(p, a.geo, P::QB0, P::QB1, P::QB2);
        a.wt = weights;
        a.ups[P::NE] = upstream_total;
        for (int f = 0; f < P::NF; ++f) a.g[f] = grads[f];
        a.partial = reinterpret_cast<double*>(ws);
        const long qtiles = (long)a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
        a.chunk = pick_chunk(p->batch * qtiles, p->frames);
        const long items = (long)cdiv(p->frames, a.chunk) * qtiles;
        const long nwg = (long)p->batch * items;
        if (nwg < (1L << 31) && (size_t)nwg * P::NE * sizeof(double) <= ws_bytes_given) {
          {
            AdjointTimer timer(stream);
            if (full_grid(p)) {
              hipLaunchKernelGGL((bwd_march_q_kernel<P, P::QWB, true, true>), dim3((unsigned)nwg), dim3(QuadCfg<P, P::QWB>::NTB), 0, stream, a);
              note_kernel("bwd_march_q_kernel", P::NAME, P::QWB == 2 ? ", 2, true, true" : ", 4, true, true");
            } else {
              hipLaunchKernelGGL((bwd_march_q_kernel<P, P::QWB, true>), dim3((unsigned)nwg), dim3(QuadCfg<P, P::QWB>::NTB), 0, stream, a);
              note_kernel("bwd_march_q_kernel", P::NAME, P::QWB == 2 ? ", 2, true" : ", 4, true");
            }
          }
          reduce_partials(a.partial, sums, totals, items, p->batch, stream);
          fused = true;
        }
      }
    }
    if (fused) return launch_status();
    st = fwd(p, fields, aux, sparams, weights, sums, totals, nullptr, ws, ws_bytes_given, stream);
    if (st) return st;
    const float* ups[MAXE + 1] = {};
    ups[P::NE] = upstream_total;
    return bwd(p, fields, aux, sparams, weights, ups, nullptr, nullptr, grads, ws, ws_bytes_given, stream);
  }

  static int bwd(const pde_amd_problem* p, const float* const* fields, const float* const* aux,
                 const float* const* sparams, const float* weights, const float* const* upstream,
                 const double* upstream_sums, const float* const* resid_in, float* const* grads,
                 void* ws, size_t ws_bytes_given, hipStream_t stream) {
    (void)ws; (void)ws_bytes_given;
    int st = check(p);
    if (st) return st;
    if (!grads) return PDE_AMD_EINVAL;
    for (int f = 0; f < P::NF; ++f)
      if (!grads[f]) return PDE_AMD_EINVAL;
    const long tr = p->frames - P::TLO - P::THI;
    if (tr <= 0) {      // the loss does not depend on the fields
      for (int f = 0; f < P::NF; ++f)
        (void)hipMemsetAsync(grads[f], 0, sizeof(float) * (size_t)p->batch * p->frames * p->n[0] * p->n[1] * p->n[2], stream);
      return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP;
    }
    if (!fields || !weights || (!upstream && !upstream_sums) || (P::NA && !aux) || (P::NSP && !sparams)) return PDE_AMD_EINVAL;
    Args a;
    fill_args(a, p, fields, aux, sparams);
    fill_geom(p, a.geo, P::BT0, P::BT1, P::BT2);
    a.wt = weights;
    a.gsums = upstream_sums;
#ifdef PDE_AMD_STAMPS
    a.partial = reinterpret_cast<double*>(ws);
#endif
    if (upstream)
      for (int k = 0; k <= P::NE; ++k) a.ups[k] = upstream[k];
    for (int f = 0; f < P::NF; ++f) a.g[f] = grads[f];
    const long tiles = (long)a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
    if (resid_in) {
      for (int k = 0; k < P::NE; ++k) {
        if (!resid_in[k]) return PDE_AMD_EINVAL;
        a.r[k] = resid_in[k];
      }
      bool launched = false;
      AdjointTimer timer(stream);
      if constexpr (ZColCfg<P, P::ZCWB, true>::BWD_OK) {
        if (zcol_usable<P::ZCWB>(p, fields) && aligned_vec<P::NF, P::ZCWB>(grads) && aligned_vec<P::NE, P::ZCWB>(resid_in)) {
          using C = ZColCfg<P, P::ZCWB, true>;
          const ZPlan z = plan_zcol<C::LPF, C::FPWB, C::T1, C::T2>(p, a, tr);
          if (p->batch * z.items >= (1L << 31)) return PDE_AMD_ESHAPE;
          if (launch_zcol(p, a, z, C::LPF, bwd_stored_zcol_kernel<P, P::ZCWB, false>, bwd_stored_zcol_kernel<P, P::ZCWB, true>,
                          C::bwd_lds(z.nfw), stream)) {
            note_kernel("bwd_stored_zcol_kernel", P::NAME, P::ZCWB == 2 ? (z.nfw == 1 ? ", 2, true" : ", 2, false") : (z.nfw == 1 ? ", 4, true" : ", 4, false"));
            launched = true;
          }
        }
      }
      if constexpr (ZCfg<P, P::ZW>::BWD_OK) {
        if (!launched && p->n[2] % P::ZW == 0 && aligned_vec<P::NF, P::ZW>(fields) && aligned_vec<P::NF, P::ZW>(grads) && aligned_vec<P::NE, P::ZW>(resid_in)) {
          using C = ZCfg<P, P::ZW>;
          const long yx = (long)cdiv(p->n[1], C::T1) * cdiv(p->n[2], C::T2);
          a.chunk = pick_plane_chunk(p->batch * p->frames * yx, p->n[0]);
          fill_geom(p, a.geo, a.chunk, C::T1, C::T2);
          const long nwgz = (long)p->batch * p->frames * a.geo.nt0 * yx;
          if (nwgz >= (1L << 31)) return PDE_AMD_ESHAPE;
          hipLaunchKernelGGL((bwd_stored_zmarch_kernel<P, P::ZW>), dim3((unsigned)nwgz), dim3(C::NT), 0, stream, a);
          note_kernel("bwd_stored_zmarch_kernel", P::NAME, P::ZW == 2 ? ", 2" : ", 4");
          launched = true;
        }
      }
      if constexpr (StoredCfg<P, P::QWB>::OK) {
        if (!launched)
        if (quad_addressable(p) && p->n[2] % P::QWB == 0 && aligned_vec<P::NF, P::QWB>(fields) && aligned_vec<P::NF, P::QWB>(grads) &&
            aligned_vec<P::NE, P::QWB>(resid_in) && (!P::NA || aligned_vec<P::NA, P::QWB>(aux))) {
          fill_geom(p, a.geo, P::QB0, P::QB1, P::QB2);
          const long nwgq = (long)p->batch * p->frames * a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
          if (nwgq >= (1L << 31)) return PDE_AMD_ESHAPE;
          hipLaunchKernelGGL((bwd_stored_q_kernel<P, P::QWB>), dim3((unsigned)nwgq), dim3(StoredCfg<P, P::QWB>::NT), 0, stream, a);
          note_kernel("bwd_stored_q_kernel", P::NAME, P::QWB == 2 ? ", 2" : ", 4");
          launched = true;
        }
      }
      if (!launched) {
        const long nwg = (long)p->batch * p->frames * tiles;
        if (nwg >= (1L << 31)) return PDE_AMD_ESHAPE;
        hipLaunchKernelGGL((bwd_stored_kernel<P>), dim3((unsigned)nwg), dim3(P::NT), 0, stream, a);
        note_kernel("bwd_stored_kernel", P::NAME, "");
      }
    } else if (QuadCfg<P, P::QWB>::BWD_OK && quad_addressable(p) && p->n[2] % P::QWB == 0 && aligned_vec<P::NF, P::QWB>(fields) &&
               aligned_vec<P::NF, P::QWB>(grads) && (!P::NA || aligned_vec<P::NA, P::QWB>(aux))) {
      if constexpr (QuadCfg<P, P::QWB>::BWD_OK) {
        fill_geom(p, a.geo, P::QB0, P::QB1, P::QB2);
        const long qtiles = (long)a.geo.nt0 * a.geo.nt1 * a.geo.nt2;
        a.chunk = pick_chunk(p->batch * qtiles, p->frames);
        const long nwg = (long)p->batch * cdiv(p->frames, a.chunk) * qtiles;
        if (nwg >= (1L << 31)) return PDE_AMD_ESHAPE;
        #ifdef PDE_AMD_STAMPS
        static const int dyn_lds = getenv("PDE_AMD_DYN_LDS") ? atoi(getenv("PDE_AMD_DYN_LDS")) : 0;   // diagnostic: caps workgroups per CU
#else
        constexpr int dyn_lds = 0;
#endif
        AdjointTimer timer(stream);
        if (full_grid(p)) {
          hipLaunchKernelGGL((bwd_march_q_kernel<P, P::QWB, false, true>), dim3((unsigned)nwg), dim3(QuadCfg<P, P::QWB>::NTB), dyn_lds, stream, a);
          note_kernel("bwd_march_q_kernel", P::NAME, P::QWB == 2 ? ", 2, false, true" : ", 4, false, true");
        } else {
          hipLaunchKernelGGL((bwd_march_q_kernel<P, P::QWB>), dim3((unsigned)nwg), dim3(QuadCfg<P, P::QWB>::NTB), dyn_lds, stream, a);
          note_kernel("bwd_march_q_kernel", P::NAME, P::QWB == 2 ? ", 2" : ", 4");
        }
      }
    } else {
      if constexpr (BwdMarchFits<P>::value) {
        a.chunk = pick_chunk(p->batch * tiles, p->frames);
        const long nwg = (long)p->batch * cdiv(p->frames, a.chunk) * tiles;
        if (nwg >= (1L << 31)) return PDE_AMD_ESHAPE;
        AdjointTimer timer(stream);
        hipLaunchKernelGGL((bwd_march_kernel<P>), dim3((unsigned)nwg), dim3(P::NTM), 0, stream, a);
        note_kernel("bwd_march_kernel", P::NAME, "");
      } else {
        return PDE_AMD_EINVAL;   // this PDE has no recomputing adjoint: pass the stored residuals
      }
    }
    return launch_status();
  }

  static PdeVTable vtable() {
    PdeVTable v;
    v.info.id = P::ID; v.info.ndim = P::ND; v.info.n_fields = P::NF; v.info.n_eqs = P::NE; v.info.n_aux = P::NA;
    v.info.n_consts = P::NC; v.info.n_sparams = P::NSP; v.info.t_lo = P::TLO; v.info.t_hi = P::THI;
    v.info.halo[0] = P::RH0 > P::RT0 ? P::RH0 : P::RT0;
    v.info.halo[1] = P::RH1 > P::RT1 ? P::RH1 : P::RT1;
    v.info.halo[2] = P::RH2 > P::RT2 ? P::RH2 : P::RT2;
    v.info.can_recompute = BwdMarchFits<P>::value ? 1 : 0;   // the quad adjoint is an accelerated form of the same mode
    v.info.name = P::NAME;
    v.ws_bytes = &ws_bytes; v.fwd = &fwd; v.bwd = &bwd; v.fwd_bwd = &fwd_bwd; v.fwd_f64 = &fwd_f64;
    return v;
  }
};

#ifdef PDE_AMD_DIAG_ONLY      // diagnostic builds (-DPDE_AMD_DIAG_ONLY=Pde_tg2d): every slot runs that PDE's kernels; compiles in seconds
#ifdef PDE_AMD_DIAG_ZCW       // ... with other lane widths of the z-column kernels (-DPDE_AMD_DIAG_ZCW=2 -DPDE_AMD_DIAG_ZCWB=2)
#ifndef PDE_AMD_DIAG_ZCT1
#define PDE_AMD_DIAG_ZCT1 PDE_AMD_DIAG_ONLY::ZCT1
#define PDE_AMD_DIAG_ZCT2 PDE_AMD_DIAG_ONLY::ZCT2
#endif
#ifndef PDE_AMD_DIAG_ZCB1
#define PDE_AMD_DIAG_ZCB1 PDE_AMD_DIAG_ZCT1
#define PDE_AMD_DIAG_ZCB2 PDE_AMD_DIAG_ZCT2
#endif
#ifndef PDE_AMD_DIAG_ZCOL
#define PDE_AMD_DIAG_ZCOL PDE_AMD_DIAG_ONLY::ZCOL
#define PDE_AMD_DIAG_ZCOLF PDE_AMD_DIAG_ONLY::ZCOLF
#endif
#ifndef PDE_AMD_DIAG_ZCOLV
#define PDE_AMD_DIAG_ZCOLV PDE_AMD_DIAG_ONLY::ZCOLV
#endif
struct DiagPde : PDE_AMD_DIAG_ONLY {
  static constexpr int ZCW = PDE_AMD_DIAG_ZCW, ZCWB = PDE_AMD_DIAG_ZCWB, ZCT1 = PDE_AMD_DIAG_ZCT1, ZCT2 = PDE_AMD_DIAG_ZCT2;
  static constexpr int ZCB1 = PDE_AMD_DIAG_ZCB1, ZCB2 = PDE_AMD_DIAG_ZCB2;
  static constexpr int ZCOL = PDE_AMD_DIAG_ZCOL, ZCOLF = PDE_AMD_DIAG_ZCOLF, ZCOLV = PDE_AMD_DIAG_ZCOLV;
};
#define PDE_AMD_VT(P) Ops<DiagPde>::vtable(),
#else
#define PDE_AMD_VT(P) Ops<PDE_AMD_DIAG_ONLY>::vtable(),
#endif
#else
#define PDE_AMD_VT(P) Ops<P>::vtable(),
#endif
static const PdeVTable kTable[] = {PDE_AMD_FOR_EACH_PDE(PDE_AMD_VT)};
static constexpr int kNumPdes = sizeof(kTable) / sizeof(kTable[0]);

static const PdeVTable* lookup(int id) {
  if (id < 0 || id >= kNumPdes) return nullptr;
  return &kTable[id];
}
}  // namespace pde_amd

using namespace pde_amd;

// fills the workgroup's whole LDS allocation with NaNs (pde_amd_diag_poison_lds)
__global__ void poison_lds_kernel(int words) {
  extern __shared__ float s_poison[];
  volatile float* p = s_poison;
  const float nan = __int_as_float(0x7fc00000);
  for (int i = threadIdx.x; i < words; i += blockDim.x) p[i] = nan;
  __builtin_amdgcn_s_sleep(32);      // stay resident a moment, so that the workgroups spread over the CUs
}

extern "C" {

int pde_amd_num_pdes(void) { return kNumPdes; }

int pde_amd_pde_id(const char* name) {
  if (!name) return PDE_AMD_EINVAL;
  for (int i = 0; i < kNumPdes; ++i)
    if (strcmp(kTable[i].info.name, name) == 0) return i;
  return PDE_AMD_EINVAL;
}

int pde_amd_get_pde_info(int pde_id, pde_amd_pde_info* out) {
  const PdeVTable* v = lookup(pde_id);
  if (!v || !out) return PDE_AMD_EINVAL;
  *out = v->info;
  return PDE_AMD_OK;
}

const char* pde_amd_status_string(int status) {
  switch (status) {
    case PDE_AMD_OK: return "ok";
    case PDE_AMD_EINVAL: return "invalid argument";
    case PDE_AMD_ESHAPE: return "unsupported shape";
    case PDE_AMD_EWORKSPACE: return "workspace too small";
    case PDE_AMD_EHIP: return "HIP launch failed";
    case PDE_AMD_EEVENT: return "recording a timing event failed (pde_amd_time_next_adjoint); the kernels were launched";
    case PDE_AMD_ESCRATCH: return "problem needs more partial-sum scratch than PDE_AMD_STATIC_SCRATCH_BYTES";
    default: return "unknown status";
  }
}

size_t pde_amd_residual_workspace_bytes(const pde_amd_problem* prob) {
  const PdeVTable* v = prob ? lookup(prob->pde_id) : nullptr;
  return v ? v->ws_bytes(prob) : 0;
}

int pde_amd_residual_forward(const pde_amd_problem* prob, const float* const* fields, const float* const* aux,
                             const float* const* sparams, const float* weights, double* sums, float* totals,
                             float* const* resid_out, void* workspace, size_t workspace_bytes, void* stream) {
  const PdeVTable* v = prob ? lookup(prob->pde_id) : nullptr;
  if (!v) return PDE_AMD_EINVAL;
  return v->fwd(prob, fields, aux, sparams, weights, sums, totals, resid_out, workspace, workspace_bytes,
                static_cast<hipStream_t>(stream));
}

int pde_amd_residual_backward(const pde_amd_problem* prob, const float* const* fields, const float* const* aux,
                              const float* const* sparams, const float* weights, const float* const* upstream,
                              const double* upstream_sums, const float* const* resid_in, float* const* grads,
                              void* workspace, size_t workspace_bytes, void* stream) {
  const PdeVTable* v = prob ? lookup(prob->pde_id) : nullptr;
  if (!v) return PDE_AMD_EINVAL;
  return v->bwd(prob, fields, aux, sparams, weights, upstream, upstream_sums, resid_in, grads, workspace,
                workspace_bytes, static_cast<hipStream_t>(stream));
}

int pde_amd_residual_forward_backward(const pde_amd_problem* prob, const float* const* fields, const float* const* aux,
                                      const float* const* sparams, const float* weights, const float* upstream_total,
                                      double* sums, float* totals, float* const* grads, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  const PdeVTable* v = prob ? lookup(prob->pde_id) : nullptr;
  if (!v) return PDE_AMD_EINVAL;
  return v->fwd_bwd(prob, fields, aux, sparams, weights, upstream_total, sums, totals, grads, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream));
}

int pde_amd_residual_forward_f64(const pde_amd_problem* prob, const double* consts, const double* const* fields,
                                 const double* const* aux, const double* const* sparams, const double* weights, double* sums,
                                 double* totals, double* const* resid_out, void* workspace, size_t workspace_bytes, void* stream) {
  const PdeVTable* v = prob ? lookup(prob->pde_id) : nullptr;
  if (!v) return PDE_AMD_EINVAL;
  return v->fwd_f64(prob, consts, fields, aux, sparams, weights, sums, totals, resid_out, workspace, workspace_bytes,
                    static_cast<hipStream_t>(stream));
}

// The device address of `scratch` on the GPU that owns `device_ptr` (one of the call's own buffers; the current device when
// the runtime cannot tell), looked up once per device with that device current: each GPU holds its own copy of the array.
static int static_scratch(const void* device_ptr, void** out) {
  constexpr int kMaxDevices = 64;
  static void* ptrs[kMaxDevices] = {};
  static std::mutex mu;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return PDE_AMD_EHIP;
  int dev = cur;
  hipPointerAttribute_t attr;
  if (device_ptr && hipPointerGetAttributes(&attr, device_ptr) == hipSuccess && attr.type == hipMemoryTypeDevice) dev = attr.device;
  else (void)hipGetLastError();
  if (dev < 0 || dev >= kMaxDevices) return PDE_AMD_EHIP;
  std::lock_guard<std::mutex> lock(mu);
  if (!ptrs[dev]) {
    void* p = nullptr;
    if (dev != cur && hipSetDevice(dev) != hipSuccess) return PDE_AMD_EHIP;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (dev != cur && hipSetDevice(cur) != hipSuccess) return PDE_AMD_EHIP;
    if (e != hipSuccess || !p) return PDE_AMD_EHIP;
    ptrs[dev] = p;
  }
  *out = ptrs[dev];
  return PDE_AMD_OK;
}

int pde_amd_residual_forward_backward_static(const pde_amd_problem* prob, const float* const* fields,
                                             const float* const* aux, const float* const* sparams, const float* weights,
                                             const float* upstream_total, double* sums, float* totals,
                                             float* const* grads, void* stream) {
  const PdeVTable* v = prob ? lookup(prob->pde_id) : nullptr;
  if (!v) return PDE_AMD_EINVAL;
  const size_t need = v->ws_bytes(prob);
  if (need > PDE_AMD_STATIC_SCRATCH_BYTES) return PDE_AMD_ESCRATCH;
  void* ws = nullptr;
  if (!grads || !grads[0]) return PDE_AMD_EINVAL;
  const int st = static_scratch(grads[0], &ws);      // the GPU of the gradients, which is where the kernels write
  if (st) return st;
  // every partial sum the reduction reads is written by the kernel before it, so nothing has to be cleared between calls
  return v->fwd_bwd(prob, fields, aux, sparams, weights, upstream_total, sums, totals, grads, ws,
                    PDE_AMD_STATIC_SCRATCH_BYTES, static_cast<hipStream_t>(stream));
}

const char* pde_amd_last_adjoint_kernel(void) { return t_kernel; }

int pde_amd_diag_poison_lds(void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  constexpr int BYTES = 160 * 1024;
  static bool granted = false;
  if (!granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES) != hipSuccess)
      return PDE_AMD_EHIP;
    granted = true;
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return PDE_AMD_EHIP;
  hipLaunchKernelGGL(poison_lds_kernel, dim3(4 * cus), dim3(1024), BYTES, stream, BYTES / 4);      // one workgroup fills a CU's LDS: each CU gets several in turn
  return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP;
}

#ifdef PDE_AMD_DIAG_ONLY
// diagnostic builds only: workgroups per CU the runtime grants the z-column kernels of the build's PDE
void pde_amd_diag_zcol_occupancy(void) {
#ifdef PDE_AMD_DIAG_ZCW
  using P = DiagPde;
#else
  using P = PDE_AMD_DIAG_ONLY;
#endif
  if constexpr (ZColCfg<P, P::ZCW>::FWD_OK) {
    using C = ZColCfg<P, P::ZCW>;
    for (int nfw = 1; nfw <= C::FPWF; ++nfw) {
      int n = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fwd_zcol_kernel<P, P::ZCW, true, false>, C::LPF * nfw, C::fwd_lds(nfw));
      printf("fwd_zcol nfw=%d threads=%d lds=%zu -> %d workgroups per CU (%s)\n", nfw, C::LPF * nfw, C::fwd_lds(nfw), n, hipGetErrorString(e));
    }
  }
  if constexpr (ZColCfg<P, P::ZCWB, true>::BWD_OK) {
    using C = ZColCfg<P, P::ZCWB, true>;
    for (int nfw = 1; nfw <= C::FPWB; ++nfw) {
      int n = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bwd_stored_zcol_kernel<P, P::ZCWB, false>, C::LPF * nfw, C::bwd_lds(nfw));
      printf("bwd_zcol nfw=%d threads=%d lds=%zu -> %d workgroups per CU (%s)\n", nfw, C::LPF * nfw, C::bwd_lds(nfw), n, hipGetErrorString(e));
    }
  }
}
#endif

int pde_amd_time_next_adjoint(void* start_event, void* stop_event) {
  t_ev0 = static_cast<hipEvent_t>(start_event);
  t_ev1 = static_cast<hipEvent_t>(stop_event);
  return PDE_AMD_OK;
}

int pde_amd_rescale_gradients(float* const* grads, int32_t n_fields, int64_t elements, const float* upstream,
                              const float* assumed, void* stream) {
  if (!grads || !upstream || !assumed || n_fields < 0 || n_fields > PDE_AMD_MAX_FIELDS || elements < 0) return PDE_AMD_EINVAL;
  if (n_fields == 0 || elements == 0) return PDE_AMD_OK;
  ScaleArgs a;
  a.n = n_fields; a.g = upstream; a.hint = assumed;
  bool vec = true;
  for (int f = 0; f < n_fields; ++f) {
    if (!grads[f]) return PDE_AMD_EINVAL;
    a.p[f] = grads[f];
    vec = vec && (reinterpret_cast<uintptr_t>(grads[f]) & 15) == 0;
  }
  vec = vec && elements % 4 == 0;
  a.n4 = vec ? elements / 4 : elements;
  long nb = (a.n4 + 256 * 8 - 1) / (256 * 8);
  if (nb > 4096) nb = 4096;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec) hipLaunchKernelGGL(scale_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(scale_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP;
}

}  // extern "C"
