// C ABI (include/pde_amd.h) of the channel-last reductions and index-list "term" losses:
//   pair statistics over [groups][rows][channels] tensors (nRMSE / VRMSE / RSE / RMSE families),
//   sums of squares over unions of index-list boxes with optional 3-tap combinations (boundary strips,
//   wall, patch-seam losses) and their gradients, and the field gather / gradient scatter between the
//   model's channel-last tensors and the per-field tensors of the residual losses.
#include "../../include/pde_amd.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pde_amd {
namespace {

constexpr int NSTAT = PDE_AMD_NSTAT;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// pair statistics.  Thread t of a workgroup of NT = m * C threads always sees channel t % C: the flat
// element index of a group advances by NT per step, so loads are contiguous dwords across the workgroup
// and the six accumulators of a thread belong to one channel.  Accumulation is fp64 throughout.
// ------------------------------------------------------------------------------------------------
constexpr int PS_UNROLL = 4;

// Optional by-product of the pass: the channels ch[f] of pred (after the affine map) written out as per-field tensors
// [groups][frames + 1][S] at frames 1..frames (frame 0 is the caller's: the trainers put the input's first frame there),
// i.e. what gather_fields produces, without a second sweep over pred.  rows = frames * S.
struct GatherSide {
  float* fields[PDE_AMD_MAX_FIELDS];
  int ch[PDE_AMD_MAX_FIELDS];
  int nf;
  long S, frames;
};

template <bool RSE>
__global__ __launch_bounds__(1024) void pair_stats_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           long rows, int C, int rows_per_step, long steps_per_block,
                                                           int blocks_per_group, float rse_eps, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, double* __restrict__ partial,
                                                           const GatherSide gs) {
  extern __shared__ double s_acc[];          // [NSTAT][NT]
  const int NT = blockDim.x;
  const int g = blockIdx.x / blocks_per_group, blk = blockIdx.x % blocks_per_group;
  // optional affine map of pred, one (scale, shift) per group and channel: pred is read as fl(fl(p * scale) + shift)
  const bool affine = scale != nullptr;
  const float sc = affine ? scale[(long)g * C + threadIdx.x % C] : 1.0f, sh = affine ? shift[(long)g * C + threadIdx.x % C] : 0.0f;
  const long n = rows * (long)C;             // elements of the group
  const float* p = pred + (long)g * n;
  const float* q = target + (long)g * n;
  const long first = (long)blk * steps_per_block * NT + threadIdx.x;
  const long last = min(n, (long)(blk + 1) * steps_per_block * NT);
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  float amax = 0.0f;
  long i = first;
  // a thread's channel never changes, so it either feeds one field or none; element i is row i / C of the group, and a
  // step of NT elements is rows_per_step rows
  float* gdst = nullptr;
  for (int f = 0; f < gs.nf; ++f)
    if (gs.ch[f] == (int)(threadIdx.x % C)) gdst = gs.fields[f] + ((long)g * (gs.frames + 1) + 1) * gs.S + first / C;
  for (; i + (long)(PS_UNROLL - 1) * NT < last; i += (long)PS_UNROLL * NT) {
    float pv[PS_UNROLL], gv[PS_UNROLL];
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {      // both tensors are read once: streaming loads
      pv[u] = __builtin_nontemporal_load(p + i + (long)u * NT);
      gv[u] = __builtin_nontemporal_load(q + i + (long)u * NT);
    }
    if (affine) {
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) pv[u] = __fadd_rn(__fmul_rn(pv[u], sc), sh);
    }
    if (gdst) {
#pragma unroll
      for (int u = 0; u < PS_UNROLL; ++u) gdst[(long)u * rows_per_step] = pv[u];
      gdst += (long)PS_UNROLL * rows_per_step;
    }
#pragma unroll
    for (int u = 0; u < PS_UNROLL; ++u) {
      const float d = pv[u] - gv[u];
      const float d2 = d * d, g2 = gv[u] * gv[u];
      a0 += (double)d2; a1 += (double)g2; a2 += (double)gv[u]; a3 += (double)pv[u];
      if (RSE) a4 += (double)(d2 / (g2 + rse_eps));
      amax = fmaxf(amax, fabsf(d));
    }
  }
  for (; i < last; i += NT) {
    const float pvv = affine ? __fadd_rn(__fmul_rn(p[i], sc), sh) : p[i], gvv = q[i];
    if (gdst) { *gdst = pvv; gdst += rows_per_step; }
    const float d = pvv - gvv;
    const float d2 = d * d, g2 = gvv * gvv;
    a0 += (double)d2; a1 += (double)g2; a2 += (double)gvv; a3 += (double)pvv;
    if (RSE) a4 += (double)(d2 / (g2 + rse_eps));
    amax = fmaxf(amax, fabsf(d));
  }
  s_acc[0 * NT + threadIdx.x] = a0;
  s_acc[1 * NT + threadIdx.x] = a1;
  s_acc[2 * NT + threadIdx.x] = a2;
  s_acc[3 * NT + threadIdx.x] = a3;
  s_acc[4 * NT + threadIdx.x] = a4;
  s_acc[5 * NT + threadIdx.x] = (double)amax;
  __syncthreads();
  // thread (s, c): fixed-order sum over the threads of channel c
  for (int w = threadIdx.x; w < NSTAT * C; w += NT) {
    const int s = w / C, c = w % C;
    double r = 0.0;
    if (s < 5) { for (int t = c; t < NT; t += C) r += s_acc[s * NT + t]; }
    else { for (int t = c; t < NT; t += C) r = fmax(r, s_acc[s * NT + t]); }
    partial[((long)blockIdx.x * C + c) * NSTAT + s] = r;
  }
}

// The same statistics with 16-byte loads: a thread owns the float4 number t + k * NT of its group (NT a multiple of C), so
// its four elements always carry the channels (4 t + j) % C and it keeps one set of accumulators per j.  Needs a group
// length that is a multiple of four elements and 16-byte aligned tensors; moves four times the bytes per load instruction
// of the dword kernel above, which is what that kernel's rate is bound by.
constexpr int PSQ_UNROLL = 4;
template <bool RSE>
__global__ __launch_bounds__(256) void pair_stats_q_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                            long rows, int C, long steps_per_block, int blocks_per_group,
                                                            float rse_eps, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, double* __restrict__ partial) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ double s_acc[];          // [NSTAT][4 * NT]
  const int NT = blockDim.x;
  const int g = blockIdx.x / blocks_per_group, blk = blockIdx.x % blocks_per_group;
  const bool affine = scale != nullptr;
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = (int)((4 * threadIdx.x + j) % C);
    sc[j] = affine ? scale[(long)g * C + c] : 1.0f;
    sh[j] = affine ? shift[(long)g * C + c] : 0.0f;
  }
  const long n4 = rows * (long)C / 4;        // float4s of the group
  const f4* p = reinterpret_cast<const f4*>(pred + (long)g * rows * C);
  const f4* q = reinterpret_cast<const f4*>(target + (long)g * rows * C);
  const long last = min(n4, (long)(blk + 1) * steps_per_block * NT);
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, a3[4] = {0, 0, 0, 0}, a4[4] = {0, 0, 0, 0};
  float amax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto add = [&](f4 pv, const f4 gv) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float pj = affine ? __fadd_rn(__fmul_rn(pv[j], sc[j]), sh[j]) : pv[j];
      const float d = pj - gv[j];
      const float d2 = d * d, g2 = gv[j] * gv[j];
      a0[j] += (double)d2; a1[j] += (double)g2; a2[j] += (double)gv[j]; a3[j] += (double)pj;
      if (RSE) a4[j] += (double)(d2 / (g2 + rse_eps));
      amax[j] = fmaxf(amax[j], fabsf(d));
    }
  };
  long i = (long)blk * steps_per_block * NT + threadIdx.x;
  for (; i + (long)(PSQ_UNROLL - 1) * NT < last; i += (long)PSQ_UNROLL * NT) {
    f4 pv[PSQ_UNROLL], gv[PSQ_UNROLL];
#pragma unroll
    for (int u = 0; u < PSQ_UNROLL; ++u) {      // both tensors are read once: streaming loads
      pv[u] = __builtin_nontemporal_load(p + i + (long)u * NT);
      gv[u] = __builtin_nontemporal_load(q + i + (long)u * NT);
    }
#pragma unroll
    for (int u = 0; u < PSQ_UNROLL; ++u) add(pv[u], gv[u]);
  }
  for (; i < last; i += NT) add(p[i], q[i]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = 4 * threadIdx.x + j;
    s_acc[0 * 4 * NT + e] = a0[j];
    s_acc[1 * 4 * NT + e] = a1[j];
    s_acc[2 * 4 * NT + e] = a2[j];
    s_acc[3 * 4 * NT + e] = a3[j];
    s_acc[4 * 4 * NT + e] = a4[j];
    s_acc[5 * 4 * NT + e] = (double)amax[j];
  }
  __syncthreads();
  // thread (s, c): fixed-order sum over the accumulator sets of channel c
  for (int w = threadIdx.x; w < NSTAT * C; w += NT) {
    const int s = w / C, c = w % C;
    double r = 0.0;
    if (s < 5) { for (int t = c; t < 4 * NT; t += C) r += s_acc[s * 4 * NT + t]; }
    else { for (int t = c; t < 4 * NT; t += C) r = fmax(r, s_acc[s * 4 * NT + t]); }
    partial[((long)blockIdx.x * C + c) * NSTAT + s] = r;
  }
}

// one wave per (group, channel, statistic): lanes stride over the workgroups' partials, then a fixed-order shuffle tree
__global__ __launch_bounds__(256) void pair_stats_finalize_kernel(const double* __restrict__ partial, int blocks_per_group, int C,
                                                                   double* __restrict__ stats) {
  const int g = blockIdx.y;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= NSTAT * C) return;
  const int c = w / NSTAT, s = w % NSTAT, lane = threadIdx.x & 63;
  const bool is_max = s == 5;
  double r = 0.0;
  for (int b = lane; b < blocks_per_group; b += 64) {
    const double v = partial[(((long)g * blocks_per_group + b) * C + c) * NSTAT + s];
    r = is_max ? fmax(r, v) : r + v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double v = __shfl_down(r, o, 64);
    r = is_max ? fmax(r, v) : r + v;
  }
  if (lane == 0) stats[((long)g * C + c) * NSTAT + s] = r;
}

struct PairPlan {
  int nt, blocks_per_group;
  long steps_per_block;
};

inline bool pair_plan(long groups, long rows, int C, PairPlan& pl) {
  if (groups <= 0 || rows <= 0 || C <= 0 || C > 1024) return false;
  int m = 768 / C;
  if (m < 1) m = 1;
  pl.nt = m * C;
  const long n = rows * (long)C;
  const long steps = (n + pl.nt - 1) / pl.nt;
  // about 2048 workgroups in total, at least 8 steps each
  long bpg = (2048 + groups - 1) / groups;
  const long maxb = (steps + 7) / 8;
  if (bpg > maxb) bpg = maxb;
  if (bpg < 1) bpg = 1;
  pl.steps_per_block = (steps + bpg - 1) / bpg;
  pl.blocks_per_group = (int)((steps + pl.steps_per_block - 1) / pl.steps_per_block);
  return groups * (long)pl.blocks_per_group < (1L << 31) && groups <= 65535;
}

// ------------------------------------------------------------------------------------------------
// terms: tensor [outer][d0][d1][d2][C]; a term is the Cartesian product of four index lists, and the value at
// a product element is sum_j coef_j * x[... + shift_j along `axis`], x = pred - target or pred alone.
// ------------------------------------------------------------------------------------------------
struct TermArgs {
  const float* pred;
  const float* target;
  const int32_t* indices;
  long outer;
  long stride[4];       // element strides of d0, d1, d2, channel
  long outer_stride;
  pde_amd_term t;
  long count;           // outer * n0 * n1 * n2 * nc
  // optional per-(sample, channel) affine map of pred, [B][C] each: the tensor's value is pred * scale + shift
  // (mul, then add, each rounded: what `output_norm * std + mean` leaves in memory); sample = outer index / outer_per_sample
  const float* scale;
  const float* shift;
  long outer_per_sample;
};

__device__ __forceinline__ float term_value(const TermArgs& a, long e, long& off, long& tap_stride, float& sc) {
  const pde_amd_term& t = a.t;
  const int ic = (int)(e % t.n[3]); e /= t.n[3];
  const int i2 = (int)(e % t.n[2]); e /= t.n[2];
  const int i1 = (int)(e % t.n[1]); e /= t.n[1];
  const int i0 = (int)(e % t.n[0]); e /= t.n[0];
  off = e * a.outer_stride + a.indices[t.idx_off[0] + i0] * a.stride[0] + a.indices[t.idx_off[1] + i1] * a.stride[1] +
        a.indices[t.idx_off[2] + i2] * a.stride[2] + a.indices[t.idx_off[3] + ic] * a.stride[3];
  tap_stride = a.stride[t.axis];
  sc = 1.0f;
  float sh = 0.0f;
  if (a.scale) {
    const long bc = (e / a.outer_per_sample) * a.stride[2] + a.indices[t.idx_off[3] + ic];      // stride[2] = C
    sc = a.scale[bc];
    sh = a.shift[bc];
  }
  float v = 0.0f;
  for (int j = 0; j < t.n_taps; ++j) {
    const long o = off + t.shift[j] * tap_stride;
    float p = a.pred[o];
    if (a.scale) p = __fadd_rn(__fmul_rn(p, sc), sh);
    const float x = t.use_target ? p - a.target[o] : p;
    v += t.coef[j] * x;
  }
  return v;
}

struct BlockTable { int start[PDE_AMD_MAX_TERMS + 1]; };

// every term of a set in ONE launch: block x belongs to the term whose block range holds it
struct TermBatch {
  TermArgs base;                       // everything but the term and its element count
  pde_amd_term t[PDE_AMD_MAX_TERMS];
  long count[PDE_AMD_MAX_TERMS];
  BlockTable blk;
  int n_terms;
};

__global__ __launch_bounds__(256) void term_fwd_batch_kernel(const TermBatch tb, double* __restrict__ partial) {
  __shared__ double s_red[4];
  int k = 0;
  while (k + 1 < tb.n_terms && (int)blockIdx.x >= tb.blk.start[k + 1]) ++k;
  TermArgs a = tb.base;
  a.t = tb.t[k];
  a.count = tb.count[k];
  const long first = (long)((int)blockIdx.x - tb.blk.start[k]) * 256 + threadIdx.x;
  const long stride = (long)(tb.blk.start[k + 1] - tb.blk.start[k]) * 256;
  double acc = 0.0;
  for (long e = first; e < a.count; e += stride) {
    long off, ts;
    float sc;
    const float v = term_value(a, e, off, ts, sc);
    acc += (double)(v * v);
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// one wave per term: lanes stride over the term's block partials, then a shuffle tree (fixed order)
__global__ __launch_bounds__(64) void term_finalize_kernel(const double* __restrict__ partial, const BlockTable blk, int n_terms, double* __restrict__ sums) {
  const int t = blockIdx.x;
  if (t >= n_terms) return;
  double r = 0.0;
  for (int b = blk.start[t] + (int)threadIdx.x; b < blk.start[t + 1]; b += 64) r += partial[b];
  r = wave_sum_d(r);
  if (threadIdx.x == 0) sums[t] = r;
}

// the backward of every term of a set in one launch, for sets whose terms touch no common element
__global__ __launch_bounds__(256) void term_bwd_batch_kernel(const TermBatch tb, const float* __restrict__ scales, float* __restrict__ grad) {
  int k = 0;
  while (k + 1 < tb.n_terms && (int)blockIdx.x >= tb.blk.start[k + 1]) ++k;
  TermArgs a = tb.base;
  a.t = tb.t[k];
  a.count = tb.count[k];
  const float scale = 2.0f * scales[k];
  const long first = (long)((int)blockIdx.x - tb.blk.start[k]) * 256 + threadIdx.x;
  const long stride = (long)(tb.blk.start[k + 1] - tb.blk.start[k]) * 256;
  for (long e = first; e < a.count; e += stride) {
    long off, ts;
    float sc;
    const float v = term_value(a, e, off, ts, sc);
    for (int j = 0; j < a.t.n_taps; ++j) grad[off + a.t.shift[j] * ts] += scale * v * a.t.coef[j] * sc;
  }
}

// grad[...] += scale * 2 * value * coef_j at every tap; offsets inside one term never collide
__global__ __launch_bounds__(256) void term_bwd_kernel(const TermArgs a, const float* __restrict__ scale_ptr, float* __restrict__ grad) {
  const float scale = 2.0f * scale_ptr[0];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < a.count; e += (long)gridDim.x * 256) {
    long off, ts;
    float sc;
    const float v = term_value(a, e, off, ts, sc);
    for (int j = 0; j < a.t.n_taps; ++j) grad[off + a.t.shift[j] * ts] += scale * v * a.t.coef[j] * sc;
  }
}

inline int term_blocks(long count) {
  long b = (count + 1023) / 1024;
  if (b < 1) b = 1;
  if (b > 1024) b = 1024;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------
// field gather / gradient scatter between [B][T][S][C] model tensors and per-field [B][T+1][S] tensors
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
  const float* output;     // [B][T_out][S][C]
  const float* input;      // [B][T_in][S][C], frame 0 is prepended
  float* fields[PDE_AMD_MAX_FIELDS];
  const float* gfields[PDE_AMD_MAX_FIELDS];
  int32_t ch[PDE_AMD_MAX_FIELDS];
  int nf, C;
  long B, T_out, T_in, S;
  int rows, cols, transpose;   // S = lead * rows * cols; transpose swaps the last two spatial axes in the fields
  // optional per-(sample, channel) affine map of `output`, [B][C] each: the gathered value is output * scale + shift
  // (mul, then add, each rounded), the scattered gradient is multiplied by scale; `input` is taken as it is
  const float* scale;
  const float* shift;
};

__device__ __forceinline__ long field_site(const GatherArgs& a, long s) {
  if (!a.transpose) return s;
  const long rc = (long)a.rows * a.cols;
  const long lead = s / rc, r = (s % rc) / a.cols, c = s % a.cols;
  return lead * rc + c * a.rows + r;
}
__device__ __forceinline__ unsigned field_site(const GatherArgs& a, unsigned s) {      // s < 2^31: inside one frame
  if (!a.transpose) return s;
  const unsigned rc = (unsigned)a.rows * (unsigned)a.cols;
  const unsigned lead = s / rc, rem = s - lead * rc, r = rem / (unsigned)a.cols, c = rem - r * (unsigned)a.cols;
  return lead * rc + c * (unsigned)a.rows + r;
}

__global__ __launch_bounds__(256) void gather_kernel(const GatherArgs a) {
  const long npts = a.B * (a.T_out + 1) * a.S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npts; i += (long)gridDim.x * 256) {
    const long s = i % a.S, t = (i / a.S) % (a.T_out + 1), b = i / (a.S * (a.T_out + 1));
    const float* src = t == 0 ? a.input + ((b * a.T_in) * a.S + s) * a.C : a.output + ((b * a.T_out + (t - 1)) * a.S + s) * a.C;
    const long dst = (b * (a.T_out + 1) + t) * a.S + field_site(a, s);
    for (int f = 0; f < a.nf; ++f) {
      float v = src[a.ch[f]];
      if (a.scale && t > 0) v = __fadd_rn(__fmul_rn(v, a.scale[b * a.C + a.ch[f]]), a.shift[b * a.C + a.ch[f]]);
      a.fields[f][dst] = v;
    }
  }
}

// grid.y = (b, t) frame of grad_output, grid.x strides over the frame's float4s; all index arithmetic inside a frame
// is 32-bit.  Every element is written (zeros outside the gathered channels).  VEC = 4 needs S*C % 4 == 0.
template <int VEC>
__global__ __launch_bounds__(256) void scatter_kernel(const GatherArgs a, float* __restrict__ grad_output) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const long frame = blockIdx.y;
  const long b = frame / a.T_out, t = frame % a.T_out;
  const unsigned n = (unsigned)(a.S * a.C);                 // elements of the frame
  const unsigned C = (unsigned)a.C;
  float* dst = grad_output + frame * (long)n;
  const long fbase = (b * (a.T_out + 1) + t + 1) * a.S;
  for (unsigned i = (blockIdx.x * 256 + threadIdx.x) * VEC; i < n; i += gridDim.x * 256 * VEC) {
    float v[VEC];
    const unsigned pt0 = i / C;
    unsigned c = i - pt0 * C, pt = pt0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float r = 0.0f;
      for (int f = 0; f < a.nf; ++f)
        if ((unsigned)a.ch[f] == c) r += a.gfields[f][fbase + field_site(a, pt)];
      if (a.scale) r *= a.scale[b * a.C + c];
      v[k] = r;
      if (++c == C) { c = 0; ++pt; }
    }
    if (VEC == 4) __builtin_nontemporal_store(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(dst + i));      // streamed once, never re-read here
    else dst[i] = v[0];
  }
}

// Tiled forms of the two kernels above for the usual case (few channels, 16-byte aligned frames).  The channel-last
// tensor is only ever touched with whole float4 rows of a 256-point chunk, which pass through an LDS image
// [256][C]; the per-channel side reads or writes one float per thread, consecutive points in consecutive lanes.
// A work item is (frame, chunk of 256 points).  Streaming rate instead of one strided access per point and channel.
constexpr int TILE_PTS = 256;

__global__ __launch_bounds__(TILE_PTS) void gather_tile_kernel(const GatherArgs a) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const unsigned C = (unsigned)a.C, tid = threadIdx.x;
  const long chunks = (a.S + TILE_PTS - 1) / TILE_PTS, items = a.B * (a.T_out + 1) * chunks;
  for (long w = blockIdx.x; w < items; w += gridDim.x) {
    const long frame = w / chunks, b = frame / (a.T_out + 1), t = frame % (a.T_out + 1);
    const unsigned p0 = (unsigned)(w % chunks) * TILE_PTS;
    const unsigned np = min((unsigned)TILE_PTS, (unsigned)a.S - p0);
    const float* src = (t == 0 ? a.input + (b * a.T_in) * a.S * C : a.output + (b * a.T_out + (t - 1)) * a.S * C) + (long)p0 * C;
    const unsigned n4 = np * C / 4;
    // read once and never again: a streaming load, so that the 18-channel tensor does not push the gathered fields (which
    // the residual kernels read next) out of the caches
    for (unsigned i = tid; i < n4; i += TILE_PTS) reinterpret_cast<f4*>(tile)[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src) + i);
    __syncthreads();
    if (tid < np) {
      const long dst = frame * a.S + field_site(a, p0 + tid);
      const bool affine = a.scale && t > 0;
      for (int f = 0; f < a.nf; ++f) {
        float v = tile[tid * C + (unsigned)a.ch[f]];
        if (affine) v = __fadd_rn(__fmul_rn(v, a.scale[b * a.C + a.ch[f]]), a.shift[b * a.C + a.ch[f]]);
        a.fields[f][dst] = v;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TILE_PTS) void scatter_tile_kernel(const GatherArgs a, float* __restrict__ grad_output) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const unsigned C = (unsigned)a.C, tid = threadIdx.x;
  for (unsigned i = tid; i < TILE_PTS * C; i += TILE_PTS) tile[i] = 0.0f;      // the other channels stay zero for good
  __syncthreads();
  const long chunks = (a.S + TILE_PTS - 1) / TILE_PTS, items = a.B * a.T_out * chunks;
  for (long w = blockIdx.x; w < items; w += gridDim.x) {
    const long frame = w / chunks, b = frame / a.T_out, t = frame % a.T_out;
    const unsigned p0 = (unsigned)(w % chunks) * TILE_PTS;
    const unsigned np = min((unsigned)TILE_PTS, (unsigned)a.S - p0);
    if (tid < np) {
      const long src = (b * (a.T_out + 1) + t + 1) * a.S + field_site(a, p0 + tid);
      for (int f = 0; f < a.nf; ++f) {
        float v = a.gfields[f][src];
        if (a.scale) v *= a.scale[b * a.C + a.ch[f]];
        bool again = false;      // a channel listed more than once collects all its gradients
        for (int g = 0; g < f; ++g) again = again || a.ch[g] == a.ch[f];
        if (again) v += tile[tid * C + (unsigned)a.ch[f]];
        tile[tid * C + (unsigned)a.ch[f]] = v;
      }
    }
    __syncthreads();
    f4* dst = reinterpret_cast<f4*>(grad_output + (frame * a.S + p0) * C);
    const unsigned n4 = np * C / 4;
    for (unsigned i = tid; i < n4; i += TILE_PTS) __builtin_nontemporal_store(reinterpret_cast<const f4*>(tile)[i], dst + i);   // streamed once
    __syncthreads();
  }
}

// The statistics pass of pair_stats_kernel and the gather of gather_tile_kernel in one sweep: a workgroup takes whole
// 256-site chunks of one sample's frames, both tensors pass through LDS images [256][C] as float4 rows, the gathered
// channels leave as one float per lane (consecutive sites in consecutive lanes), and the statistics are accumulated from
// the images by the first (256 / C) * C threads, whose channel (thread % C) then never changes.  Accumulation in fp64.
// partial has the layout of pair_stats_kernel's, so the same finalize kernel follows.
__global__ __launch_bounds__(TILE_PTS) void stats_gather_tile_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                     int C, int blocks_per_group, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, double* __restrict__ partial,
                                                                     const GatherSide gs) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float tile[];      // [2][TILE_PTS * C]; the statistics' scratch afterwards
  float* const tp = tile;
  float* const tq = tile + TILE_PTS * C;
  const unsigned tid = threadIdx.x, NTS = (TILE_PTS / C) * C;
  const int g = blockIdx.x / blocks_per_group, blk = blockIdx.x % blocks_per_group;
  const long chunks = (gs.S + TILE_PTS - 1) / TILE_PTS, items = gs.frames * chunks;
  const bool affine = scale != nullptr;
  const unsigned cm = tid % C;
  const float sc = affine ? scale[(long)g * C + cm] : 1.0f, sh = affine ? shift[(long)g * C + cm] : 0.0f;
  float gsc[PDE_AMD_MAX_FIELDS], gsh[PDE_AMD_MAX_FIELDS];
  for (int f = 0; f < gs.nf; ++f) {
    gsc[f] = affine ? scale[(long)g * C + gs.ch[f]] : 1.0f;
    gsh[f] = affine ? shift[(long)g * C + gs.ch[f]] : 0.0f;
  }
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  float amax = 0.0f;
  for (long item = blk; item < items; item += blocks_per_group) {
    const long t = item / chunks;
    const unsigned p0 = (unsigned)(item % chunks) * TILE_PTS;
    const unsigned np = min((unsigned)TILE_PTS, (unsigned)gs.S - p0);
    const long src = (((long)g * gs.frames + t) * gs.S + p0) * C;
    const unsigned n4 = np * C / 4;
    for (unsigned i = tid; i < n4; i += TILE_PTS) {
      reinterpret_cast<f4*>(tp)[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(pred + src) + i);      // streamed once
      reinterpret_cast<f4*>(tq)[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(target + src) + i);
    }
    __syncthreads();
    if (tid < np) {
      const long dst = ((long)g * (gs.frames + 1) + t + 1) * gs.S + p0 + tid;
      for (int f = 0; f < gs.nf; ++f) {
        float v = tp[tid * C + (unsigned)gs.ch[f]];
        if (affine) v = __fadd_rn(__fmul_rn(v, gsc[f]), gsh[f]);
        gs.fields[f][dst] = v;
      }
    }
    if (tid < NTS) {
      for (unsigned e = tid; e < np * C; e += NTS) {
        const float pv = affine ? __fadd_rn(__fmul_rn(tp[e], sc), sh) : tp[e], gv = tq[e];
        const float d = pv - gv;
        a0 += (double)(d * d); a1 += (double)(gv * gv); a2 += (double)gv; a3 += (double)pv;
        amax = fmaxf(amax, fabsf(d));
      }
    }
    __syncthreads();
  }
  double* s_acc = reinterpret_cast<double*>(tile);      // [NSTAT][NTS]
  if (tid < NTS) {
    s_acc[0 * NTS + tid] = a0; s_acc[1 * NTS + tid] = a1; s_acc[2 * NTS + tid] = a2; s_acc[3 * NTS + tid] = a3;
    s_acc[4 * NTS + tid] = 0.0; s_acc[5 * NTS + tid] = (double)amax;
  }
  __syncthreads();
  for (unsigned w = tid; w < (unsigned)(NSTAT * C); w += TILE_PTS) {
    const unsigned s = w / C, c = w % C;
    double r = 0.0;
    if (s < 5) { for (unsigned k = c; k < NTS; k += C) r += s_acc[s * NTS + k]; }
    else { for (unsigned k = c; k < NTS; k += C) r = fmax(r, s_acc[s * NTS + k]); }
    partial[((long)blockIdx.x * C + c) * NSTAT + s] = r;
  }
}

// the tiled kernels need whole float4 rows per chunk and frame, and an LDS image of 256 points
inline bool tile_path_ok(long S, int C, const void* p0, const void* p1) {
  return C <= 40 && (S * C) % 4 == 0 && (TILE_PTS * C) % 4 == 0 && S < (1L << 31) &&
         ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1)) & 15) == 0;
}

// (vrmse, rmse) of the listed channels from one group's statistics: a single wave, one channel per lane
struct VrmseArgs {
  const double* stats;      // [C][NSTAT]
  int32_t ch[64];
  int n;
  double rows;
};
__global__ __launch_bounds__(64) void vrmse_kernel(const VrmseArgs a, float* __restrict__ out) {
  const int i = threadIdx.x;
  double v = 0.0, d2 = 0.0;
  if (i < a.n) {
    const double* st = a.stats + (long)a.ch[i] * NSTAT;
    d2 = st[0];
    const double mean_g = st[2] / a.rows;
    const double var = st[1] / a.rows - mean_g * mean_g;
    v = sqrt((d2 / a.rows) / (var + 1e-8));
  }
  v = wave_sum_d(v);
  d2 = wave_sum_d(d2);
  if (i == 0) {
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    out[0] = a.n ? (float)(v / a.n) : (float)nan;                                   // mean over an empty channel list
    out[1] = a.n ? (float)sqrt(d2 / (a.rows * a.n) + 1e-8) : (float)nan;
  }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP; }


// ------------------------------------------------------------------------------------------------
// Radial Fourier-band sums: out[g][bin][t] = sum over the wavenumbers of the bin of |F[g][point][t]|^2.
// F is complex64 [groups][points][nt] (re, im interleaved); the bins arrive as a CSR list (bin_start, point index),
// so every (group, bin) pair is one workgroup that adds its members in a fixed order: repeatable, no atomics.
// Lanes are laid out as (member, frame) with TT frames side by side, TT a power of two <= 64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_power_kernel(const float2* __restrict__ F, const int* __restrict__ bin_start,
                                                         const int* __restrict__ point, long points, int nt, int TT,
                                                         int nbins, float* __restrict__ out) {
  __shared__ double s_acc[256];
  const int bin = blockIdx.x;
  const long g = blockIdx.y;
  const int lo = bin_start[bin], hi = bin_start[bin + 1];
  const int tl = threadIdx.x % TT, ml = threadIdx.x / TT, MS = 256 / TT;
  const float2* Fg = F + g * points * nt;
  for (int t0 = 0; t0 < nt; t0 += TT) {
    const int t = t0 + tl;
    double acc = 0.0;
    if (t < nt) {
      for (int m = lo + ml; m < hi; m += MS) {
        const float2 v = Fg[(long)point[m] * nt + t];
        acc += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
      }
    }
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < TT && t < nt) {
      double s = 0.0;
      for (int j = 0; j < MS; ++j) s += s_acc[j * TT + threadIdx.x];
      out[(g * nbins + bin) * nt + t] = (float)s;
    }
    __syncthreads();
  }
}
}  // namespace
}  // namespace pde_amd

using namespace pde_amd;

extern "C" {

size_t pde_amd_pair_stats_workspace_bytes(int64_t groups, int64_t rows, int32_t channels) {
  PairPlan pl;
  if (!pair_plan(groups, rows, channels, pl)) return 0;
  return (size_t)groups * pl.blocks_per_group * channels * NSTAT * sizeof(double);
}

int pde_amd_pair_stats(const float* pred, const float* target, int64_t groups, int64_t rows, int32_t channels,
                       float rse_eps, double* stats, void* workspace, size_t workspace_bytes, void* stream) {
  return pde_amd_pair_stats_affine(pred, target, groups, rows, channels, rse_eps, nullptr, nullptr, stats, workspace, workspace_bytes, stream);
}

int pde_amd_pair_stats_affine(const float* pred, const float* target, int64_t groups, int64_t rows, int32_t channels,
                              float rse_eps, const float* scale, const float* shift, double* stats, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return pde_amd_pair_stats_gather(pred, target, groups, rows, channels, rse_eps, scale, shift, stats, 0, nullptr, 0, nullptr,
                                   workspace, workspace_bytes, stream);
}

int pde_amd_pair_stats_gather(const float* pred, const float* target, int64_t groups, int64_t rows, int32_t channels,
                              float rse_eps, const float* scale, const float* shift, double* stats, int64_t frames,
                              const int32_t* field_channels, int32_t n_fields, float* const* fields_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if ((scale == nullptr) != (shift == nullptr)) return PDE_AMD_EINVAL;
  GatherSide gs = {};
  if (n_fields != 0) {
    if (n_fields < 0 || n_fields > PDE_AMD_MAX_FIELDS || !field_channels || !fields_out || frames <= 0 || rows % frames != 0) return PDE_AMD_EINVAL;
    gs.nf = n_fields; gs.frames = frames; gs.S = rows / frames;
    for (int f = 0; f < n_fields; ++f) {
      if (field_channels[f] < 0 || field_channels[f] >= channels || !fields_out[f]) return PDE_AMD_EINVAL;
      for (int k = 0; k < f; ++k)
        if (field_channels[k] == field_channels[f]) return PDE_AMD_EINVAL;      // one field per channel
      gs.ch[f] = field_channels[f]; gs.fields[f] = fields_out[f];
    }
  }
  if (groups == 0 || rows == 0) {
    if (groups < 0 || rows < 0 || channels <= 0) return PDE_AMD_EINVAL;
    if (groups > 0 && stats) (void)hipMemsetAsync(stats, 0, (size_t)groups * channels * NSTAT * sizeof(double), (hipStream_t)stream);
    return PDE_AMD_OK;
  }
  PairPlan pl;
  if (!pair_plan(groups, rows, channels, pl)) return PDE_AMD_ESHAPE;
  if (!pred || !target || !stats || !workspace) return PDE_AMD_EINVAL;
  if (workspace_bytes < pde_amd_pair_stats_workspace_bytes(groups, rows, channels)) return PDE_AMD_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace;
  if (gs.nf > 0 && rse_eps < 0.0f && tile_path_ok(gs.S, channels, pred, target) && TILE_PTS / channels >= 1 &&
      (size_t)2 * TILE_PTS * channels * sizeof(float) <= 64 * 1024 &&      // two LDS images; wider tensors take the dword kernel's gather
      (size_t)NSTAT * TILE_PTS * sizeof(double) <= (size_t)2 * TILE_PTS * channels * sizeof(float)) {
    // one sweep with coalesced field writes; a sample's (frame, chunk) items are dealt to at most as many workgroups as the
    // workspace was sized for
    const long items = gs.frames * ((gs.S + TILE_PTS - 1) / TILE_PTS);
    const int bpg = (int)(items < pl.blocks_per_group ? items : pl.blocks_per_group);
    hipLaunchKernelGGL(stats_gather_tile_kernel, dim3((unsigned)(groups * bpg)), dim3(TILE_PTS), (size_t)2 * TILE_PTS * channels * sizeof(float),
                       st, pred, target, (int)channels, bpg, scale, shift, partial, gs);
    hipLaunchKernelGGL(pair_stats_finalize_kernel, dim3((unsigned)((NSTAT * channels + 3) / 4), (unsigned)groups), dim3(256), 0, st, partial,
                       bpg, (int)channels, stats);
    return check_launch();
  }
  if (gs.nf == 0 && channels <= 256 && (rows * (int64_t)channels) % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target)) & 15) == 0) {
    // 16-byte loads; the workgroups of a group share its float4s in contiguous runs of whole steps, at most as many
    // workgroups as the workspace was sized for
    const int ntq = (256 / channels) * channels;
    const long n4 = rows * (long)channels / 4, steps = (n4 + ntq - 1) / ntq;
    const long spb = (steps + pl.blocks_per_group - 1) / pl.blocks_per_group;
    const int bpg = (int)((steps + spb - 1) / spb);
    const size_t ldsq = (size_t)NSTAT * 4 * ntq * sizeof(double);
    if (rse_eps >= 0.0f)
      hipLaunchKernelGGL(pair_stats_q_kernel<true>, dim3((unsigned)(groups * bpg)), dim3(ntq), ldsq, st, pred, target, (long)rows,
                         (int)channels, spb, bpg, rse_eps, scale, shift, partial);
    else
      hipLaunchKernelGGL(pair_stats_q_kernel<false>, dim3((unsigned)(groups * bpg)), dim3(ntq), ldsq, st, pred, target, (long)rows,
                         (int)channels, spb, bpg, rse_eps, scale, shift, partial);
    hipLaunchKernelGGL(pair_stats_finalize_kernel, dim3((unsigned)((NSTAT * channels + 3) / 4), (unsigned)groups), dim3(256), 0, st, partial,
                       bpg, (int)channels, stats);
    return check_launch();
  }
  const size_t lds = (size_t)NSTAT * pl.nt * sizeof(double);
  // a negative rse_eps leaves the relative-squared-error statistic out (0): its division is a fifth of the kernel's arithmetic
  if (rse_eps >= 0.0f)
    hipLaunchKernelGGL(pair_stats_kernel<true>, dim3((unsigned)(groups * pl.blocks_per_group)), dim3(pl.nt), lds, st, pred, target, (long)rows,
                       (int)channels, pl.nt / channels, pl.steps_per_block, pl.blocks_per_group, rse_eps, scale, shift, partial, gs);
  else
    hipLaunchKernelGGL(pair_stats_kernel<false>, dim3((unsigned)(groups * pl.blocks_per_group)), dim3(pl.nt), lds, st, pred, target, (long)rows,
                       (int)channels, pl.nt / channels, pl.steps_per_block, pl.blocks_per_group, rse_eps, scale, shift, partial, gs);
  hipLaunchKernelGGL(pair_stats_finalize_kernel, dim3((unsigned)((NSTAT * channels + 3) / 4), (unsigned)groups), dim3(256), 0, st, partial,
                     pl.blocks_per_group, (int)channels, stats);
  return check_launch();
}

int pde_amd_vrmse_from_stats(const double* stats, int32_t channels, const int32_t* valid, int32_t n_valid, int64_t rows,
                             float* out, void* stream) {
  if (!stats || !out || channels <= 0 || n_valid < 0 || n_valid > 64 || rows <= 0 || (n_valid && !valid)) return PDE_AMD_EINVAL;
  VrmseArgs a;
  a.stats = stats; a.n = n_valid; a.rows = (double)rows;
  for (int i = 0; i < n_valid; ++i) {
    if (valid[i] < 0 || valid[i] >= channels) return PDE_AMD_EINVAL;
    a.ch[i] = valid[i];
  }
  hipLaunchKernelGGL(vrmse_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
  return check_launch();
}

int pde_amd_band_power(const float* spectrum, int64_t groups, int64_t points, int32_t nt, const int32_t* bin_start,
                       const int32_t* point_index, int32_t nbins, float* out, void* stream) {
  if (groups < 0 || points < 0 || nt < 0 || nbins < 0) return PDE_AMD_EINVAL;
  if (groups == 0 || nbins == 0 || nt == 0) return PDE_AMD_OK;
  if (!spectrum || !bin_start || !point_index || !out) return PDE_AMD_EINVAL;
  if (groups > 65535 || points * (int64_t)nt >= (1LL << 31)) return PDE_AMD_ESHAPE;
  int TT = 1;
  while (TT < nt && TT < 64) TT *= 2;
  hipLaunchKernelGGL(band_power_kernel, dim3((unsigned)nbins, (unsigned)groups), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(spectrum), bin_start, point_index, (long)points, (int)nt, TT, (int)nbins, out);
  return check_launch();
}

static int fill_term_args(TermArgs& a, const float* pred, const float* target, int64_t outer, const int32_t* dims,
                          const pde_amd_term* t, const int32_t* indices, const float* scale = nullptr,
                          const float* shift = nullptr, int64_t outer_per_sample = 1) {
  if ((scale == nullptr) != (shift == nullptr) || (scale && outer_per_sample < 1)) return PDE_AMD_EINVAL;
  a.scale = scale; a.shift = shift; a.outer_per_sample = outer_per_sample;
  if (t->n_taps < 1 || t->n_taps > 3 || t->axis < 0 || t->axis > 2) return PDE_AMD_EINVAL;
  if (t->use_target && !target) return PDE_AMD_EINVAL;
  a.pred = pred; a.target = target; a.indices = indices; a.outer = outer;
  a.stride[3] = 1; a.stride[2] = dims[3]; a.stride[1] = (long)dims[2] * dims[3]; a.stride[0] = (long)dims[1] * dims[2] * dims[3];
  a.outer_stride = (long)dims[0] * dims[1] * dims[2] * dims[3];
  a.t = *t;
  a.count = outer;
  for (int k = 0; k < 4; ++k) {
    if (t->n[k] < 0) return PDE_AMD_EINVAL;
    a.count *= t->n[k];
  }
  return PDE_AMD_OK;
}

size_t pde_amd_terms_workspace_bytes(int32_t n_terms) {
  return n_terms <= 0 ? 0 : (size_t)n_terms * 1024 * sizeof(double);
}

int pde_amd_terms_forward(const float* pred, const float* target, int64_t outer, const int32_t* dims,
                          const pde_amd_term* terms, int32_t n_terms, const int32_t* indices, double* sums,
                          void* workspace, size_t workspace_bytes, void* stream) {
  return pde_amd_terms_forward_affine(pred, target, outer, dims, terms, n_terms, indices, nullptr, nullptr, 1, sums, workspace,
                                      workspace_bytes, stream);
}

int pde_amd_terms_forward_affine(const float* pred, const float* target, int64_t outer, const int32_t* dims,
                                 const pde_amd_term* terms, int32_t n_terms, const int32_t* indices, const float* scale,
                                 const float* shift, int64_t outer_per_sample, double* sums, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!pred || !dims || !terms || !sums || !indices || n_terms < 1 || n_terms > PDE_AMD_MAX_TERMS || outer < 0) return PDE_AMD_EINVAL;
  if (!workspace || workspace_bytes < pde_amd_terms_workspace_bytes(n_terms)) return PDE_AMD_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)workspace;
  TermBatch tb;
  tb.n_terms = n_terms;
  int* blk_start = tb.blk.start;
  for (int k = 0; k <= PDE_AMD_MAX_TERMS; ++k) blk_start[k] = 0;
  for (int k = 0; k < n_terms; ++k) {
    const int rc = fill_term_args(tb.base, pred, target, outer, dims, &terms[k], indices, scale, shift, outer_per_sample);
    if (rc) return rc;
    tb.t[k] = terms[k];
    tb.count[k] = tb.base.count;
    const int nb = tb.base.count > 0 ? term_blocks(tb.base.count) : 0;
    blk_start[k + 1] = blk_start[k] + nb;
  }
  for (int k = n_terms; k < PDE_AMD_MAX_TERMS; ++k) blk_start[k + 1] = blk_start[n_terms];
  if (blk_start[n_terms] > 0)
    hipLaunchKernelGGL(term_fwd_batch_kernel, dim3((unsigned)blk_start[n_terms]), dim3(256), 0, st, tb, partial);
  hipLaunchKernelGGL(term_finalize_kernel, dim3((unsigned)n_terms), dim3(64), 0, st, partial, tb.blk, (int)n_terms, sums);
  return check_launch();
}

int pde_amd_terms_backward(const float* pred, const float* target, int64_t outer, const int32_t* dims,
                           const pde_amd_term* terms, int32_t n_terms, const int32_t* indices, const float* scales,
                           float* grad, void* stream) {
  return pde_amd_terms_backward_affine(pred, target, outer, dims, terms, n_terms, indices, nullptr, nullptr, 1, 0, scales, grad, stream);
}

int pde_amd_terms_backward_affine(const float* pred, const float* target, int64_t outer, const int32_t* dims,
                                  const pde_amd_term* terms, int32_t n_terms, const int32_t* indices, const float* scale,
                                  const float* shift, int64_t outer_per_sample, int32_t disjoint, const float* scales,
                                  float* grad, void* stream) {
  if (!pred || !dims || !terms || !scales || !grad || !indices || n_terms < 1 || n_terms > PDE_AMD_MAX_TERMS || outer < 0) return PDE_AMD_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (disjoint && n_terms > 1) {
    TermBatch tb;
    tb.n_terms = n_terms;
    int* blk_start = tb.blk.start;
    blk_start[0] = 0;
    for (int k = 0; k < n_terms; ++k) {
      const int rc = fill_term_args(tb.base, pred, target, outer, dims, &terms[k], indices, scale, shift, outer_per_sample);
      if (rc) return rc;
      tb.t[k] = terms[k];
      tb.count[k] = tb.base.count;
      blk_start[k + 1] = blk_start[k] + (tb.base.count > 0 ? term_blocks(tb.base.count) : 0);
    }
    for (int k = n_terms; k < PDE_AMD_MAX_TERMS; ++k) blk_start[k + 1] = blk_start[n_terms];
    if (blk_start[n_terms] > 0)
      hipLaunchKernelGGL(term_bwd_batch_kernel, dim3((unsigned)blk_start[n_terms]), dim3(256), 0, st, tb, scales, grad);
    return check_launch();
  }
  for (int k = 0; k < n_terms; ++k) {      // one launch per term, in order: terms may overlap each other, a term never overlaps itself
    TermArgs a;
    const int rc = fill_term_args(a, pred, target, outer, dims, &terms[k], indices, scale, shift, outer_per_sample);
    if (rc) return rc;
    if (a.count > 0) hipLaunchKernelGGL(term_bwd_kernel, dim3(term_blocks(a.count)), dim3(256), 0, st, a, scales + k, grad);
  }
  return check_launch();
}

static int fill_gather(GatherArgs& a, int64_t B, int64_t T_out, int64_t T_in, int64_t S, int32_t C, const int32_t* channels,
                       int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose) {
  if (B < 0 || T_out < 0 || T_in < 1 || S < 0 || C < 1 || !channels || n_fields < 1 || n_fields > PDE_AMD_MAX_FIELDS) return PDE_AMD_EINVAL;
  if (transpose && (rows < 1 || cols < 1 || S % ((long)rows * cols) != 0)) return PDE_AMD_ESHAPE;
  a.nf = n_fields; a.C = C; a.B = B; a.T_out = T_out; a.T_in = T_in; a.S = S;
  a.rows = rows; a.cols = cols; a.transpose = transpose;
  a.scale = a.shift = nullptr;
  for (int f = 0; f < n_fields; ++f) {
    if (channels[f] < 0 || channels[f] >= C) return PDE_AMD_EINVAL;
    a.ch[f] = channels[f];
  }
  return PDE_AMD_OK;
}

int pde_amd_gather_fields(const float* output, const float* input, int64_t B, int64_t T_out, int64_t T_in, int64_t S, int32_t C,
                          const int32_t* channels, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                          float* const* fields, void* stream) {
  return pde_amd_gather_fields_affine(output, input, B, T_out, T_in, S, C, channels, n_fields, rows, cols, transpose, nullptr, nullptr,
                                      fields, stream);
}

int pde_amd_gather_fields_affine(const float* output, const float* input, int64_t B, int64_t T_out, int64_t T_in, int64_t S, int32_t C,
                                 const int32_t* channels, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                                 const float* scale, const float* shift, float* const* fields, void* stream) {
  GatherArgs a;
  const int rc = fill_gather(a, B, T_out, T_in, S, C, channels, n_fields, rows, cols, transpose);
  if (rc) return rc;
  if ((scale == nullptr) != (shift == nullptr)) return PDE_AMD_EINVAL;
  a.scale = scale; a.shift = shift;
  if (!output || !input || !fields) return PDE_AMD_EINVAL;
  a.output = output; a.input = input;
  for (int f = 0; f < n_fields; ++f) { if (!fields[f]) return PDE_AMD_EINVAL; a.fields[f] = fields[f]; }
  const long npts = B * (T_out + 1) * S;
  if (npts == 0) return PDE_AMD_OK;
  if (tile_path_ok(S, C, output, input)) {
    long items = B * (T_out + 1) * ((S + TILE_PTS - 1) / TILE_PTS);
    if (items > 256 * 32) items = 256 * 32;
    hipLaunchKernelGGL(gather_tile_kernel, dim3((unsigned)items), dim3(TILE_PTS), (size_t)TILE_PTS * C * sizeof(float), (hipStream_t)stream, a);
    return check_launch();
  }
  long nb = (npts + 255) / 256;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

int pde_amd_scatter_field_grads(const float* const* field_grads, int64_t B, int64_t T_out, int64_t S, int32_t C,
                                const int32_t* channels, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                                float* grad_output, void* stream) {
  return pde_amd_scatter_field_grads_affine(field_grads, B, T_out, S, C, channels, n_fields, rows, cols, transpose, nullptr,
                                            grad_output, stream);
}

int pde_amd_scatter_field_grads_affine(const float* const* field_grads, int64_t B, int64_t T_out, int64_t S, int32_t C,
                                       const int32_t* channels, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                                       const float* scale, float* grad_output, void* stream) {
  GatherArgs a;
  const int rc = fill_gather(a, B, T_out, 1, S, C, channels, n_fields, rows, cols, transpose);
  if (rc) return rc;
  a.scale = scale;
  if (!field_grads || !grad_output) return PDE_AMD_EINVAL;
  for (int f = 0; f < n_fields; ++f) { if (!field_grads[f]) return PDE_AMD_EINVAL; a.gfields[f] = field_grads[f]; }
  const long n = B * T_out * S * (long)C;
  if (n == 0) return PDE_AMD_OK;
  if (tile_path_ok(S, C, grad_output, grad_output)) {
    long items = B * T_out * ((S + TILE_PTS - 1) / TILE_PTS);
    if (items > 256 * 32) items = 256 * 32;
    hipLaunchKernelGGL(scatter_tile_kernel, dim3((unsigned)items), dim3(TILE_PTS), (size_t)TILE_PTS * C * sizeof(float), (hipStream_t)stream, a, grad_output);
    return check_launch();
  }
  const long frame_elems = S * (long)C;
  if (frame_elems >= (1L << 31) || B * T_out > 65535) return PDE_AMD_ESHAPE;
  const bool vec = frame_elems % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_output) & 15) == 0;
  long nbx = (frame_elems / (vec ? 4 : 1) + 256 * 8 - 1) / (256 * 8);      // about 8 stores per thread
  if (nbx > 4096) nbx = 4096;
  const dim3 grid((unsigned)nbx, (unsigned)(B * T_out));
  if (vec) hipLaunchKernelGGL(scatter_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a, grad_output);
  else hipLaunchKernelGGL(scatter_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a, grad_output);
  return check_launch();
}

}  // extern "C"
