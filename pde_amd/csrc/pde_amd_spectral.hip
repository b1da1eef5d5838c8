// Spectral Laplacian of a pair of real planes, one workgroup per pair, the whole transform resident in LDS.
//
// The reference's Gray-Scott loss takes the Laplacian of both concentrations as ifft2(-K2 * fft2(f)).real
// (finetune/pde_loss_verified.py:200-214).  The two real planes A and B of one sample and frame are packed as z = A + iB:
// the operator is linear and real (K2 is even in the wavenumbers), so one complex transform pair gives
// lap(A) = Re L(z) and lap(B) = Im L(z).  A 128 x 128 complex plane is 128 KB: it stays in LDS from the first load to
// the last store, and HBM sees each plane once in and once out.
//
// A length-N transform (N = RA * RB, both radices done in registers) is two steps around one barrier:
//   forward   step 1: thread t of RB takes x[t + RB j], j < RA: DFT_RA over j, times W_N^(t k1), stored at k1 RB + t
//             step 2: for its k1's, DFT_RB over t: X[k1 + RA k2], left at position k1 RB + k2 (digit-swapped order)
//   inverse   step A: for its k1's, inverse DFT_RB over k2, times conj W_N^(t k1), stored at k1 RB + t
//             step B: thread t: inverse DFT_RA over k1: x[t + RB j], stored at its natural place
// Every step reads and writes the same LDS positions of its thread, so one barrier between steps is enough, and the
// spectrum is never brought into natural order: step 2 of the second axis, the multiplication by -K2 / (NX NY) and
// step A of the way back happen in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pde_amd.h"

namespace pde_amd_spectral {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

template <int R> constexpr int bitrev(int i) {
  int r = 0;
  for (int b = 1; b < R; b <<= 1) { r = (r << 1) | (i & 1); i >>= 1; }
  return r;
}

// exp(2 pi i m / 16), m < 8
__device__ constexpr float C16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                     0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
__device__ constexpr float S16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                     1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// DFT of R <= 16 values held in registers, natural order in and out; SIGN -1: forward (e^-i), +1: inverse (unscaled)
template <int R, int SIGN>
__device__ __forceinline__ void dft(float2 (&v)[R]) {
  static_assert(R == 2 || R == 4 || R == 8 || R == 16, "radix");
  float2 w[R];
#pragma unroll
  for (int i = 0; i < R; ++i) w[bitrev<R>(i)] = v[i];
#pragma unroll
  for (int h = 1; h < R; h *= 2) {
#pragma unroll
    for (int s = 0; s < R; s += 2 * h) {
#pragma unroll
      for (int j = 0; j < h; ++j) {
        const int m = j * (8 / h);
        const float2 x = w[s + j + h];
        float2 b;
        if (m == 0) b = x;
        else if (m == 4) b = SIGN < 0 ? make_float2(x.y, -x.x) : make_float2(-x.y, x.x);      // times -+ i
        else b = cmul(x, make_float2(C16[m], SIGN * S16[m]));
        const float2 a = w[s + j];
        w[s + j] = cadd(a, b);
        w[s + j + h] = csub(a, b);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] = w[i];
}

template <int N> struct Split {
  static_assert(N == 16 || N == 32 || N == 64 || N == 128, "length");
  static constexpr int RA = N >= 128 ? 16 : (N >= 32 ? 8 : 4), RB = N / RA;      // 16 x 8, 8 x 8, 8 x 4, 4 x 4
  static constexpr int PER = RA / RB;                                            // step-2 transforms per thread
  __device__ static constexpr int freq(int pos) { return pos / RB + RA * (pos % RB); }      // frequency held at a position after step 2
};

// tw[m] = exp(-2 pi i m / N)
template <int N, int SIGN>
__device__ __forceinline__ void step_outer_fwd(float2* line, int es, int t, const float2* tw) {      // forward step 1
  using S = Split<N>;
  float2 v[S::RA];
#pragma unroll
  for (int j = 0; j < S::RA; ++j) v[j] = line[(t + S::RB * j) * es];
  dft<S::RA, -1>(v);
#pragma unroll
  for (int k1 = 0; k1 < S::RA; ++k1) line[(k1 * S::RB + t) * es] = cmul(v[k1], tw[t * k1]);
}
template <int N>
__device__ __forceinline__ void step_outer_inv(float2* line, int es, int t) {      // inverse step B
  using S = Split<N>;
  float2 v[S::RA];
#pragma unroll
  for (int k1 = 0; k1 < S::RA; ++k1) v[k1] = line[(k1 * S::RB + t) * es];
  dft<S::RA, 1>(v);
#pragma unroll
  for (int j = 0; j < S::RA; ++j) line[(t + S::RB * j) * es] = v[j];
}
template <int N>
__device__ __forceinline__ void step_inner_fwd(float2* line, int es, int u) {      // forward step 2
  using S = Split<N>;
#pragma unroll
  for (int m = 0; m < S::PER; ++m) {
    const int k1 = u + S::RB * m;
    float2 v[S::RB];
#pragma unroll
    for (int t = 0; t < S::RB; ++t) v[t] = line[(k1 * S::RB + t) * es];
    dft<S::RB, -1>(v);
#pragma unroll
    for (int k2 = 0; k2 < S::RB; ++k2) line[(k1 * S::RB + k2) * es] = v[k2];
  }
}
template <int N>
__device__ __forceinline__ void step_inner_inv(float2* line, int es, int u, const float2* tw) {      // inverse step A
  using S = Split<N>;
#pragma unroll
  for (int m = 0; m < S::PER; ++m) {
    const int k1 = u + S::RB * m;
    float2 v[S::RB];
#pragma unroll
    for (int k2 = 0; k2 < S::RB; ++k2) v[k2] = line[(k1 * S::RB + k2) * es];
    dft<S::RB, 1>(v);
#pragma unroll
    for (int t = 0; t < S::RB; ++t) {
      const float2 w = tw[t * k1];
      line[(k1 * S::RB + t) * es] = cmul(v[t], make_float2(w.x, -w.y));
    }
  }
}

template <int NX, int NY> struct Cfg {
  using SX = Split<NX>;
  using SY = Split<NY>;
  static constexpr int LS = NY + 1;                                   // line stride in complex elements: odd, so that lanes
                                                                      // on consecutive rows hit different banks
  static constexpr int TROW = NX * SY::RB, TCOL = NY * SX::RB;        // threads of the row passes / of the column passes
  static constexpr int NT = TROW > TCOL ? TROW : TCOL;
  static constexpr size_t LDS = sizeof(float2) * ((size_t)NX * LS + NX + NY);
};

// a, b -> la, lb: [planes][NX][NY]; mult: [NX][NY], the factor of every wavenumber pair (the reference's -K2)
template <int NX, int NY>
__global__ __launch_bounds__((Cfg<NX, NY>::NT)) void laplacian_pair_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                           float* __restrict__ la, float* __restrict__ lb,
                                                                           const float* __restrict__ mult) {
  using C = Cfg<NX, NY>;
  using SX = typename C::SX;
  using SY = typename C::SY;
  constexpr int LS = C::LS, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  float2* const z = reinterpret_cast<float2*>(s_raw);
  float2* const twx = z + NX * LS;      // exp(-2 pi i m / NX)
  float2* const twy = twx + NX;
  const int id = threadIdx.x;
  const size_t plane = (size_t)blockIdx.x * NX * NY;

  for (int m = id; m < NX + NY; m += NT) {
    const bool isx = m < NX;
    const int k = isx ? m : m - NX;
    float s, c;
    sincospif(2.0f * (float)k / (float)(isx ? NX : NY), &s, &c);
    (isx ? twx : twy)[k] = make_float2(c, -s);
  }
  for (int i = id; i < NX * NY / 4; i += NT) {
    const int r = (4 * i) / NY, c = (4 * i) % NY;
    const float4 va = *reinterpret_cast<const float4*>(a + plane + 4 * (size_t)i);
    const float4 vb = *reinterpret_cast<const float4*>(b + plane + 4 * (size_t)i);
    float2* d = z + r * LS + c;
    d[0] = make_float2(va.x, vb.x); d[1] = make_float2(va.y, vb.y); d[2] = make_float2(va.z, vb.z); d[3] = make_float2(va.w, vb.w);
  }
  __syncthreads();

  const bool row_on = id < C::TROW, col_on = id < C::TCOL;
  const int rr = id % NX, rt = id / NX;      // row passes: lanes <-> rows
  const int cc = id % NY, ct = id / NY;      // column passes: lanes <-> columns
  if (row_on) step_outer_fwd<NY, -1>(z + rr * LS, 1, rt, twy);
  __syncthreads();
  if (row_on) step_inner_fwd<NY>(z + rr * LS, 1, rt);
  __syncthreads();
  if (col_on) step_outer_fwd<NX, -1>(z + cc, LS, ct, twx);
  __syncthreads();
  if (col_on) {      // column step 2, the multiplier, column step A: in registers
    const int kc = SY::freq(cc);
    constexpr float norm = 1.0f / (float)(NX * NY);
#pragma unroll
    for (int m = 0; m < SX::PER; ++m) {
      const int k1 = ct + SX::RB * m;
      float2 v[SX::RB];
#pragma unroll
      for (int t = 0; t < SX::RB; ++t) v[t] = z[(k1 * SX::RB + t) * LS + cc];
      dft<SX::RB, -1>(v);
#pragma unroll
      for (int k2 = 0; k2 < SX::RB; ++k2) {
        const float f = mult[(k1 + SX::RA * k2) * NY + kc] * norm;
        v[k2] = make_float2(v[k2].x * f, v[k2].y * f);
      }
      dft<SX::RB, 1>(v);
#pragma unroll
      for (int t = 0; t < SX::RB; ++t) {
        const float2 w = twx[t * k1];
        z[(k1 * SX::RB + t) * LS + cc] = cmul(v[t], make_float2(w.x, -w.y));
      }
    }
  }
  __syncthreads();
  if (col_on) step_outer_inv<NX>(z + cc, LS, ct);
  __syncthreads();
  if (row_on) step_inner_inv<NY>(z + rr * LS, 1, rt, twy);
  __syncthreads();
  if (row_on) step_outer_inv<NY>(z + rr * LS, 1, rt);
  __syncthreads();

  for (int i = id; i < NX * NY / 4; i += NT) {
    const int r = (4 * i) / NY, c = (4 * i) % NY;
    const float2* d = z + r * LS + c;
    const float2 p0 = d[0], p1 = d[1], p2 = d[2], p3 = d[3];
    *reinterpret_cast<float4*>(la + plane + 4 * (size_t)i) = make_float4(p0.x, p1.x, p2.x, p3.x);
    *reinterpret_cast<float4*>(lb + plane + 4 * (size_t)i) = make_float4(p0.y, p1.y, p2.y, p3.y);
  }
}

template <int NX, int NY>
static int launch(const float* a, const float* b, float* la, float* lb, const float* mult, int64_t planes, hipStream_t stream) {
  using C = Cfg<NX, NY>;
  auto kern = laplacian_pair_kernel<NX, NY>;
  if (C::LDS > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS) != hipSuccess)
    return PDE_AMD_EHIP;
  hipLaunchKernelGGL(kern, dim3((unsigned)planes), dim3(C::NT), C::LDS, stream, a, b, la, lb, mult);
  return hipGetLastError() == hipSuccess ? PDE_AMD_OK : PDE_AMD_EHIP;
}

template <int NX>
static int launch_y(int ny, const float* a, const float* b, float* la, float* lb, const float* mult, int64_t planes, hipStream_t stream) {
  switch (ny) {
    case 16: return launch<NX, 16>(a, b, la, lb, mult, planes, stream);
    case 32: return launch<NX, 32>(a, b, la, lb, mult, planes, stream);
    case 64: return launch<NX, 64>(a, b, la, lb, mult, planes, stream);
    case 128: return launch<NX, 128>(a, b, la, lb, mult, planes, stream);
    default: return PDE_AMD_ESHAPE;
  }
}

}  // namespace pde_amd_spectral

extern "C" {

int pde_amd_spectral_pair_supported(int32_t nx, int32_t ny) {
  auto ok = [](int n) { return n == 16 || n == 32 || n == 64 || n == 128; };
  return ok(nx) && ok(ny) ? 1 : 0;
}

int pde_amd_spectral_laplacian_pair(const float* a, const float* b, float* lap_a, float* lap_b, const float* multiplier,
                                    int64_t planes, int32_t nx, int32_t ny, void* stream_) {
  using namespace pde_amd_spectral;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!a || !b || !lap_a || !lap_b || !multiplier || planes < 0) return PDE_AMD_EINVAL;
  if (!pde_amd_spectral_pair_supported(nx, ny) || planes >= (1LL << 31)) return PDE_AMD_ESHAPE;
  for (const void* p : {(const void*)a, (const void*)b, (const void*)lap_a, (const void*)lap_b})
    if (reinterpret_cast<uintptr_t>(p) & 15) return PDE_AMD_ESHAPE;
  if (planes == 0) return PDE_AMD_OK;
  switch (nx) {
    case 16: return launch_y<16>(ny, a, b, lap_a, lap_b, multiplier, planes, stream);
    case 32: return launch_y<32>(ny, a, b, lap_a, lap_b, multiplier, planes, stream);
    case 64: return launch_y<64>(ny, a, b, lap_a, lap_b, multiplier, planes, stream);
    default: return launch_y<128>(ny, a, b, lap_a, lap_b, multiplier, planes, stream);
  }
}

}  // extern "C"
