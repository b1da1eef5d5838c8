// Microbenchmark: VALU issue rate per SIMD vs resident waves (plain f32 FMA and packed f32 FMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a[8];
  f2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.0f}; }
  const f2 s2 = {s, s};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], s, 0.5f);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], s2, s2);
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i] + p[i][0] + p[i][1];
  if (r == 12345.678f) out[0] = r;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode)
    for (int wps = 1; wps <= 8; wps *= 2) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999f);
        else hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double insts = (double)iters * 32 * wps;          // VALU instructions per SIMD
      printf("mode %s waves/SIMD %d: %.3f ms  -> %.2f ns per VALU instr per SIMD (%.1f TFLOP/s)\n", mode ? "pk_fma" : "fma", wps, best,
             best * 1e6 / insts, (mode ? 4.0 : 2.0) * 64 * insts * 1024 / (best * 1e-3) / 1e12);
    }
  return 0;
}
