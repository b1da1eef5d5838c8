// Microbenchmark: dependent-chain f32 FMA issue rate vs number of independent chains and resident waves.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a[CH];
  for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / CH; ++u) {
#pragma unroll
      for (int i = 0; i < CH; ++i) a[i] = __builtin_fmaf(a[i], s, 0.5f);
    }
  }
  float r = 0;
  for (int i = 0; i < CH; ++i) r += a[i];
  if (r == 12345.678f) out[0] = r;
}
template <int CH>
void run(float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<CH>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("chains %d waves/SIMD %d: %.3f ms -> %.2f ns per instr per wave, %.2f ns per instr per SIMD\n", CH, wps, best,
           best * 1e6 / (iters * 32.0), best * 1e6 / (iters * 32.0 * wps));
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  run<1>(out, e0, e1); run<2>(out, e0, e1); run<4>(out, e0, e1); run<8>(out, e0, e1);
  return 0;
}
