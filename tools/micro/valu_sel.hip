// Microbenchmark: SIMD issue rate of selects (v_cndmask forms, compare + select pairs) against v_fma, at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0: v_fma (reference)  1: v_cndmask_b32_e32 (vcc)  2: v_cndmask_b32_e64 (sgpr pair)  3: v_cmp_gt_f32 + v_cndmask (vcc)
// MODE 4: v_cmp_gt_f32_e64 into an sgpr pair + v_cndmask_e64  5: compiler select (b > 0.5 ? c : a)  6: v_max_f32  7: v_med3_f32
// MODE 8: v_mul_f32 with clamp
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a[8], b[8], c[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 0.5f + s; c[i] = a[i] * 0.25f + 2.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 1) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 3) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]) : "vcc");
        if (MODE == 4) asm volatile("v_cmp_gt_f32_e64 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]) : "s20", "s21");
        if (MODE == 5) a[i] = (b[i] > 0.5f) ? c[i] * a[i] : a[i];
        if (MODE == 6) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 7) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 8) asm volatile("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i] + b[i] + c[i];
  if (r == 12345.678f) out[0] = r;
}
template <int MODE>
void run(const char* name, int per, float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s waves/SIMD %d: %.2f ns per instruction per SIMD\n", name, wps, best * 1e6 / (iters * 32.0 * per * wps));
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  run<0>("v_fma_f32", 1, out, e0, e1);
  run<1>("v_cndmask_b32_e32 (vcc)", 1, out, e0, e1);
  run<2>("v_cndmask_b32_e64 (sgpr pair)", 1, out, e0, e1);
  run<3>("v_cmp + v_cndmask via vcc (2 instr)", 2, out, e0, e1);
  run<4>("v_cmp_e64 + v_cndmask_e64 via sgprs (2 instr)", 2, out, e0, e1);
  run<5>("compiler: b > 0.5 ? c * a : a", 3, out, e0, e1);
  run<6>("v_max_f32", 1, out, e0, e1);
  run<7>("v_med3_f32", 1, out, e0, e1);
  run<8>("v_mul_f32 clamp", 1, out, e0, e1);
  return 0;
}
