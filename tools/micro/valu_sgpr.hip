// Microbenchmark: does an SGPR / literal operand slow a f32 VALU instruction down?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a[8], b[8], c[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 0.5f + 1.0f; c[i] = a[i] * 0.25f + 2.0f; }
  unsigned long long m = (threadIdx.x & 1) ? 0x5555555555555555ull : 0xaaaaaaaaaaaaaaaaull;
  m = __builtin_amdgcn_readfirstlane((unsigned)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(c[i]));
        if (MODE == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "s"(s), "v"(c[i]));
        if (MODE == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 4) asm volatile("v_fmamk_f32 %0, %1, 0x41000000, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 5) asm volatile("v_mul_f32 %0, 0x41000000, %1" : "=v"(a[i]) : "v"(c[i]));
        if (MODE == 6) asm volatile("v_mul_f32 %0, 2.0, %1" : "=v"(a[i]) : "v"(c[i]));
        if (MODE == 7) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]), "s"(m));
        if (MODE == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "s"(s), "v"(c[i]));
        if (MODE == 9) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i] + b[i] + c[i];
  if (r == 12345.678f) out[0] = r;
}
template <int MODE>
void run(const char* name, float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-34s waves/SIMD %d: %.2f ns per instr per SIMD\n", name, wps, best * 1e6 / (iters * 32.0 * wps));
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  run<0>("fmac v += v*v", out, e0, e1);
  run<1>("fmac v += s*v", out, e0, e1);
  run<8>("fma d = v*s+v (VOP3)", out, e0, e1);
  run<3>("mul v*v", out, e0, e1);
  run<2>("mul s*v", out, e0, e1);
  run<5>("mul literal*v", out, e0, e1);
  run<6>("mul 2.0*v", out, e0, e1);
  run<4>("fmamk v*K+v", out, e0, e1);
  run<9>("sub v-v", out, e0, e1);
  run<7>("cndmask sgpr mask", out, e0, e1);
  return 0;
}
