// Microbenchmark: issue cost of the f32 VALU forms the stencil kernels are made of, alone and mixed, at 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0 add vop2   1 add via fma(a,1.0,b)   2 sub   3 mul   4 fma vvv   5 fmac   6 mix fma,add alternate   7 mix fma,fma(1.0)   8 add e64
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[8], b[8], c[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 0.5f + 1.0f; c[i] = a[i] * 0.25f + 2.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 1) asm volatile("v_fma_f32 %0, %1, 1.0, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 2) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]), "v"(a[(i + 1) & 7]));
        if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 6) { if (i & 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                         else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]), "v"(a[(i + 1) & 7])); }
        if (MODE == 7) { if (i & 1) asm volatile("v_fma_f32 %0, %1, 1.0, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
                         else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]), "v"(a[(i + 1) & 7])); }
        if (MODE == 8) asm volatile("v_add_f32_e64 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 9) asm volatile("v_mul_f32 %0, 0.5, %1" : "=v"(a[i]) : "v"(b[i]));
        if (MODE == 10) asm volatile("v_add_f32 %0, %1, %1" : "=v"(a[i]) : "v"(b[i]));
        if (MODE == 11) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
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
      hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-34s waves/SIMD %d: %.2f ns per instr per SIMD\n", name, wps, best * 1e6 / (iters * 32.0 * wps));
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  run<0>("add v+v (vop2)", out, e0, e1);
  run<8>("add v+v (e64)", out, e0, e1);
  run<10>("add v+same v", out, e0, e1);
  run<1>("add as fma(v,1.0,v)", out, e0, e1);
  run<2>("sub v-v", out, e0, e1);
  run<3>("mul v*v", out, e0, e1);
  run<9>("mul 0.5*v", out, e0, e1);
  run<4>("fma d=v*v+v", out, e0, e1);
  run<11>("fma d=v*w+w (2 distinct)", out, e0, e1);
  run<5>("fmac v+=v*v", out, e0, e1);
  run<6>("mix fma/add", out, e0, e1);
  run<7>("mix fma/fma1.0", out, e0, e1);
  return 0;
}
