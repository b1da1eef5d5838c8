// Microbenchmark: SIMD issue rate of f32 VALU forms (operand kinds) at 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
// MODE 0: fma(v, s, const)   1: fmac v += v*v (3 VGPR reads)   2: fma d = v*v + v (4 distinct regs)   3: mul v*v   4: add v+v
// MODE 5: pk_fma 3 VGPR pairs   6: v*s + v (2 VGPR + SGPR)   7: cndmask
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s, float s2) {
  float a[8], b[8], c[8];
  f2 pa[8], pb[8], pc[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = threadIdx.x * 0.001f + i; b[i] = a[i] * 0.5f + 1.0f; c[i] = a[i] * 0.25f + 2.0f;
    pa[i] = f2{a[i], b[i]}; pb[i] = f2{b[i], c[i]}; pc[i] = f2{c[i], a[i]};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);
        if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]), "v"(a[(i + 1) & 7]));
        if (MODE == 3) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[i]) : "v"(pb[i]), "v"(pc[i]));
        if (MODE == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(c[i]));
        if (MODE == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(c[i]));
        if (MODE == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "s"(s), "v"(c[i]), "s"(s));
        if (MODE == 9) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[i]));
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i] + b[i] + c[i] + pa[i][0] + pa[i][1];
  if (r == 12345.678f) out[0] = r;
}
template <int MODE>
void run(const char* name, float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  for (int wps = 2; wps <= 4; wps *= 2) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999f, 0.5f);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-34s waves/SIMD %d: %.2f ns per instr per SIMD\n", name, wps, best * 1e6 / (iters * 32.0 * wps));
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  run<0>("fma v,s,const", out, e0, e1);
  run<1>("fmac v += v*v", out, e0, e1);
  run<2>("fma d = v*v+v", out, e0, e1);
  run<3>("mul v*v", out, e0, e1);
  run<4>("add v+v", out, e0, e1);
  run<9>("sub v-v (same reg)", out, e0, e1);
  run<5>("pk_fma 3 vgpr pairs", out, e0, e1);
  run<6>("fmac v += s*v", out, e0, e1);
  run<8>("fma d = s*v+s", out, e0, e1);
  run<7>("cndmask", out, e0, e1);
  return 0;
}
