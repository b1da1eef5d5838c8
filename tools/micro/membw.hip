// Microbenchmark: streaming read / write / copy rates of plain grid-stride kernels (float4), for use as practical ceilings.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 read, 1 write, 2 write nontemporal, 3 copy, 4 copy with nontemporal store
__global__ __launch_bounds__(256) void k(const f4* __restrict__ src, f4* __restrict__ dst, long n, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (MODE == 0) acc += src[i];
    if (MODE == 1) dst[i] = f4{1.f, 2.f, 3.f, (float)i};
    if (MODE == 2) __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)i}, dst + i);
    if (MODE == 3) dst[i] = src[i];
    if (MODE == 4) __builtin_nontemporal_store(src[i], dst + i);
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}
template <int MODE>
void run(const char* name, const f4* src, f4* dst, long n, float* sink, int bytes_per_elem) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks : {2048, 8192, 32768}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, dst, n, sink);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s blocks %6d: %.3f ms  %.2f TB/s\n", name, blocks, best, (double)n * bytes_per_elem / best / 1e9);
  }
}
int main() {
  const long n = (1200L << 20) / 16;   // 1.2 GiB
  f4 *a, *b; float* sink;
  (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&sink, 4);
  (void)hipMemset(a, 0, n * 16); (void)hipMemset(b, 0, n * 16);
  run<0>("read", a, b, n, sink, 16);
  run<1>("write", a, b, n, sink, 16);
  run<2>("write nontemporal", a, b, n, sink, 16);
  run<3>("copy (bytes moved)", a, b, n, sink, 32);
  run<4>("copy nontemporal store", a, b, n, sink, 32);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); (void)hipMemsetAsync(b, 0, n * 16, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("hipMemsetAsync: %.3f ms  %.2f TB/s\n", ms, (double)n * 16 / ms / 1e9);
  return 0;
}
