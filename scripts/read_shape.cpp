#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
template <int SV, bool NT>
__global__ __launch_bounds__(256) void stream_read(const v4* __restrict__ src, size_t n4_per_wg, float* sink) {
  extern __shared__ char dyn[];
  const v4* p = src + (size_t)blockIdx.x * n4_per_wg + threadIdx.x;
  v4 acc = (v4)(0.f);
  for (size_t i = 0; i < n4_per_wg; i += 256 * SV) {
    v4 v[SV];
#pragma unroll
    for (int u = 0; u < SV; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + 256 * u) : p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < SV; ++u) acc += v[u];
  }
  if (acc.x == 123.456f) sink[0] = acc.y + dyn[threadIdx.x];
}
int main() {
  const size_t GB = 1ull << 30;
  v4* big; CK(hipMalloc(&big, 6 * GB)); CK(hipMemset(big, 0, 6 * GB));
  float* sink; CK(hipMalloc(&sink, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t total : {6 * GB, 3 * GB, 3 * GB / 2, GB / 2}) {
    for (size_t chunk : {(size_t)32 << 10, (size_t)96 << 10, (size_t)256 << 10, (size_t)1 << 20}) {
      for (int lds : {0, 41}) {
        const size_t wgs = total / chunk, n4 = chunk / 16;
        float best = 1e9;
        for (int it = 0; it < 6; ++it) {
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL((stream_read<8, true>), dim3(wgs), dim3(256), lds * 1024, 0, big, n4, sink);
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        printf("total %5.2f GB  chunk %5zu KB per workgroup (%6zu workgroups)  LDS %2d KB: %7.1f us  %5.0f GB/s\n", total / 1e9, chunk >> 10, wgs, lds, best * 1e3, total / best / 1e6);
      }
    }
  }
  return 0;
}
