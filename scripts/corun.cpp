// What does a projection kernel take from the state pass when both run at once?  Two streams, kernels alone and together:
//   S  : HBM streaming read with the read pass's resource shape (256 threads, `SV` float4 per lane in flight, `SLDS` KB of LDS)
//   M0 : MFMA only (no memory, no LDS): 128 VGPRs, f16 32x32x16
//   M1 : M0 + 32 KB of LDS allocated (occupancy effect only)
//   D  : the pre-split GEMM's tile delivery (L2 -> LDS DMA of K-tile-major planes), no MFMA
//   G  : D + 24 MFMAs per wave and K tile (the GEMM's loop without its epilogue)
// Reported: duration alone, duration together (both kernels re-launched back to back on their streams for the same wall window).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SV, int SLDS>
__global__ __launch_bounds__(256) void stream_read(const v4* __restrict__ src, size_t n4_per_wg, float* sink) {
  extern __shared__ char dyn[];
  const v4* p = src + (size_t)blockIdx.x * n4_per_wg + threadIdx.x;
  v4 acc = (v4)(0.f);
  for (size_t i = 0; i < n4_per_wg; i += 256 * SV) {
    v4 v[SV];
#pragma unroll
    for (int u = 0; u < SV; ++u) v[u] = __builtin_nontemporal_load(p + i + 256 * u);
#pragma unroll
    for (int u = 0; u < SV; ++u) acc += v[u];
  }
  if (acc.x == 123.456f) sink[0] = acc.y + dyn[threadIdx.x];
}

template <int LDSKB>
__global__ __launch_bounds__(256, 2) void mfma_only(int iters, float* sink) {
  __shared__ char lds[LDSKB > 0 ? LDSKB * 1024 : 4];
  f16v acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  h8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = (_Float16)(threadIdx.x * 0.001f), b[e] = (_Float16)(e * 0.01f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 123.456f) sink[0] = s + lds[threadIdx.x];
}

template <bool MFMA>
__global__ __launch_bounds__(256, 4) void gemm_like(const _Float16* A2, const _Float16* W2, int M, int N, int K, float* sink) {
  __shared__ __attribute__((aligned(1024))) char lds[32768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = N / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int64_t a_plane = (int64_t)M * K, w_plane = (int64_t)N * K;
  f16v acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int kt = 0; kt < K / 32; ++kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = wave + 4 * i;
      const int op = p >> 4, plane = (p >> 3) & 1, rb = p & 7;
      const int row = 16 * rb + (lane >> 2);
      const _Float16* src = (op == 0 ? A2 + plane * a_plane + ((int64_t)kt * M + m0 + row) * 32 : W2 + plane * w_plane + ((int64_t)kt * N + n0 + row) * 32) + 8 * (lane & 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
    }
    __syncthreads();
    if (MFMA) {
      const h8* l = reinterpret_cast<const h8*>(lds) + lane;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        h8 a0 = l[64 * ks], a1 = l[128 + 64 * ks], b0 = l[512 + 64 * ks], b1 = l[640 + 64 * ks];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 123.456f) sink[blockIdx.x] = s;
}

int main() {
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const size_t GB = 1ull << 30;
  const size_t bytes = 3 * GB;                       // one "read pass": 2048 envs x 7 x ... ~ 3 GB
  v4* big; CK(hipMalloc(&big, bytes)); CK(hipMemset(big, 0, bytes));
  float* sink; CK(hipMalloc(&sink, 1 << 20));
  const int M = 6144, N = 2048, K = 512;
  _Float16 *A2, *W2; CK(hipMalloc(&A2, (size_t)2 * M * K * 2)); CK(hipMalloc(&W2, (size_t)2 * N * K * 2));
  CK(hipMemset(A2, 0, (size_t)2 * M * K * 2)); CK(hipMemset(W2, 0, (size_t)2 * N * K * 2));
  const int swgs = 32768;
  const size_t n4_per_wg = bytes / 16 / swgs;
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  auto launch_s = [&](int sv, int ldskb, hipStream_t st) {
    if (sv == 8) hipLaunchKernelGGL((stream_read<8, 0>), dim3(swgs), dim3(256), ldskb * 1024, st, big, n4_per_wg, sink);
    else hipLaunchKernelGGL((stream_read<4, 0>), dim3(swgs), dim3(256), ldskb * 1024, st, big, n4_per_wg, sink);
  };
  auto launch_o = [&](int kind, hipStream_t st) {
    const int tiles = (M / 128) * (N / 128);
    switch (kind) {
      case 0: hipLaunchKernelGGL((mfma_only<0>), dim3(768), dim3(256), 0, st, 420, sink); break;
      case 1: hipLaunchKernelGGL((mfma_only<32>), dim3(768), dim3(256), 0, st, 420, sink); break;
      case 2: hipLaunchKernelGGL((gemm_like<false>), dim3(tiles), dim3(256), 0, st, A2, W2, M, N, K, sink); break;
      default: hipLaunchKernelGGL((gemm_like<true>), dim3(tiles), dim3(256), 0, st, A2, W2, M, N, K, sink); break;
    }
  };
  const char* names[] = {"M0 mfma only", "M1 mfma + 32 KB LDS", "D  tile delivery", "G  delivery + mfma"};
  for (int ldskb : {41, 8}) {
    for (int sv : {8, 4}) {
      // S alone
      float s_alone = 1e9;
      for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0, s1)); launch_s(sv, ldskb, s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < s_alone) s_alone = ms;
      }
      printf("S (%d float4 in flight per lane, %d KB LDS): alone %.1f us = %.0f GB/s\n", sv, ldskb, s_alone * 1e3, bytes / s_alone / 1e6);
      for (int kind = 0; kind < 4; ++kind) {
        float o_alone = 1e9;
        const int reps = 8;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(f0, s2)); for (int r = 0; r < reps; ++r) launch_o(kind, s2); CK(hipEventRecord(f1, s2)); CK(hipEventSynchronize(f1));
          float ms; CK(hipEventElapsedTime(&ms, f0, f1)); if (it && ms / reps < o_alone) o_alone = ms / reps;
        }
        // together: S once on s1, the other kernel `reps` times back to back on s2
        float s_tog = 1e9, o_tog = 1e9;
        for (int it = 0; it < 4; ++it) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, s1)); CK(hipEventRecord(f0, s2));
          launch_s(sv, ldskb, s1);
          for (int r = 0; r < reps; ++r) launch_o(kind, s2);
          CK(hipEventRecord(e1, s1)); CK(hipEventRecord(f1, s2));
          CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
          float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, f0, f1));
          if (it && a < s_tog) s_tog = a, o_tog = b / reps;
        }
        printf("   + %-22s alone %6.1f us x %d | together: S %.1f us (x %.2f), other %.1f us per launch (x %.2f) | serial %.0f us, overlapped %.0f us\n", names[kind], o_alone * 1e3, reps,
               s_tog * 1e3, s_tog / s_alone, o_tog * 1e3, o_tog / o_alone, (s_alone + reps * o_alone) * 1e3, fmaxf(s_tog, o_tog * reps) * 1e3);
      }
    }
  }
  return 0;
}
