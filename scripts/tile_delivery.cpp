// L2 / HBM -> LDS delivery rate of the pre-split projection kernel's tile loads, no MFMA: 4 waves x 8 global_load_lds_dwordx4 per
// K tile (A hi, A lo, W hi, W lo planes of a 128 x 128 x 32 f16 tile = 32 KB), 16 K tiles per workgroup, M x N = 6144 x 2048, K = 512.
//   strided : row-major planes [rows][K]   -- a K tile is 128 rows x 64 B at a 1 KB row stride (what gemm_f16x2p reads today)
//   blocked : K-tile-major planes [K/32][rows][32] -- a K tile's 128 rows are one contiguous 8 KB run
//   strided128 : BK = 64 on row-major planes -- 128 B per row (16 K-tile pairs -> 8 iterations of 64 KB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h;

template <int MODE, int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void deliver(const h* A2, const h* W2, int M, int N, int K, float* sink) {
  constexpr int BYTES = MODE == 2 ? 65536 : 32768;
  __shared__ __attribute__((aligned(1024))) char lds[BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = N / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int nk = K / 32;
  const int64_t a_plane = (int64_t)M * K, w_plane = (int64_t)N * K;
  if (MODE == 2) {
    for (int kt = 0; kt < nk / 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int p = wave + 4 * i;                     // 64 pieces of 1 KB: 8 rows x 128 B
        const int op = p >> 5, plane = (p >> 4) & 1, rb = p & 15;
        const int row = 8 * rb + (lane >> 3);
        const h* src = (op == 0 ? A2 + plane * a_plane + (int64_t)(m0 + row) * K : W2 + plane * w_plane + (int64_t)(n0 + row) * K) + kt * 64 + 8 * (lane & 7);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
      }
      __syncthreads();
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = wave + 4 * i;                     // 32 pieces of 1 KB: 16 rows x 64 B
        const int op = p >> 4, plane = (p >> 3) & 1, rb = p & 7;
        const int row = 16 * rb + (lane >> 2);
        const h* src;
        if (MODE == 0)
          src = (op == 0 ? A2 + plane * a_plane + (int64_t)(m0 + row) * K : W2 + plane * w_plane + (int64_t)(n0 + row) * K) + kt * 32 + 8 * (lane & 3);
        else
          src = (op == 0 ? A2 + plane * a_plane + ((int64_t)kt * M + m0 + row) * 32 : W2 + plane * w_plane + ((int64_t)kt * N + n0 + row) * 32) + 8 * (lane & 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
      }
      __syncthreads();  // (drains the DMAs of the issuing waves)
      __syncthreads();
    }
  }
  if (sink != nullptr && threadIdx.x == 0) sink[blockIdx.x] = ((float*)lds)[blockIdx.x & 63];
}

int main() {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct Shape { const char* tag; int M, N, K; } shapes[] = {{"16m_up", 6144, 2048, 512}, {"mamba_in_s", 3072, 3072, 768}, {"up_b4096", 12288, 2048, 512}};
  for (auto& s : shapes) {
    h *A2, *W2; float* sink;
    CK(hipMalloc(&A2, (size_t)2 * s.M * s.K * 2)); CK(hipMalloc(&W2, (size_t)2 * s.N * s.K * 2)); CK(hipMalloc(&sink, 1 << 20));
    CK(hipMemset(A2, 0, (size_t)2 * s.M * s.K * 2)); CK(hipMemset(W2, 0, (size_t)2 * s.N * s.K * 2));
    const int tiles = (s.M / 128) * (s.N / 128);
    auto run = [&](const char* name, auto kern) {
      float best = 1e9;
      for (int it = 0; it < 8; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, A2, W2, s.M, s.N, s.K, sink);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms;
      }
      const double bytes = (double)tiles * (s.K / 32) * 32768.0;
      printf("%-12s %-22s %7.1f us  %6.2f TB/s delivered  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", s.tag, name, best * 1e3, bytes / best / 1e9,
             bytes / (best * 1e-3) / 2.4e9 / 256);
    };
    run("strided, 4 WG/CU", deliver<0, 4>);
    run("blocked, 4 WG/CU", deliver<1, 4>);
    run("strided, 2 WG/CU", deliver<0, 2>);
    run("blocked, 2 WG/CU", deliver<1, 2>);
    run("strided128 BK=64, 2/CU", deliver<2, 2>);
    CK(hipFree(A2)); CK(hipFree(W2)); CK(hipFree(sink));
  }
  return 0;
}
