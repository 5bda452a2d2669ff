// Skeleton of the pre-split GEMM (6144 x 2048 x 512, 128 x 128 x 32 tiles, K-tile-major planes): where do 51 us go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int FRAGS, int EPI, int WGPC>
__global__ __launch_bounds__(256, WGPC) void skel(const _Float16* A2, const _Float16* W2, int M, int N, int K, float* C, const float* inv) {
  __shared__ __attribute__((aligned(1024))) _Float16 lds[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int64_t a_plane = (int64_t)M * K, w_plane = (int64_t)N * K;
  f16v acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int li = lane & 31, lh = lane >> 5, sw = (li >> 2) & 3;
  const _Float16* a_base = lds + (64 * wm + li) * 32;
  const _Float16* b_base = lds + 2 * 4096 + (64 * wn + li) * 32;
  for (int kt = 0; kt < K / 32; ++kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = wave + 4 * i;
      const int op = p >> 4, plane = (p >> 3) & 1, rb = p & 7;
      const int row = 16 * rb + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const _Float16* src = (op == 0 ? A2 + plane * a_plane + ((int64_t)kt * M + m0 + row) * 32 : W2 + plane * w_plane + ((int64_t)kt * N + n0 + row) * 32) + 8 * chunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + p * 512), 16, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ((2 * ks + lh) ^ sw) << 3;
      h8 af[2][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (FRAGS == 8 || t == 0) {
            af[t][p] = *reinterpret_cast<const h8*>(a_base + p * 4096 + 32 * t * 32 + ko);
            bf[t][p] = *reinterpret_cast<const h8*>(b_base + p * 4096 + 32 * t * 32 + ko);
          } else {
            af[t][p] = af[0][p], bf[t][p] = bf[0][p];
          }
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  if (EPI == 0) {
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0];
    if (s == 123.456f) C[blockIdx.x] = s;
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + 64 * wn + 32 * j + li;
        const float wi = EPI == 2 ? inv[col] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float ai = EPI == 2 ? inv[N + row] : 1.f;
          C[(int64_t)row * N + col] = acc[i][j][r] * (wi * ai);
        }
      }
  }
}

int main() {
  const int M = 6144, N = 2048, K = 512;
  _Float16 *A2, *W2; float *C, *inv;
  CK(hipMalloc(&A2, (size_t)2 * M * K * 2)); CK(hipMalloc(&W2, (size_t)2 * N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&inv, (size_t)(M + N) * 4));
  CK(hipMemset(A2, 0, (size_t)2 * M * K * 2)); CK(hipMemset(W2, 0, (size_t)2 * N * K * 2)); CK(hipMemset(inv, 0, (size_t)(M + N) * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, auto kern) {
    float best = 1e9;
    for (int it = 0; it < 8; ++it) {
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(kern, dim3((M / 128) * (N / 128)), dim3(256), 0, 0, A2, W2, M, N, K, C, inv);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms;
    }
    printf("%-60s %6.1f us\n", name, best * 1e3);
  };
  run("4 fragment reads per k16, no epilogue, 4 WG/CU", skel<4, 0, 4>);
  run("8 fragment reads per k16, no epilogue, 4 WG/CU", skel<8, 0, 4>);
  run("8 fragment reads, C stored (plain), 4 WG/CU", skel<8, 1, 4>);
  run("8 fragment reads, C stored with row / column scales, 4 WG/CU", skel<8, 2, 4>);
  run("8 fragment reads, C stored with scales, 3 WG/CU", skel<8, 2, 3>);
  run("8 fragment reads, C stored with scales, 2 WG/CU", skel<8, 2, 2>);
  return 0;
}
