// Does a projection kernel beside a read pass want MORE tiles in flight?  (round 5; follow-up of scripts/corun.cpp)
//
// Inside the two-slice pipelines the small-grid projections take 2-4 x as long as alone (206M proj_up, 768 x 5120 x 1280: 38 us alone,
// 167 us beside the other slice's read pass).  Two streams: S = a streaming read with the read pass's resource shape (256 threads,
// SV float4 per lane in flight, SLDS KB of LDS per workgroup), G = the pre-split GEMM's K loop (LDS-DMA of K-tile-major f16 planes +
// 24 MFMAs per wave and K tile, no epilogue) as a ring of NS LDS stages (NS - 1 K tiles of DMA in flight per workgroup; NS = 1: the
// one-stage form, two barriers per K tile) with 128- or 64-row tiles.  Reported: G alone, G beside S, S beside G.
//   hipcc --offload-arch=gfx950 -O3 -o corun_depth scripts/corun_depth.cpp && ./corun_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SV>
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

// TI = 32-row blocks per wave in M (2: 128-row tile, 1: 64-row tile); tile = (64 TI) x 128, four waves (2 x 2)
template <int NS, int TI>
__global__ __launch_bounds__(256) void gemm_ring(const _Float16* A2, const _Float16* W2, int M, int N, int K, float* sink) {
  constexpr int BMT = 64 * TI;
  constexpr int NPA = 2 * (BMT / 16), NPIECE = NPA + 16, PPW = NPIECE / 4;   // 1 KB pieces per stage: A planes, W planes
  constexpr int STG = NPIECE * 1024;
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BMT, n0 = tn * 128;
  const int64_t a_plane = (int64_t)M * K, w_plane = (int64_t)N * K;
  f16v acc[TI][2];
  for (int i = 0; i < TI; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / 32;
  auto dma = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 4 * i;
      const bool is_a = p < NPA;
      const int q = is_a ? p : p - NPA, rbs = is_a ? BMT / 16 : 8;
      const int plane = q / rbs, rb = q % rbs;
      const int row = 16 * rb + (lane >> 2);
      const _Float16* src = (is_a ? A2 + plane * a_plane + ((int64_t)kt * M + m0 + row) * 32
                                  : W2 + plane * w_plane + ((int64_t)kt * N + n0 + row) * 32) + 8 * (lane & 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + stage * STG + p * 1024), 16, 0, 0);
    }
  };
  auto mfma = [&](int stage) {
    const h8* la = reinterpret_cast<const h8*>(lds + stage * STG) + lane + 64 * (2 * TI * wm);
    const h8* lw = reinterpret_cast<const h8*>(lds + stage * STG + NPA * 1024) + lane + 64 * (4 * wn);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 a[TI][2], b[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (t < TI) a[t][p] = la[64 * (t + TI * 2 * p) + 32 * ks];   // (addresses only have to be distinct and in range: a timing skeleton)
          b[t][p] = lw[64 * (t + 8 * p) + 32 * ks];
        }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
        }
    }
  };
  if (NS == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      dma(0, kt);
      __syncthreads();
      mfma(0);
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int st = 0; st < NS - 1; ++st)
      if (st < nk) dma(st, st);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = min(NS - 2, nk - 1 - kt);
      if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
      else if (NS >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int nxt = cur == 0 ? NS - 1 : cur - 1;
      if (kt + NS - 1 < nk) dma(nxt, kt + NS - 1);
      mfma(cur);
      cur = cur + 1 == NS ? 0 : cur + 1;
    }
  }
  float s = 0.f;
  for (int i = 0; i < TI; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0];
  if (s == 123.456f) sink[blockIdx.x] = s;
}

template <int NS, int TI>
static void launch_g(const _Float16* A2, const _Float16* W2, int M, int N, int K, float* sink, hipStream_t st) {
  constexpr int BMT = 64 * TI;
  const size_t shmem = (size_t)NS * (2 * (BMT / 16) + 16) * 1024;
  static bool raised = false;
  if (!raised && shmem > 48 * 1024) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring<NS, TI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  hipLaunchKernelGGL((gemm_ring<NS, TI>), dim3((M / BMT) * (N / 128)), dim3(256), shmem, st, A2, W2, M, N, K, sink);
}

int main(int argc, char** argv) {
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const size_t GB = 1ull << 30;
  const size_t bytes = 4 * GB;   // (8192 float4 per workgroup: whole trips for 8 and for 16 rows in flight)
  v4* big; CK(hipMalloc(&big, bytes)); CK(hipMemset(big, 0, bytes));
  float* sink; CK(hipMalloc(&sink, 1 << 20));
  struct Shape { const char* name; int M, N, K; };
  const Shape shapes[] = {{"206M proj_up, 768 rows", 768, 5120, 1280}, {"206M proj_down, 768 rows", 768, 1280, 2560},
                          {"16M proj_down, 1536 rows (C2)", 1536, 512, 1024}, {"Mamba in_proj, 3072 rows", 3072, 3072, 768}};
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  const int swgs = 32768;
  const size_t n4_per_wg = bytes / 16 / swgs;
  auto launch_s = [&](int sv, int ldskb, hipStream_t st) {
    if (sv == 16) hipLaunchKernelGGL((stream_read<16>), dim3(swgs), dim3(256), ldskb * 1024, st, big, n4_per_wg, sink);
    else hipLaunchKernelGGL((stream_read<8>), dim3(swgs), dim3(256), ldskb * 1024, st, big, n4_per_wg, sink);
  };
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    _Float16 *A2, *W2; CK(hipMalloc(&A2, (size_t)2 * M * K * 2)); CK(hipMalloc(&W2, (size_t)2 * N * K * 2));
    CK(hipMemset(A2, 0, (size_t)2 * M * K * 2)); CK(hipMemset(W2, 0, (size_t)2 * N * K * 2));
    auto launch_o = [&](int kind, hipStream_t st) {
      switch (kind) {
        case 0: launch_g<1, 2>(A2, W2, M, N, K, sink, st); break;
        case 1: launch_g<2, 2>(A2, W2, M, N, K, sink, st); break;
        case 2: launch_g<3, 2>(A2, W2, M, N, K, sink, st); break;
        case 3: launch_g<4, 2>(A2, W2, M, N, K, sink, st); break;
        case 4: launch_g<1, 1>(A2, W2, M, N, K, sink, st); break;
        case 5: launch_g<2, 1>(A2, W2, M, N, K, sink, st); break;
        case 6: launch_g<3, 1>(A2, W2, M, N, K, sink, st); break;
        default: launch_g<4, 1>(A2, W2, M, N, K, sink, st); break;
      }
    };
    const char* names[] = {"128-row tile, 1 stage ", "128-row tile, 2 stages", "128-row tile, 3 stages", "128-row tile, 4 stages",
                           " 64-row tile, 1 stage ", " 64-row tile, 2 stages", " 64-row tile, 3 stages", " 64-row tile, 4 stages"};
    printf("== %s: %d x %d x %d (%d / %d workgroups)\n", sh.name, M, N, K, (M / 128) * (N / 128), (M / 64) * (N / 128));
    for (int cfg = 0; cfg < 2; ++cfg) {
      const int sv = cfg == 0 ? 16 : 8, ldskb = cfg == 0 ? 29 : 41;
      float s_alone = 1e9;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, s1)); launch_s(sv, ldskb, s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < s_alone) s_alone = ms;
      }
      printf(" S (%d float4 in flight per lane, %d KB LDS): alone %.1f us = %.0f GB/s\n", sv, ldskb, s_alone * 1e3, bytes / s_alone / 1e6);
      for (int kind = 0; kind < 8; ++kind) {
        float o_alone = 1e9;
        const int reps = 6;
        for (int it = 0; it < 4; ++it) {
          CK(hipEventRecord(f0, s2)); for (int r = 0; r < reps; ++r) launch_o(kind, s2); CK(hipEventRecord(f1, s2)); CK(hipEventSynchronize(f1));
          float ms; CK(hipEventElapsedTime(&ms, f0, f1)); if (it && ms / reps < o_alone) o_alone = ms / reps;
        }
        float s_tog = 1e9, o_tog = 1e9;
        for (int it = 0; it < 4; ++it) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, s1)); CK(hipEventRecord(f0, s2));
          for (int r = 0; r < 3; ++r) launch_s(sv, ldskb, s1);   // (long enough to cover the G launches)
          for (int r = 0; r < reps; ++r) launch_o(kind, s2);
          CK(hipEventRecord(e1, s1)); CK(hipEventRecord(f1, s2));
          CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
          float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, f0, f1));
          if (it && a < s_tog) s_tog = a, o_tog = b / reps;
        }
        printf("   G %s alone %6.1f us | beside S: %6.1f us (x %.2f), S x %.2f\n", names[kind], o_alone * 1e3, o_tog * 1e3, o_tog / o_alone,
               s_tog / (3 * s_alone));
      }
    }
    CK(hipFree(A2)); CK(hipFree(W2));
  }
  return 0;
}
