// Phase profile of the pre-split GEMM's one-stage loop (128 x 128 x 32 tiles, four workgroups per CU, K-tile-major planes, RANDOM
// operands so that the chip holds the clock it holds under the real kernel): per wave and K tile, s_memtime (100 MHz) stamps at
//   t0 -> t1  issuing the tile's 8 LDS-DMA instructions          ("dma issue")
//   t1 -> t2  s_waitcnt vmcnt(0) + barrier: the tile has landed   ("wait data")
//   t2 -> t3  16 fragment reads + 24 MFMAs issued                 ("mfma")
//   t3 -> t4  barrier before the stage is overwritten             ("barrier")
// summed over the K loop, averaged over all waves; plus the kernel's wall time and what 24 MFMAs x 32 cycles per K tile would take
// at 2.4 GHz.  hipcc --offload-arch=gfx950 -O3 -o gemm_phases scripts/gemm_phases.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int WGPC, bool STAMP, int STAGGER = 0>
__global__ __launch_bounds__(256, WGPC) void skel(const _Float16* A2, const _Float16* W2, int M, int N, int K, float* C, unsigned* phases) {
  __shared__ __attribute__((aligned(1024))) _Float16 lds[16384];
  if (STAGGER > 0) {
    // de-phase the workgroups that share a CU: wave 0's slot number on its SIMD (HW_REG_HW_ID bits 3:0) x STAGGER x 64 cycles
    __shared__ unsigned slot;
    if (threadIdx.x == 0) slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 3;   // size 4, offset 0, HW_REG_HW_ID = 4
    __syncthreads();
    const unsigned n = slot;
    for (unsigned i = 0; i < n * STAGGER; ++i) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
  }
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
  unsigned p_dma = 0, p_wait = 0, p_mfma = 0, p_bar = 0;
  const unsigned long long tstart = STAMP ? __builtin_amdgcn_s_memtime() : 0;
  for (int kt = 0; kt < K / 32; ++kt) {
    const unsigned long long t0 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = wave + 4 * i;
      const int op = p >> 4, plane = (p >> 3) & 1, rb = p & 7;
      const int row = 16 * rb + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const _Float16* src = (op == 0 ? A2 + plane * a_plane + ((int64_t)kt * M + m0 + row) * 32 : W2 + plane * w_plane + ((int64_t)kt * N + n0 + row) * 32) + 8 * chunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + p * 512), 16, 0, 0);
    }
    const unsigned long long t1 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    const unsigned long long t2 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ((2 * ks + lh) ^ sw) << 3;
      h8 af[2][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          af[t][p] = *reinterpret_cast<const h8*>(a_base + p * 4096 + 32 * t * 32 + ko);
          bf[t][p] = *reinterpret_cast<const h8*>(b_base + p * 4096 + 32 * t * 32 + ko);
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
    const unsigned long long t3 = STAMP ? __builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    if (STAMP) {
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      p_dma += (unsigned)(t1 - t0), p_wait += (unsigned)(t2 - t1), p_mfma += (unsigned)(t3 - t2), p_bar += (unsigned)(t4 - t3);
    }
  }
  const unsigned long long tloop = STAMP ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        C[(int64_t)row * N + col] = acc[i][j][r];
      }
    }
  if (STAMP && lane == 0) {
    const unsigned long long tend = __builtin_amdgcn_s_memtime();
    unsigned* o = phases + ((int64_t)blockIdx.x * 4 + wave) * 8;
    o[0] = p_dma, o[1] = p_wait, o[2] = p_mfma, o[3] = p_bar, o[4] = (unsigned)(tloop - tstart), o[5] = (unsigned)(tend - tloop);
  }
}


// ---- the same loop for WM x WN waves of 64 x 64 outputs (128 x 128: 2 x 2; 256 x 128: 4 x 2; 256 x 256: 4 x 4), one or two LDS stages ----
template <int WM, int WN, int NSTAGE, int WPE>
__global__ __launch_bounds__(64 * WM * WN, WPE) void skelg(const _Float16* A2, const _Float16* W2, int M, int N, int K, float* C, unsigned* phases) {
  constexpr int NW = WM * WN, BMT = 64 * WM, BNT = 64 * WN, APL = BMT * 32, PLANE = BNT * 32, STG = 2 * APL + 2 * PLANE;
  constexpr int NPA = 2 * (BMT / 16), NPIECE = NPA + 2 * (BNT / 16), PPW = NPIECE / NW;
  __shared__ __attribute__((aligned(1024))) _Float16 lds[(NSTAGE == 5 ? 1 : NSTAGE >= 3 ? 2 : NSTAGE) * STG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = N / BNT;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * BMT, n0 = tn * BNT;
  const int64_t a_plane = (int64_t)M * K, w_plane = (int64_t)N * K;
  f16v acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int li = lane & 31, lh = lane >> 5, sw = (li >> 2) & 3;
  const _Float16* a_base = lds + (64 * wm + li) * 32;
  const _Float16* b_base = lds + 2 * APL + (64 * wn + li) * 32;
  unsigned p_dma = 0, p_wait = 0, p_mfma = 0, p_bar = 0;
  // buffer form of the LDS-DMA: one resource per operand, per-lane byte offsets loop-invariant in 32-bit VGPRs, the K tile's offset
  // in the scalar operand -- no per-instruction 64-bit vector address arithmetic
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A2, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W2, 0, 0x7fffffff, 0x00020000);
#endif
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave + NW * i;
    const bool is_a = p < NPA;
    const int q = is_a ? p : p - NPA, rbs = is_a ? BMT / 16 : BNT / 16;
    const int plane = q / rbs, rb = q % rbs;
    const int row = 16 * rb + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    voff[i] = (unsigned)(((is_a ? plane * a_plane + (int64_t)(m0 + row) * 32 : plane * w_plane + (int64_t)(n0 + row) * 32) + 8 * chunk) * 2);
  }
  auto dmab = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + NW * i;
      const bool is_a = NW * i < NPA;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? ra : rw, (__attribute__((address_space(3))) void*)(lds + stage * STG + p * 512), 16, voff[i],
                                               (unsigned)kt * (unsigned)(is_a ? M : N) * 64u, 0, 0);
#endif
    }
  };
  auto dma = [&](int stage, int kt) {
    if (NSTAGE >= 5) { dmab(stage, kt); return; }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + NW * i;
      const bool is_a = p < NPA;
      const int q = is_a ? p : p - NPA, rbs = is_a ? BMT / 16 : BNT / 16;
      const int plane = q / rbs, rb = q % rbs;
      const int row = 16 * rb + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const _Float16* src = (is_a ? A2 + plane * a_plane + ((int64_t)kt * M + m0 + row) * 32 : W2 + plane * w_plane + ((int64_t)kt * N + n0 + row) * 32) + 8 * chunk;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + stage * STG + p * 512), 16, 0, 0);
    }
  };
  auto mfma = [&](int stage) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ((2 * ks + lh) ^ sw) << 3;
      h8 af[2][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          af[t][p] = *reinterpret_cast<const h8*>(a_base + stage * STG + p * APL + 32 * t * 32 + ko);
          bf[t][p] = *reinterpret_cast<const h8*>(b_base + stage * STG + p * PLANE + 32 * t * 32 + ko);
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
  };
  // two stages, the next tile's DMA pieces issued ONE AT A TIME between the accumulator groups of the current tile's MFMAs
  auto mfma_dma = [&](int stage, int nstage, int kt_next, bool more) {
    int piece = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ((2 * ks + lh) ^ sw) << 3;
      h8 af[2][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          af[t][p] = *reinterpret_cast<const h8*>(a_base + stage * STG + p * APL + 32 * t * 32 + ko);
          bf[t][p] = *reinterpret_cast<const h8*>(b_base + stage * STG + p * PLANE + 32 * t * 32 + ko);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          __builtin_amdgcn_sched_barrier(0);
          if (piece < PPW && more) {
            const int p = wave + NW * piece;
            const bool is_a = p < NPA;
            const int q = is_a ? p : p - NPA, rbs = is_a ? BMT / 16 : BNT / 16;
            const int plane = q / rbs, rb = q % rbs;
            const int row = 16 * rb + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            const _Float16* src = (is_a ? A2 + plane * a_plane + ((int64_t)kt_next * M + m0 + row) * 32 : W2 + plane * w_plane + ((int64_t)kt_next * N + n0 + row) * 32) + 8 * chunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + nstage * STG + p * 512), 16, 0, 0);
          }
          ++piece;
          __builtin_amdgcn_sched_barrier(0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
        }
    }
  };
  // two stages, REGISTER staging: the next tile's 16-byte pieces by global_load_dwordx4 into registers at the top of the iteration
  // (cheap to issue, in flight under the MFMAs), ds_write_b128 into the other stage after the MFMAs -- no LDS-DMA instruction at all
  uint4 rs[PPW];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + NW * i;
      const bool is_a = p < NPA;
      const int q = is_a ? p : p - NPA, rbs = is_a ? BMT / 16 : BNT / 16;
      const int plane = q / rbs, rb = q % rbs;
      const int row = 16 * rb + (lane >> 2);
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      const _Float16* src = (is_a ? A2 + plane * a_plane + ((int64_t)kt * M + m0 + row) * 32 : W2 + plane * w_plane + ((int64_t)kt * N + n0 + row) * 32) + 8 * chunk;
      rs[i] = *reinterpret_cast<const uint4*>(src);
    }
  };
  auto lstore = [&](int stage) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) *reinterpret_cast<uint4*>(lds + stage * STG + (wave + NW * i) * 512 + lane * 8) = rs[i];
  };
  const unsigned long long tstart = __builtin_amdgcn_s_memtime();
  const int nk = K / 32;
  if (NSTAGE == 4) {
    gload(0);
    lstore(0);
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      if (kt + 1 < nk) gload(kt + 1);
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      mfma(kt & 1);
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      if (kt + 1 < nk) lstore((kt + 1) & 1);
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      p_wait += (unsigned)(t1 - t0), p_dma += (unsigned)(t2 - t1), p_mfma += (unsigned)(t3 - t2), p_bar += (unsigned)(t4 - t3);
    }
  } else if (NSTAGE == 3) {
    static_assert(NSTAGE != 3 || PPW <= 8, "one DMA piece per accumulator group");
    dma(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      mfma_dma(kt & 1, (kt + 1) & 1, kt + 1, kt + 1 < nk);
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      p_wait += (unsigned)(t1 - t0), p_mfma += (unsigned)(t3 - t1);
    }
  } else if (NSTAGE == 2 || NSTAGE == 6) {
    dma(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      if (kt + 1 < nk) dma((kt + 1) & 1, kt + 1);
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      mfma(kt & 1);
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      p_wait += (unsigned)(t1 - t0), p_dma += (unsigned)(t2 - t1), p_mfma += (unsigned)(t3 - t2);
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      dma(0, kt);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      mfma(0);
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      p_dma += (unsigned)(t1 - t0), p_wait += (unsigned)(t2 - t1), p_mfma += (unsigned)(t3 - t2), p_bar += (unsigned)(t4 - t3);
    }
  }
  const unsigned long long tloop = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        C[(int64_t)row * N + col] = acc[i][j][r];
      }
    }
  if (lane == 0) {
    const unsigned long long tend = __builtin_amdgcn_s_memtime();
    unsigned* o = phases + ((int64_t)blockIdx.x * NW + wave) * 8;
    o[0] = p_dma, o[1] = p_wait, o[2] = p_mfma, o[3] = p_bar, o[4] = (unsigned)(tloop - tstart), o[5] = (unsigned)(tend - tloop);
  }
}

int main() {
  struct Shape { const char* tag; int M, N, K; } shapes[] = {{"16m_up", 6144, 2048, 512}, {"prefill_up", 24576, 2048, 512}, {"mamba_in", 6144, 3072, 768}, {"c5_up", 16128, 5120, 1280}};
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (auto& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    _Float16 *A2, *W2; float* C; unsigned* ph;
    const size_t na = (size_t)2 * M * K, nw = (size_t)2 * N * K;
    CK(hipMalloc(&A2, na * 2)); CK(hipMalloc(&W2, nw * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
    const int wgs = (M / 128) * (N / 128);
    CK(hipMalloc(&ph, (size_t)wgs * 4 * 8 * 4));
    std::vector<_Float16> ha(na), hw(nw);
    unsigned x = 12345;
    for (size_t i = 0; i < na; ++i) { x = x * 1664525u + 1013904223u; ha[i] = (_Float16)(((int)(x >> 9) % 2048 - 1024) / 64.0f); }
    for (size_t i = 0; i < nw; ++i) { x = x * 1664525u + 1013904223u; hw[i] = (_Float16)(((int)(x >> 9) % 2048 - 1024) / 1024.0f); }
    CK(hipMemcpy(A2, ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W2, hw.data(), nw * 2, hipMemcpyHostToDevice));
    auto run = [&](auto kern) {
      float best = 1e9;
      for (int it = 0; it < 8; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, A2, W2, M, N, K, C, ph);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms;
      }
      return best * 1e3;
    };
    const float plain = run(skel<4, false>), stamped = run(skel<4, true>);
    printf("   staggered start (slot x n x 64 cycles): n = 4: %6.1f us, 8: %6.1f, 16: %6.1f, 24: %6.1f, 32: %6.1f\n", run(skel<4, false, 4>), run(skel<4, false, 8>),
           run(skel<4, false, 16>), run(skel<4, false, 24>), run(skel<4, false, 32>));
    std::vector<unsigned> hp((size_t)wgs * 4 * 8);
    CK(hipMemcpy(hp.data(), ph, hp.size() * 4, hipMemcpyDeviceToHost));
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (size_t w = 0; w < (size_t)wgs * 4; ++w) for (int k = 0; k < 6; ++k) sum[k] += hp[w * 8 + k];
    const double nwv = (double)wgs * 4, tick = 0.01;   // us per s_memtime tick (100 MHz)
    const double loop = sum[4] / nwv * tick, epi = sum[5] / nwv * tick;
    const double ideal = (K / 32) * 24 * 32 / 2400.0;   // us of back-to-back MFMAs per wave at 2.4 GHz
    printf("%-11s %5d x %4d x %4d  kernel %6.1f us (with stamps %6.1f), %d workgroups = %.2f rounds of 1024 slots\n", s.tag, M, N, K, plain, stamped, wgs, wgs / 1024.0);
    printf("   per wave: K loop %6.2f us = dma issue %5.2f (%2.0f %%) + wait data %5.2f (%2.0f %%) + frag reads & mfma %5.2f (%2.0f %%) + barrier %5.2f (%2.0f %%); "
           "epilogue %5.2f us; 24 MFMAs x %d K tiles back to back at 2.4 GHz: %5.2f us\n",
           loop, sum[0] / nwv * tick, 100 * sum[0] / sum[4], sum[1] / nwv * tick, 100 * sum[1] / sum[4], sum[2] / nwv * tick, 100 * sum[2] / sum[4],
           sum[3] / nwv * tick, 100 * sum[3] / sum[4], epi, K / 32, ideal);
    auto rung = [&](const char* name, auto kern, int bm, int bn, int nw) {
      const int wg = (M / bm) * (N / bn);
      if (M % bm || N % bn) { printf("   %-34s (shape not divisible)\n", name); return; }
      float best = 1e9;
      for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(wg), dim3(64 * nw), 0, 0, A2, W2, M, N, K, C, ph);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms;
      }
      std::vector<unsigned> q((size_t)wg * nw * 8);
      CK(hipMemcpy(q.data(), ph, q.size() * 4, hipMemcpyDeviceToHost));
      double t[6] = {0, 0, 0, 0, 0, 0};
      for (size_t w = 0; w < (size_t)wg * nw; ++w) for (int k = 0; k < 6; ++k) t[k] += q[w * 8 + k];
      printf("   %-34s %6.1f us | per wave and K tile (ticks): dma issue %5.0f, wait / barrier %5.0f, frag + mfma %5.0f, 2nd barrier %4.0f | loop %6.0f, epilogue %5.0f ticks per wave\n",
             name, best * 1e3, t[0] / (wg * nw) / (K / 32), t[1] / (wg * nw) / (K / 32), t[2] / (wg * nw) / (K / 32), t[3] / (wg * nw) / (K / 32), t[4] / (wg * nw), t[5] / (wg * nw));
    };
    CK(hipFree(ph)); CK(hipMalloc(&ph, (size_t)wgs * 16 * 8 * 4));
    rung("128 x 128, one stage (4 WG/CU)", skelg<2, 2, 1, 4>, 128, 128, 4);
    rung("128 x 128, two stages (2 WG/CU)", skelg<2, 2, 2, 2>, 128, 128, 4);
    rung("128 x 128, one stage, BUFFER lds loads", skelg<2, 2, 5, 4>, 128, 128, 4);
    rung("128 x 128, two stages, BUFFER lds loads", skelg<2, 2, 6, 2>, 128, 128, 4);
    rung("256 x 128, one stage, BUFFER lds loads", skelg<4, 2, 5, 4>, 256, 128, 8);
    rung("256 x 256, two stages, BUFFER lds loads", skelg<4, 4, 6, 4>, 256, 256, 16);
    rung("128 x 128, two stages, DMA interleaved", skelg<2, 2, 3, 2>, 128, 128, 4);
    rung("256 x 128, two stages, DMA interleaved", skelg<4, 2, 3, 2>, 256, 128, 8);
    rung("256 x 256, two stages, DMA interleaved", skelg<4, 4, 3, 4>, 256, 256, 16);
    rung("256 x 128, one stage (2 WG/CU)", skelg<4, 2, 1, 4>, 256, 128, 8);
    rung("256 x 128, two stages (1 WG/CU)", skelg<4, 2, 2, 2>, 256, 128, 8);
    rung("256 x 256, one stage (1 WG/CU)", skelg<4, 4, 1, 4>, 256, 256, 16);
    rung("256 x 256, two stages (1 WG/CU)", skelg<4, 4, 2, 4>, 256, 256, 16);
    CK(hipFree(A2)); CK(hipFree(W2)); CK(hipFree(C)); CK(hipFree(ph));
  }
  return 0;
}
