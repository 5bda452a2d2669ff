// fp32-accurate GEMM on the bf16 matrix cores ("bf16x3"):  C[M,N] = A[M,K] * W[N,K]^T (+ bias) (+ residual)
//
// Every fp32 operand is split exactly into three bf16 pieces x = hi + mid + lo (8 + 8 + 8 significand bits;
// the residuals x - hi and x - hi - mid are exact in fp32), and the product is accumulated in fp32 from the
// six largest piece products  hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi ; the dropped terms
// (mid*lo, lo*mid, lo*lo) are <= 2^-24 relative -- the size of one fp32 rounding.  `v_mfma_f32_32x32x16_bf16`
// runs at 16x the rate of the fp32-input MFMA, so six of them cost 6/16 of `v_mfma_f32_32x32x2_f32` work.
// Accuracy is fp32-level (verified against fp64 in tests/test_gpu_parity.py::test_gemm_bf16x3_matches_fp64);
// results are not bit-identical to an fp32 fma chain, which the parity bars (actions exact / 1e-4, states 2e-4
// against a CPU oracle that sums in yet another order) do not require.
//
// W is split once when the weights are finalised (three [N,K] bf16 planes); A is split on the fly while its
// fp32 tile is staged into LDS.  Tiling as gemm_f32.hip: 128 x 128 x 32 block tile, 4 waves, 64 x 64 per wave.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BN = 128, BK = 32;
constexpr int PITCH = 40;                 // bf16 elements per LDS row (80 B: conflict-free ds_read_b128)
constexpr int PLANE = BN * PITCH;         // elements per 128-row plane (W; A when BM = 128)

__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  const float r2 = r1 - (float)mid;
  lo = (__bf16)r2;
}

// (A pre-split A operand -- three bf16 planes written by its producer -- was built in round 2, measured 1-2 % slower end to end
// than the split on the fly and removed in round 5: profiles/EXPERIMENTS.md.)
// BM = 128: 64 x 64 per wave (2 x 2 MFMA tiles).  BM = 64: 32 x 64 per wave -- twice the workgroups for outputs with few
// 128-wide column tiles (proj_down, out_proj, ffn_down: N = 512 .. 1280 gives 192 .. 290 tiles of 128 x 128 for 512
// workgroup slots), 46 KB of LDS instead of 61 KB.
// GATE: the fp32 A operand is multiplied element-wise by g.gate while it is staged (mLSTM output gate).
template <bool HAS_BIAS, bool HAS_RES, int BM, bool GATE = false>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmArgs g) {
  constexpr int WM = BM / 2;          // rows per wave
  constexpr int TI = WM / 32;         // MFMA row tiles per wave
  constexpr int APLANE = BM * PITCH;
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * APLANE + 3 * PLANE];
  __bf16* As = lds;                // [3][BM][PITCH]
  __bf16* Bs = lds + 3 * APLANE;   // [3][128][PITCH]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int z = blockIdx.y;
  const int z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  // (operand tables: the outer batch index picks unrelated base pointers -- the four sLSTM gate projections as ONE launch)
  const bool tab = g.w3_tab[0] != nullptr;
  const float* A = tab ? g.a_tab[z2] + z1 * g.sA1 : g.a + z1 * g.sA1 + z2 * g.sA2;
  const __bf16* W3 = tab ? reinterpret_cast<const __bf16*>(g.w3_tab[z2]) + z1 * g.sW1
                         : reinterpret_cast<const __bf16*>(g.w3) + z1 * g.sW1 + z2 * g.sW2;
  float* C = tab ? g.c_tab[z2] + z1 * g.sC1 : g.c + z1 * g.sC1 + z2 * g.sC2;
  const float* R = HAS_RES ? g.residual + z1 * g.sC1 + z2 * g.sC2 : nullptr;
  const float* bias = HAS_BIAS ? g.bias + z1 * g.sBias1 + z2 * g.sBias2 : nullptr;

  const int tiles_n = (g.n + BN - 1) / BN;
  const int tiles_m = (g.m + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x;
  if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);  // XCD-aware tile order (see gemm_f32.hip)
  const int tm_idx = bid / tiles_n;
  const int tn_idx = bid - tm_idx * tiles_n;
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;

  // staging registers: A 4 x float4 (fp32), W 6 x 16 B (bf16 planes)
  const int lr = tid >> 3;        // A: row within a 32-row slab
  const int lc = (tid & 7) << 2;  // A: k offset 0,4,..,28
  constexpr int NA = BM / 32;          // fp32 A: float4 per thread and K tile
  float4 ra[NA];
  float4 rz[GATE ? NA : 1];
  (void)rz;
  uint4 rw[6];
  auto load_tile = [&](int k0) {
    {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int gm = m0 + lr + 32 * i, kk = k0 + lc;
        ra[i] = (gm < g.m && kk < g.k) ? *reinterpret_cast<const float4*>(A + (int64_t)gm * g.lda + kk)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        if (GATE)
          rz[i] = (gm < g.m && kk < g.k) ? *reinterpret_cast<const float4*>(g.gate + (int64_t)gm * g.ldg + kk)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int q = tid + 256 * j;          // 16-byte chunk id: 3 planes x 128 rows x 4 chunks
      const int plane = q >> 9, rem = q & 511;
      const int r = rem >> 2, c = (rem & 3) << 3;
      const int gn = n0 + r, kk = k0 + c;
      rw[j] = (gn < g.n && kk < g.k)
                  ? *reinterpret_cast<const uint4*>(W3 + (int64_t)plane * g.w3_plane + (int64_t)gn * g.ldw + kk)
                  : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      bf16x4 hi, mid, lo;
      float xs[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
      if (GATE) xs[0] *= rz[i].x, xs[1] *= rz[i].y, xs[2] *= rz[i].z, xs[3] *= rz[i].w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 h, m, l;
        split3(xs[e], h, m, l);
        hi[e] = h;
        mid[e] = m;
        lo[e] = l;
      }
      __bf16* dst = As + (lr + 32 * i) * PITCH + lc;
      *reinterpret_cast<bf16x4*>(dst) = hi;
      *reinterpret_cast<bf16x4*>(dst + APLANE) = mid;
      *reinterpret_cast<bf16x4*>(dst + 2 * APLANE) = lo;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int q = tid + 256 * j;
      const int plane = q >> 9, rem = q & 511;
      const int r = rem >> 2, c = (rem & 3) << 3;
      *reinterpret_cast<uint4*>(Bs + plane * PLANE + r * PITCH + c) = rw[j];
    }
  };

  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31;
  const int lh = lane >> 5;
  // lane (row li, half lh) holds k = 8*lh + j (j = 0..7) of a 16-deep MFMA step for both operands
  const __bf16* a_base = As + (WM * wm + li) * PITCH + 8 * lh;
  const __bf16* b_base = Bs + (64 * wn + li) * PITCH + 8 * lh;

  const int nk_all = (g.k + BK - 1) / BK;
  const int kt0 = g.split_k > 1 ? blockIdx.z * g.k_tiles_per_split : 0;
  const int nk = g.split_k > 1 ? min(nk_all, kt0 + g.k_tiles_per_split) : nk_all;
  load_tile(kt0 * BK);
  for (int kt = kt0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    if (g.mfma_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[TI][3], bf[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (t < TI) af[t][p] = *reinterpret_cast<const bf16x8*>(a_base + p * APLANE + 32 * t * PITCH + 16 * ks);
          bf[t][p] = *reinterpret_cast<const bf16x8*>(b_base + p * PLANE + 32 * t * PITCH + 16 * ks);
        }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);  // lo * hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);  // hi * lo
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);  // mid * mid
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);  // mid * hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);  // hi * mid
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);  // hi * hi
        }
    }
    if (g.mfma_prio) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
  if (g.split_k > 1) {  // raw partial sums into this split's slab [M][N]; bias / residual are applied by the reduce
    float* S = g.splitk_ws + (int64_t)blockIdx.z * g.m * g.n;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + 64 * wn + 32 * j + li;
        if (col >= g.n) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + WM * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < g.m) S[(int64_t)row * g.n + col] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + li;
      if (col >= g.n) continue;
      const float bv = HAS_BIAS ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + WM * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < g.m) {
          float v = acc[i][j][r] + bv;
          if (HAS_RES) v += R[(int64_t)row * g.ldc + col];
          if (g.act_silu_from >= 0 && col >= g.act_silu_from) v = silu_hw(v);
          C[(int64_t)row * g.ldc + col] = v;
        }
      }
    }
}

// out[p][i] = piece p of w[i]  (three consecutive planes of n bf16 each)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* w, __bf16* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    __bf16 hi, mid, lo;
    split3(w[i], hi, mid, lo);
    out[i] = hi;
    out[n + i] = mid;
    out[2 * n + i] = lo;
  }
}
}  // namespace

bool gemm_bf16x3_supported(const GemmArgs& g) {
  return g.w3 != nullptr && (g.k & 7) == 0 && (g.ldw & 7) == 0 && (g.lda & 3) == 0 &&
         ((g.sW1 | g.sW2) & 7) == 0 && ((g.sA1 | g.sA2) & 3) == 0 && (g.w3_plane & 7) == 0;
}

template <int BM>
static void launch_bm(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  const bool hb = g.bias != nullptr, hr = g.residual != nullptr;
  dim3 block(256);
  if (hb && hr)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true, BM>), grid, block, 0, stream, g);
  else if (hb)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true, false, BM>), grid, block, 0, stream, g);
  else if (hr)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, BM>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false, BM>), grid, block, 0, stream, g);
}

void launch_gemm_bf16x3(const GemmArgs& g_in, hipStream_t stream) {
  GemmArgs g = g_in;
  // waves inside their MFMA block issue ahead of the co-resident workgroup's staging code (s_setprio 1): Mamba-48M
  // 319.1k -> 322.5k env-steps/s, 16M 403.8k -> 404.4k
  g.mfma_prio = 1;
  int S = 1;
  if (g.act_silu_from >= 0)
    g.split_k = 1, g.k_tiles_per_split = 0;  // the output activation is applied by this kernel's epilogue: K unsplit
  else
    S = gemm_choose_split_k(g);
  LRAM_REQUIRE(g.m > 0 && g.n > 0 && g.k > 0, "gemm: empty problem");
  LRAM_REQUIRE(gemm_bf16x3_supported(g), "gemm bf16x3: K, ldw and W strides must be multiples of 8");
  const int tiles_n = (g.n + BN - 1) / BN;
  const int tiles128 = ((g.m + 127) / 128) * tiles_n;
  // Outputs with at most four 128-wide column tiles and fewer 128-row tiles than two per CU (16M proj_down / ffn_down:
  // 192 tiles): 64-row tiles give twice the workgroups.  Measured standalone 67 -> 62 us (proj_down), 141 -> 101 us
  // (206M proj_down), 131 -> 117 us (Mamba out_proj); inside the two-slice pipelines, where the other slice fills the
  // chip anyway, only the narrow 16M shapes gain (+1 % end to end) while Mamba-48M and 206M lose 2.5 % to the smaller
  // tile's lower reuse -- hence the tiles_n limit.
  // ... and every GEMM of at most 1024 operand rows (slices of up to 341 envs, where a step is a
  // chain of short launches and twice the workgroups per launch shorten each: 16M at 192 / 256 / 341 envs +1.6 / +2.1 / +6 %,
  // Mamba-48M at 64 / 256 envs +4.7 / +7.3 %; weights above 2.5M elements -- the 206M stack -- keep the 128-row tile's reuse:
  // 128 envs -3.8 % otherwise)
  constexpr int bm64_rows = 1024;
  const bool small = g.m > 64 && ((S == 1 && g.nb1 * g.nb2 == 1 && tiles128 < 512 && tiles_n <= 4) ||
                                  (g.m <= bm64_rows && (int64_t)g.n * g.k <= 2500000));
  const int tiles = small ? ((g.m + 63) / 64) * tiles_n : tiles128;
  dim3 grid(tiles, g.nb1 * g.nb2, S);
  if (g.gate != nullptr) {
    LRAM_REQUIRE(g.nb1 * g.nb2 == 1 && (g.ldg & 3) == 0 && g.residual != nullptr && g.bias == nullptr,
                 "gemm bf16x3: the gated operand form is the un-batched fp32-A residual GEMM (proj_down)");
    if (small)
      hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, 64, true>), grid, dim3(256), 0, stream, g);
    else
      hipLaunchKernelGGL((gemm_bf16x3_kernel<false, true, 128, true>), grid, dim3(256), 0, stream, g);
  } else if (small) {
    launch_bm<64>(g, grid, stream);
  } else {
    launch_bm<128>(g, grid, stream);
  }
  LRAM_HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce(g, stream);
}

void launch_split_bf16x3(const float* w, uint16_t* out, size_t n, hipStream_t stream) {
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, stream, w, reinterpret_cast<__bf16*>(out), n);
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
