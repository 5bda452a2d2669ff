// Narrow-output projection: C[M, N] = A[M, K] W[N, K]^T (+ bias) for N <= 96 (Mamba's x_proj: N = dt_rank + 2 d_state = 80,
// K = d_inner = 1536), exact fp32 products on v_mfma_f32_16x16x4_f32 (bit for bit a k-ordered fp32 fma chain per partial sum).
//
// Why its own kernel.  As a tile GEMM the projection is 24 (x 80 of 128 columns used) tiles of 128 x 128 per 3072-row slice, so it
// ran as a split-K launch + a reduce launch: 28 + 7 us per layer and slice inside the Mamba-48M pipeline at 0.04-0.08 matrix-pipe
// busy -- 16 % of the C3 step for 1.5 % of its FLOPs (profiles/r05_kernel_stats_mamba48m_b2048.csv).  Round 4's
// few-row form (operands straight into registers in MFMA order: 16-byte pieces of 16 different rows per load instruction) was
// latency-bound on its A operand (43 us for 6144 rows alone).  Here:
//   * workgroup = 16 rows x all N columns (J = ceil(N / 16) accumulator tiles of 16 x 16), K in chunks of 64 channels dealt round
//     robin to the 4 waves -- no barrier in the K loop, every wave runs its own chunks through its own 4 KB of LDS, and requests
//     a chunk's operands a whole chunk ahead (one wave per SIMD: nothing else hides the latency);
//   * A is loaded COALESCED (a chunk = 16 rows x 256 bytes: four dwordx4 per lane, whole 256-byte row pieces), written to LDS
//     with the 16-byte slot XOR-ed by the row (conflict-free for the fragment read) and read back as one ds_read_b128 per 16
//     channels: lane (row r = l & 15, quarter q = l >> 4) takes channels 16 g + 4 q .. + 3 and uses element i in the i-th of four
//     MFMAs of the group -- the MFMA's k index is a label, both operands just have to agree on it;
//   * W is packed once at upload in exactly that order ([K / 16][J][lane] float4: W[16 j + (l & 15)][16 g + 4 (l >> 4) + i], zero
//     rows beyond N), so a B operand is one contiguous 1 KB load per (group, tile) out of L2;
//   * the four waves' partial tiles are summed in a fixed order through LDS (deterministic), bias added, stored.
// 3072 x 80 x 1536: 192 workgroups of 4 waves, 480 MFMAs of 32 cycles per wave.
//
// Replaces on the reference path: mamba_ssm.Mamba.x_proj (nn.Linear(d_inner, dt_rank + 2 d_state, bias=False)), reached from
// src/algos/models/decision_mamba.py:130-147 (`Block` -> `Mamba.step`).
#include <algorithm>

#include "common.h"
#include "device_math.h"

namespace lram {

namespace {
typedef float nf4 __attribute__((ext_vector_type(4)));

// out[(G * J + j) * 64 + lane] = float4 { W[16 j + (lane & 15)][16 G + 4 (lane >> 4) + i] : i = 0..3 } (0 beyond N)
__global__ __launch_bounds__(256) void narrow_pack_kernel(const float* w, int N, int K, int J, float4* out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)(K / 16) * J * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int64_t gj = idx >> 6;
  const int j = (int)(gj % J);
  const int64_t G = gj / J;
  const int n = 16 * j + (lane & 15);
  const int64_t k0 = 16 * G + 4 * (lane >> 4);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < N) v = *reinterpret_cast<const float4*>(w + (int64_t)n * K + k0);
  out[idx] = v;
}

// (4 waves = one per SIMD: the two register sets of a chunk-ahead prefetch need ~270 registers at J = 5, beyond the 256 a wave
// gets at two per SIMD -- 8 waves spilled; the matrix-pipe time per SIMD is the same either way: 192 workgroups x 480 MFMAs)
constexpr int kNarrowWaves = 4;

template <int J, bool HAS_BIAS>
__global__ __launch_bounds__(64 * kNarrowWaves) void gemm_narrow_kernel(GemmArgs g, const float4* __restrict__ wp) {
  constexpr int NW = kNarrowWaves;
  __shared__ __attribute__((aligned(16))) float lds[NW * 1024 + NW * J * 256];   // 4 KB of A per wave, then the partial tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 16;
  float* abuf = lds + wave * 1024;
  const int nchunks = g.k >> 6;
  // A loads: instruction i covers rows 4 i + (lane >> 4), 16-byte slot lane & 15 of the chunk's 256-byte row piece
  const float* arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = g.a + (int64_t)min(m0 + 4 * i + (lane >> 4), g.m - 1) * g.lda + 4 * (lane & 15);
  const int r = lane & 15, q = lane >> 4;
  nf4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = nf4{0.f, 0.f, 0.f, 0.f};

  // One wave per SIMD at most (192 workgroups for 3072 rows): nothing but this wave's own requests hides memory latency, so a
  // chunk's operands -- 4 float4 of A, 4 J float4 of W -- are requested a whole chunk (16 J MFMAs = 2560 cycles at J = 5) ahead.
  // (two named register sets, the loop body written out twice: arrays handed to lambdas by reference ended up in scratch memory)
  nf4 a0[4], a1[4], w0[4 * J], w1[4 * J];   // (native vector values: HIP's float4 class arrays of the A set went to scratch too)
#define LRAM_NARROW_LOAD(AR, WR, CH)                                                             \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) AR[i] = *reinterpret_cast<const nf4*>(arow[i] + 64 * (CH)); \
    const nf4* wg_ = reinterpret_cast<const nf4*>(wp) + (int64_t)(4 * (CH)) * J * 64 + lane;     \
    _Pragma("unroll") for (int x = 0; x < 4 * J; ++x) WR[x] = wg_[x * 64];                       \
  }
  // (LDS operations of one wave execute in order: a chunk's stores follow the previous chunk's fragment reads)
#define LRAM_NARROW_COMPUTE(AR, WR)                                                              \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
      const int row_ = 4 * i + (lane >> 4);                                                      \
      *reinterpret_cast<nf4*>(abuf + row_ * 64 + 4 * ((lane & 15) ^ row_)) = AR[i];              \
    }                                                                                            \
    _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                           \
      const nf4 a4_ = *reinterpret_cast<const nf4*>(abuf + r * 64 + 4 * ((4 * gq + q) ^ r));   \
      /* (consecutive MFMAs on DIFFERENT accumulators: a dependent 16 x 16 x 4 pair issues every 40 cycles, not 32) */ \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
        _Pragma("unroll") for (int j = 0; j < J; ++j)                                            \
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4_[i], WR[gq * J + j][i], acc[j], 0, 0, 0); \
    }                                                                                            \
  }
  int c = wave;
  if (c < nchunks) LRAM_NARROW_LOAD(a0, w0, c);
  while (c < nchunks) {
    if (c + NW < nchunks) LRAM_NARROW_LOAD(a1, w1, c + NW);
    LRAM_NARROW_COMPUTE(a0, w0);
    c += NW;
    if (c >= nchunks) break;
    if (c + NW < nchunks) LRAM_NARROW_LOAD(a0, w0, c + NW);
    LRAM_NARROW_COMPUTE(a1, w1);
    c += NW;
  }
#undef LRAM_NARROW_LOAD
#undef LRAM_NARROW_COMPUTE
  // partial tiles -> LDS [wave][j][lane] float4, summed in wave order
  float* red = lds + NW * 1024;
#pragma unroll
  for (int j = 0; j < J; ++j) *reinterpret_cast<nf4*>(red + ((wave * J + j) * 64 + lane) * 4) = acc[j];
  __syncthreads();
  // C/D layout of the 16 x 16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg; tile j has 256 values: one per thread
  {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int ln = tid >> 2, reg = tid & 3;
      float s = red[((0 * J + j) * 64 + ln) * 4 + reg];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += red[((w * J + j) * 64 + ln) * 4 + reg];
      const int row = m0 + 4 * (ln >> 4) + reg, col = 16 * j + (ln & 15);
      if (row < g.m && col < g.n) {
        if (HAS_BIAS) s += g.bias[col];
        g.c[(int64_t)row * g.ldc + col] = s;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// f16x2 form: the same decomposition with the products as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 (the arithmetic of
// gemm_f16x2.hip: rows scaled by a power of two from their maxima, every scaled element split exactly into two binary16 pieces,
// fp32 accumulation, exact un-scaling) -- 15 MFMAs of 16 cycles per 32 channels and wave instead of 40 of 32: inside the Mamba
// pipeline the exact-fp32 form held the matrix pipes of its CUs for 25 us per launch beside the other slice's projection.
//   * A: each lane converts the float4s it loaded (one rounding per piece, 14 VALU instructions per float4) and writes two 8-byte
//     pieces into the wave's hi / lo planes ([16 rows][64 channels] binary16, 16-byte slot XOR-ed with (row >> 1) & 7: the
//     ds_read_b128 fragment reads of a 16-lane group then cover all 16 slots of the 256-byte bank row);
//   * W: the K-tile-major planes the f16x2 tile GEMMs use ([K / 32][N][32] binary16) ARE the B fragments of the 16 x 16 x 32
//     MFMA: rows 16 j .. 16 j + 15 of a K tile are 1 KB contiguous, lane l takes 16 bytes at row l & 15, chunk l >> 4;
//   * row maxima of A: handed over by A's producer as `amax_parts` partial maxima per row (the conv kernel: one per 64 channels).
typedef _Float16 nh8 __attribute__((ext_vector_type(8)));
typedef _Float16 nh4 __attribute__((ext_vector_type(4)));

template <int J, bool HAS_BIAS>
__global__ __launch_bounds__(64 * kNarrowWaves) void gemm_narrow16_kernel(GemmArgs g) {
  constexpr int NW = kNarrowWaves;
  static_assert(NW == 4, "16 rows x 16 lanes take the row maxima");
  __shared__ __attribute__((aligned(16))) float lds[NW * 1024 + 32 + NW * J * 256];   // per wave 2 planes x 16 x 64 f16; scales; partial tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 16;
  _Float16* abuf = reinterpret_cast<_Float16*>(lds + wave * 1024);
  float* scl = lds + NW * 1024;   // [0..15] scale, [16..31] inverse
  const int nchunks = g.k >> 6;
  {  // row scales: thread (row = tid >> 4, p = tid & 15) walks parts p, p + 16, ...; 16-lane maxima
    const int row = tid >> 4, p = tid & 15;
    const float* am = g.a_amax + (int64_t)min(m0 + row, g.m - 1) * g.amax_parts;
    float mx = 0.f;
    for (int x = p; x < g.amax_parts; x += 16) mx = fmaxf(mx, am[x]);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 16));
    const float sc = pow2_scale(mx);
    if (p == 0) scl[row] = sc, scl[16 + row] = 1.f / sc;
  }
  __syncthreads();
  const float* arow[4];
  float asc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    arow[i] = g.a + (int64_t)min(m0 + 4 * i + (lane >> 4), g.m - 1) * g.lda + 4 * (lane & 15);
    asc[i] = scl[4 * i + (lane >> 4)];
  }
  const int r = lane & 15, q = lane >> 4;
  nf4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = nf4{0.f, 0.f, 0.f, 0.f};
  // W fragment sources: tile j, plane pl of K tile kt at w2 + pl * w2_plane + kt * w2_kt + (16 j + (l & 15)) * 32 + 8 (l >> 4)
  const _Float16* w2 = reinterpret_cast<const _Float16*>(g.w2);
  int64_t woff[J];
#pragma unroll
  for (int j = 0; j < J; ++j) woff[j] = (int64_t)min(16 * j + r, g.n - 1) * 32 + 8 * q;

  nf4 a0[4], a1[4];
  nh8 w0[4 * J], w1[4 * J];   // [step s][tile j][plane]
#define LRAM_NARROW16_LOAD(AR, WR, CH)                                                           \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) AR[i] = *reinterpret_cast<const nf4*>(arow[i] + 64 * (CH)); \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                             \
      _Pragma("unroll") for (int j = 0; j < J; ++j)                                              \
        _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                         \
          WR[(s_ * J + j) * 2 + pl] = *reinterpret_cast<const nh8*>(w2 + pl * g.w2_plane + (int64_t)(2 * (CH) + s_) * g.w2_kt + woff[j]); \
  }
#define LRAM_NARROW16_COMPUTE(AR, WR)                                                            \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
      const int row_ = 4 * i + (lane >> 4), p_ = lane & 15;                                      \
      nh4 hi_, lo_;                                                                              \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                         \
        const float x_ = AR[i][e_] * asc[i];                                                     \
        const _Float16 h_ = (_Float16)x_;                                                        \
        hi_[e_] = h_, lo_[e_] = (_Float16)(x_ - (float)h_);                                      \
      }                                                                                          \
      _Float16* d_ = abuf + row_ * 64 + (((p_ >> 1) ^ ((row_ >> 1) & 7)) << 3) + ((p_ & 1) << 2); \
      *reinterpret_cast<nh4*>(d_) = hi_;                                                         \
      *reinterpret_cast<nh4*>(d_ + 1024) = lo_;                                                  \
    }                                                                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                           \
      const _Float16* f_ = abuf + r * 64 + (((4 * s_ + q) ^ ((r >> 1) & 7)) << 3);               \
      const nh8 ah_ = *reinterpret_cast<const nh8*>(f_), al_ = *reinterpret_cast<const nh8*>(f_ + 1024); \
      /* smallest terms first, consecutive MFMAs on different accumulators */                    \
      _Pragma("unroll") for (int j = 0; j < J; ++j)                                              \
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al_, WR[(s_ * J + j) * 2 + 0], acc[j], 0, 0, 0); \
      _Pragma("unroll") for (int j = 0; j < J; ++j)                                              \
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah_, WR[(s_ * J + j) * 2 + 1], acc[j], 0, 0, 0); \
      _Pragma("unroll") for (int j = 0; j < J; ++j)                                              \
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah_, WR[(s_ * J + j) * 2 + 0], acc[j], 0, 0, 0); \
    }                                                                                            \
  }
  int c = wave;
  if (c < nchunks) LRAM_NARROW16_LOAD(a0, w0, c);
  while (c < nchunks) {
    if (c + NW < nchunks) LRAM_NARROW16_LOAD(a1, w1, c + NW);
    LRAM_NARROW16_COMPUTE(a0, w0);
    c += NW;
    if (c >= nchunks) break;
    if (c + NW < nchunks) LRAM_NARROW16_LOAD(a0, w0, c + NW);
    LRAM_NARROW16_COMPUTE(a1, w1);
    c += NW;
  }
#undef LRAM_NARROW16_LOAD
#undef LRAM_NARROW16_COMPUTE
  float* red = lds + NW * 1024 + 32;
#pragma unroll
  for (int j = 0; j < J; ++j) *reinterpret_cast<nf4*>(red + ((wave * J + j) * 64 + lane) * 4) = acc[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int ln = tid >> 2, reg = tid & 3;
    float s = red[((0 * J + j) * 64 + ln) * 4 + reg];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red[((w * J + j) * 64 + ln) * 4 + reg];
    const int rl = 4 * (ln >> 4) + reg, row = m0 + rl, col = 16 * j + (ln & 15);
    if (row < g.m && col < g.n) {
      s *= scl[16 + rl] * g.w_inv[col];   // exact: powers of two
      if (HAS_BIAS) s += g.bias[col];
      g.c[(int64_t)row * g.ldc + col] = s;
    }
  }
}

template <int J>
void launch16_j(const GemmArgs& g, hipStream_t stream) {
  dim3 grid((unsigned)((g.m + 15) / 16)), block(64 * kNarrowWaves);
  if (g.bias != nullptr)
    hipLaunchKernelGGL((gemm_narrow16_kernel<J, true>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_narrow16_kernel<J, false>), grid, block, 0, stream, g);
}

template <int J>
void launch_j(const GemmArgs& g, const float4* wp, hipStream_t stream) {
  dim3 grid((unsigned)((g.m + 15) / 16)), block(64 * kNarrowWaves);
  if (g.bias != nullptr)
    hipLaunchKernelGGL((gemm_narrow_kernel<J, true>), grid, block, 0, stream, g, wp);
  else
    hipLaunchKernelGGL((gemm_narrow_kernel<J, false>), grid, block, 0, stream, g, wp);
}
}  // namespace

size_t gemm_narrow_pack_elems(int n, int k) { return (size_t)((n + 15) / 16) * 16 * (size_t)k; }

bool gemm_narrow_shape(int n, int k) { return n >= 1 && n <= 96 && k >= 256 && (k & 63) == 0; }

bool gemm_narrow_supported(const GemmArgs& g) {
  return gemm_narrow_shape(g.n, g.k) && g.nb1 * g.nb2 == 1 && g.gate == nullptr && g.act_silu_from < 0 && g.residual == nullptr &&
         g.norm_g == nullptr && (g.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(g.a) & 15) == 0 && g.a != nullptr;
}

void launch_gemm_narrow_pack(const float* w, int n, int k, float* packed, hipStream_t stream) {
  LRAM_REQUIRE(gemm_narrow_shape(n, k) && (reinterpret_cast<uintptr_t>(w) & 15) == 0, "narrow-output pack: N <= 96, K a multiple of 64");
  const int J = (n + 15) / 16;
  const int64_t total = (int64_t)(k / 16) * J * 64;
  hipLaunchKernelGGL(narrow_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, n, k, J,
                     reinterpret_cast<float4*>(packed));
  LRAM_HIP_CHECK(hipGetLastError());
}

bool gemm_narrow16_supported(const GemmArgs& g) {
  return gemm_narrow_supported(g) && g.w2 != nullptr && g.w_inv != nullptr && g.a_amax != nullptr && g.amax_parts >= 1 &&
         g.w2_kt >= 32 * (int64_t)g.n && (g.w2_kt & 7) == 0 && (g.w2_plane & 7) == 0 && (reinterpret_cast<uintptr_t>(g.w2) & 15) == 0;
}

void launch_gemm_narrow16(const GemmArgs& g, hipStream_t stream) {
  LRAM_REQUIRE(g.m > 0 && gemm_narrow16_supported(g), "narrow-output projection (f16x2): unsupported operands");
  switch ((g.n + 15) / 16) {
    case 1: launch16_j<1>(g, stream); break;
    case 2: launch16_j<2>(g, stream); break;
    case 3: launch16_j<3>(g, stream); break;
    case 4: launch16_j<4>(g, stream); break;
    case 5: launch16_j<5>(g, stream); break;
    default: launch16_j<6>(g, stream); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_gemm_narrow(const GemmArgs& g, const float* packed, hipStream_t stream) {
  LRAM_REQUIRE(g.m > 0 && gemm_narrow_supported(g) && packed != nullptr, "narrow-output projection: unsupported operands");
  const float4* wp = reinterpret_cast<const float4*>(packed);
  switch ((g.n + 15) / 16) {
    case 1: launch_j<1>(g, wp, stream); break;
    case 2: launch_j<2>(g, wp, stream); break;
    case 3: launch_j<3>(g, wp, stream); break;
    case 4: launch_j<4>(g, wp, stream); break;
    case 5: launch_j<5>(g, wp, stream); break;
    default: launch_j<6>(g, wp, stream); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
