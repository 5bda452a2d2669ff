// fp32-accurate GEMM on the f16 matrix cores ("f16x2"):  C[M,N] = A[M,K] * W[N,K]^T (+ bias) (+ residual)
//
// Each fp32 operand row is scaled by a power of two so that its largest element lands in [2^14, 2^15) and every
// scaled element is split exactly into two binary16 pieces  s x = hi + lo  (hi = rn16(s x), lo = rn16(s x - hi); the
// residual s x - hi is exact in fp32).  hi carries 11 significand bits, lo the next 11: 22 of fp32's 24, with an
// absolute error <= 2^-25 for elements so small that lo leaves the normal range -- 2^-39 of the row's largest element.
// The product is accumulated in fp32 from the three largest piece products  hi*hi + hi*lo + lo*hi ; the dropped
// lo*lo is <= 2^-22 relative, and random in sign.  Measured against fp64 the result is as close as the bf16x3 kernel's
// and the exact fp32 fma chain's (tests/test_gpu_parity.py::test_gemm_f16x2_matches_fp64; the CPU emulation in
// tests/test_oracle_selfchecks.py states the arithmetic), at HALF the matrix-core work of bf16x3 (3 products instead
// of 6 at the same `v_mfma_f32_32x32x16_*` rate) and 2/3 of its LDS bytes (4 B per element instead of 6).
//
// Why that matters here (profiles/r03_gemm_*): the 128 x 128 x 32 bf16x3 kernel is not matrix-core bound.  Its loop
// is [wait for the next tile's global loads] -> split + LDS write -> barrier -> issue loads -> 48 MFMAs -> barrier, the
// loads have only the MFMA block (1.5-3 k cycles) to come back, and 61 KB of LDS + 240 VGPRs allow two workgroups per
// CU: two waves per SIMD cannot cover an L2 / HBM round trip of that length, so the matrix pipe idles ~55 % of the
// time (exact-fp32 MFMA kernel, 2.7 x the MFMA time per tile: 75 % busy).  This kernel's tile is 40 KB of LDS and
// <= 168 VGPRs: THREE workgroups per CU, three waves per SIMD, while one waits for its tile two others compute.
//
// W is split once when the weights are finalised (two [N,K] f16 planes + the per-row inverse scale); A is split on
// the fly while its fp32 tile is staged into LDS; its per-row scale is derived from the row's largest magnitude, which
// the kernel that produced A hands over (norm kernels: one wave per row; the Mamba conv / state-update kernels: one
// partial maximum per wave, plain stores, reduced in this kernel's prologue via `amax_parts`) or `launch_row_amax` computes.  Scales are powers of two: un-scaling the accumulator is exact.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BN = 128, BK = 32;
constexpr int kPitchPadded = 40;   // f16 elements per padded LDS row (80 B: conflict-free ds_read_b128)

// BM = 128: 64 x 64 per wave (2 x 2 MFMA tiles).  BM = 64: 32 x 64 per wave.  GATE: the fp32 A operand is multiplied
// element-wise by g.gate while it is staged (mLSTM output gate); the row scales then are those of the gated rows.
// (The ablation variants this kernel carried while it was tuned -- no conversion / no loads / no MFMA / no LDS writes / no
// barriers, profiles/EXPERIMENTS.md -- are in the git history: commit "gemm_f16x2: double-buffered, swizzled-LDS variant".)
// DB: two LDS stages and two register sets, ONE barrier per K tile: while the matrix cores work on stage s, the same
// wave converts the next tile (already in registers) into stage s ^ 1 and the tile after that is in flight from
// memory.  Without it the fp32 -> f16 conversion, the LDS writes and the MFMA block of a workgroup are separated by
// barriers and overlap only with OTHER workgroups' phases: measured, the MFMA time was simply added on top of the rest
// (16M proj_up: 66 us with, 52 us without the MFMAs, 15.5 us of pure MFMA time).
template <bool HAS_BIAS, bool HAS_RES, int BM, bool GATE, int PF, bool DB = false>
__global__ __launch_bounds__(256, (PF == 1 && !DB) ? 3 : 2) void gemm_f16x2_kernel(GemmArgs g) {
  constexpr int WM = BM / 2;   // rows per wave
  constexpr int TI = WM / 32;  // MFMA row tiles per wave
  // DB: un-padded 64-byte rows with the 16-byte chunk index XOR-ed by (row >> 2) & 3 (the 16 lanes of a ds_read_b128
  // pass then cover all 64 banks: rows r, r + 4, r + 8, r + 12 of a pass would otherwise share their banks) -- 32 KB per
  // stage, two stages of two workgroups fit a CU with room to spare; else rows padded to 80 bytes.
  constexpr int PITCH = DB ? 32 : kPitchPadded;
  constexpr int PLANE = BN * PITCH;
  constexpr int APLANE = BM * PITCH;
  constexpr int STAGE = 2 * APLANE + 2 * PLANE;  // f16 elements of one LDS stage
  __shared__ __attribute__((aligned(16))) _Float16 lds[(DB ? 2 : 1) * STAGE];
  // power-of-two scale of each of the tile's A rows (0 beyond M): lives in the first stage's memory before the first
  // tile is written and again after the last one is read (two stages of the 128-row tile are exactly half the CU's LDS)
  float* srow = reinterpret_cast<float*>(lds);
  // single-stage instances keep the tile's inverse row scales and inverse column scales in their own 1 KB from the start, so
  // the epilogue neither asks the memory for them after the last MFMA nor divides (the two-stage instances fill the CU's LDS
  // with two workgroups exactly and take them into registers instead)
  __shared__ float sinv[DB ? 1 : BM + BN];
  _Float16* As = lds;               // [2][BM][PITCH]   hi, lo
  _Float16* Bs = lds + 2 * APLANE;  // [2][128][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const float* A = g.a;
  const _Float16* W2 = reinterpret_cast<const _Float16*>(g.w2);
  float* C = g.c;

  const int tiles_n = (g.n + BN - 1) / BN;
  const int tiles_m = (g.m + BM - 1) / BM;
  int tm_idx, tn_idx;
  gemm_tile_of(g, blockIdx.x, tiles_m, tiles_n, tm_idx, tn_idx);  // XCD-aware tile order (common.h)
  const int m0 = tm_idx * BM, n0 = tn_idx * BN;

  // row scales from the producers' row maxima: a row's maximum may arrive in `amax_parts` partial maxima (one per wave
  // of a producer whose workgroups each cover a slice of the row: plain stores there, no atomics, no zeroing)
  if (tid < BM) {
    const int gm = m0 + tid;
    float mx = 0.f;
    if (gm < g.m) {
      const float* ap = g.a_amax + (int64_t)gm * g.amax_parts;
      for (int q = 0; q < g.amax_parts; ++q) mx = fmaxf(mx, ap[q]);
    }
    srow[tid] = gm < g.m ? pow2_scale(mx) : 0.f;
    if (!DB) sinv[tid] = gm < g.m ? 1.f / pow2_scale(mx) : 0.f;  // (exact: a power of two)
  } else if (!DB && tid < BM + BN) {
    sinv[tid] = g.w_inv[min(n0 + tid - BM, g.n - 1)];
  }
  __syncthreads();
  const int lr = tid >> 3;        // A: row within a 32-row slab
  const int lc = (tid & 7) << 2;  // A: k offset 0,4,..,28
  constexpr int NA = BM / 32;     // float4 per thread and K tile
  // PF register sets: tile kt + PF is requested while tile kt is computed, so PF tiles of global loads are in flight
  // per workgroup at any time (PF = 2 keeps the memory pipe fed across the split / LDS-write / barrier stretch)
  constexpr int NSET = DB ? 2 : PF;
  float4 ra[NSET][NA];
  float4 rz[NSET][GATE ? NA : 1];
  (void)rz;
  uint4 rw[NSET][4];
  float sa[NA];
  const float* arow[NA];
  const float* grow[GATE ? NA : 1];
  (void)grow;
  bool aok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int gm = m0 + lr + 32 * i;
    aok[i] = gm < g.m;
    const int gmc = aok[i] ? gm : g.m - 1;  // clamped: loads are unconditional (no branch around them), masked after
    sa[i] = srow[lr + 32 * i];
    arow[i] = A + (int64_t)gmc * g.lda + lc;
    if (GATE) grow[i] = g.gate + (int64_t)gmc * g.ldg + lc;
  }
  // two-stage instances (256 VGPRs): the inverse scales of this lane's epilogue rows / columns into registers while `srow` is
  // still valid -- the epilogue then neither refills it from memory nor divides
  float sinv_r[DB ? TI : 1][DB ? 16 : 1], winv_r[2];
  if (DB) {
    const int li_ = lane & 31, lh_ = lane >> 5;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sr = srow[WM * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh_];
        sinv_r[DB ? i : 0][DB ? r : 0] = sr != 0.f ? 1.f / sr : 0.f;  // (exact: a power of two; 0 marks rows beyond M)
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) winv_r[j] = g.w_inv[min(n0 + 64 * wn + 32 * j + li_, g.n - 1)];
  }
  __syncthreads();  // every thread has its row scales: the tile stages may overwrite srow now
  const _Float16* wrow[4];
  bool wok[4];
  int wkc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = tid + 256 * j;  // 16-byte chunk id: 2 planes x 128 rows x 4 chunks
    const int plane = q >> 9, rem = q & 511;
    const int r = rem >> 2;
    wkc[j] = (rem & 3) << 3;
    const int gn = n0 + r;
    wok[j] = gn < g.n;
    wrow[j] = W2 + (int64_t)plane * g.w2_plane + (int64_t)(wok[j] ? gn : g.n - 1) * 32 + wkc[j];  // (K-tile-major planes)
  }
  // Loads are unconditional (clamped addresses, never a branch around a load: hipcc would otherwise wait for ALL
  // outstanding loads at the next use and the second register set would buy nothing); what lies outside the problem is
  // zeroed arithmetically -- rows beyond M carry scale 0, a K tail (K is a multiple of 8, not of 32) zeroes its columns
  // through the scale of the tile / an AND mask on the weight chunks.
  const bool k_tail = (g.k & (BK - 1)) != 0;
  auto load_tile = [&](int set, int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int ko = (!k_tail || k0 + lc < g.k) ? k0 : 0;
      ra[set][i] = *reinterpret_cast<const float4*>(arow[i] + ko);
      if (GATE) rz[set][i] = *reinterpret_cast<const float4*>(grow[i] + ko);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool kin = !k_tail || k0 + wkc[j] < g.k;
      rw[set][j] = *reinterpret_cast<const uint4*>(wrow[j] + (kin ? (int64_t)(k0 >> 5) * g.w2_kt : 0));
    }
  };
  auto store_tile = [&](int set, int k0, int buf = 0) {
    const float kmask = (!k_tail || k0 + lc < g.k) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float xs[4] = {ra[set][i].x, ra[set][i].y, ra[set][i].z, ra[set][i].w};
      if (GATE) xs[0] *= rz[set][i].x, xs[1] *= rz[set][i].y, xs[2] *= rz[set][i].z, xs[3] *= rz[set][i].w;
      const float se = sa[i] * kmask;
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = xs[e] * se;
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
      }
      const int arow = lr + 32 * i;
      _Float16* dst = As + buf * STAGE + arow * PITCH + (DB ? ((((lc >> 3) ^ ((arow >> 2) & 3)) << 3) | (lc & 4)) : lc);
      *reinterpret_cast<f16x4*>(dst) = hi;
      *reinterpret_cast<f16x4*>(dst + APLANE) = lo;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + 256 * j;
      const int plane = q >> 9, rem = q & 511;
      const int r = rem >> 2, c = (rem & 3) << 3;
      const unsigned msk = (wok[j] && (!k_tail || k0 + wkc[j] < g.k)) ? 0xffffffffu : 0u;
      uint4 v = rw[set][j];
      v.x &= msk, v.y &= msk, v.z &= msk, v.w &= msk;
      const int cs = DB ? (((c >> 3) ^ ((r >> 2) & 3)) << 3) : c;
      *reinterpret_cast<uint4*>(Bs + buf * STAGE + plane * PLANE + r * PITCH + cs) = v;
    }
  };

  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  // lane (row li, half lh) holds k = 8*lh + j (j = 0..7) of a 16-deep MFMA step for both operands
  const _Float16* a_base = As + (WM * wm + li) * PITCH + (DB ? 0 : 8 * lh);
  const _Float16* b_base = Bs + (64 * wn + li) * PITCH + (DB ? 0 : 8 * lh);
  const int sw = (li >> 2) & 3;  // DB: chunk swizzle of this lane's rows (row offsets of the tiles are multiples of 32)

  const int nk_all = (g.k + BK - 1) / BK;
  const int kt0 = g.split_k > 1 ? blockIdx.z * g.k_tiles_per_split : 0;
  const int nk = g.split_k > 1 ? min(nk_all, kt0 + g.k_tiles_per_split) : nk_all;
  auto mfma_tile = [&]() {
    if (g.mfma_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f16x8 af[TI][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (t < TI) af[t][p] = *reinterpret_cast<const f16x8*>(a_base + p * APLANE + 32 * t * PITCH + 16 * ks);
          bf[t][p] = *reinterpret_cast<const f16x8*>(b_base + p * PLANE + 32 * t * PITCH + 16 * ks);
        }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);  // lo * hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);  // hi * lo
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);  // hi * hi
        }
    }
    if (g.mfma_prio) __builtin_amdgcn_s_setprio(0);
  };
  if (DB) {
    // step(kt): tile kt sits in stage cur, register set cur ^ 1 holds tile kt + 1, set cur is free.
    auto frag_mfma = [&](int buf, int ks) {
      f16x8 af[TI][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int ko = ((2 * ks + lh) ^ sw) << 3;
          if (t < TI) af[t][p] = *reinterpret_cast<const f16x8*>(a_base + buf * STAGE + p * APLANE + 32 * t * PITCH + ko);
          bf[t][p] = *reinterpret_cast<const f16x8*>(b_base + buf * STAGE + p * PLANE + 32 * t * PITCH + ko);
        }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);  // lo * hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);  // hi * lo
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);  // hi * hi
        }
    };
    auto step = [&](int cur, int kt) {
      load_tile(cur, min(kt + 2, nk - 1) * BK);       // into the free set; lands during the NEXT step
      frag_mfma(cur, 0);
      if (kt + 1 < nk) store_tile(cur ^ 1, (kt + 1) * BK, cur ^ 1);  // conversion + LDS writes between the MFMA halves
      frag_mfma(cur, 1);
      __syncthreads();
    };
    load_tile(0, kt0 * BK);
    load_tile(1, min(kt0 + 1, nk - 1) * BK);
    store_tile(0, kt0 * BK, 0);
    __syncthreads();
    int kt = kt0;
    for (; kt + 2 <= nk; kt += 2) {
      step(0, kt);
      step(1, kt + 1);
    }
    if (kt < nk) step(0, kt);
  } else {
  // The prefetch of a half is unconditional (past the last tile it re-reads the last one; nothing consumes it) and the
  // loop body always runs all PF halves: hipcc's wait counts are then exact -- a load behind a condition makes it
  // assume the shorter queue on every path and wait for the NEWEST register set where the oldest is needed.
  auto half = [&](int u, int kt) {
    // (scheduling fence: hipcc otherwise hoists the NEXT register set's conversion above this half's barriers and
    // with it the wait for that set's loads -- every load would again have to land before the first barrier)
    __builtin_amdgcn_sched_barrier(0);
    store_tile(u, kt * BK);
    __syncthreads();
    load_tile(u, min(kt + PF, nk - 1) * BK);
    mfma_tile();
    __syncthreads();
  };
  load_tile(0, kt0 * BK);
  if (PF > 1) load_tile(PF - 1, min(kt0 + 1, nk - 1) * BK);
  int kt = kt0;
  for (; kt + PF <= nk; kt += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) half(u, kt + u);
  }
  if (PF > 1 && kt < nk) half(0, kt);  // odd tile count: one half left
  }

  // epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5));
  // un-scale with the exact inverse powers of two of the row (A) and column (W) scales
  float* S = g.split_k > 1 ? g.splitk_ws + (int64_t)blockIdx.z * g.m * g.n : nullptr;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + li;
      if (col >= g.n) continue;
      const float wi = DB ? winv_r[j] : sinv[BM + col - n0];
      const float bv = HAS_BIAS ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + WM * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < g.m) {
          float v = acc[i][j][r] * (wi * (DB ? sinv_r[DB ? i : 0][DB ? r : 0] : sinv[row - m0]));
          if (S != nullptr) {  // raw partial sums into this split's slab [M][N]; bias / residual are applied by the reduce
            S[(int64_t)row * g.n + col] = v;
            continue;
          }
          v += bv;
          if (HAS_RES) v += g.residual[(int64_t)row * g.ldc + col];
          if (g.act_silu_from >= 0 && col >= g.act_silu_from) v = silu_hw(v);
          C[(int64_t)row * g.ldc + col] = v;
        }
      }
    }
}

// One wave per row: amax[r] = max_k |a[r][k] (* gate[r][k])|.
__global__ __launch_bounds__(256) void row_amax_kernel(const float* a, int64_t lda, const float* gate, int64_t ldg, int rows,
                                                       int k, float* amax) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* ar = a + (int64_t)row * lda;
  const float* gr = gate != nullptr ? gate + (int64_t)row * ldg : nullptr;
  float mx = 0.f;
  for (int c = lane * 4; c < k; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(ar + c);
    if (gr != nullptr) {
      const float4 z = *reinterpret_cast<const float4*>(gr + c);
      v.x *= z.x, v.y *= z.y, v.z *= z.z, v.w *= z.w;
    }
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  mx = wave_max(mx);
  if (lane == 0) amax[row] = mx;
}

// One wave per weight row: planes[0] = hi, planes[1] = lo of scale * w[n][k], K-tile-major ((n, k) at ((k / 32) * rows + n) * 32
// + k % 32; the columns of a last partial K tile beyond K are never used: the GEMM masks them); inv[n] = 1 / scale.
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* w, int rows, int k, _Float16* planes, int64_t plane,
                                                          float* inv) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* wr = w + (int64_t)row * k;
  float mx = 0.f;
  for (int c = lane; c < k; c += 64) mx = fmaxf(mx, fabsf(wr[c]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  const float s = pow2_scale(mx);
  for (int c = lane; c < k; c += 64) {
    const float v = wr[c] * s;
    const _Float16 h = (_Float16)v;
    const int64_t at = ((int64_t)(c >> 5) * rows + row) * 32 + (c & 31);
    planes[at] = h;
    planes[plane + at] = (_Float16)(v - (float)h);
  }
  if (lane == 0) inv[row] = 1.f / s;
}
}  // namespace

bool gemm_f16x2_supported(const GemmArgs& g) {
  return g.w2 != nullptr && g.w_inv != nullptr && g.nb1 * g.nb2 == 1 && (g.k & 7) == 0 && g.w2_kt >= 32 * (int64_t)g.n &&
         (g.w2_kt & 7) == 0 && (g.lda & 3) == 0 && (g.w2_plane & 7) == 0 && (g.gate == nullptr || (g.ldg & 3) == 0);
}

void launch_row_amax(const float* a, int64_t lda, const float* gate, int64_t ldg, int rows, int k, float* amax,
                     hipStream_t stream) {
  LRAM_REQUIRE((k & 3) == 0 && (lda & 3) == 0 && (gate == nullptr || (ldg & 3) == 0), "row amax: K, lda must be multiples of 4");
  hipLaunchKernelGGL(row_amax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a, lda, gate, ldg, rows, k, amax);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_split_f16x2(const float* w, int rows, int k, uint16_t* planes, float* inv, hipStream_t stream) {
  hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, w, rows, k,
                     reinterpret_cast<_Float16*>(planes), (int64_t)split_f16x2_plane_elems(rows, k), inv);
  LRAM_HIP_CHECK(hipGetLastError());
}

template <int BM, bool GATE, int PF>
static void launch_bm_pf(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  const bool hb = g.bias != nullptr, hr = g.residual != nullptr;
  dim3 block(256);
  if (hb && hr)
    hipLaunchKernelGGL((gemm_f16x2_kernel<true, true, BM, GATE, PF>), grid, block, 0, stream, g);
  else if (hb)
    hipLaunchKernelGGL((gemm_f16x2_kernel<true, false, BM, GATE, PF>), grid, block, 0, stream, g);
  else if (hr)
    hipLaunchKernelGGL((gemm_f16x2_kernel<false, true, BM, GATE, PF>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f16x2_kernel<false, false, BM, GATE, PF>), grid, block, 0, stream, g);
}

template <int BM, bool GATE>
static void launch_bm(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  // Two LDS stages + one barrier per K tile (the kernel's DB note) pay where the launch cannot fill the chip with
  // workgroups anyway -- at most one per CU: 206M proj_down (240 workgroups) 109 -> 86 us -- and lose where three
  // single-stage workgroups per CU can overlap each other's phases (16M proj_up 62 -> 75 us, Mamba in_proj 128 -> 140).
  // (Two K tiles of global loads in flight per workgroup -- PF = 2 -- pushed the kernel from three workgroups per CU to two: 94 ->
  // 118 us on proj_up; removed.)
  // (also tried for split-K launches -- Mamba's x_proj, 45 -> 35 us standalone: no end-to-end difference, 365.4k vs 365.1k)
  const long wgs = (long)grid.x * grid.y * grid.z;
  if (wgs <= 256 && !GATE) {
    const bool hb = g.bias != nullptr, hr = g.residual != nullptr;
    dim3 block(256);
    if (hb && hr)
      hipLaunchKernelGGL((gemm_f16x2_kernel<true, true, BM, false, 2, true>), grid, block, 0, stream, g);
    else if (hb)
      hipLaunchKernelGGL((gemm_f16x2_kernel<true, false, BM, false, 2, true>), grid, block, 0, stream, g);
    else if (hr)
      hipLaunchKernelGGL((gemm_f16x2_kernel<false, true, BM, false, 2, true>), grid, block, 0, stream, g);
    else
      hipLaunchKernelGGL((gemm_f16x2_kernel<false, false, BM, false, 2, true>), grid, block, 0, stream, g);
    return;
  }
  launch_bm_pf<BM, GATE, 1>(g, grid, stream);
}

// g.a_amax: per-row largest magnitude of A (of the gated rows when g.gate is set), from launch_row_amax or A's producer.
void launch_gemm_f16x2(const GemmArgs& g_in, hipStream_t stream) {
  GemmArgs g = g_in;
  g.mfma_prio = 1;
  LRAM_REQUIRE(g.m > 0 && g.n > 0 && g.k > 0, "gemm: empty problem");
  LRAM_REQUIRE(gemm_f16x2_supported(g) && g.a_amax != nullptr, "gemm f16x2: unsupported operand layout");
  int S = 1;
  if (g.act_silu_from >= 0 || g.gate != nullptr)
    g.split_k = 1, g.k_tiles_per_split = 0;  // output activation / gated operand: K unsplit
  else
    S = gemm_choose_split_k(g);
  const int tiles_n = (g.n + BN - 1) / BN;
  const int tiles128 = ((g.m + 127) / 128) * tiles_n;
  // 64-row tiles where 128-row tiles leave workgroup slots empty (three workgroups per CU: 768 slots) and the launch is narrow, or
  // where they would leave CUs without any workgroup and K is short (1536 x 1024 x 512: 31.6 us with 96 tiles of 128 rows, 20.5
  // with 192 of 64; long-K launches: see launch_gemm_f16x2p)
  // (LRAM_GEMM_TILE = 64 / 128 forces one: the knob of launch_gemm_f16x2p)
  const int force_bm = gemm_knobs().tile;
  constexpr int bm64_below = 256, bm64_anyk = 128;
  const bool small = force_bm == 64 || (force_bm != 128 && g.m > 64 && ((S == 1 && tiles128 < 768 && tiles_n <= 6) || ((long)tiles128 * S < bm64_below && g.k <= 768) || (long)tiles128 * S < bm64_anyk));
  const int tiles = small ? ((g.m + 63) / 64) * tiles_n : tiles128;
  dim3 grid(tiles, 1, S);
  gemm_choose_xcd_split(g, small ? 64 : 128, BN, 4);
  if (g.gate != nullptr) {
    if (small) launch_bm<64, true>(g, grid, stream); else launch_bm<128, true>(g, grid, stream);
  } else {
    if (small) launch_bm<64, false>(g, grid, stream); else launch_bm<128, false>(g, grid, stream);
  }
  LRAM_HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce(g, stream);
}

}  // namespace lram
