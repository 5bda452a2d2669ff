// f16x2 projection GEMM, pre-split operands, 256 x 256 workgroup tile, 8 waves in two staggered groups, 8 phases per two K tiles.
//
// Same arithmetic as gemm_f16x2p.hip -- C[M,N] = A[M,K] W[N,K]^T with both operands handed over as two binary16 planes (hi, lo)
// of the row-scaled value, hi*hi + hi*lo + lo*hi accumulated in fp32 by v_mfma_f32_32x32x16_f16 in the SAME order per output
// element (K tile ascending, k16 step ascending, lo*hi, hi*lo, hi*hi), exact un-scaling: results are bit-identical to that
// kernel's and to gemm_f16x2.hip's.  What changes is the schedule.  The 128 x 128 kernels run two to four workgroups per CU
// whose barriers drain every wave's LDS-DMA (`__syncthreads` waits vmcnt(0)): 0.44-0.51 matrix-pipe busy on the largest
// launches, a wave spending 38-40 % of its K loop ISSUING staging instructions (profiles/r05_gemm_presplit_phase_profile.txt).
// Here (the "8-phase" structure of cdna_hip_programming.md section 5, re-derived for split operands):
//
//   * one workgroup of 8 waves per CU, 256 x 256 outputs, wave (wr, wc) of a 2 x 4 grid owns 128 x 64 of them as 2 x 2
//     QUADRANTS of 64 x 32 (two 32 x 32 accumulator tiles each; 128 accumulator registers per lane);
//   * a K tile is 32 deep: per operand 256 rows x 32 x two planes = 32 KB, staged as FOUR 16 KB half-tiles -- A0 / A1 = the
//     first / second 64 rows of both wave rows, B0 / B1 = the first / second 32 columns of all four wave columns, i.e. the
//     operand sub-tile ONE quadrant step needs from every wave -- into a two-tile ring of 8 slots (128 KB of LDS) by
//     `buffer_load ... lds` (1 KB per wave instruction, swizzle on the source address);
//   * a K tile is four phases, one quadrant each: (a0,b0) (a0,b1) (a1,b1) (a1,b0).  A phase = { ds_read the sub-tile that
//     changed (B0 + A0: 12 reads; B1: 4; A1: 8; none), issue TWO DMA pieces (one half-tile per phase over the 8 waves) }
//     s_barrier { 12 MFMAs } s_barrier.  Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave's MFMA block
//     (384 cycles) covers its partner's reads and DMA issue;
//   * half-tile i (tile i / 4, half i % 4 in the order A0 B0 B1 A1) is issued SIX phases before the tile that reads it starts
//     its phase i % 4; the only VM wait of the loop is a counted `s_waitcnt vmcnt(4)` in the last phase of each K tile (two
//     half-tiles stay in flight across the tile boundary; vmcnt(0) only before the final tile); barriers are raw s_barrier.
//     Placement rules (MI355X_MICROARCH.md "Two waves per SIMD" item 7, cdna_hip_programming.md "Read a staged buffer one phase
//     AFTER the wait that retires it"): RAW -- a half-tile's pieces are retired by every issuing wave's vmcnt in phase 3 of the
//     previous tile, before that phase's first barrier, and read from phase 0 of the next tile on; WAR -- slot (tile & 1, half j)
//     is re-filled for tile t + 2 in phase (j + 2) & 3 ... of tile t (A0: phase 2, B0: 3, B1: phase 0 of t + 1, A1: phase 1 of
//     t + 1), at least two phases after its last read (A0, B0: phase 0; B1: phase 1; A1: phase 2).
//
// Arithmetic intensity: 64 KB of operand planes per 96 MFMAs of 32 cycles per SIMD-pair ... per K tile and CU: 3072 matrix-pipe
// cycles per SIMD for 64 KB delivered = 21 B / clock / CU (the 128 x 128 tiles needed 43).
//
// Replaces on the reference path: the same nn.Linear projections as gemm_f16x2p.hip (src/algos/models/decision_mamba.py:78-93
// in_proj / out_proj; xlstm proj_up / proj_down / FFN, call site src/algos/models/decision_xlstm.py:159-163).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int BK = 32;
constexpr int BMT = 256, BNT = 256;
constexpr int HALF = 128 * BK * 2;       // f16 elements of one half-tile: 128 rows x 32 x two planes (16 KB)
constexpr int PLN = 128 * BK;            // ... of one plane of it
constexpr int TILE = 4 * HALF;           // one K tile of the ring (64 KB)
// slot order inside a K tile of the ring = issue order
constexpr int S_A0 = 0, S_B0 = 1, S_B1 = 2, S_A1 = 3;

#if defined(__HIP_DEVICE_COMPILE__)
#define LRAM_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#else
#define LRAM_WAIT_VM(n)
#endif

// ABL (measurement builds only, scripts/gemm8p_ablate.cpp; the library instantiates 0): 1 = DMA descriptors with zero records (the
// instruction stream and the waits stay, no byte moves), 2 = no DMA instructions, 3 = fragments read once (no ds_read in the
// loop), 4 = s_memtime stamps of waves 0 and 4 of workgroup 0 into splitk_ws (phase shares), 5 = both ring tiles staged once, no
// DMA in the loop (random operands WITHOUT memory traffic: separates delivery from the clock effect of 1 / 2, whose operands are
// zeros); outputs meaningless in 1-5
template <bool HAS_BIAS, bool HAS_RES, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_f16x2_8p_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(1024))) _Float16 lds[2 * TILE];   // the ONLY LDS object of the kernel (128 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long real_entry = 0;
  if (ABL == 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real_entry)::"memory");
#endif
  }
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 31, lh = lane >> 5;

  const int tiles_n = (g.n + BNT - 1) / BNT;
  const int tiles_m = (g.m + BMT - 1) / BMT;
  int tm_idx, tn_idx;
  gemm_tile_of(g, blockIdx.x, tiles_m, tiles_n, tm_idx, tn_idx);
  const int m0 = tm_idx * BMT, n0 = tn_idx * BNT;

  // ---- DMA sources.  Wave w stages rows 16 w .. 16 w + 15 of every half-tile, one 1 KB piece per plane.  Half-tile row rho:
  // A half h: tile row (rho >> 6) * 128 + 64 h + (rho & 63); B half h: tile column (rho >> 5) * 64 + 32 h + (rho & 31).
  // Lane l fills linear position (row 16 w + l / 4, 16-byte chunk l & 3) and fetches logical chunk (l & 3) ^ ((rho >> 2) & 3)
  // of that row (the fragment reads apply the same involution).  Rows beyond M / N are clamped (their products land in rows /
  // columns the epilogue drops).  Buffer form: resource per operand based at the tile's first row, loop-invariant per-lane
  // byte offsets, K tile + plane offsets in the scalar operand.
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(g.a2) + (int64_t)m0 * 32, 0, ABL == 1 ? 0u : 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(g.w2) + (int64_t)n0 * 32, 0, ABL == 1 ? 0u : 0xffffffffu, 0x00020000);
#endif
  const int rho = 16 * wave + (lane >> 2);
  const int chunk = (lane & 3) ^ ((rho >> 2) & 3);
  unsigned voff_a[2], voff_b[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ra = (rho >> 6) * 128 + 64 * h + (rho & 63);
    const int rb = (rho >> 5) * 64 + 32 * h + (rho & 31);
    voff_a[h] = (unsigned)(((min(m0 + ra, g.m - 1) - m0) * 32 + 8 * chunk) * 2);
    voff_b[h] = (unsigned)(((min(n0 + rb, g.n - 1) - n0) * 32 + 8 * chunk) * 2);
  }
  const unsigned kt_bytes_a = (unsigned)(g.a2_kt * 2), kt_bytes_w = (unsigned)(g.w2_kt * 2);
  const unsigned pl_bytes_a = (unsigned)(g.a2_plane * 2), pl_bytes_w = (unsigned)(g.w2_plane * 2);

  // stage half-tile `slot` of K tile kt into ring tile `buf` (2 DMA pieces per wave)
  bool dma_on = true;
  auto stage = [&](int buf, int slot, int kt) {
    if (ABL == 2 || (ABL == 5 && !dma_on)) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const bool is_a = slot == S_A0 || slot == S_A1;
    const int h = (slot == S_A1 || slot == S_B1) ? 1 : 0;
    _Float16* dst = lds + buf * TILE + slot * HALF + 16 * wave * BK;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const unsigned soff = (unsigned)kt * (is_a ? kt_bytes_a : kt_bytes_w) + (unsigned)p * (is_a ? pl_bytes_a : pl_bytes_w);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? rsrc_a : rsrc_w, (__attribute__((address_space(3))) void*)(dst + p * PLN), 16,
                                               is_a ? voff_a[h] : voff_b[h], soff, 0, 0);
    }
#endif
  };

  f32x16 acc[2][2][2];   // [a sub-tile][row tile i][b sub-tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;

  // fragment addresses (f16 units inside a half-tile): row * 32 + ((2 ks + lh) ^ sw) * 8; sw from the row inside its 32-row tile
  const int sw = (li >> 2) & 3;
  const int ko0 = ((0 + lh) ^ sw) << 3, ko1 = ((2 + lh) ^ sw) << 3;
  const int a_row = (wr * 64 + li) * BK;   // + i * 32 * BK
  const int b_row = (wc * 32 + li) * BK;

  const int nk_all = g.k / BK;
  const int kt0 = g.split_k > 1 ? blockIdx.z * g.k_tiles_per_split : 0;
  const int nk = g.split_k > 1 ? min(nk_all, kt0 + g.k_tiles_per_split) : nk_all;

  f16x8 af[2][2][2];     // [row tile i][ks][plane]
  f16x8 bf0[2][2], bf1[2][2];   // [ks][plane]

  bool reads_on = true;
  auto read_a = [&](int buf, int slot) {
    if (ABL == 3 && !reads_on) return;
    const _Float16* base = lds + buf * TILE + slot * HALF + a_row;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        af[i][0][p] = *reinterpret_cast<const f16x8*>(base + p * PLN + i * 32 * BK + ko0);
        af[i][1][p] = *reinterpret_cast<const f16x8*>(base + p * PLN + i * 32 * BK + ko1);
      }
  };
  auto read_b = [&](int buf, int slot, f16x8 (&bf)[2][2]) {
    if (ABL == 3 && !reads_on) return;
    const _Float16* base = lds + buf * TILE + slot * HALF + b_row;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      bf[0][p] = *reinterpret_cast<const f16x8*>(base + p * PLN + ko0);
      bf[1][p] = *reinterpret_cast<const f16x8*>(base + p * PLN + ko1);
    }
  };
  auto quadrant = [&](f32x16 (&c0), f32x16 (&c1), const f16x8 (&bf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // smallest terms first (the order of gemm_f16x2p.hip: bit-identical sums)
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][ks][1], bf[ks][0], c0, 0, 0, 0);  // lo * hi
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][ks][1], bf[ks][0], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][ks][0], bf[ks][1], c0, 0, 0, 0);  // hi * lo
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][ks][0], bf[ks][1], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][ks][0], bf[ks][0], c0, 0, 0, 0);  // hi * hi
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][ks][0], bf[ks][0], c1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: half-tiles 0 .. 5 (all of tile kt0, A0 / B0 of tile kt0 + 1); tile kt0 landed before the loop
  if (kt0 < nk) {
    stage(0, S_A0, kt0), stage(0, S_B0, kt0), stage(0, S_B1, kt0), stage(0, S_A1, kt0);
    if (kt0 + 1 < nk) {
      stage(1, S_A0, kt0 + 1), stage(1, S_B0, kt0 + 1);
      if (ABL == 5) {
        stage(1, S_B1, kt0 + 1), stage(1, S_A1, kt0 + 1);
        LRAM_WAIT_VM(0);
        dma_on = false;
      }
      LRAM_WAIT_VM(4);
    } else {
      LRAM_WAIT_VM(0);
    }
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind waves 0-3 from here on

  if (ABL == 3) {   // fragments once, from whatever the prologue staged
    read_a(0, S_A0), read_b(0, S_B0, bf0), read_b(0, S_B1, bf1);
    reads_on = false;
  }
  // stamps (ABL 4): per phase slot p (0..3) sums of [reads + DMA issue (+ VM wait)] [barrier 1] [MFMA issue] [barrier 2]
  unsigned long long st_sum[4][4] = {}, st_prev = 0;
  const bool stamping = ABL == 4 && blockIdx.x == 0 && blockIdx.z == 0 && (wave == 0 || wave == 4);
  auto stamp = [&](int p, int seg) {
    if (ABL != 4) return;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (seg >= 0) st_sum[p][seg] += t - st_prev;
    st_prev = t;
#endif
  };
  unsigned long long real0 = 0, cyc0 = 0;
  if (ABL == 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0), "=s"(cyc0)::"memory");
#endif
  }
  stamp(0, -1);
  // one K tile = four phases on ring tile BUF
#define LRAM_PHASE_SYNC_1()                 \
  __builtin_amdgcn_sched_barrier(0);        \
  __builtin_amdgcn_s_barrier();             \
  __builtin_amdgcn_sched_barrier(0)
#define LRAM_PHASE_SYNC_2()                 \
  __builtin_amdgcn_sched_barrier(0);        \
  __builtin_amdgcn_s_barrier();             \
  __builtin_amdgcn_sched_barrier(0)
#define LRAM_K_TILE(BUF, kt)                                                                  \
  {                                                                                           \
    /* phase 0: (a0, b0) */                                                                   \
    read_b(BUF, S_B0, bf0);                                                                   \
    read_a(BUF, S_A0);                                                                        \
    if ((kt) + 1 < nk) stage((BUF) ^ 1, S_B1, (kt) + 1);                                      \
    stamp(0, 0);                                                                              \
    LRAM_PHASE_SYNC_1();                                                                      \
    stamp(0, 1);                                                                              \
    quadrant(acc[0][0][0], acc[0][1][0], bf0);                                                \
    stamp(0, 2);                                                                              \
    LRAM_PHASE_SYNC_2();                                                                      \
    stamp(0, 3);                                                                              \
    /* phase 1: (a0, b1) */                                                                   \
    read_b(BUF, S_B1, bf1);                                                                   \
    if ((kt) + 1 < nk) stage((BUF) ^ 1, S_A1, (kt) + 1);                                      \
    stamp(1, 0);                                                                              \
    LRAM_PHASE_SYNC_1();                                                                      \
    stamp(1, 1);                                                                              \
    quadrant(acc[0][0][1], acc[0][1][1], bf1);                                                \
    stamp(1, 2);                                                                              \
    LRAM_PHASE_SYNC_2();                                                                      \
    stamp(1, 3);                                                                              \
    /* phase 2: (a1, b1) */                                                                   \
    read_a(BUF, S_A1);                                                                        \
    if ((kt) + 2 < nk) stage(BUF, S_A0, (kt) + 2);                                            \
    stamp(2, 0);                                                                              \
    LRAM_PHASE_SYNC_1();                                                                      \
    stamp(2, 1);                                                                              \
    quadrant(acc[1][0][1], acc[1][1][1], bf1);                                                \
    stamp(2, 2);                                                                              \
    LRAM_PHASE_SYNC_2();                                                                      \
    stamp(2, 3);                                                                              \
    /* phase 3: (a1, b0); the next tile's half-tiles are retired here, read from its phase 0 on */ \
    if ((kt) + 2 < nk) {                                                                      \
      stage(BUF, S_B0, (kt) + 2);                                                             \
      LRAM_WAIT_VM(4);                                                                        \
    } else {                                                                                  \
      LRAM_WAIT_VM(0);                                                                        \
    }                                                                                         \
    stamp(3, 0);                                                                              \
    LRAM_PHASE_SYNC_1();                                                                      \
    stamp(3, 1);                                                                              \
    quadrant(acc[1][0][0], acc[1][1][0], bf0);                                                \
    stamp(3, 2);                                                                              \
    LRAM_PHASE_SYNC_2();                                                                      \
    stamp(3, 3);                                                                              \
  }

  for (int kt = kt0; kt < nk; kt += 2) {
    LRAM_K_TILE(0, kt);
    if (kt + 1 < nk) LRAM_K_TILE(1, kt + 1);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();   // (every wave has passed the same number of barriers)
  if (ABL == 4) {
    unsigned long long real1 = 0, cyc1 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real1), "=s"(cyc1)::"memory");
#endif
    if (stamping && lane == 0) {
      unsigned long long* out = reinterpret_cast<unsigned long long*>(g.splitk_ws) + (wave == 0 ? 0 : 16);
      for (int p = 0; p < 4; ++p)
        for (int sg = 0; sg < 4; ++sg) out[4 * p + sg] = st_sum[p][sg];
      if (wave == 0) out[32] = real1 - real0, out[33] = cyc1 - cyc0;   // K loop: 100 MHz ticks, s_memtime ticks
    }
    if (wave == 0 && lane == 0) {   // every workgroup: entry, loop start, loop end (s_memrealtime); the exit stamp follows the epilogue
      unsigned long long* tl = reinterpret_cast<unsigned long long*>(g.splitk_ws) + 64 + 4 * (int64_t)blockIdx.x;
      tl[0] = real_entry, tl[1] = real0, tl[2] = real1;
    }
  }
#undef LRAM_K_TILE
#undef LRAM_PHASE_SYNC_1
#undef LRAM_PHASE_SYNC_2

  // ---- epilogue.  C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) -- a lane
  // holds ONE column of 16 rows, so storing from the accumulators is 128 single-dword store instructions per lane, two 128-byte
  // row pieces each: measured 21-24 us per workgroup (38 k cycles for 256 KB, ~7 B / clock / CU: store-issue-bound) against a
  // K loop of 39-92 us, with nothing to overlap it at one workgroup per CU (scripts/gemm8p_ablate.cpp, per-workgroup timeline;
  // profiles/r06_gemm_8phase_ablation.txt).  The operand ring is free once every wave has left the K loop, so each wave turns
  // its 64 x 64 accumulator half through 16 KB of it: ds_write_b32 in [row][col] order (a lane half writes 32 consecutive
  // floats: conflict-free), ds_read_b128 of four whole rows per instruction, un-scale (exact powers of two) + bias + residual
  // + activation on float4s, dwordx4 stores of 256 contiguous bytes per row.  Same arithmetic per element as before:
  // acc * (w_inv * a_inv) + bias, + residual, silu.
  float* C = g.c;
  float* S = g.split_k > 1 ? g.splitk_ws + (int64_t)blockIdx.z * g.m * g.n : nullptr;
  typedef float __attribute__((may_alias)) lds_f32;
  typedef float4 __attribute__((may_alias)) lds_f32x4;
  const int64_t ld_out = S != nullptr ? (int64_t)g.n : g.ldc;
  float* out = S != nullptr ? S : C;
  const bool vec = (ld_out & 3) == 0 && (g.n & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(g.w_inv + n0) & 15) == 0 &&
                   (!HAS_BIAS || (reinterpret_cast<uintptr_t>(g.bias + n0) & 15) == 0) &&
                   (!HAS_RES || (reinterpret_cast<uintptr_t>(g.residual) & 15) == 0);
  if (vec) {
    __builtin_amdgcn_s_barrier();   // every wave's fragment reads of the ring are behind it (each was waited for by its MFMAs)
    lds_f32* stg = reinterpret_cast<lds_f32*>(lds) + wave * 4096;
    const int c4 = (lane & 15) * 4, rq = lane >> 4;
    const int gcol = n0 + wc * 64 + c4;
    const bool col_in = gcol < g.n;   // (n is a multiple of 4: a float4 is inside or outside as a whole)
    float4 wi4 = make_float4(0.f, 0.f, 0.f, 0.f), bv4 = wi4;
    if (col_in) {
      wi4 = *reinterpret_cast<const float4*>(g.w_inv + gcol);
      if (HAS_BIAS && S == nullptr) bv4 = *reinterpret_cast<const float4*>(g.bias + gcol);
    }
    const int act_from = S == nullptr ? g.act_silu_from : -1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stg[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + b * 32 + li] = acc[a][i][b][r];
      const int grow0 = m0 + wr * 128 + a * 64 + rq;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int grow = grow0 + 4 * it;
        const float4 t = *reinterpret_cast<const lds_f32x4*>(stg + (4 * it + rq) * 64 + c4);
        if (grow < g.m && col_in) {
          const float ai = g.a2_inv[grow];
          float4 v;
          v.x = t.x * (wi4.x * ai) + bv4.x, v.y = t.y * (wi4.y * ai) + bv4.y;
          v.z = t.z * (wi4.z * ai) + bv4.z, v.w = t.w * (wi4.w * ai) + bv4.w;
          if (HAS_RES && S == nullptr) {
            const float4 rr = *reinterpret_cast<const float4*>(g.residual + (int64_t)grow * g.ldc + gcol);
            v.x += rr.x, v.y += rr.y, v.z += rr.z, v.w += rr.w;
          }
          if (act_from >= 0) {
            if (gcol + 0 >= act_from) v.x = silu_hw(v.x);
            if (gcol + 1 >= act_from) v.y = silu_hw(v.y);
            if (gcol + 2 >= act_from) v.z = silu_hw(v.z);
            if (gcol + 3 >= act_from) v.w = silu_hw(v.w);
          }
          *reinterpret_cast<float4*>(out + (int64_t)grow * ld_out + gcol) = v;
        }
      }
    }
  } else {
    // (row pitches / column counts that are not multiples of 4: the accumulator-order stores)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row0 = m0 + wr * 128 + a * 64 + i * 32 + 4 * lh;
        float ainv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ainv[r] = g.a2_inv[min(row0 + (r & 3) + 8 * (r >> 2), g.m - 1)];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int col = n0 + wc * 64 + b * 32 + li;
          if (col >= g.n) continue;
          const float wi = g.w_inv[col];
          const float bv = HAS_BIAS ? g.bias[col] : 0.f;
          const bool act = g.act_silu_from >= 0 && col >= g.act_silu_from;
          const bool rows_in = row0 + 27 < g.m;
          const f32x16& t = acc[a][i][b];
          if (S != nullptr) {
            float* sp = S + (int64_t)row0 * g.n + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ro = (r & 3) + 8 * (r >> 2);
              if (rows_in || row0 + ro < g.m) sp[(int64_t)ro * g.n] = t[r] * (wi * ainv[r]);
            }
            continue;
          }
          float* cp = C + (int64_t)row0 * g.ldc + col;
          const float* rp = HAS_RES ? g.residual + (int64_t)row0 * g.ldc + col : nullptr;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (rows_in || row0 + ro < g.m) {
              float v = t[r] * (wi * ainv[r]) + bv;
              if (HAS_RES) v += rp[(int64_t)ro * g.ldc];
              if (act) v = silu_hw(v);
              cp[(int64_t)ro * g.ldc] = v;
            }
          }
        }
      }
  }
  if (ABL == 4) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long real_exit;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real_exit)::"memory");
    if (wave == 0 && lane == 0) (reinterpret_cast<unsigned long long*>(g.splitk_ws) + 64 + 4 * (int64_t)blockIdx.x)[3] = real_exit;
#endif
  }
}
#undef LRAM_WAIT_VM
}  // namespace

bool gemm_f16x2_8p_supported(const GemmArgs& g) { return gemm_f16x2p_supported(g) && g.m >= 1 && g.n >= 1; }

// K splits of the 256 x 256 kernel: only where the tile grid leaves most of the chip without a workgroup and K is deep
int gemm_f16x2_8p_split_k(GemmArgs& g) {
  g.split_k = 1, g.k_tiles_per_split = 0;
  if (g.splitk_ws == nullptr || g.act_silu_from >= 0) return 1;
  const int tiles = ((g.m + BMT - 1) / BMT) * ((g.n + BNT - 1) / BNT);
  const int nk = g.k / BK;
  if (tiles >= 96 || nk < 16) return 1;
  int S = std::min(std::min(nk / 8, 8), (224 + tiles - 1) / tiles);
  while (S > 1 && (int64_t)S * g.m * g.n > g.splitk_ws_elems) --S;
  if (S < 2) return 1;
  const int per = (nk + S - 1) / S;
  S = (nk + per - 1) / per;
  if (S < 2) return 1;
  g.split_k = S, g.k_tiles_per_split = per;
  return S;
}

void launch_gemm_f16x2_8p(const GemmArgs& g_in, hipStream_t stream) {
  GemmArgs g = g_in;
  LRAM_REQUIRE(g.m > 0 && g.n > 0 && g.k > 0, "gemm: empty problem");
  LRAM_REQUIRE(gemm_f16x2_8p_supported(g), "gemm f16x2 (8-phase): unsupported operand layout");
  const int S = gemm_f16x2_8p_split_k(g);
  const int tiles = ((g.m + BMT - 1) / BMT) * ((g.n + BNT - 1) / BNT);
  dim3 grid(tiles, 1, S), block(512);
  gemm_choose_xcd_split(g, BMT, BNT, 4);
  const bool hb = g.bias != nullptr, hr = g.residual != nullptr;
  if (hb && hr)
    hipLaunchKernelGGL((gemm_f16x2_8p_kernel<true, true>), grid, block, 0, stream, g);
  else if (hb)
    hipLaunchKernelGGL((gemm_f16x2_8p_kernel<true, false>), grid, block, 0, stream, g);
  else if (hr)
    hipLaunchKernelGGL((gemm_f16x2_8p_kernel<false, true>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f16x2_8p_kernel<false, false>), grid, block, 0, stream, g);
  LRAM_HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce(g, stream);
}

}  // namespace lram
