// f16x2 projection GEMM with BOTH operands pre-split: C[M,N] = A[M,K] * W[N,K]^T, fp32-accurate on the f16 matrix cores.
//
// Same arithmetic as gemm_f16x2.hip (rows scaled by a power of two, every scaled element split exactly into two
// binary16 pieces, hi*hi + hi*lo + lo*hi accumulated in fp32, exact un-scaling) -- what changes is WHERE the A operand is
// split.  gemm_f16x2_kernel converts its fp32 A tile while it stages it: ~10 vector instructions per element, in every
// workgroup that touches the tile, i.e. N / 128 times per element (16M proj_up: 16 times, Mamba in_proj: 24 times) -- as
// many vector cycles per K tile as the tile's 24 MFMAs take, behind barriers that keep the two from overlapping inside a
// workgroup (matrix pipe 0.27-0.33 busy, profiles/r03_gemm_mfma_busy.json).  Two f16 planes are 4 bytes per element: the
// same bytes as the fp32 value.  So the kernel that PRODUCES A -- a row norm, which holds the whole row and its maximum in
// one wave -- writes the two planes and the row's inverse scale instead of fp32 (launch_row_norm / launch_add_rms_norm,
// `h2`), or launch_row_split_f16x2 does it once per row where the producer cannot, and this kernel's staging is a pure copy
// for both operands: global -> LDS directly (`global_load_lds_dwordx4`, 1 KiB per wave instruction, no VGPR round trip, no
// conversion, no LDS-write instructions), an MFMA-only inner loop, two LDS stages with the next tile's DMA in flight
// under the current tile's MFMAs and ONE barrier per K tile.  110 VGPRs.
//
// LDS image of a stage: [A hi][A lo][W hi][W lo], each 128 rows x 32 f16 (64-byte rows, un-padded: the DMA writes 1 KiB
// contiguously = 16 rows).  A ds_read_b128 fragment read of rows r, r + 4, r + 8, r + 12 would hit the same banks, so
// the 16-byte chunk index is XOR-ed with (row >> 2) & 3 -- applied on the SOURCE address of the DMA (the lane that fills
// linear position (row, c) fetches logical chunk c ^ ((row >> 2) & 3)) and on the fragment read, the same involution on
// both sides (cdna_hip_programming.md section 5.4 rule 21).
//
// Replaces on the reference path: nn.Linear in_proj / out_proj of mamba_ssm.Mamba (src/algos/models/decision_mamba.py:78-93),
// xlstm proj_up / proj_down / FFN projections ([3P], call site src/algos/models/decision_xlstm.py:159-163).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int BK = 32;

// Workgroup tile = WM x WN waves, each wave a (32 TI) x 64 block of the output (TI x 2 accumulator tiles of the 32 x 32 MFMA):
//   TI 1, 2 x 2 waves:  64 x 128  (launches that would leave CUs without a workgroup)
//   TI 2, 2 x 2 waves: 128 x 128  (rounds 3-4: 32 KB per K tile for 96 MFMAs = 341 B of operand delivery per MFMA -- the
//                                  kernel is bound by global -> LDS delivery, ~29 B / clock / CU out of L2, i.e. 11.8 clocks per
//                                  MFMA against the 8 the four SIMDs need: profiles/EXPERIMENTS.md "Projection GEMM ...")
// Round 5 instantiated this template for 256 x 128 (4 x 2 waves, 256 B per MFMA) and 256 x 256 (4 x 4 waves, 171 B per MFMA)
// tiles too, and added a 3 / 4 stage LDS ring with counted vmcnt waits: every form bit-identical, none faster -- within 3 % on
// the largest launches (24576 x 2048 x 512: 158-166 us against 163; 16128 x 5120 x 1280: 629-632 against 643), 10-30 % slower
// on everything a step launches, the rings (one or two waves per SIMD) 1.3-1.7 x slower.  What the sweep says: the kernel needs
// four waves per SIMD to cover its LDS-read -> MFMA dependency and its barrier, and four co-resident 32 KB stages ARE the
// bytes in flight that set the delivery rate; a taller tile trades one for the other.  profiles/r05_gemm_presplit_tile_sweep.txt,
// r05_gemm_presplit_ring_sweep.txt, EXPERIMENTS.md "Taller tiles and deeper rings"; the code is in git history (commit "Experiment:
// 256-row / 256x256 workgroup tiles ...").  So was a loader / consumer form (two or four extra waves per workgroup that only issue
// the LDS-DMA, the MFMA waves never touching global memory in the K loop, two stages, one barrier per K tile): bit-identical,
// 1.3-1.5 x slower (r05_gemm_presplit_loader_consumer_sweep.txt; commit "Experiment: loader / consumer ...").
// And the per-WAVE tile: waves of 128 x 64 (TI = 4: 512 B of LDS fragment reads per MFMA instead of 683) as 128 x 128 tiles of two
// waves or 256 x 128 of four: bit-identical, 10-50 % slower on every shape a step launches, equal on the largest launch
// (r05_gemm_presplit_wave_tile_sweep.txt).
template <bool HAS_BIAS, bool HAS_RES, int NSTAGE, int TI, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, NSTAGE == 1 ? 4 : 2) void gemm_f16x2p_kernel(GemmArgs g) {
  constexpr int NW = WM * WN;                       // waves per workgroup
  constexpr int BMT = 32 * TI * WM, BN = 64 * WN;   // workgroup tile
  constexpr int APL = BMT * BK;                     // f16 elements of one A plane tile
  constexpr int PLANE = BN * BK;                    // ... of one W plane tile
  constexpr int STG = 2 * APL + 2 * PLANE;          // one LDS stage: A hi, A lo, W hi, W lo
  constexpr int NPA = 2 * (BMT / 16);               // 1 KiB DMA pieces of the A planes
  constexpr int NPIECE = NPA + 2 * (BN / 16), PPW = NPIECE / NW;  // ... of a stage, per wave
  static_assert(NPA % NW == 0 && NPIECE % NW == 0, "a wave's DMA pieces of one round belong to one operand");
  static_assert(NSTAGE != 1 || BMT + BN <= 64 * NW, "scale staging: one thread per row / column of the tile");
  constexpr int SCL = NSTAGE == 1 ? 0 : 2 * ((BMT + BN + 63) / 64 * 64);  // (two-stage form: the scale table, f16 units)
  __shared__ __attribute__((aligned(1024))) _Float16 lds[NSTAGE * STG + SCL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int tiles_n = (g.n + BN - 1) / BN;
  const int tiles_m = (g.m + BMT - 1) / BMT;
  int tm_idx, tn_idx;
  gemm_tile_of(g, blockIdx.x, tiles_m, tiles_n, tm_idx, tn_idx);  // XCD-aware tile order (common.h)
  const int m0 = tm_idx * BMT, n0 = tn_idx * BN;
  // the tile's inverse row / column scales, requested before the first K tile and in LDS from its barrier on: the epilogue
  // used to request them only after the last MFMA -- a memory round trip at the end of every workgroup, and on grids of at
  // most one round of workgroups nothing overlaps an epilogue (skeleton of this kernel, 6144 x 2048 x 512: K loop 32 us,
  // + C stores 5, + the scale loads at the end 5; scripts/gemm_skeleton.cpp)
  // (one-stage instances, 128 VGPRs: through 1 KB of LDS; two-stage instances have 256 VGPRs and take them into registers --
  // a second LDS object with ordinary stores makes hipcc wait for the in-flight tile DMA before every fragment read, which
  // serialises the two stages: +16 ... +34 % on the small grids they serve)
  __shared__ float scl[NSTAGE == 1 ? BMT + BN : 1];
  if (NSTAGE == 1) {
    const int t = threadIdx.x;
    if (t < BMT) scl[t] = g.a2_inv[min(m0 + t, g.m - 1)];
    else if (t < BMT + BN) scl[t] = g.w_inv[min(n0 + t - BMT, g.n - 1)];
  }

  // ---- DMA sources: pieces of 1 KiB per stage (2 A planes x BMT / 16 row blocks of 16 rows, then 2 W planes x BN / 16), PPW per
  // wave.  Wave w takes pieces w, w + NW, ...  Lane l fills linear position (row = 16 rb + l / 4,
  // chunk position l & 3) and fetches logical chunk (l & 3) ^ ((row >> 2) & 3) of that row.  Rows beyond M / N are clamped
  // (their products land in rows / columns the epilogue drops).
  // BUFFER form of the LDS-DMA (round 5): one buffer resource per operand, based at the workgroup's first row; the per-lane byte
  // offsets are loop-invariant 32-bit registers, the K tile's offset rides in the instruction's SCALAR offset.  The global form
  // (`global_load_lds_dwordx4` on a 64-bit per-lane address + k0) cost the issuing wave ~220 ticks per instruction -- as long as the
  // arithmetic it feeds; this form costs half (scripts/gemm_phases.cpp, profiles/r05_gemm_presplit_phase_profile.txt: 24576 x
  // 2048 x 512 160 -> 140 us, Mamba in_proj 102 -> 87).  Offsets stay below 2^31: launch_gemm_f16x2p checks the plane sizes.
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(g.a2) + (int64_t)m0 * 32, 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(g.w2) + (int64_t)n0 * 32, 0, 0xffffffffu, 0x00020000);
#endif
  unsigned voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave + NW * i;
    const bool is_a = p < NPA;
    const int q = is_a ? p : p - NPA;                 // piece within its operand
    const int rbs = is_a ? BMT / 16 : BN / 16;        // row blocks per plane
    const int plane = q / rbs, rb = q % rbs;
    const int row = 16 * rb + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const int lrow = is_a ? min(m0 + row, g.m - 1) - m0 : min(n0 + row, g.n - 1) - n0;   // row relative to the resource's base
    voff[i] = (unsigned)(((int64_t)plane * (is_a ? g.a2_plane : g.w2_plane) + (int64_t)lrow * 32 + 8 * chunk) * 2);
  }
  const unsigned kt_bytes_a = (unsigned)(g.a2_kt * 2), kt_bytes_w = (unsigned)(g.w2_kt * 2);
  auto dma_tile = [&](int stage, int kt) {  // (K-tile-major planes: the tile's rows of a plane are one contiguous run)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + NW * i;
      const bool is_a = NW * i < NPA;   // (pieces NW i .. NW i + NW - 1 belong to one operand)
      // (LDS destination: wave-uniform base of the piece -- pieces lie in the stage in piece order: A hi, A lo, W hi, W lo;
      // the hardware adds lane * 16)
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? rsrc_a : rsrc_w, (__attribute__((address_space(3))) void*)(lds + stage * STG + p * 512), 16,
                                               voff[i], (unsigned)kt * (is_a ? kt_bytes_a : kt_bytes_w), 0, 0);
#endif
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
  const int sw = (li >> 2) & 3;  // chunk swizzle of this lane's rows (tile row offsets are multiples of 32)
  // Two-stage instances: the tile's BMT + BN inverse scales go into LDS by DMA too (4 bytes per lane, behind the tile stages
  // of the same array; the first barrier of the K loop publishes them), read by the epilogue.  Round 5: they used to sit in
  // 32 + 2 registers through the K loop (181 VGPRs) -- one register too many for a workgroup to start beside two read-pass
  // workgroups of the 206M geometry (2 x 177), so the projection of one env slice WAITED for the other slice's read pass to
  // retire workgroups instead of running under it.
  const float* sclf = reinterpret_cast<const float*>(lds + NSTAGE * STG);
  if (NSTAGE != 1) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc_ai = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a2_inv), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_wi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w_inv), 0, 0xffffffffu, 0x00020000);
#pragma unroll
    for (int c = wave; c < (BMT + BN + 63) / 64; c += NW) {  // chunk c: table entries 64 c .. 64 c + 63 (BMT is a multiple of 64)
      const bool is_a = 64 * c < BMT;
      const int e = 64 * c + lane - (is_a ? 0 : BMT);
      const unsigned off = 4u * (unsigned)(is_a ? min(m0 + e, g.m - 1) : min(n0 + e, g.n - 1));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? rsrc_ai : rsrc_wi,
                                               (__attribute__((address_space(3))) void*)(lds + NSTAGE * STG + 128 * c), 4, off, 0, 0, 0);
    }
#endif
  }
  const _Float16* a_base = lds + (32 * TI * wm + li) * BK;
  const _Float16* b_base = lds + 2 * APL + (64 * wn + li) * BK;

  const int nk_all = g.k / BK;
  const int kt0 = g.split_k > 1 ? blockIdx.z * g.k_tiles_per_split : 0;
  const int nk = g.split_k > 1 ? min(nk_all, kt0 + g.k_tiles_per_split) : nk_all;

  auto mfma_tile = [&](int stage) {
    if (g.mfma_prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int ko = ((2 * ks + lh) ^ sw) << 3;
      f16x8 af[TI][2], bf[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (t < TI) af[t][p] = *reinterpret_cast<const f16x8*>(a_base + stage * STG + p * APL + 32 * t * BK + ko);
          bf[t][p] = *reinterpret_cast<const f16x8*>(b_base + stage * STG + p * PLANE + 32 * t * BK + ko);
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

  if (NSTAGE == 1 && kt0 >= nk) __syncthreads();  // (an empty K range: the scales are still read by other threads below)
  if (NSTAGE == 2) {
    // tile kt sits in stage kt & 1 once the barrier at the top of its iteration is passed (__syncthreads drains the issuing
    // waves' DMAs: an LDS-DMA is a pending LDS write on the VM counter); the DMA of tile kt + 1 is issued right behind that
    // barrier -- every wave has then finished reading that stage (tile kt - 1) -- and is in flight under tile kt's MFMAs
    if (kt0 < nk) dma_tile(0, kt0);  // (an empty K split must not read past the operand planes)
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = (kt - kt0) & 1;
      __syncthreads();
      if (kt + 1 < nk) dma_tile(cur ^ 1, kt + 1);
      mfma_tile(cur);
    }
  } else {
    for (int kt = kt0; kt < nk; ++kt) {
      dma_tile(0, kt);
      __syncthreads();
      mfma_tile(0);
      __syncthreads();
    }
  }

  // epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)): a lane holds ONE
  // column of 16 rows, so storing from the accumulators is 16 single-dword stores per tile and lane, two 128-byte row pieces each --
  // store-issue-bound (~7 B / clock / CU: measured on the 256 x 256 kernel, gemm_f16x2_8p.hip, where it was 21-24 us per workgroup),
  // and on grids of one round of workgroups every workgroup stores at the same time with nothing left to overlap it.  Round 6: each
  // wave turns its tile through the operand stages (free once every wave has left the K loop) -- ds_write_b32 in [row][col]
  // order, ds_read_b128 of whole 256-byte rows, float4 un-scale + bias + residual + activation, dwordx4 stores.  Same arithmetic
  // per element: acc * (w_inv * a_inv) + bias, + residual, silu.  Row pitches / column counts that are not multiples of 4 keep
  // the accumulator-order stores.
  float* C = g.c;
  float* S = g.split_k > 1 ? g.splitk_ws + (int64_t)blockIdx.z * g.m * g.n : nullptr;
  typedef float __attribute__((may_alias)) lds_f32;
  typedef float4 __attribute__((may_alias)) lds_f32x4;
  const int64_t ld_out = S != nullptr ? (int64_t)g.n : g.ldc;
  float* outp = S != nullptr ? S : C;
  const bool vec = (ld_out & 3) == 0 && (g.n & 3) == 0 && (reinterpret_cast<uintptr_t>(outp) & 15) == 0 &&
                   (!HAS_BIAS || (reinterpret_cast<uintptr_t>(g.bias + n0) & 15) == 0) &&
                   (!HAS_RES || (reinterpret_cast<uintptr_t>(g.residual) & 15) == 0);
  constexpr int PR = (NSTAGE * STG * 2 >= NW * 32 * 64 * 4) ? 32 : 16;   // rows a wave stages per pass (its share of the stages)
  if (vec) {
    if (NSTAGE == 2) __syncthreads();   // (the one-stage loop ends on a barrier; here the last tile's fragment reads are still under way)
    lds_f32* stg = reinterpret_cast<lds_f32*>(lds) + wave * (PR * 64);
    const int c4 = (lane & 15) * 4, rq = lane >> 4;
    const int gcol = n0 + 64 * wn + c4;
    const bool col_in = gcol < g.n;   // (n is a multiple of 4: a float4 is inside or outside as a whole)
    float4 wi4 = make_float4(0.f, 0.f, 0.f, 0.f), bv4 = wi4;
    if (col_in) {
      const int cl = BMT + gcol - n0;
      wi4 = make_float4(NSTAGE == 1 ? scl[cl] : sclf[cl], NSTAGE == 1 ? scl[cl + 1] : sclf[cl + 1],
                        NSTAGE == 1 ? scl[cl + 2] : sclf[cl + 2], NSTAGE == 1 ? scl[cl + 3] : sclf[cl + 3]);
      if (HAS_BIAS && S == nullptr) bv4 = *reinterpret_cast<const float4*>(g.bias + gcol);
    }
    const int act_from = S == nullptr ? g.act_silu_from : -1;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int h = 0; h < 32 / PR; ++h) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2);            // + 4 lh: row inside the 32-row tile
            if (PR == 32 || (ro >> 4) == h) stg[((ro & (PR - 1)) + 4 * lh) * 64 + 32 * j + li] = acc[i][j][r];
          }
        const int rl0 = 32 * TI * wm + 32 * i + PR * h + rq;   // row inside the workgroup tile
#pragma unroll
        for (int it = 0; it < PR / 4; ++it) {
          const int rl = rl0 + 4 * it, grow = m0 + rl;
          const float4 t = *reinterpret_cast<const lds_f32x4*>(stg + (4 * it + rq) * 64 + c4);
          if (grow < g.m && col_in) {
            const float ai = NSTAGE == 1 ? scl[rl] : sclf[rl];
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
            *reinterpret_cast<float4*>(outp + (int64_t)grow * ld_out + gcol) = v;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    float ainv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * TI * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
      ainv[r] = NSTAGE == 1 ? scl[row - m0] : sclf[row - m0];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 64 * wn + 32 * j + li;
      if (col >= g.n) continue;
      const float wi = NSTAGE == 1 ? scl[BMT + col - n0] : sclf[BMT + col - n0];
      const float bv = HAS_BIAS ? g.bias[col] : 0.f;
      // (per (i, j) one base pointer; the 16 rows of the accumulator tile are compile-time multiples of the row pitch from it --
      // the per-element 64-bit row * pitch products, bounds checks and libm SiLU of the first form were ~60 instructions per
      // stored value, as many in the epilogue as in the whole K loop)
      const int row0 = m0 + 32 * TI * wm + 32 * i + 4 * lh;
      const bool act = g.act_silu_from >= 0 && col >= g.act_silu_from;
      const bool rows_in = row0 + 27 < g.m;  // (uniform but for tiles on the lower edge)
      if (S != nullptr) {  // raw partial sums into this split's slab [M][N]; bias / residual are applied by the reduce
        float* sp = S + (int64_t)row0 * g.n + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          if (rows_in || row0 + ro < g.m) sp[(int64_t)ro * g.n] = acc[i][j][r] * (wi * ainv[r]);
        }
        continue;
      }
      float* cp = C + (int64_t)row0 * g.ldc + col;
      const float* rp = HAS_RES ? g.residual + (int64_t)row0 * g.ldc + col : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
        if (rows_in || row0 + ro < g.m) {
          float v = acc[i][j][r] * (wi * ainv[r]) + bv;
          if (HAS_RES) v += rp[(int64_t)ro * g.ldc];
          if (act) v = silu_hw(v);
          cp[(int64_t)ro * g.ldc] = v;
        }
      }
    }
  }
}

// One wave per row (K <= 3072): planes[0] = hi, planes[1] = lo of scale * a[r][k] (* gate[r][k]), K-tile-major ((r, k) at
// (k / 32) * kt + r * 32 + k % 32); inv[r] = 1 / scale (exact power of two).  The stand-alone form of what the norm kernels
// do in their epilogue.
constexpr int kSplitMaxV = 12;
__global__ __launch_bounds__(256) void row_split_f16x2_kernel(const float* a, int64_t lda, const float* gate, int64_t ldg, int rows,
                                                              int k, _Float16* planes, int64_t kt, int64_t plane, float* inv) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int nv = k >> 2;
  const float* ar = a + (int64_t)row * lda;
  const float* gr = gate != nullptr ? gate + (int64_t)row * ldg : nullptr;
  float4 v[kSplitMaxV];
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < kSplitMaxV; ++j) {
    const int i = lane + 64 * j;
    v[j] = i < nv ? *reinterpret_cast<const float4*>(ar + 4 * i) : f4_zero();
    if (gr != nullptr && i < nv) {
      const float4 z = *reinterpret_cast<const float4*>(gr + 4 * i);
      v[j].x *= z.x, v[j].y *= z.y, v[j].z *= z.z, v[j].w *= z.w;
    }
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
  }
  const float s = pow2_scale(wave_max(mx));
#pragma unroll
  for (int j = 0; j < kSplitMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) split2_store4(v[j], s, planes + (int64_t)(i >> 3) * kt + (int64_t)row * 32 + 4 * (i & 7), plane);
  }
  if (lane == 0) inv[row] = 1.f / s;
}
}  // namespace

bool gemm_f16x2p_supported(const GemmArgs& g) {
  // (byte offsets of the buffer-form LDS-DMA are 32-bit: both planes of an operand within 2 GiB of its first row)
  if (2 * g.a2_plane * 2 >= (1ll << 31) || 2 * g.w2_plane * 2 >= (1ll << 31)) return false;
  return g.a2 != nullptr && g.a2_inv != nullptr && g.w2 != nullptr && g.w_inv != nullptr && g.nb1 * g.nb2 == 1 &&
         (g.k % BK) == 0 && g.w2_kt >= 32 * (int64_t)g.n && g.a2_kt >= 32 * (int64_t)g.m && (g.w2_plane & 7) == 0 && (g.a2_plane & 7) == 0 &&
         (g.w2_kt & 7) == 0 && (g.a2_kt & 7) == 0 &&
         g.gate == nullptr && (reinterpret_cast<uintptr_t>(g.a2) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(g.w2) & 15) == 0;
}

void launch_row_split_f16x2(const float* a, int64_t lda, const float* gate, int64_t ldg, int rows, int k, uint16_t* planes,
                            int64_t kt, int64_t plane, float* inv, hipStream_t stream) {
  LRAM_REQUIRE((k & 3) == 0 && k <= 4 * 64 * kSplitMaxV && (lda & 3) == 0 && kt >= 32 * (int64_t)rows && (gate == nullptr || (ldg & 3) == 0),
               "row split: K must be a multiple of 4, <= 3072; row pitches multiples of 4; K-tile pitch >= 32 x rows");
  hipLaunchKernelGGL(row_split_f16x2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a, lda, gate, ldg, rows, k,
                     reinterpret_cast<_Float16*>(planes), kt, plane, inv);
  LRAM_HIP_CHECK(hipGetLastError());
}

template <int NSTAGE, int TI, int WM, int WN>
static void launch_stage(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  const bool hb = g.bias != nullptr, hr = g.residual != nullptr;
  dim3 block(64 * WM * WN);
  if (hb && hr)
    hipLaunchKernelGGL((gemm_f16x2p_kernel<true, true, NSTAGE, TI, WM, WN>), grid, block, 0, stream, g);
  else if (hb)
    hipLaunchKernelGGL((gemm_f16x2p_kernel<true, false, NSTAGE, TI, WM, WN>), grid, block, 0, stream, g);
  else if (hr)
    hipLaunchKernelGGL((gemm_f16x2p_kernel<false, true, NSTAGE, TI, WM, WN>), grid, block, 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f16x2p_kernel<false, false, NSTAGE, TI, WM, WN>), grid, block, 0, stream, g);
}

// Workgroup tile of a launch.  64 x 128 where 128-row tiles would leave CUs without a workgroup AND K is short (16M at 1024
// slots: proj_up's 1536 x 1024 x 512 halves are 96 tiles of 128 x 128 -- 26.0 us -- or 192 of 64 x 128 -- 17.8 us); long-K
// launches keep 128 rows (inside the two-slice pipeline the small tile's 1.5 x operand traffic per flop costs Mamba-48M 2.3 %
// and the 206M stack 1.5 %: profiles/r04_ab_tile_height.txt).  LRAM_GEMM_TILE (measurement knob): 64 / 128 force one.
int gemm_f16x2p_tile(const GemmArgs& g, int S) {
  const int force = gemm_knobs().tile;  // (cached: common.h GemmKnobs)
  if (force == 64 || force == 128) return force;
  const long tiles128 = (long)((g.m + 127) / 128) * ((g.n + 127) / 128);
  if (g.m > 64 && ((tiles128 * S < 256 && g.k <= 768) || tiles128 * S < 128)) return 64;
  return 128;
}

void launch_gemm_f16x2p(const GemmArgs& g_in, hipStream_t stream) {
  GemmArgs g = g_in;
  // LRAM_F16P_STAGES (measurement knob): 1 = one LDS stage, two barriers per K tile, up to four workgroups per CU;
  // 2 = two stages, the next tile's DMA under the current tile's MFMAs, one barrier per K tile, two workgroups per CU
  // default 0 = by grid size (same box, standalone: 16M proj_up 768 tiles 55 us with one stage / 65 with two; Mamba in_proj 576
  // tiles 67 / 80; 16M proj_down 192 tiles 50 / 40; Mamba out_proj 144 tiles 65 / 51 -- profiles/r04_gemm_f16x2p_durations.txt)
  const int stages_env = gemm_knobs().stages;
  g.mfma_prio = 1;
  LRAM_REQUIRE(g.m > 0 && g.n > 0 && g.k > 0, "gemm: empty problem");
  LRAM_REQUIRE(gemm_f16x2p_supported(g), "gemm f16x2 (pre-split operands): unsupported operand layout");
  // The 8-phase 256 x 256 kernel (gemm_f16x2_8p.hip; bit-identical results) where it is faster: launches of several rounds of
  // 256 x 256 tiles with a deep K -- one workgroup per CU exposes its prologue and its 256 KB epilogue (2 + 10 us per round,
  // HBM-write-bound when every CU stores at once), so it wins from ~4 rounds on: C5's proj_up 16128 x 5120 x 1280 567 -> 509 us,
  // 24576 x 2048 x 512 155 -> 147; everything a step launches is 0.6-1.0 x (profiles/r06_gemm_8phase_durations.txt).
  // LRAM_GEMM_TILE=256 forces it (tests, measurements), -1 keeps it off.
  {
    const int force = gemm_knobs().tile;
    const long t256 = (long)((g.m + 255) / 256) * ((g.n + 255) / 256);
    const bool big = t256 >= 700 && g.k >= 512 && g.n >= 1024;
    // ... and where the caller says the launch shares the chip with another slice's memory-bound kernels and covers 0.25-0.63 of
    // the CUs as 256 x 256 tiles (Mamba's in_proj in the two-slice pipeline: 1024 slots +1 %, 2048 +2.8 %; 4096 slots = 288
    // tiles -5 %, hence the upper limit)
    // (2: a chunk of lram_prefill beside the other chunk lanes' state passes -- 206M, 64 envs x 63 tokens = 320 tiles: 308 -> 304 ms per prefill)
    const bool beside = g.beside_memory_bound != 0 && t256 >= 64 && t256 <= (g.beside_memory_bound == 2 ? 400 : 160) && g.k >= 512;
    if (force == 256 || (force == 0 && (big || beside))) {
      launch_gemm_f16x2_8p(g_in, stream);
      return;
    }
  }
  int S = 1;
  if (g.act_silu_from >= 0)
    g.split_k = 1, g.k_tiles_per_split = 0;  // output activation: K unsplit
  else
    S = gemm_choose_split_k(g);
  // (the split-K chooser counts K tiles of its own BK: an empty split would read past the operand planes)
  LRAM_REQUIRE(S == 1 || (int64_t)(S - 1) * g.k_tiles_per_split < g.k / BK, "gemm f16x2 (pre-split operands): empty K split");
  const int bm = gemm_f16x2p_tile(g, S), bn = 128;
  const int tiles = ((g.m + bm - 1) / bm) * ((g.n + bn - 1) / bn);
  dim3 grid(tiles, 1, S);
  gemm_choose_xcd_split(g, bm, bn, 4);
  const int stages = stages_env == 1 || stages_env == 2 ? stages_env : ((long)tiles * S >= 384 ? 1 : 2);
  if (bm == 64) {
    if (stages == 1) launch_stage<1, 1, 2, 2>(g, grid, stream); else launch_stage<2, 1, 2, 2>(g, grid, stream);
  } else {
    if (stages == 1) launch_stage<1, 2, 2, 2>(g, grid, stream); else launch_stage<2, 2, 2, 2>(g, grid, stream);
  }
  LRAM_HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce(g, stream);
}

}  // namespace lram
