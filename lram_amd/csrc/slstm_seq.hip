// sLSTM recurrence of a whole env-step (T tokens) as ONE launch for large env slices (head dims 128 / 192 / 256 / 320 / 384) on gfx950.
//
// The generic path runs, per token, a batched per-head GEMM  ry = R_g h_{t-1}  (bf16x3 tile kernel, [envs, 4 gates, H]
// through HBM) and the pointwise cell kernel after it: 2 T = 6 dependent launches of 25-90 us each beside a read pass,
// ~300 us of a 2048-env slice's chain during which the state-pass queue has nothing but folds to run
// (profiles/r03_step_timeline_xlstm16m_b4096_verbose.txt, 1.2 ms "sLSTM stretch" of a 9.7 ms step).  slstm_token_kernel
// (xlstm_kernels.hip) fuses one token for slices of <= 512 envs; this kernel does the whole step for any slice size:
//
//   * one workgroup = 32 envs x one head; the head's recurrence is closed over its own 128 channels (R is block-diagonal
//     per head: [3P] sLSTMCell, `_recurrent_kernel_` (head, in, gate, out)), so the T tokens run back to back INSIDE the
//     workgroup with h_t handed over in LDS -- no launch boundary, no [envs, 4 H] round trip;
//   * R_g h on the exact fp32 matrix instruction (v_mfma_f32_32x32x2_f32): wave w owns channels 32 w .. 32 w + 31 of the head,
//     its four 32 x 32 accumulator tiles are the four GATES of those channels -- the B operand is read from a copy of R
//     re-packed [head][k][channel][gate] (one float4 per lane and k step, 512 contiguous bytes per half wave, L2-resident:
//     256 KB per head), so accumulator column `lane & 31` of tile g is gate g of the lane's channel and the pointwise cell
//     ([3P] slstm_pointwise: per-element n == 0 first-step rule) is lane-local: no exchange of gate sums at all;
//   * the cell state (c, n, m) of a lane's 16 (env, channel) pairs lives in registers across the T tokens and is written
//     back once; h_t goes to LDS (next token's A operand) and to the output rows.
// The k index of an MFMA step is split over the lane halves as k = (SDH / 2) (lane >> 5) + j (j = 0 .. SDH / 2 - 1): each lane reads its A
// operands as float4 runs of its env's h row in LDS.
// Round 5: templated on the head dim (SDH / 32 waves per workgroup: 4 at 128 -- the 16M model --, 10 at 320 -- the 206M model's
// sLSTM blocks --; from nine waves the kernel is built for three waves per SIMD, 168 registers); 448 (xlstm_huge_half) would need
// fourteen waves at 128 registers and keeps the per-token path.
//
// Arithmetic: products and sums in fp32 exactly as an fma chain per lane half, the two halves added inside the matrix
// instruction; the gate pre-activations differ from the GEMM path's (bf16x3, different summation order) by fp32 rounding.
// Reference call site: xLSTMBlockStack.step -> sLSTMLayer.step (src/algos/models/decision_xlstm.py:155-166).
#include <algorithm>

#include "common.h"
#include "device_math.h"

namespace lram {

typedef float sq_f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kEnv = 32;

// rt2[((head * SDH + k) * SDH + ch) * 4 + g] = rt[((head * 4 + g) * SDH + ch) * SDH + k]     (rt: [NH, 4, out, in])
__global__ __launch_bounds__(256) void slstm_pack_rt_kernel(const float* rt, float* rt2, int NH, int SDH) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)NH * 4 * SDH * SDH;
  if (i >= n) return;
  const int g = (int)(i & 3);
  const int ch = (int)((i >> 2) % SDH);
  const int k = (int)((i >> 2) / SDH % SDH);
  const int head = (int)(i / (4 * (int64_t)SDH * SDH));
  rt2[i] = rt[(((int64_t)head * 4 + g) * SDH + ch) * SDH + k];
}

template <int T, int kSDH>
__global__ __launch_bounds__(2 * kSDH, kSDH <= 256 ? 2 : 3) void slstm_seq_kernel(SlstmSeqArgs a) {
  constexpr int kPitch = kSDH + 4, kHalf = kSDH / 2;   // K per lane half
  static_assert(kSDH % 32 == 0 && kHalf % 8 == 0, "head dim: a multiple of 32 whose half is a multiple of the 8-deep product round");
  __shared__ __attribute__((aligned(16))) float hs[2][kEnv][kPitch];
  const int H = a.H;
  const int head = blockIdx.x % a.NH, b0 = (blockIdx.x / a.NH) * kEnv;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int ch = head * kSDH + 32 * w + li;  // this lane's channel (column of H)
  const int64_t BH = (int64_t)a.state_B * H;

  // ---- h_{-1} of the workgroup's envs and head -> LDS (32 envs x SDH / 4 float4) ----
  for (int idx = tid; idx < kEnv * (kSDH / 4); idx += 2 * kSDH) {
    const int e = idx / (kSDH / 4), c4 = 4 * (idx % (kSDH / 4));
    const int b = min(b0 + e, a.B - 1);
    *reinterpret_cast<float4*>(&hs[0][e][c4]) = *reinterpret_cast<const float4*>(a.state + (int64_t)b * H + head * kSDH + c4);
  }
  // ---- this lane's cells: envs e_r = (r & 3) + 8 (r >> 2) + 4 lh, r = 0 .. 15 (the accumulator rows it holds) ----
  float cs[16], ns[16], ms[16];
  auto env_of = [&](int r) { return min(b0 + (r & 3) + 8 * (r >> 2) + 4 * lh, a.B - 1); };  // (clamped: rows beyond B are dropped at the stores)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* st = a.state + (int64_t)env_of(r) * H + ch;
    cs[r] = st[BH], ns[r] = st[2 * BH], ms[r] = st[3 * BH];
  }
  const float bi = a.bias[ch], bf = a.bias[H + ch], bz = a.bias[2 * H + ch], bo = a.bias[3 * H + ch];
  const float* bp = a.rt2 + (((int64_t)head * kSDH + kHalf * lh) * kSDH + 32 * w + li) * 4;  // k = kHalf lh + j: + j * SDH * 4
  __syncthreads();

  // (the token loop stays a loop and the kernel is built for two waves per SIMD -- 256 registers, accumulators included: fully
  // unrolled it took 469 and a workgroup had a CU's register file to itself, stalling the folds and the other slice's chain
  // beside it: 109 us alone, 389 us in the pipeline, profiles/r04_ab_slstm_seq.txt)
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    sq_f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    const float* ap = &hs[cur][li][kHalf * lh];
    // B operands: 4 k steps (4 float4) in flight ahead of the products that consume them, two register sets, the loop
    // NOT unrolled further (unrolled, hipcc hoists all 64 float4 of the token above the first product: 256 registers)
    constexpr int PF = 4;
    float4 bq0[PF], bq1[PF];
    auto load_b = [&](float4* dst, int j0) {
#pragma unroll
      for (int jj = 0; jj < PF; ++jj) dst[jj] = *reinterpret_cast<const float4*>(bp + (unsigned)((j0 + jj) * kSDH * 4));
    };
    auto products = [&](const float4* bq, int j0) {
      const float4 a0 = *reinterpret_cast<const float4*>(ap + j0);
      const float av[PF] = {a0.x, a0.y, a0.z, a0.w};
#pragma unroll
      for (int jj = 0; jj < PF; ++jj) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bq[jj].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bq[jj].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bq[jj].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[jj], bq[jj].w, acc[3], 0, 0, 0);
      }
    };
    load_b(bq0, 0);
#pragma unroll 1
    for (int j0 = 0; j0 < kHalf; j0 += 2 * PF) {
      load_b(bq1, j0 + PF);
      products(bq0, j0);
      load_b(bq0, min(j0 + 2 * PF, kHalf - PF));   // (unconditional: past the end it re-requests the last set and drops it)
      products(bq1, j0 + PF);
    }
    // pointwise cell ([3P] slstm_pointwise), lane-local: accumulator row r of tile g = gate g of (env e_r, this channel);
    // the token's input pre-activations (Wx, from the gate projections) are requested four envs at a time
    // (two cells at a time behind a scheduling fence, 32-bit offsets from uniform bases: left alone, hipcc hoists all 64
    // requests of the token and interleaves the 16 cells' exp / log / tanh expansions -- 480 registers' worth of live values)
    const unsigned grow = 4u * (unsigned)H, ycol = (unsigned)ch;
#pragma unroll
    for (int r2 = 0; r2 < 8; ++r2) {
      __builtin_amdgcn_sched_barrier(0);
      float gi[2], gf[2], gz[2], go[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned off = ((unsigned)env_of(2 * r2 + q) * (unsigned)T + (unsigned)t) * grow + ycol;
        gi[q] = a.gates[off], gf[q] = a.gates[off + H], gz[q] = a.gates[off + 2 * H], go[q] = a.gates[off + 3 * H];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = 2 * r2 + q;
        const int e = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float iraw = gi[q] + acc[0][r] + bi, fraw = gf[q] + acc[1][r] + bf;
        const float zraw = gz[q] + acc[2][r] + bz, oraw = go[q] + acc[3][r] + bo;
        const float logfplusm = ms[r] + log_sigmoid(fraw);
        const float mnew = (ns[r] == 0.f) ? iraw : fmaxf(iraw, logfplusm);
        const float ogate = sigmoid_f(oraw);
        const float igate = fminf(expf(iraw - mnew), 1.f);
        const float fgate = fminf(expf(logfplusm - mnew), 1.f);
        const float cnew = fgate * cs[r] + igate * tanhf(zraw);
        const float nnew = fgate * ns[r] + igate;
        const float ynew = ogate * cnew / nnew;
        cs[r] = cnew, ns[r] = nnew, ms[r] = mnew;
        hs[cur ^ 1][e][32 * w + li] = ynew;
        if (b0 + e < a.B) {
          a.yout[((unsigned)(b0 + e) * (unsigned)T + (unsigned)t) * (unsigned)H + ycol] = ynew;
          if (t == T - 1) a.state[(unsigned)(b0 + e) * (unsigned)H + ycol] = ynew;   // the state's h plane: the step's last h
        }
      }
    }
    __syncthreads();  // h_t complete in hs[cur ^ 1]; every wave is done reading hs[cur]
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int e = (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (b0 + e < a.B) {
      float* st = a.state + (int64_t)(b0 + e) * H + ch;
      st[BH] = cs[r], st[2 * BH] = ns[r], st[3 * BH] = ms[r];
    }
  }
}

// =============================================================================================
// f16x2 form (round 5): R_g h as hi * hi + hi * lo + lo * hi on v_mfma_f32_16x16x32_f16, like the projection GEMMs.
//
// The fp32 form above spends 64 cycles per 32 x 32 x 2 product round -- 128 us of matrix-pipe time per workgroup at the 206M
// model's 320-wide heads -- and, with 32 envs per workgroup, runs a 256-env slice on 32 of the 256 CUs: 480 us per launch on
// each slice's chain, three times per step, while the state-pass queue has only folds to run.  Here:
//   * one workgroup = 16 envs x one head (twice the workgroups), wave w owns channels 32 w .. 32 w + 31 as 2 column tiles x 4 gates
//     of 16 x 16 accumulators (32 registers); a lane's 8 (env, channel) cells keep all four gates lane-local as before;
//   * h_t lives in LDS as two f16 planes of 2^12 h (|h| < 1: the scale is fixed, no row maxima), R is split at upload into two
//     f16 planes per (gate, channel) row scaled by its power of two, laid out so that a wave's B operands are ONE contiguous
//     stream ([head][wave][k step][column tile][gate][plane][lane][8]): 12 MFMAs of 16 cycles per 8 KB -- the kernel is bound by
//     streaming R from L2 (the same bytes as fp32), not by the matrix pipe.
// Differences from the fp32 form are those of the f16x2 GEMMs (22-bit operands, lo * lo dropped); LRAM_GEMM=f32 keeps the fp32 form.
// =============================================================================================
typedef float sq_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sq_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sq_f16x4 __attribute__((ext_vector_type(4)));
constexpr int kEnv16 = 16;
constexpr float kHScale = 4096.f;

// rt: [NH, 4, out, in] fp32 -> rt2h planes (layout above), rinv[(head * 4 + g) * SDH + ch] = 1 / (row scale * kHScale)
__global__ __launch_bounds__(256) void slstm_pack_rt16_kernel(const float* rt, _Float16* rt2h, float* rinv, int NH, int SDH) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // row = (head * 4 + g) * SDH + ch
  if (row >= NH * 4 * SDH) return;
  const int ch = row % SDH, g = (row / SDH) & 3, head = row / (4 * SDH);
  const float* r = rt + (int64_t)row * SDH;
  float mx = 0.f;
  for (int k = lane; k < SDH; k += 64) mx = fmaxf(mx, fabsf(r[k]));
  const float sc = pow2_scale(wave_max(mx));
  const int NW = SDH / 32, KS = SDH / 32;
  const int w = ch >> 5, ct = (ch >> 4) & 1, c = ch & 15;
  for (int k = lane; k < SDH; k += 64) {
    const float v = r[k] * sc;
    const _Float16 hi = (_Float16)v;
    const int ks = k >> 5, kg = (k >> 3) & 3, j = k & 7;
    const int64_t at = ((((((int64_t)head * NW + w) * KS + ks) * 2 + ct) * 4 + g) * 2) * 512 + (kg * 16 + c) * 8 + j;
    rt2h[at] = hi;
    rt2h[at + 512] = (_Float16)(v - (float)hi);
  }
  if (lane == 0) rinv[row] = 1.f / (sc * kHScale);
}

template <int T, int kSDH, bool GE>
__global__ __launch_bounds__(2 * kSDH, kSDH <= 128 ? 2 : 1) void slstm_seq16_kernel(SlstmSeqArgs a) {
  constexpr int KS = kSDH / 32, NU = 2 * KS, kPitch = kSDH + 8;   // K steps of 32, B units (k step, column tile), LDS row pitch (f16)
  __shared__ __attribute__((aligned(16))) _Float16 hs[2][2][kEnv16][kPitch];   // [buffer][plane][env][k]
  const int H = a.H;
  const int head = blockIdx.x % a.NH, b0 = (blockIdx.x / a.NH) * kEnv16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lc = lane & 15, lg = lane >> 4;
  const int64_t BH = (int64_t)a.state_B * H;

  // ---- h_{-1} of the workgroup's envs and head -> the two LDS planes ----
  for (int idx = tid; idx < kEnv16 * (kSDH / 4); idx += 2 * kSDH) {
    const int e = idx / (kSDH / 4), c4 = 4 * (idx % (kSDH / 4));
    const int b = min(b0 + e, a.B - 1);
    const float4 v = *reinterpret_cast<const float4*>(a.state + (int64_t)b * H + head * kSDH + c4);
    const float xs[4] = {v.x * kHScale, v.y * kHScale, v.z * kHScale, v.w * kHScale};
    sq_f16x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hi[q] = (_Float16)xs[q];
      lo[q] = (_Float16)(xs[q] - (float)hi[q]);
    }
    *reinterpret_cast<sq_f16x4*>(&hs[0][0][e][c4]) = hi;
    *reinterpret_cast<sq_f16x4*>(&hs[0][1][e][c4]) = lo;
  }
  // ---- this lane's cells: column tile ct, accumulator row r -> env 4 lg + r, channel 32 w + 16 ct + lc of the head ----
  auto env_of = [&](int r) { return min(b0 + 4 * lg + r, a.B - 1); };   // (clamped: rows beyond B are dropped at the stores)
  const int chl = 32 * w + lc;             // channel within the head of column tile 0 (+ 16 for tile 1)
  const int ch0 = head * kSDH + chl;       // ... within H
  float cs[2][4], ns[2][4], ms[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* st = a.state + (int64_t)env_of(r) * H + ch0 + 16 * ct;
      cs[ct][r] = st[BH], ns[ct][r] = st[2 * BH], ms[ct][r] = st[3 * BH];
    }
  float bias[2][4], ri[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bias[ct][g] = a.bias[g * H + ch0 + 16 * ct];
      ri[ct][g] = a.rinv[(head * 4 + g) * kSDH + chl + 16 * ct];
    }
  const _Float16* bp = reinterpret_cast<const _Float16*>(a.rt2h) + ((int64_t)(head * (kSDH / 32) + w) * KS) * (16 * 512) + lane * 8;
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    const int cur = t & 1;
    sq_f32x4 acc[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[ct][g] = (sq_f32x4)(0.f);
    // the token's input pre-activations (Wx): requested before the K loop where the register budget allows (GE), so that their
    // round trip hides behind it
    const unsigned grow = 4u * (unsigned)H;
    float gin[GE ? 2 : 1][GE ? 4 : 1][GE ? 4 : 1];
    (void)gin;
    if (GE) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned off = ((unsigned)env_of(r) * (unsigned)T + (unsigned)t) * grow + (unsigned)(ch0 + 16 * ct);
#pragma unroll
          for (int g = 0; g < 4; ++g) gin[GE ? ct : 0][GE ? r : 0][GE ? g : 0] = a.gates[off + g * H];
        }
    }
    // B units (k step, column tile): 8 x 1 KB (gate x plane) each, two register sets, one unit ahead
    sq_f16x8 bq0[8], bq1[8];
    auto load_b = [&](sq_f16x8* dst, int u) {
      const _Float16* q = bp + (unsigned)(u * 8 * 512);
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const sq_f16x8*>(q + i * 512);
    };
    auto products = [&](const sq_f16x8* bq, int ct, const sq_f16x8& ah, const sq_f16x8& al) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // smallest terms first
        acc[ct][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bq[2 * g], acc[ct][g], 0, 0, 0);      // lo * hi
        acc[ct][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[2 * g + 1], acc[ct][g], 0, 0, 0);  // hi * lo
        acc[ct][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[2 * g], acc[ct][g], 0, 0, 0);      // hi * hi
      }
    };
    load_b(bq0, 0);
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
      load_b(bq1, 2 * ks + 1);
      const sq_f16x8 ah = *reinterpret_cast<const sq_f16x8*>(&hs[cur][0][lc][32 * ks + 8 * lg]);
      const sq_f16x8 al = *reinterpret_cast<const sq_f16x8*>(&hs[cur][1][lc][32 * ks + 8 * lg]);
      products(bq0, 0, ah, al);
      load_b(bq0, min(2 * ks + 2, NU - 2));   // (unconditional: past the end it re-requests a unit and drops it)
      products(bq1, 1, ah, al);
    }
    // pointwise cell ([3P] slstm_pointwise), lane-local
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      if (!GE) __builtin_amdgcn_sched_barrier(0);
      float gl[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned off = ((unsigned)env_of(r) * (unsigned)T + (unsigned)t) * grow + (unsigned)(ch0 + 16 * ct);
#pragma unroll
        for (int g = 0; g < 4; ++g) gl[r][g] = GE ? gin[GE ? ct : 0][GE ? r : 0][GE ? g : 0] : a.gates[off + g * H];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * lg + r;
        const float iraw = gl[r][0] + acc[ct][0][r] * ri[ct][0] + bias[ct][0], fraw = gl[r][1] + acc[ct][1][r] * ri[ct][1] + bias[ct][1];
        const float zraw = gl[r][2] + acc[ct][2][r] * ri[ct][2] + bias[ct][2], oraw = gl[r][3] + acc[ct][3][r] * ri[ct][3] + bias[ct][3];
        const float logfplusm = ms[ct][r] + log_sigmoid(fraw);
        const float mnew = (ns[ct][r] == 0.f) ? iraw : fmaxf(iraw, logfplusm);
        const float ogate = sigmoid_f(oraw);
        const float igate = fminf(expf(iraw - mnew), 1.f);
        const float fgate = fminf(expf(logfplusm - mnew), 1.f);
        const float cnew = fgate * cs[ct][r] + igate * tanhf(zraw);
        const float nnew = fgate * ns[ct][r] + igate;
        const float ynew = ogate * cnew / nnew;
        cs[ct][r] = cnew, ns[ct][r] = nnew, ms[ct][r] = mnew;
        const float ysc = ynew * kHScale;
        const _Float16 yh = (_Float16)ysc;
        hs[cur ^ 1][0][e][chl + 16 * ct] = yh;
        hs[cur ^ 1][1][e][chl + 16 * ct] = (_Float16)(ysc - (float)yh);
        if (b0 + e < a.B) {
          const unsigned ycol = (unsigned)(ch0 + 16 * ct);
          a.yout[((unsigned)(b0 + e) * (unsigned)T + (unsigned)t) * (unsigned)H + ycol] = ynew;
          if (t == T - 1) a.state[(unsigned)(b0 + e) * (unsigned)H + ycol] = ynew;   // the state's h plane: the step's last h
        }
      }
    }
    __syncthreads();  // h_t complete in hs[cur ^ 1]; every wave is done reading hs[cur]
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = 4 * lg + r;
      if (b0 + e < a.B) {
        float* st = a.state + (int64_t)(b0 + e) * H + ch0 + 16 * ct;
        st[BH] = cs[ct][r], st[2 * BH] = ns[ct][r], st[3 * BH] = ms[ct][r];
      }
    }
}
}  // namespace

// flag != 0 when an element of h[n] is NOT inside (-limit, limit) -- NaN included (`!(|v| < limit)`; a maximum would drop NaNs)
__global__ __launch_bounds__(256) void slstm_h_range_kernel(const float* h, int64_t n, float limit, int* flag) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= !(fabsf(h[i]) < limit);
  if (bad) atomicOr(flag, 1);
}

void launch_slstm_h_range(const float* h, int64_t n, float limit, int* flag, hipStream_t stream) {
  const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(slstm_h_range_kernel, dim3(blocks), dim3(256), 0, stream, h, n, limit, flag);
  LRAM_HIP_CHECK(hipGetLastError());
}

bool slstm_seq_supported(int H, int NH, int T) {
  if (NH <= 0 || H % NH != 0 || T < 1 || T > 4) return false;
  const int sdh = H / NH;
  return sdh == 128 || sdh == 192 || sdh == 256 || sdh == 320 || sdh == 384;
}

void launch_slstm_pack_rt(const float* rt, float* rt2, int NH, int SDH, hipStream_t stream) {
  const int64_t n = (int64_t)NH * 4 * SDH * SDH;
  hipLaunchKernelGGL(slstm_pack_rt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rt, rt2, NH, SDH);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_slstm_pack_rt16(const float* rt, uint16_t* rt2h, float* rinv, int NH, int SDH, hipStream_t stream) {
  hipLaunchKernelGGL(slstm_pack_rt16_kernel, dim3((unsigned)((NH * 4 * SDH + 3) / 4)), dim3(256), 0, stream, rt,
                     reinterpret_cast<_Float16*>(rt2h), rinv, NH, SDH);
  LRAM_HIP_CHECK(hipGetLastError());
}

template <int SDH>
static void launch_seq_sdh(const SlstmSeqArgs& a, hipStream_t stream) {
  if (a.rt2h != nullptr) {   // f16x2 form: 16 envs per workgroup; the gate rows requested early where 8 waves or fewer share the registers
    constexpr bool GE = SDH <= 256;
    const dim3 grid((unsigned)(a.NH * ((a.B + kEnv16 - 1) / kEnv16))), block(2 * SDH);
    switch (a.T) {
      case 1: hipLaunchKernelGGL((slstm_seq16_kernel<1, SDH, GE>), grid, block, 0, stream, a); break;
      case 2: hipLaunchKernelGGL((slstm_seq16_kernel<2, SDH, GE>), grid, block, 0, stream, a); break;
      case 3: hipLaunchKernelGGL((slstm_seq16_kernel<3, SDH, GE>), grid, block, 0, stream, a); break;
      default: hipLaunchKernelGGL((slstm_seq16_kernel<4, SDH, GE>), grid, block, 0, stream, a); break;
    }
    return;
  }
  const dim3 grid((unsigned)(a.NH * ((a.B + kEnv - 1) / kEnv))), block(2 * SDH);
  switch (a.T) {
    case 1: hipLaunchKernelGGL((slstm_seq_kernel<1, SDH>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((slstm_seq_kernel<2, SDH>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((slstm_seq_kernel<3, SDH>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((slstm_seq_kernel<4, SDH>), grid, block, 0, stream, a); break;
  }
}

void launch_slstm_seq(const SlstmSeqArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(slstm_seq_supported(a.H, a.NH, a.T) && (a.rt2 != nullptr || (a.rt2h != nullptr && a.rinv != nullptr)),
               "sLSTM step kernel: head dim 128 / 192 / 256 / 320 / 384, 1..4 tokens");
  // (the kernel's row offsets into gates / yout / state are 32-bit: slices beyond ~700k envs at H = 512 would wrap)
  LRAM_REQUIRE((int64_t)a.B * a.T * 4 * a.H < (1ll << 32) && (int64_t)a.state_B * a.H < (1ll << 30),
               "sLSTM step kernel: slice too large for its 32-bit row offsets");
  switch (a.H / a.NH) {
    case 128: launch_seq_sdh<128>(a, stream); break;
    case 192: launch_seq_sdh<192>(a, stream); break;
    case 256: launch_seq_sdh<256>(a, stream); break;
    case 320: launch_seq_sdh<320>(a, stream); break;
    default: launch_seq_sdh<384>(a, stream); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
