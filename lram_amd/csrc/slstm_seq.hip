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
}  // namespace

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

template <int SDH>
static void launch_seq_sdh(const SlstmSeqArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.NH * ((a.B + kEnv - 1) / kEnv))), block(2 * SDH);
  switch (a.T) {
    case 1: hipLaunchKernelGGL((slstm_seq_kernel<1, SDH>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((slstm_seq_kernel<2, SDH>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((slstm_seq_kernel<3, SDH>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((slstm_seq_kernel<4, SDH>), grid, block, 0, stream, a); break;
  }
}

void launch_slstm_seq(const SlstmSeqArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(slstm_seq_supported(a.H, a.NH, a.T) && a.rt2 != nullptr, "sLSTM step kernel: head dim 128 / 192 / 256 / 320 / 384, 1..4 tokens");
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
