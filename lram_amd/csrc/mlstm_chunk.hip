// Chunkwise mLSTM for context prefill on gfx950: L <= 64 tokens of one env per state pass, on the fp32 matrix
// cores.
//
// The recurrence the step kernels apply token by token ([3P] recurrent_step_stabilized_simple, SURVEY.md 3.4;
// reference call site src/algos/models/decision_xlstm.py:138-169 with `chunkwise_step`, :110,158-159)
//     C_t = f_t C_{t-1} + i_t khat_t v_t^T ,   h_t = q_t^T C_t / den_t ,   khat = k / sqrt(DH)
// with the stabilised per-token factors f_t = exp(logsig(f~_t) + m_{t-1} - m_t) <= 1, i_t = exp(i~_t - m_t) <= 1
// unrolls over a chunk of T tokens into three dense contractions (all factors are products of numbers <= 1, so
// nothing can overflow and no extra stabiliser is needed):
//     A[t][s] = (f_{s+1} ... f_t) i_s (q_t . khat_s)          s <= t          intra-chunk weights
//     H       = diag(fcum) Q C_0 + A V                          fcum_t = f_0 ... f_t
//     C_T     = fcum_{T-1} C_0 + (diag(w) Khat)^T V             w_s = (f_{s+1} ... f_{T-1}) i_s
//     q_t.n_t = fcum_t (q_t . n_0) + sum_s A[t][s]              (the normaliser is the v == 1 column of C)
// The matrix memory is read and written once per chunk of up to 21 timesteps instead of once per 4 (the
// token-sequential kernels of xlstm_kernels.hip), and the work is `v_mfma_f32_32x32x2_f32` (exact fp32 products,
// fp32 accumulation) instead of VALU fma chains.  Same inputs, same state layout, results equal to the
// token-by-token kernels up to fp32 summation order (tests/test_gpu_parity.py::test_chunkwise_prefill_*).
//
// Three launches per block and chunk:
//   mlstm_pre_tok_kernel    (env, token)       conv / q,k,v / gate pre-activations, token-parallel
//   mlstm_chunk_scan_kernel (env, head)        gate scan, A (MFMA), denominators, n / m / conv state
//   mlstm_cell_chunk_kernel (env, head, 128 columns of C)   the three contractions above; HBM: C once in, once out
#include <algorithm>

#include "common.h"
#include "device_math.h"

namespace lram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

constexpr int kLp = kChunkMaxTokens;  // 64: padded chunk length (two 32-row MFMA tiles)
constexpr int kThreads = 256;
constexpr int kPreMaxGroups = 3;      // channel groups of 4 per thread: inner <= 3072

__device__ __forceinline__ float dot4(const float4& x, const float4& y) {
  return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
}

// row of the 32x32 MFMA accumulator held in register r by a lane of half lh
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// =============================================================================================
// Front end, token-parallel: one workgroup per (env, group of kTB consecutive tokens).  Same arithmetic (and the
// same summation order) as mlstm_pre_seq_kernel; the conv window of the first three tokens reaches back into
// conv_state.  Per-channel weights are loaded once per workgroup and reused for its kTB tokens.
// Writes q, k, v, xa rows and the raw gate pre-activations gates[row][head] = (i~, f~).
// =============================================================================================
constexpr int kTB = 4;

template <int NH>
__global__ __launch_bounds__(kThreads) void mlstm_pre_tok_kernel(MlstmPreArgs a) {
  __shared__ float red[4][kTB * 2 * NH];
  const int t0 = blockIdx.x * kTB, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = a.inner, T = a.T, ngroups = inner >> 2;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  float pi[kTB][NH], pf[kTB][NH];
#pragma unroll
  for (int j = 0; j < kTB; ++j)
#pragma unroll
    for (int h = 0; h < NH; ++h) pi[j][h] = pf[j][h] = 0.f;
#pragma unroll
  for (int g = 0; g < kPreMaxGroups; ++g) {
    const int cg = tid + g * kThreads;
    if (cg >= ngroups) continue;
    const int c0 = cg << 2;
    auto input_row = [&](int tt) -> float4 {  // x_mlstm of token tt (tt < 0: the conv state's tap 4 + tt)
      if (tt >= 0) return *reinterpret_cast<const float4*>(a.u + ((int64_t)b * T + tt) * 2 * inner + c0);
      return rs ? f4_zero() : *reinterpret_cast<const float4*>(a.conv_state + ((int64_t)b * 4 + 4 + tt) * inner + c0);
    };
    float4 win[4];
    win[1] = input_row(t0 - 3);
    win[2] = input_row(t0 - 2);
    win[3] = input_row(t0 - 1);
    float4 cw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cw[c] = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)(c0 + c) * 4);
    const float4 cb = *reinterpret_cast<const float4*>(a.conv_b + c0);
    const float4* wq = reinterpret_cast<const float4*>(a.wq + (int64_t)cg * 16);
    const float4* wk = reinterpret_cast<const float4*>(a.wk + (int64_t)cg * 16);
    const float4* wv = reinterpret_cast<const float4*>(a.wv + (int64_t)cg * 16);
    const float4 wq0 = wq[0], wq1 = wq[1], wq2 = wq[2], wq3 = wq[3];
    const float4 wk0 = wk[0], wk1 = wk[1], wk2 = wk[2], wk3 = wk[3];
    const float4 wv0 = wv[0], wv1 = wv[1], wv2 = wv[2], wv3 = wv[3];
    float4 qs[kTB], ks[kTB], vs[kTB];
#pragma unroll
    for (int j = 0; j < kTB; ++j) {
      const int t = t0 + j;
      qs[j] = f4_zero();
      ks[j] = f4_zero();
      vs[j] = f4_zero();
      if (t >= T) continue;
      const int64_t row = (int64_t)b * T + t;
      const float4 xm = input_row(t);
      win[0] = win[1];
      win[1] = win[2];
      win[2] = win[3];
      win[3] = xm;
      float4 y;
      y.x = win[0].x * cw[0].x + win[1].x * cw[0].y + win[2].x * cw[0].z + win[3].x * cw[0].w + cb.x;
      y.y = win[0].y * cw[1].x + win[1].y * cw[1].y + win[2].y * cw[1].z + win[3].y * cw[1].w + cb.y;
      y.z = win[0].z * cw[2].x + win[1].z * cw[2].y + win[2].z * cw[2].z + win[3].z * cw[2].w + cb.z;
      y.w = win[0].w * cw[3].x + win[1].w * cw[3].y + win[2].w * cw[3].z + win[3].w * cw[3].w + cb.w;
      const float4 xa = make_float4(silu_f(y.x), silu_f(y.y), silu_f(y.z), silu_f(y.w));
      const float4 q = make_float4(dot4(wq0, xa), dot4(wq1, xa), dot4(wq2, xa), dot4(wq3, xa));
      const float4 k = make_float4(dot4(wk0, xa), dot4(wk1, xa), dot4(wk2, xa), dot4(wk3, xa));
      const float4 v = make_float4(dot4(wv0, xm), dot4(wv1, xm), dot4(wv2, xm), dot4(wv3, xm));
      *reinterpret_cast<float4*>(a.q + row * inner + c0) = q;
      *reinterpret_cast<float4*>(a.k + row * inner + c0) = k;
      *reinterpret_cast<float4*>(a.v + row * inner + c0) = v;
      *reinterpret_cast<float4*>(a.xa + row * inner + c0) = xa;
      qs[j] = q;
      ks[j] = k;
      vs[j] = v;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float* wi = a.wi + (int64_t)h * 3 * inner + c0;
      const float* wf = a.wf + (int64_t)h * 3 * inner + c0;
      const float4 iq = *reinterpret_cast<const float4*>(wi), ik = *reinterpret_cast<const float4*>(wi + inner),
                   iv = *reinterpret_cast<const float4*>(wi + 2 * inner);
      const float4 fq = *reinterpret_cast<const float4*>(wf), fk = *reinterpret_cast<const float4*>(wf + inner),
                   fv = *reinterpret_cast<const float4*>(wf + 2 * inner);
#pragma unroll
      for (int j = 0; j < kTB; ++j) {
        pi[j][h] += dot4(iq, qs[j]) + dot4(ik, ks[j]) + dot4(iv, vs[j]);
        pf[j][h] += dot4(fq, qs[j]) + dot4(fk, ks[j]) + dot4(fv, vs[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kTB; ++j)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float si = wave_sum(pi[j][h]), sf = wave_sum(pf[j][h]);
      if (lane == 0) {
        red[wave][(j * NH + h) * 2] = si;
        red[wave][(j * NH + h) * 2 + 1] = sf;
      }
    }
  __syncthreads();
  if (tid < kTB * NH) {
    const int j = tid / NH, h = tid - j * NH;
    if (t0 + j < T) {
      const int o = (j * NH + h) * 2;
      const float gi = red[0][o] + red[1][o] + red[2][o] + red[3][o] + a.bi[h];
      const float gf = red[0][o + 1] + red[1][o + 1] + red[2][o + 1] + red[3][o + 1] + a.bf[h];
      *reinterpret_cast<float2*>(a.gates + (((int64_t)b * T + t0 + j) * NH + h) * 2) = make_float2(gi, gf);
    }
  }
}

// =============================================================================================
// Per (env, head): gate scan, intra-chunk weight matrix A (Q Khat^T on the matrix cores, masked and decayed),
// denominators, and the small states (n, m; conv for head 0).
// =============================================================================================
constexpr int kSP = 36;  // LDS row pitch of the 32-wide K tiles (floats): conflict-free ds_read_b128
constexpr int kDP = 65;  // LDS row pitch of the 64 x 64 decay / A matrix

__global__ __launch_bounds__(kThreads) void mlstm_chunk_scan_kernel(MlstmPreArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[kLp * kSP];
  __shared__ __attribute__((aligned(16))) float Ks[kLp * kSP];
  __shared__ float Dm[kLp * kDP];
  __shared__ float s_m[kLp], s_fc[kLp], s_w[kLp], s_qn[kLp];

  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = a.inner, T = a.T, NH = a.NH, DH = inner / NH;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  const float sqrt_dh = sqrtf((float)DH);
  const float* qb = a.q + (int64_t)b * T * inner + (int64_t)h * DH;  // row t at + t * inner
  const float* kb = a.k + (int64_t)b * T * inner + (int64_t)h * DH;

  // ---- 1 + 2. wave 0: stabilised gate factors and every chain over the tokens, in registers.  Lane t owns
  // token t; the sequential parts run unrolled over the 64 lanes with v_readlane broadcasts (no LDS round trips).
  if (wave == 0) {
    const int t = lane;
    const bool valid = t < T;
    float gi = 0.f, lf = 0.f;
    if (valid) {
      const float2 g = *reinterpret_cast<const float2*>(a.gates + (((int64_t)b * T + t) * NH + h) * 2);
      gi = g.x;
      lf = log_sigmoid(g.y);
    }
    // m_t = max(lf_t + m_{t-1}, gi_t)   (all lanes run the same chain; lane t keeps its own step)
    float m = rs ? 0.f : a.m_state[(int64_t)b * NH + h];
    float fa = 0.f, mt = 0.f;
#pragma unroll
    for (int u = 0; u < kLp; ++u) {
      if (u < T) {
        const float lfu = __shfl(lf, u, 64), giu = __shfl(gi, u, 64);
        const float mn = fmaxf(lfu + m, giu);
        if (lane == u) {
          fa = lfu + m - mn;
          mt = mn;
        }
        m = mn;
      }
    }
    if (lane == 0) a.m_state[(int64_t)b * NH + h] = m;
    const float f = valid ? expf(fa) : 1.f;
    const float ig = valid ? expf(gi - mt) : 0.f;
    // fcum_t = f_0 .. f_t ;  w_s = i_s f_{s+1} .. f_{T-1} ;  decay column D[t][s = lane] = i_s f_{s+1} .. f_t
    float fc = 1.f, fcum = 0.f;
    float d = ig;
#pragma unroll
    for (int u = 0; u < kLp; ++u) {
      const float fu = __shfl(f, u, 64);
      fc *= fu;  // f == 1 beyond T
      if (lane == u) fcum = fc;
      if (u > lane) d *= fu;
      Dm[u * kDP + lane] = (u >= lane && u < T && valid) ? d : 0.f;
    }
    s_fc[t] = valid ? fcum : 0.f;
    s_w[t] = valid ? d : 0.f;  // after the last step d = i_s f_{s+1} .. f_63 with f == 1 beyond T
    s_m[t] = mt;
  }

  // ---- 3. S = Q Khat^T (64 x 64, K = DH) on the matrix cores: wave (tm, tn) owns one 32 x 32 tile ----
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lr = tid >> 3, lc = (tid & 7) << 2;  // staging: rows lr, lr + 32; k offset lc
  float4 rq[2], rk[2];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = lr + 32 * i;
      if (t < T) {
        rq[i] = *reinterpret_cast<const float4*>(qb + (int64_t)t * inner + k0 + lc);
        rk[i] = *reinterpret_cast<const float4*>(kb + (int64_t)t * inner + k0 + lc);
      } else {
        rq[i] = f4_zero();
        rk[i] = f4_zero();
      }
    }
  };
  const int nk = DH / 32;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = lr + 32 * i;
      *reinterpret_cast<float4*>(Qs + t * kSP + lc) = rq[i];
      *reinterpret_cast<float4*>(Ks + t * kSP + lc) =
          make_float4(rk[i].x / sqrt_dh, rk[i].y / sqrt_dh, rk[i].z / sqrt_dh, rk[i].w / sqrt_dh);
    }
    __syncthreads();
    if (kt + 1 < nk) load_tile((kt + 1) * 32);
    // (the tile t < 32 <= s is masked out entirely; its wave computes it anyway and drops it below)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const float4 af = *reinterpret_cast<const float4*>(Qs + (32 * wm + li) * kSP + 8 * kc + 4 * lh);
      const float4 bf = *reinterpret_cast<const float4*>(Ks + (32 * wn + li) * kSP + 8 * kc + 4 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- 4. A = D o S, in place in LDS (the tile (0, 1) of D is zero already) ----
  {
    const bool live = !(wm == 0 && wn == 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = 32 * wm + acc_row(r, lh), s = 32 * wn + li;
      const float d = Dm[t * kDP + s];
      Dm[t * kDP + s] = live ? d * acc[r] : 0.f;
    }
  }
  __syncthreads();
  // ---- 5. A to global (row pitch 64), q_t . n_0 (one wave per row, lanes over DH) ----
  float* Ag = a.amat + ((int64_t)b * NH + h) * kLp * kLp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + kThreads * i;
    const int t = idx >> 4, s4 = (idx & 15) << 2;
    const float* src = Dm + t * kDP + s4;
    *reinterpret_cast<float4*>(Ag + t * kLp + s4) = make_float4(src[0], src[1], src[2], src[3]);
  }
  const float* n0 = a.n_state + (int64_t)b * inner + (int64_t)h * DH;
  {
    // rows t = wave, wave + 4, ...: all loads of four rows are issued before the first reduction
    const int nq = DH >> 2;
    for (int t0 = wave; t0 < kLp; t0 += 16) {
      float p[4] = {0.f, 0.f, 0.f, 0.f};
      if (!rs) {
        for (int r4 = lane; r4 < nq; r4 += 64) {
          const float4 nv = *reinterpret_cast<const float4*>(n0 + 4 * r4);
          float4 qv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int t = t0 + 4 * j;
            qv[j] = t < T ? *reinterpret_cast<const float4*>(qb + (int64_t)t * inner + 4 * r4) : f4_zero();
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) p[j] += dot4(qv[j], nv);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sum = wave_sum(p[j]);
        if (lane == 0) s_qn[t0 + 4 * j] = sum;
      }
    }
  }
  __syncthreads();
  // ---- 6. denominators, chunk vectors; n_T = fcum_{T-1} n_0 + sum_s w_s khat_s ----
  float* vec = a.vec + ((int64_t)b * NH + h) * 3 * kLp;
  if (tid < kLp) {
    const int t = tid;
    float den = 1.f;
    if (t < T) {
      float rowsum = 0.f;  // entries beyond the diagonal are zero
#pragma unroll 16
      for (int s = 0; s < kLp; ++s) rowsum += Dm[t * kDP + s];
      den = fmaxf(fabsf(s_fc[t] * s_qn[t] + rowsum), expf(-s_m[t])) + 1e-6f;
    }
    vec[t] = s_fc[t];
    vec[kLp + t] = s_w[t];
    vec[2 * kLp + t] = den;
  }
  const float fall = s_fc[T - 1];
  float* nst = a.n_state + (int64_t)b * inner + (int64_t)h * DH;
  for (int r = tid; r < DH; r += kThreads) {
    float n = rs ? 0.f : fall * nst[r];
    for (int s0 = 0; s0 < T; s0 += 8) {
      float kv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) kv[j] = kb[(int64_t)min(s0 + j, T - 1) * inner + r];  // w == 0 beyond T
#pragma unroll
      for (int j = 0; j < 8; ++j) n += s_w[s0 + j] * (kv[j] / sqrt_dh);
    }
    nst[r] = n;
  }
  // ---- 7. conv state = the chunk's last 4 inputs (T >= 4 on this path) ----
  if (h == 0) {
    for (int idx = tid; idx < inner; idx += kThreads * 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        a.conv_state[((int64_t)b * 4 + k) * inner + idx] = a.u[((int64_t)b * T + T - 4 + k) * 2 * inner + idx];
    }
  }
}

// =============================================================================================
// Cell: one workgroup per (env, head, 128 columns of C).  C streams through LDS in tiles of 32 rows: each tile
// is read once (non-temporal), contributes to H = diag(fcum) Q C_0 (wave w: columns 32w..32w+31, both 32-token
// tiles) and leaves as fcum_{T-1} C_0 + (w Khat)^T V.  A V is added from the A matrix of the scan kernel.
// =============================================================================================
constexpr int kCW = 128;  // columns of C per workgroup
constexpr int kRT = 32;   // rows of C per tile
constexpr int kPV = 136;  // LDS row pitch of the 128-wide tiles (V, C): the two lane halves hit different banks
constexpr int kPK = 40;   // LDS row pitch of the (w Khat) tile
constexpr int kCellChunkLds = (kLp * kPV + kRT * kPV + kLp * kSP + kLp * kPK) * 4;  // 71,680 B

__global__ __launch_bounds__(kThreads) void mlstm_cell_chunk_kernel(MlstmCellArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                 // [64][kPV]   v_t, columns of this slice
  float* Cs = Vs + kLp * kPV;       // [32][kPV]   C_0 row tile
  float* Qs = Cs + kRT * kPV;       // [64][kSP]   fcum_t q_t, 32 rows of DH
  float* Ks = Qs + kLp * kSP;       // [64][kPK]   w_t khat_t, 32 rows of DH
  __shared__ float s_fc[kLp], s_w[kLp], s_den[kLp];

  const int T = a.T, NH = a.NH, DH = a.DH, inner = NH * DH;
  const int nslices = DH / kCW;
  // XCD-aware order: consecutive work items (the column slices of one (env, head), which share Q, K and A) run on
  // the same XCD back to back, so the second slice finds them in that XCD's L2
  int wid = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
  const int slice = wid % nslices;
  const int bh = wid / nslices;
  const int h = bh % NH, b = bh / NH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  const bool two = T > 32;
  const int kt8 = (T + 7) >> 3;  // 8-token groups in the contractions over tokens
  const float sqrt_dh = sqrtf((float)DH);

  const float* vec = a.vec + ((int64_t)b * NH + h) * 3 * kLp;
  if (tid < kLp) {
    s_fc[tid] = vec[tid];
    s_w[tid] = vec[kLp + tid];
    s_den[tid] = vec[2 * kLp + tid];
  }
  const int col0 = slice * kCW;
  const float* qb = a.q + (int64_t)b * T * inner + (int64_t)h * DH;
  const float* kb = a.k + (int64_t)b * T * inner + (int64_t)h * DH;
  const float* vb = a.v + (int64_t)b * T * inner + (int64_t)h * DH + col0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + kThreads * i;
    const int t = idx >> 5, c4 = (idx & 31) << 2;
    const float4 v = t < T ? *reinterpret_cast<const float4*>(vb + (int64_t)t * inner + c4) : f4_zero();
    *reinterpret_cast<float4*>(Vs + t * kPV + c4) = v;
  }
  __syncthreads();
  const float fall = s_fc[T - 1];

  f32x16 hacc0, hacc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) hacc0[r] = hacc1[r] = 0.f;

  // ---- H += A V  (A rows straight from global / L2; tile 0 only sees s < 32) ----
  {
    const float* Ag = a.amat + ((int64_t)b * NH + h) * kLp * kLp;
    for (int j = 0; j < kt8; ++j) {
      float vbv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) vbv[i] = Vs[(8 * j + 4 * lh + i) * kPV + 32 * w + li];
      if (j < 4) {
        const float4 a0 = *reinterpret_cast<const float4*>(Ag + li * kLp + 8 * j + 4 * lh);
        hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, vbv[0], hacc0, 0, 0, 0);
        hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, vbv[1], hacc0, 0, 0, 0);
        hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, vbv[2], hacc0, 0, 0, 0);
        hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, vbv[3], hacc0, 0, 0, 0);
      }
      if (two) {
        const float4 a1 = *reinterpret_cast<const float4*>(Ag + (32 + li) * kLp + 8 * j + 4 * lh);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, vbv[0], hacc1, 0, 0, 0);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, vbv[1], hacc1, 0, 0, 0);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, vbv[2], hacc1, 0, 0, 0);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, vbv[3], hacc1, 0, 0, 0);
      }
    }
  }

  // ---- main loop over 32-row tiles of C ----
  float* Cg = a.C + (((int64_t)b * NH + h) * DH) * DH + col0;
  const int crow = tid >> 5, cc4 = (tid & 31) << 2;  // C tile: rows crow + 8 i, 4 columns at cc4
  const int qrow = tid >> 3, qc4 = (tid & 7) << 2;   // Q / K tiles: tokens qrow + 32 i, 4 rows of DH at qc4
  v4f rc[4];
  float4 rq[2], rk[2];
  auto load_tile = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rc[i] = rs ? (v4f)(0.f)
                 : __builtin_nontemporal_load(reinterpret_cast<const v4f*>(Cg + (int64_t)(r0 + crow + 8 * i) * DH + cc4));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = qrow + 32 * i;
      if (t < T) {
        rq[i] = *reinterpret_cast<const float4*>(qb + (int64_t)t * inner + r0 + qc4);
        rk[i] = *reinterpret_cast<const float4*>(kb + (int64_t)t * inner + r0 + qc4);
      } else {
        rq[i] = f4_zero();
        rk[i] = f4_zero();
      }
    }
  };
  // Stores of tile i are issued one iteration late, right after the barrier of tile i + 1 and before the prefetch of
  // tile i + 2: on gfx9 loads and stores share vmcnt, so the wait for a prefetched tile also waits for every
  // store issued before it -- this way those stores have had a whole compute phase to drain.
  const int64_t lane_off = (int64_t)(4 * lh) * DH + 32 * w + li;
  auto store_tile = [&](const f32x16& c, int r0) {
    float* dst = Cg + (int64_t)r0 * DH + lane_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(c[r], dst + (int64_t)((r & 3) + 8 * (r >> 2)) * DH);
  };
  const int ntiles = DH / kRT;
  f32x16 cprev;
#pragma unroll
  for (int r = 0; r < 16; ++r) cprev[r] = 0.f;
  load_tile(0);
  for (int it = 0; it < ntiles; ++it) {
    const int r0 = it * kRT;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<v4f*>(Cs + (crow + 8 * i) * kPV + cc4) = rc[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = qrow + 32 * i;
      const float fc = s_fc[t], wt = s_w[t];
      *reinterpret_cast<float4*>(Qs + t * kSP + qc4) = make_float4(fc * rq[i].x, fc * rq[i].y, fc * rq[i].z, fc * rq[i].w);
      *reinterpret_cast<float4*>(Ks + t * kPK + qc4) =
          make_float4(wt * (rk[i].x / sqrt_dh), wt * (rk[i].y / sqrt_dh), wt * (rk[i].z / sqrt_dh), wt * (rk[i].w / sqrt_dh));
    }
    __syncthreads();
    if (it > 0) store_tile(cprev, r0 - kRT);
    if (it + 1 < ntiles) load_tile(r0 + kRT);
    // H += (fcum Q)[:, tile] C_0[tile, cols]   (all LDS operands of the tile first, then the MFMAs)
    float cb[16];
    float4 q0[4], q1[4];
#pragma unroll
    for (int j = 0; j < kRT / 8; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) cb[4 * j + i] = Cs[(8 * j + 4 * lh + i) * kPV + 32 * w + li];
      q0[j] = *reinterpret_cast<const float4*>(Qs + li * kSP + 8 * j + 4 * lh);
      q1[j] = two ? *reinterpret_cast<const float4*>(Qs + (32 + li) * kSP + 8 * j + 4 * lh) : f4_zero();
    }
    // first operands of the token contraction, in flight while the MFMAs above run
    float ka[4], va[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ka[i] = Ks[(4 * lh + i) * kPK + li];
      va[i] = Vs[(4 * lh + i) * kPV + 32 * w + li];
    }
#pragma unroll
    for (int j = 0; j < kRT / 8; ++j) {
      hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0[j].x, cb[4 * j + 0], hacc0, 0, 0, 0);
      hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0[j].y, cb[4 * j + 1], hacc0, 0, 0, 0);
      hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0[j].z, cb[4 * j + 2], hacc0, 0, 0, 0);
      hacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(q0[j].w, cb[4 * j + 3], hacc0, 0, 0, 0);
    }
    if (two) {
#pragma unroll
      for (int j = 0; j < kRT / 8; ++j) {
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1[j].x, cb[4 * j + 0], hacc1, 0, 0, 0);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1[j].y, cb[4 * j + 1], hacc1, 0, 0, 0);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1[j].z, cb[4 * j + 2], hacc1, 0, 0, 0);
        hacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q1[j].w, cb[4 * j + 3], hacc1, 0, 0, 0);
      }
    }
    // C_T[tile, cols] = fcum_{T-1} C_0[tile, cols] + (w Khat)[:, tile]^T V[:, cols]
    // (register r of the accumulator holds row acc_row(r, lh) = the row cb[r] was read from)
    f32x16 cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = fall * cb[r];
    for (int j = 0; j < kt8; ++j) {
      float kn[4], vn[4];
      const int jn = min(j + 1, kt8 - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kn[i] = Ks[(8 * jn + 4 * lh + i) * kPK + li];
        vn[i] = Vs[(8 * jn + 4 * lh + i) * kPV + 32 * w + li];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[i], va[i], cacc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ka[i] = kn[i];
        va[i] = vn[i];
      }
    }
    cprev = cacc;
    __syncthreads();
  }
  store_tile(cprev, (ntiles - 1) * kRT);

  // ---- h_t = H[t] / den_t ----
  float* hb = a.h + (int64_t)b * T * inner + (int64_t)h * DH + col0 + 32 * w + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t0 = acc_row(r, lh);
    if (t0 < T) hb[(int64_t)t0 * inner] = hacc0[r] / s_den[t0];
    if (two && 32 + t0 < T) hb[(int64_t)(32 + t0) * inner] = hacc1[r] / s_den[32 + t0];
  }
}


// =============================================================================================
// Cell, bf16x3 form (the default): the same three contractions on the bf16 matrix cores.  Every fp32 operand is
// split exactly into three bf16 pieces and a product is the six largest piece products, as in gemm_bf16x3.hip
// (error <= 2^-23 relative: one fp32 rounding) -- `v_mfma_f32_32x32x16_bf16` runs at 16 x the rate of the fp32-input
// MFMA, so the pass drops from matrix-core-bound (85 TF/s of exact fp32 products) to the HBM time of C once in, once out.
//   * C_0 never goes through LDS: lane (column li, half lh) loads exactly the 16 rows of its 32 x 32 accumulator
//     tile, acc_row(i, lh); they seed the C_T accumulator (fall * C_0) AND, split in registers, are the B operand of
//     Q C_0 -- the contraction runs over the tile's rows in THAT order, and the Q planes are stored in LDS in the
//     same order (position 16 lh + i <-> row acc_row(i, lh)).
//   * V planes [column][token] stay in LDS for the whole pass (B operand of both token contractions), (fcum Q) and
//     (w Khat)^T planes are staged per 32-row tile; 16-byte granules XOR-swizzled so ds_read_b128 / ds_write_b128 /
//     ds_write_b64 are conflict-free without padding: 72 KB of LDS, two workgroups per CU.
// =============================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kVPlane = kCW * kLp;   // bf16 elements per V plane   [128 columns][64 tokens]
constexpr int kQPlane = kLp * kRT;   // per (fcum Q) plane          [64 tokens][32 row slots]
constexpr int kKPlane = kRT * kLp;   // per (w Khat)^T plane        [32 rows][64 tokens]
constexpr int kCell3Lds = 3 * (kVPlane + kQPlane + kKPlane) * 2;  // 73,728 B

__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)x;
  const float r1 = x - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}
template <typename V, int N>
__device__ __forceinline__ void split3v(const float (&x)[N], V& hi, V& mid, V& lo) {
#pragma unroll
  for (int e = 0; e < N; ++e) {
    __bf16 h, m, l;
    split3(x[e], h, m, l);
    hi[e] = h, mid[e] = m, lo[e] = l;
  }
}
// acc += a * b with a = a[0] + a[1] + a[2], b likewise; smallest terms first
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

__global__ __launch_bounds__(kThreads) void mlstm_cell_chunk3_kernel(MlstmCellArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* Vp = reinterpret_cast<__bf16*>(smem_raw);  // [3][128][64]: granule g of row c at g ^ ((c >> 1) & 7)
  __bf16* Qp = Vp + 3 * kVPlane;                     // [3][64][32]:  granule g of row t at g ^ ((t >> 2) & 3)
  __bf16* Kp = Qp + 3 * kQPlane;                     // [3][32][64]:  granule g of row r at g ^ ((r >> 1) & 7)
  __shared__ float s_fc[kLp], s_w[kLp], s_den[kLp];

  const int T = a.T, NH = a.NH, DH = a.DH, inner = NH * DH;
  const int nslices = DH / kCW;
  int wid = blockIdx.x;  // XCD-aware order, as the fp32 form
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
  const int slice = wid % nslices;
  const int bh = wid / nslices;
  const int h = bh % NH, b = bh / NH;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  const bool two = T > 32;
  const int kt16 = (T + 15) >> 4;  // 16-token steps of the contractions over tokens
  const float sqrt_dh = sqrtf((float)DH);

  const float* vec = a.vec + ((int64_t)b * NH + h) * 3 * kLp;
  if (tid < kLp) {
    s_fc[tid] = vec[tid];
    s_w[tid] = vec[kLp + tid];
    s_den[tid] = vec[2 * kLp + tid];
  }
  const int col0 = slice * kCW;
  // Buffer resources: one 32-bit lane offset + a scalar offset per access (no 64-bit address arithmetic in the loop), and the
  // range check does the masking -- rows of q / k / v beyond T and the matrix of a reset env read as zero without a branch
  // (the check sees the LANE offset only, not the scalar one: token offsets ride in the lane offset).
  const uint32_t tok_bytes = (uint32_t)T * (uint32_t)inner * 4u;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q) + (int64_t)b * T * inner, 0, tok_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.k) + (int64_t)b * T * inner, 0, tok_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v) + (int64_t)b * T * inner, 0, tok_bytes, 0x00020000);
  float* Cm = a.C + (((int64_t)b * NH + h) * DH) * DH;
  const uint32_t c_bytes = (uint32_t)DH * (uint32_t)DH * 4u;
  const __amdgpu_buffer_rsrc_t rc_in = __builtin_amdgcn_make_buffer_rsrc(Cm, 0, rs ? 0u : c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc_out = __builtin_amdgcn_make_buffer_rsrc(Cm, 0, c_bytes, 0x00020000);
  constexpr int kNt = 2;  // cache policy: non-temporal (C is touched once per pass)
  const int row_b = DH * 4, tok_b = inner * 4;
  // ---- V planes: thread (column, token half) gathers 8 consecutive tokens of its column per granule ----
  {
    const int col = tid & 127, th = tid >> 7;
    const int voff = (h * DH + col0 + col) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff + (32 * th + 8 * g + j) * tok_b, 0, 0));
      bf16x8 p0, p1, p2;
      split3v(x, p0, p1, p2);
      __bf16* d = Vp + col * kLp + (((4 * th + g) ^ ((col >> 1) & 7)) << 3);
      *reinterpret_cast<bf16x8*>(d) = p0;
      *reinterpret_cast<bf16x8*>(d + kVPlane) = p1;
      *reinterpret_cast<bf16x8*>(d + 2 * kVPlane) = p2;
    }
  }
  __syncthreads();
  const float fall = s_fc[T - 1];

  f32x16 hacc0, hacc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) hacc0[r] = hacc1[r] = 0.f;
  float ws[8];   // w_t / sqrt(DH) of this thread's eight tokens: (w_t / s) k instead of w_t (k / s), one division per pass
#pragma unroll
  for (int j = 0; j < 8; ++j) ws[j] = s_w[8 * (tid >> 5) + j] / sqrt_dh;

  const int c_voff = (4 * lh * DH + col0 + 32 * w + li) * 4;   // + (r0 + (i & 3) + 8 (i >> 2)) rows as the scalar offset
  const int qt = tid >> 2, qd = tid & 3;   // Q staging: token qt, rows 8 qd .. 8 qd + 7 of the tile
  const int kr = tid & 31, kg = tid >> 5;  // K staging: row kr, tokens 8 kg .. 8 kg + 7
  const int q_voff = (qt * inner + h * DH + 8 * qd) * 4;
  const int k_voff = (8 * kg * inner + h * DH + kr) * 4;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  float kn[8];
  v4f qn[2];
  auto load_c = [&](float (&c)[16], int r0) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      c[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc_in, c_voff, (r0 + (i & 3) + 8 * (i >> 2)) * row_b, kNt));
  };
  auto load_qk = [&](int r0) {
    qn[0] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rq, q_voff, r0 * 4, 0));
    qn[1] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rq, q_voff, r0 * 4 + 16, 0));
#pragma unroll
    for (int j = 0; j < 8; ++j)
      kn[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rk, k_voff + j * tok_b, r0 * 4, 0));
  };
  auto store_tile = [&](const f32x16& c, int r0) {   // (one iteration late: see the fp32 form)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = c[r];
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rc_out, c_voff, (r0 + (r & 3) + 8 * (r >> 2)) * row_b, kNt);
    }
  };
  const int ntiles = DH / kRT;   // a multiple of 4 (DH % 128 == 0)
  f32x16 cprev;
#pragma unroll
  for (int r = 0; r < 16; ++r) cprev[r] = 0.f;
  // one tile of the main loop; cx holds C_0 of tile `it` on entry and is refilled with tile it + 2 (two tiles of C in flight)
  auto tile = [&](int it, float (&cx)[16]) {
    const int r0 = it * kRT;
    // ---- stage (fcum Q) and (w Khat)^T planes of this tile ----
    {
      const float fc = s_fc[qt];
#pragma unroll
      for (int u = 0; u < 2; ++u) {   // u = lane half the four rows belong to: row 8 qd + 4 u + e <-> slot 16 u + 4 qd + e
        const float x[4] = {fc * qn[u][0], fc * qn[u][1], fc * qn[u][2], fc * qn[u][3]};
        bf16x4 p0, p1, p2;
        split3v(x, p0, p1, p2);
        __bf16* d = Qp + qt * kRT + ((((2 * u + (qd >> 1)) ^ ((qt >> 2) & 3))) << 3) + 4 * (qd & 1);
        *reinterpret_cast<bf16x4*>(d) = p0;
        *reinterpret_cast<bf16x4*>(d + kQPlane) = p1;
        *reinterpret_cast<bf16x4*>(d + 2 * kQPlane) = p2;
      }
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = ws[j] * kn[j];
      bf16x8 p0, p1, p2;
      split3v(x, p0, p1, p2);
      __bf16* d = Kp + kr * kLp + ((kg ^ ((kr >> 1) & 7)) << 3);
      *reinterpret_cast<bf16x8*>(d) = p0;
      *reinterpret_cast<bf16x8*>(d + kKPlane) = p1;
      *reinterpret_cast<bf16x8*>(d + 2 * kKPlane) = p2;
    }
    float cc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cc[i] = cx[i];
    __syncthreads();
    if (it > 0) store_tile(cprev, r0 - kRT);
    if (it + 2 < ntiles) load_c(cx, r0 + 2 * kRT);
    if (it + 1 < ntiles) load_qk(r0 + kRT);
    // ---- H += (fcum Q)[:, tile] C_0[tile, cols] ----
    f32x16 cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = fall * cc[r];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 cb[3], qf[3];
      const float x[8] = {cc[8 * ks], cc[8 * ks + 1], cc[8 * ks + 2], cc[8 * ks + 3],
                          cc[8 * ks + 4], cc[8 * ks + 5], cc[8 * ks + 6], cc[8 * ks + 7]};
      split3v(x, cb[0], cb[1], cb[2]);
      {
        const __bf16* src = Qp + li * kRT + (((2 * lh + ks) ^ ((li >> 2) & 3)) << 3);
#pragma unroll
        for (int p = 0; p < 3; ++p) qf[p] = *reinterpret_cast<const bf16x8*>(src + p * kQPlane);
        mfma6(hacc0, qf, cb);
      }
      if (two) {
        const __bf16* src = Qp + (32 + li) * kRT + (((2 * lh + ks) ^ ((li >> 2) & 3)) << 3);   // ((32 + li) >> 2) & 3 == (li >> 2) & 3
#pragma unroll
        for (int p = 0; p < 3; ++p) qf[p] = *reinterpret_cast<const bf16x8*>(src + p * kQPlane);
        mfma6(hacc1, qf, cb);
      }
    }
    // ---- C_T[tile, cols] = fcum_{T-1} C_0[tile, cols] + (w Khat)[:, tile]^T V[:, cols] ----
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < kt16) {
        bf16x8 kf[3], vf[3];
        const __bf16* ksrc = Kp + li * kLp + (((2 * ks + lh) ^ ((li >> 1) & 7)) << 3);
        const int vc = 32 * w + li;
        const __bf16* vsrc = Vp + vc * kLp + (((2 * ks + lh) ^ ((vc >> 1) & 7)) << 3);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          kf[p] = *reinterpret_cast<const bf16x8*>(ksrc + p * kKPlane);
          vf[p] = *reinterpret_cast<const bf16x8*>(vsrc + p * kVPlane);
        }
        mfma6(cacc, kf, vf);
      }
    }
    cprev = cacc;
    __syncthreads();
  };
  float ca[16], cb2[16];
  load_c(ca, 0);
  load_c(cb2, kRT);
  load_qk(0);
  for (int it = 0; it < ntiles; it += 2) {
    tile(it, ca);
    tile(it + 1, cb2);
  }
  store_tile(cprev, (ntiles - 1) * kRT);

  // ---- H += A V  (A rows from global / L2, split in registers; A[t][s] == 0 for s > t: token tile 0 stops at s < 32) ----
  {
    const float* Ag = a.amat + ((int64_t)b * NH + h) * kLp * kLp;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < kt16) {
        bf16x8 vf[3], af[3];
        const int vc = 32 * w + li;
        const __bf16* vsrc = Vp + vc * kLp + (((2 * ks + lh) ^ ((vc >> 1) & 7)) << 3);
#pragma unroll
        for (int p = 0; p < 3; ++p) vf[p] = *reinterpret_cast<const bf16x8*>(vsrc + p * kVPlane);
        if (ks < 2) {
          const float4 a0 = *reinterpret_cast<const float4*>(Ag + li * kLp + 16 * ks + 8 * lh);
          const float4 a1 = *reinterpret_cast<const float4*>(Ag + li * kLp + 16 * ks + 8 * lh + 4);
          const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          split3v(x, af[0], af[1], af[2]);
          mfma6(hacc0, af, vf);
        }
        if (two) {
          const float4 a0 = *reinterpret_cast<const float4*>(Ag + (32 + li) * kLp + 16 * ks + 8 * lh);
          const float4 a1 = *reinterpret_cast<const float4*>(Ag + (32 + li) * kLp + 16 * ks + 8 * lh + 4);
          const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          split3v(x, af[0], af[1], af[2]);
          mfma6(hacc1, af, vf);
        }
      }
    }
  }

  // ---- h_t = H[t] / den_t ----
  float* hb = a.h + (int64_t)b * T * inner + (int64_t)h * DH + col0 + 32 * w + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t0 = acc_row(r, lh);
    if (t0 < T) hb[(int64_t)t0 * inner] = hacc0[r] / s_den[t0];
    if (two && 32 + t0 < T) hb[(int64_t)(32 + t0) * inner] = hacc1[r] / s_den[32 + t0];
  }
}

}  // namespace

bool mlstm_chunk_supported(int inner, int NH, int K) {
  const int DH = NH > 0 ? inner / NH : 0;
  return K == 4 && NH > 0 && inner % NH == 0 && DH % kCW == 0 && inner <= 4 * kThreads * kPreMaxGroups &&
         (NH == 1 || NH == 2 || NH == 4 || NH == 8);
}

void launch_mlstm_chunk_pre(const MlstmPreArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(a.T >= 4 && a.T <= kChunkMaxTokens, "chunkwise mLSTM: 4..64 tokens per chunk");
  LRAM_REQUIRE(mlstm_chunk_supported(a.inner, a.NH, a.K), "chunkwise mLSTM: unsupported geometry");
  LRAM_REQUIRE(a.gates != nullptr && a.amat != nullptr && a.vec != nullptr, "chunkwise mLSTM: missing work buffers");
  dim3 grid((a.T + kTB - 1) / kTB, a.B), block(kThreads);
  switch (a.NH) {
    case 1: hipLaunchKernelGGL(mlstm_pre_tok_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(mlstm_pre_tok_kernel<2>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(mlstm_pre_tok_kernel<4>, grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL(mlstm_pre_tok_kernel<8>, grid, block, 0, stream, a); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(mlstm_chunk_scan_kernel, dim3(a.NH, a.B), block, 0, stream, a);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_mlstm_chunk_cell(const MlstmCellArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(a.T >= 4 && a.T <= kChunkMaxTokens, "chunkwise mLSTM: 4..64 tokens per chunk");
  LRAM_REQUIRE(a.DH % kCW == 0, "chunkwise mLSTM: head dim must be a multiple of 128");
  LRAM_REQUIRE(a.amat != nullptr && a.vec != nullptr, "chunkwise mLSTM: missing work buffers");
  static uint64_t raised = 0;
  if (first_use_on_device(raised)) {
    LRAM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlstm_cell_chunk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kCellChunkLds));
    LRAM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlstm_cell_chunk3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kCell3Lds));
  }
  const long nwg = (long)a.B * a.NH * (a.DH / kCW);
  if (a.chunk_exact_fp32)
    hipLaunchKernelGGL(mlstm_cell_chunk_kernel, dim3((unsigned)nwg), dim3(kThreads), kCellChunkLds, stream, a);
  else
    hipLaunchKernelGGL(mlstm_cell_chunk3_kernel, dim3((unsigned)nwg), dim3(kThreads), kCell3Lds, stream, a);
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
