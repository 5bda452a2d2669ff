// Recurrent xLSTM kernels for gfx950: the mLSTM matrix-memory update (the HBM-bound hot kernel), its
// conv / qkv / gate front end, the multi-head group norm epilogue, and the sLSTM scalar cell.
//
// What they replace on the reference path: N3 recurrent_step_stabilized_simple
// (~15 eager ops), N5 conv1d_step, N6 LinearHeadwiseExpand, N7 MultiHeadLayerNorm, N1 sLSTM pointwise.
// All of a timestep's T (= 3) tokens are applied inside one launch per layer: the matrix memory C is
// read once and written once per env-step instead of once per token.
#include <algorithm>

#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {

// =============================================================================================
// mLSTM front end: one workgroup per env.
//   conv1d_step (K taps, depthwise) -> SiLU -> block-diagonal q/k (from the conv branch) and v (from the
//   pre-conv branch) -> gate pre-activations Wi.[q,k,v]+bi, Wf.[q,k,v]+bf (block reduction) -> stabilised
//   gate scalars f_t, i_t, m_t -> normaliser state n_t and denominators max(|q.n_t|, exp(-m_t)) + 1e-6.
// Writes q,k,v,xa rows, the per-(token, head) scalars, and the updated conv / n / m state.
// =============================================================================================
namespace {

constexpr int kPreThreads = 256;
constexpr int kMaxGroups = 4;  // channel groups (of 4) per thread: inner <= 4096

// (body as a device function of the env index: the kernel calls it with blockIdx.x)
template <int T, int NH, bool SINGLE = false>
__device__ __forceinline__ void mlstm_pre_body(const MlstmPreArgs& a, const int b) {
  constexpr int NRED = 2 * T * NH;
  __shared__ float red[4][NRED];
  __shared__ float gate_i[T][NH], gate_f[T][NH];
  __shared__ float s_f[T][NH], s_i[T][NH], s_m[T][NH];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int inner = a.inner;
  const int DH = inner / NH;
  const int ngroups = inner >> 2;
  // (the flag is read through a pointer select, not under a branch on a.reset: a conditional load is waited for on the
  // spot, ahead of every other request of the workgroup)
  const uint8_t* rsp = a.reset != nullptr ? a.reset + b : reinterpret_cast<const uint8_t*>(a.conv_w);
  const uint8_t rsb = *rsp;
  const bool rs = a.reset != nullptr && rsb != 0;

  float pi[T][NH], pf[T][NH];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < NH; ++h) pi[t][h] = pf[t][h] = 0.f;
  // One channel group per thread (inner <= 4 * kPreThreads, e.g. the 16M geometry): q and k stay in registers for
  // phase 3 and the recurrent scalars / the normaliser state are requested before the block reductions instead of after
  // them -- one dependent memory round trip per workgroup instead of three (beside a read pass each costs microseconds).
  constexpr bool single = SINGLE;  // the launcher guarantees ngroups <= kPreThreads
  // (state is requested whether or not the env restarts and zeroed afterwards: a load that waits for the reset flag's own
  // round trip puts one more dependent memory round trip at the head of every workgroup)
  // (zeroed by a bit mask, not a branch or a multiply: no control flow on the flag -- the compiler otherwise waits for it
  // before issuing anything else -- and no NaN * 0)
  auto masked = [](float v, unsigned keep) { return __uint_as_float(__float_as_uint(v) & keep); };
  auto masked4 = [&](const float4& v, unsigned keep) {
    return make_float4(masked(v.x, keep), masked(v.y, keep), masked(v.z, keep), masked(v.w, keep));
  };
  float m_pre = single ? a.m_state[(int64_t)b * NH + min(tid, NH - 1)] : 0.f;  // (every lane: no branch; masked below)
  float4 q_keep[T], k_keep[T], n_pre = f4_zero();
#pragma unroll
  for (int t = 0; t < T; ++t) q_keep[t] = k_keep[t] = f4_zero();

  // ---- phase 1: conv, qkv, gate partial sums -------------------------------------------------
  for (int cg = tid; cg < ngroups; cg += kPreThreads) {
    const int c0 = cg << 2;
    // conv window per channel: win[k] holds tap k for the 4 channels (oldest first)
    float4 win[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      win[k] = (k >= a.K) ? f4_zero()
                          : *reinterpret_cast<const float4*>(a.conv_state + ((int64_t)b * a.K + k) * inner + c0);
    if (single) n_pre = *reinterpret_cast<const float4*>(a.n_state + (int64_t)b * inner + c0);
    // conv weights [inner, K] (K == 4 taps contiguous per channel)
    float4 cw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cw[c] = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)(c0 + c) * 4);
    const float4 cb = *reinterpret_cast<const float4*>(a.conv_b + c0);
    float4 wq[4], wk[4], wv[4];  // row o of the 4x4 block: w[o] . x
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      wq[o] = *reinterpret_cast<const float4*>(a.wq + (int64_t)cg * 16 + o * 4);
      wk[o] = *reinterpret_cast<const float4*>(a.wk + (int64_t)cg * 16 + o * 4);
      wv[o] = *reinterpret_cast<const float4*>(a.wv + (int64_t)cg * 16 + o * 4);
    }
    {  // the restart mask is applied only now, with every request above already issued (the scheduler otherwise resolves
       // the flag first and the workgroup starts with a round trip for one byte)
      __builtin_amdgcn_sched_barrier(0);
      unsigned flag = rsb;
      asm volatile("" : "+v"(flag));  // opaque here: the comparison cannot be hoisted (with its wait) above the requests
      const unsigned keep = (a.reset != nullptr && flag != 0) ? 0u : 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < 4; ++k) win[k] = masked4(win[k], keep);
      n_pre = masked4(n_pre, keep);
      m_pre = masked(m_pre, keep);
    }
    float4 qt[T], kt[T], vt[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t row = (int64_t)b * T + t;
      const float4 xm = *reinterpret_cast<const float4*>(a.u + row * 2 * inner + c0);
      win[0] = win[1];
      win[1] = win[2];
      win[2] = win[3];
      win[3] = xm;
      float4 y;
      y.x = win[0].x * cw[0].x + win[1].x * cw[0].y + win[2].x * cw[0].z + win[3].x * cw[0].w + cb.x;
      y.y = win[0].y * cw[1].x + win[1].y * cw[1].y + win[2].y * cw[1].z + win[3].y * cw[1].w + cb.y;
      y.z = win[0].z * cw[2].x + win[1].z * cw[2].y + win[2].z * cw[2].z + win[3].z * cw[2].w + cb.z;
      y.w = win[0].w * cw[3].x + win[1].w * cw[3].y + win[2].w * cw[3].z + win[3].w * cw[3].w + cb.w;
      float4 xa;
      xa.x = silu_f(y.x);
      xa.y = silu_f(y.y);
      xa.z = silu_f(y.z);
      xa.w = silu_f(y.w);
      auto bd = [](const float4* w, const float4& x) {
        float4 r;
        r.x = w[0].x * x.x + w[0].y * x.y + w[0].z * x.z + w[0].w * x.w;
        r.y = w[1].x * x.x + w[1].y * x.y + w[1].z * x.z + w[1].w * x.w;
        r.z = w[2].x * x.x + w[2].y * x.y + w[2].z * x.z + w[2].w * x.w;
        r.w = w[3].x * x.x + w[3].y * x.y + w[3].z * x.z + w[3].w * x.w;
        return r;
      };
      qt[t] = bd(wq, xa);
      kt[t] = bd(wk, xa);
      vt[t] = bd(wv, xm);
      q_keep[t] = qt[t], k_keep[t] = kt[t];
      if (!a.lean) {  // lean: the consumer (lazy read pass) rebuilds q, k, v from xa and the x half of u
        *reinterpret_cast<float4*>(a.q + row * inner + c0) = qt[t];
        *reinterpret_cast<float4*>(a.k + row * inner + c0) = kt[t];
        *reinterpret_cast<float4*>(a.v + row * inner + c0) = vt[t];
      }
      *reinterpret_cast<float4*>(a.xa + row * inner + c0) = xa;
    }
    // conv state after the T tokens (reference layout [B, K, inner], newest tap last)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < a.K) *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * a.K + k) * inner + c0) = win[k];
    // gate partial sums over this thread's 4 channels of q, k and v
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float* wi = a.wi + (int64_t)h * 3 * inner + c0;
      const float* wf = a.wf + (int64_t)h * 3 * inner + c0;
      const float4 iq = *reinterpret_cast<const float4*>(wi), ik = *reinterpret_cast<const float4*>(wi + inner),
                   iv = *reinterpret_cast<const float4*>(wi + 2 * inner);
      const float4 fq = *reinterpret_cast<const float4*>(wf), fk = *reinterpret_cast<const float4*>(wf + inner),
                   fv = *reinterpret_cast<const float4*>(wf + 2 * inner);
      auto dot = [](const float4& x, const float4& y) { return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; };
#pragma unroll
      for (int t = 0; t < T; ++t) {
        pi[t][h] += dot(iq, qt[t]) + dot(ik, kt[t]) + dot(iv, vt[t]);
        pf[t][h] += dot(fq, qt[t]) + dot(fk, kt[t]) + dot(fv, vt[t]);
      }
    }
  }
  // block reduction of the 2*T*NH partial sums
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float si = wave_sum(pi[t][h]);
      const float sf = wave_sum(pf[t][h]);
      if (lane == 0) {
        red[wave][(t * NH + h) * 2 + 0] = si;
        red[wave][(t * NH + h) * 2 + 1] = sf;
      }
    }
  __syncthreads();
  if (tid < T * NH) {
    const int t = tid / NH, h = tid - t * NH;
    float si = 0.f, sf = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      si += red[w][tid * 2 + 0];
      sf += red[w][tid * 2 + 1];
    }
    gate_i[t][h] = si + a.bi[h];
    gate_f[t][h] = sf + a.bf[h];
  }
  __syncthreads();
  // ---- phase 2: stabilised gate scalars (sequential over the T tokens, one thread per head) --
  if (tid < NH) {
    const int h = tid;
    float m = single ? m_pre : (rs ? 0.f : a.m_state[(int64_t)b * NH + h]);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float lf = log_sigmoid(gate_f[t][h]);
      const float mn = fmaxf(lf + m, gate_i[t][h]);
      s_f[t][h] = expf(lf + m - mn);
      s_i[t][h] = expf(gate_i[t][h] - mn);
      s_m[t][h] = mn;
      m = mn;
    }
    a.m_state[(int64_t)b * NH + h] = m;
  }
  __syncthreads();
  // ---- phase 3: normaliser state n_t = f_t n_{t-1} + i_t k_t/sqrt(DH) and q_t . n_t -----------
  float pq[T][NH];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < NH; ++h) pq[t][h] = 0.f;
  const float sqrt_dh = sqrtf((float)DH);
  for (int cg = tid; cg < ngroups; cg += kPreThreads) {
    const int c0 = cg << 2;
    const int hh = c0 / DH;
    float4 n = single ? n_pre : (rs ? f4_zero() : *reinterpret_cast<const float4*>(a.n_state + (int64_t)b * inner + c0));
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t row = (int64_t)b * T + t;
      float4 qv, kv;
      if (single) {
        qv = q_keep[t], kv = k_keep[t];
      } else if (a.lean) {  // same 4 x 4 block products as phase 1 (this thread's own xa rows)
        const float4 xa = *reinterpret_cast<const float4*>(a.xa + row * inner + c0);
        const float* wq = a.wq + (int64_t)cg * 16;
        const float* wk = a.wk + (int64_t)cg * 16;
        qv.x = wq[0] * xa.x + wq[1] * xa.y + wq[2] * xa.z + wq[3] * xa.w;
        qv.y = wq[4] * xa.x + wq[5] * xa.y + wq[6] * xa.z + wq[7] * xa.w;
        qv.z = wq[8] * xa.x + wq[9] * xa.y + wq[10] * xa.z + wq[11] * xa.w;
        qv.w = wq[12] * xa.x + wq[13] * xa.y + wq[14] * xa.z + wq[15] * xa.w;
        kv.x = wk[0] * xa.x + wk[1] * xa.y + wk[2] * xa.z + wk[3] * xa.w;
        kv.y = wk[4] * xa.x + wk[5] * xa.y + wk[6] * xa.z + wk[7] * xa.w;
        kv.z = wk[8] * xa.x + wk[9] * xa.y + wk[10] * xa.z + wk[11] * xa.w;
        kv.w = wk[12] * xa.x + wk[13] * xa.y + wk[14] * xa.z + wk[15] * xa.w;
      } else {
        qv = *reinterpret_cast<const float4*>(a.q + row * inner + c0);
        kv = *reinterpret_cast<const float4*>(a.k + row * inner + c0);
      }
      float f = 0.f, i = 0.f;
#pragma unroll
      for (int h = 0; h < NH; ++h)
        if (h == hh) {
          f = s_f[t][h];
          i = s_i[t][h];
        }
      n.x = f * n.x + i * (kv.x / sqrt_dh);
      n.y = f * n.y + i * (kv.y / sqrt_dh);
      n.z = f * n.z + i * (kv.z / sqrt_dh);
      n.w = f * n.w + i * (kv.w / sqrt_dh);
      const float d = qv.x * n.x + qv.y * n.y + qv.z * n.z + qv.w * n.w;
#pragma unroll
      for (int h = 0; h < NH; ++h) pq[t][h] += (h == hh) ? d : 0.f;
    }
    *reinterpret_cast<float4*>(a.n_state + (int64_t)b * inner + c0) = n;
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float s = wave_sum(pq[t][h]);
      if (lane == 0) red[wave][t * NH + h] = s;
    }
  __syncthreads();
  if (tid < T * NH) {
    const int t = tid / NH, h = tid - t * NH;
    const float qn = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const float denom = fmaxf(fabsf(qn), expf(-s_m[t][h])) + 1e-6f;
    float4 o = make_float4(s_f[t][h], s_i[t][h], denom, s_m[t][h]);
    *reinterpret_cast<float4*>(a.scal + (((int64_t)b * T + t) * NH + h) * 4) = o;
  }
}

template <int T, int NH, bool SINGLE = false>
__global__ __launch_bounds__(kPreThreads) void mlstm_pre_kernel(MlstmPreArgs a) {
  mlstm_pre_body<T, NH, SINGLE>(a, blockIdx.x);
}

// Large-T variant of the front end (context prefill: T = 3 x timesteps-per-chunk, up to kMaxTokens): the same
// arithmetic token by token with a block reduction per token, so the register footprint does not grow with T.
template <int NH>
__global__ __launch_bounds__(kPreThreads) void mlstm_pre_seq_kernel(MlstmPreArgs a) {
  __shared__ float red[4][2 * NH];
  __shared__ float s_gi[NH], s_gf[NH], s_f[NH], s_i[NH], s_m[NH];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = a.inner, T = a.T, DH = inner / NH, ngroups = inner >> 2;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  const float sqrt_dh = sqrtf((float)DH);
  float4 win[kMaxGroups][4], nst[kMaxGroups];
#pragma unroll
  for (int g = 0; g < kMaxGroups; ++g) {
    const int cg = tid + g * kPreThreads;
    const int c0 = cg << 2;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      win[g][k] = (rs || cg >= ngroups) ? f4_zero()
                                        : *reinterpret_cast<const float4*>(a.conv_state + ((int64_t)b * 4 + k) * inner + c0);
    nst[g] = (rs || cg >= ngroups) ? f4_zero() : *reinterpret_cast<const float4*>(a.n_state + (int64_t)b * inner + c0);
  }
  if (tid < NH) s_m[tid] = rs ? 0.f : a.m_state[(int64_t)b * NH + tid];
  auto dot = [](const float4& x, const float4& y) { return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; };
  for (int t = 0; t < T; ++t) {
    const int64_t row = (int64_t)b * T + t;
    float pi[NH], pf[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) pi[h] = pf[h] = 0.f;
    float4 qg[kMaxGroups], kg[kMaxGroups];
#pragma unroll
    for (int g = 0; g < kMaxGroups; ++g) {
      const int cg = tid + g * kPreThreads;
      qg[g] = kg[g] = f4_zero();
      if (cg < ngroups) {
        const int c0 = cg << 2;
        const float4 xm = *reinterpret_cast<const float4*>(a.u + row * 2 * inner + c0);
        win[g][0] = win[g][1];
        win[g][1] = win[g][2];
        win[g][2] = win[g][3];
        win[g][3] = xm;
        float4 cw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) cw[c] = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)(c0 + c) * 4);
        const float4 cb = *reinterpret_cast<const float4*>(a.conv_b + c0);
        float4 y;
        y.x = win[g][0].x * cw[0].x + win[g][1].x * cw[0].y + win[g][2].x * cw[0].z + win[g][3].x * cw[0].w + cb.x;
        y.y = win[g][0].y * cw[1].x + win[g][1].y * cw[1].y + win[g][2].y * cw[1].z + win[g][3].y * cw[1].w + cb.y;
        y.z = win[g][0].z * cw[2].x + win[g][1].z * cw[2].y + win[g][2].z * cw[2].z + win[g][3].z * cw[2].w + cb.z;
        y.w = win[g][0].w * cw[3].x + win[g][1].w * cw[3].y + win[g][2].w * cw[3].z + win[g][3].w * cw[3].w + cb.w;
        const float4 xa = make_float4(silu_f(y.x), silu_f(y.y), silu_f(y.z), silu_f(y.w));
        float4 q, k, v;
        {
          const float4* wq = reinterpret_cast<const float4*>(a.wq + (int64_t)cg * 16);
          const float4* wk = reinterpret_cast<const float4*>(a.wk + (int64_t)cg * 16);
          const float4* wv = reinterpret_cast<const float4*>(a.wv + (int64_t)cg * 16);
          q = make_float4(dot(wq[0], xa), dot(wq[1], xa), dot(wq[2], xa), dot(wq[3], xa));
          k = make_float4(dot(wk[0], xa), dot(wk[1], xa), dot(wk[2], xa), dot(wk[3], xa));
          v = make_float4(dot(wv[0], xm), dot(wv[1], xm), dot(wv[2], xm), dot(wv[3], xm));
        }
        *reinterpret_cast<float4*>(a.q + row * inner + c0) = q;
        *reinterpret_cast<float4*>(a.k + row * inner + c0) = k;
        *reinterpret_cast<float4*>(a.v + row * inner + c0) = v;
        *reinterpret_cast<float4*>(a.xa + row * inner + c0) = xa;
        qg[g] = q;
        kg[g] = k;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const float* wi = a.wi + (int64_t)h * 3 * inner + c0;
          const float* wf = a.wf + (int64_t)h * 3 * inner + c0;
          pi[h] += dot(*reinterpret_cast<const float4*>(wi), q) + dot(*reinterpret_cast<const float4*>(wi + inner), k) +
                   dot(*reinterpret_cast<const float4*>(wi + 2 * inner), v);
          pf[h] += dot(*reinterpret_cast<const float4*>(wf), q) + dot(*reinterpret_cast<const float4*>(wf + inner), k) +
                   dot(*reinterpret_cast<const float4*>(wf + 2 * inner), v);
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float si = wave_sum(pi[h]), sf = wave_sum(pf[h]);
      if (lane == 0) {
        red[wave][2 * h] = si;
        red[wave][2 * h + 1] = sf;
      }
    }
    __syncthreads();
    if (tid < NH) {
      const int h = tid;
      const float gi = red[0][2 * h] + red[1][2 * h] + red[2][2 * h] + red[3][2 * h] + a.bi[h];
      const float gf = red[0][2 * h + 1] + red[1][2 * h + 1] + red[2][2 * h + 1] + red[3][2 * h + 1] + a.bf[h];
      const float m = s_m[h];
      const float lf = log_sigmoid(gf);
      const float mn = fmaxf(lf + m, gi);
      s_f[h] = expf(lf + m - mn);
      s_i[h] = expf(gi - mn);
      s_m[h] = mn;
    }
    __syncthreads();
    float pq[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) pq[h] = 0.f;
#pragma unroll
    for (int g = 0; g < kMaxGroups; ++g) {
      const int cg = tid + g * kPreThreads;
      if (cg < ngroups) {
        const int hh = (cg << 2) / DH;
        float f = 0.f, i = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h)
          if (h == hh) {
            f = s_f[h];
            i = s_i[h];
          }
        float4& n = nst[g];
        n.x = f * n.x + i * (kg[g].x / sqrt_dh);
        n.y = f * n.y + i * (kg[g].y / sqrt_dh);
        n.z = f * n.z + i * (kg[g].z / sqrt_dh);
        n.w = f * n.w + i * (kg[g].w / sqrt_dh);
        const float d = dot(qg[g], n);
#pragma unroll
        for (int h = 0; h < NH; ++h) pq[h] += (h == hh) ? d : 0.f;
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float sq = wave_sum(pq[h]);
      if (lane == 0) red[wave][h] = sq;   // slots [0, NH) of red are free again: gates were consumed above
    }
    __syncthreads();
    if (tid < NH) {
      const int h = tid;
      const float qn = red[0][h] + red[1][h] + red[2][h] + red[3][h];
      const float denom = fmaxf(fabsf(qn), expf(-s_m[h])) + 1e-6f;
      *reinterpret_cast<float4*>(a.scal + (row * NH + h) * 4) = make_float4(s_f[h], s_i[h], denom, s_m[h]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int g = 0; g < kMaxGroups; ++g) {
    const int cg = tid + g * kPreThreads;
    if (cg < ngroups) {
      const int c0 = cg << 2;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * 4 + k) * inner + c0) = win[g][k];
      *reinterpret_cast<float4*>(a.n_state + (int64_t)b * inner + c0) = nst[g];
    }
  }
  if (tid < NH) a.m_state[(int64_t)b * NH + tid] = s_m[tid];
}

// =============================================================================================
// mLSTM cell: C_t = f_t C_{t-1} + i_t (k_t/sqrt(DH)) v_t^T ;  h_t = (q_t^T C_t) / denom_t   for t = 1..T
// One workgroup per (env, head, column slice of CW = 4*LPR columns).  Each lane owns 4 adjacent columns
// (16 B, coalesced: a wave reads 64/LPR rows x LPR*16 B contiguous), the 256/LPR row groups stride over
// the DH rows; every C element is loaded once, updated T times in registers and stored once.
// =============================================================================================
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kCellThreads = 256;

template <int T, int LPR, int kCellUnroll>
__device__ __forceinline__ void mlstm_cell_body(const MlstmCellArgs& a, const int slice, const int h, const int b,
                                                float* smem) {
  constexpr int CW = 4 * LPR;
  constexpr int RP = kCellThreads / LPR;
  const int DH = a.DH;
  float* qs = smem;                 // [T][DH]
  float* aks = smem + T * DH;       // [T][DH]   i_t * k_t / sqrt(DH)
  float* red = smem + 2 * T * DH;   // [RP][T][CW]

  const int tid = threadIdx.x;
  const int cl = tid % LPR, rg = tid / LPR;
  const int inner = a.NH * DH;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;

  float f[T], ig[T], den[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float4 s = *reinterpret_cast<const float4*>(a.scal + (((int64_t)b * T + t) * a.NH + h) * 4);
    f[t] = s.x;
    ig[t] = s.y;
    den[t] = s.z;
  }
  const float sqrt_dh = sqrtf((float)DH);
  for (int r = tid; r < DH; r += kCellThreads) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t off = ((int64_t)b * T + t) * inner + (int64_t)h * DH + r;
      qs[t * DH + r] = a.q[off];
      aks[t * DH + r] = ig[t] * (a.k[off] / sqrt_dh);
    }
  }
  const int col0 = slice * CW + 4 * cl;
  v4f vv[T], acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    vv[t] = *reinterpret_cast<const v4f*>(a.v + ((int64_t)b * T + t) * inner + (int64_t)h * DH + col0);
    acc[t] = (v4f)(0.f);
  }
  __syncthreads();

  // C is streamed exactly once per env-step: non-temporal loads/stores keep it out of the caches' way.
  float* Cb = a.C + (((int64_t)b * a.NH + h) * DH) * DH + col0;
  for (int r0 = rg; r0 < DH; r0 += RP * kCellUnroll) {
    v4f c[kCellUnroll];
#pragma unroll
    for (int u = 0; u < kCellUnroll; ++u) {
      const int r = r0 + u * RP;
      c[u] = (r < DH && !rs) ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(Cb + (int64_t)r * DH))
                             : (v4f)(0.f);
    }
#pragma unroll
    for (int u = 0; u < kCellUnroll; ++u) {
      const int r = r0 + u * RP;
      if (r < DH) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float av = aks[t * DH + r];
          const float qv = qs[t * DH + r];
          c[u] = f[t] * c[u] + av * vv[t];
          acc[t] += qv * c[u];
        }
        __builtin_nontemporal_store(c[u], reinterpret_cast<v4f*>(Cb + (int64_t)r * DH));
      }
    }
  }
  // reduce the RP row groups' partial q^T C
#pragma unroll
  for (int t = 0; t < T; ++t) *reinterpret_cast<v4f*>(red + ((rg * T + t) * CW) + 4 * cl) = acc[t];
  __syncthreads();
  for (int idx = tid; idx < T * CW; idx += kCellThreads) {
    const int t = idx / CW, c = idx - t * CW;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < RP; ++g) s += red[(g * T + t) * CW + c];
    float d = den[0];
#pragma unroll
    for (int tt = 1; tt < T; ++tt) d = (t == tt) ? den[tt] : d;
    a.h[((int64_t)b * T + t) * inner + (int64_t)h * DH + slice * CW + c] = s / d;
  }
}

template <int T, int LPR, int kCellUnroll>
__global__ __launch_bounds__(kCellThreads) void mlstm_cell_kernel(MlstmCellArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  mlstm_cell_body<T, LPR, kCellUnroll>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// =============================================================================================
// MultiHeadLayerNorm (group norm over DH per head) + mLSTM output gating, or + residual (sLSTM).
// One wave per (row, head); DH <= 768.
// =============================================================================================
constexpr int kGnMaxV = 4;  // float4 per lane: head dim <= 1024

__global__ __launch_bounds__(64) void group_norm_kernel(GroupNormArgs a) {
  const int row = blockIdx.x, h = blockIdx.y;
  const int lane = threadIdx.x;
  const int DH = a.DH, D = a.NH * a.DH;
  const int nv = DH >> 2;
  const float* src = a.h + (int64_t)row * D + (int64_t)h * DH;
  float4 v[kGnMaxV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = lane + 64 * j;
    v[j] = i < nv ? *reinterpret_cast<const float4*>(src + 4 * i) : f4_zero();
    s += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  const float mean = wave_sum(s) / (float)DH;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
      const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
      q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
  }
  const float var = wave_sum(q) / (float)DH;
  const float rstd = 1.f / sqrtf(var + a.eps);
  float omax = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i >= nv) continue;
    const int hd = h * DH + 4 * i;
    const float4 g = *reinterpret_cast<const float4*>(a.gamma + hd);
    float4 o;
    o.x = (v[j].x - mean) * rstd * g.x;
    o.y = (v[j].y - mean) * rstd * g.y;
    o.z = (v[j].z - mean) * rstd * g.z;
    o.w = (v[j].w - mean) * rstd * g.w;
    if (a.beta != nullptr) {
      const float4 bb = *reinterpret_cast<const float4*>(a.beta + hd);
      o.x += bb.x;
      o.y += bb.y;
      o.z += bb.z;
      o.w += bb.w;
    }
    float* dst = a.out + (int64_t)row * D + hd;
    if (a.mode == 0) {
      const float4 sk = *reinterpret_cast<const float4*>(a.skip + hd);
      const float4 xa = *reinterpret_cast<const float4*>(a.xa + (int64_t)row * D + hd);
      const float4 z = *reinterpret_cast<const float4*>(a.u + (int64_t)row * 2 * D + D + hd);
      o.x = (o.x + sk.x * xa.x) * silu_f(z.x);
      o.y = (o.y + sk.y * xa.y) * silu_f(z.y);
      o.z = (o.z + sk.z * xa.z) * silu_f(z.z);
      o.w = (o.w + sk.w * xa.w) * silu_f(z.w);
      *reinterpret_cast<float4*>(dst) = o;
      omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    } else {
      float4 x = *reinterpret_cast<const float4*>(dst);
      x.x += o.x;
      x.y += o.y;
      x.z += o.z;
      x.w += o.w;
      *reinterpret_cast<float4*>(dst) = x;
    }
  }
  if (a.mode == 0 && a.amax != nullptr) {
    const float m = wave_max(omax);
    if (lane == 0) a.amax[(int64_t)row * a.NH + h] = m;
  }
}

// =============================================================================================
// sLSTM: conv1d_step + SiLU on the block input (per env, T tokens), reset of the scalar state.
// =============================================================================================
template <int T>
__global__ __launch_bounds__(256) void slstm_conv_kernel(SlstmConvArgs a) {
  const int ngroups = a.D >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)a.B * ngroups) return;
  const int b = (int)(gid / ngroups);
  const int c0 = (int)(gid - (int64_t)b * ngroups) << 2;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  const int D = a.D;
  float4 win[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    win[k] = (rs || k >= a.K) ? f4_zero()
                              : *reinterpret_cast<const float4*>(a.conv_state + ((int64_t)b * a.K + k) * D + c0);
  float4 cw[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) cw[c] = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)(c0 + c) * 4);
  const float4 cb = *reinterpret_cast<const float4*>(a.conv_b + c0);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int64_t row = (int64_t)b * T + t;
    const float4 x = *reinterpret_cast<const float4*>(a.xn + row * D + c0);
    win[0] = win[1];
    win[1] = win[2];
    win[2] = win[3];
    win[3] = x;
    float4 y;
    y.x = win[0].x * cw[0].x + win[1].x * cw[0].y + win[2].x * cw[0].z + win[3].x * cw[0].w + cb.x;
    y.y = win[0].y * cw[1].x + win[1].y * cw[1].y + win[2].y * cw[1].z + win[3].y * cw[1].w + cb.y;
    y.z = win[0].z * cw[2].x + win[1].z * cw[2].y + win[2].z * cw[2].z + win[3].z * cw[2].w + cb.z;
    y.w = win[0].w * cw[3].x + win[1].w * cw[3].y + win[2].w * cw[3].z + win[3].w * cw[3].w + cb.w;
    float4 o;
    o.x = silu_f(y.x);
    o.y = silu_f(y.y);
    o.z = silu_f(y.z);
    o.w = silu_f(y.w);
    *reinterpret_cast<float4*>(a.xc + row * D + c0) = o;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < a.K) *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * a.K + k) * D + c0) = win[k];
  if (rs) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      *reinterpret_cast<float4*>(a.slstm_state + ((int64_t)s * a.state_B + b) * D + c0) = f4_zero();
  }
}

// same, runtime T (prefill chunks)
__global__ __launch_bounds__(256) void slstm_conv_rt_kernel(SlstmConvArgs a) {
  const int ngroups = a.D >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)a.B * ngroups) return;
  const int b = (int)(gid / ngroups);
  const int c0 = (int)(gid - (int64_t)b * ngroups) << 2;
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  const int D = a.D;
  float4 win[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    win[k] = rs ? f4_zero() : *reinterpret_cast<const float4*>(a.conv_state + ((int64_t)b * 4 + k) * D + c0);
  float4 cw[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) cw[c] = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)(c0 + c) * 4);
  const float4 cb = *reinterpret_cast<const float4*>(a.conv_b + c0);
  for (int t = 0; t < a.T; ++t) {
    const int64_t row = (int64_t)b * a.T + t;
    const float4 x = *reinterpret_cast<const float4*>(a.xn + row * D + c0);
    win[0] = win[1];
    win[1] = win[2];
    win[2] = win[3];
    win[3] = x;
    float4 y;
    y.x = win[0].x * cw[0].x + win[1].x * cw[0].y + win[2].x * cw[0].z + win[3].x * cw[0].w + cb.x;
    y.y = win[0].y * cw[1].x + win[1].y * cw[1].y + win[2].y * cw[1].z + win[3].y * cw[1].w + cb.y;
    y.z = win[0].z * cw[2].x + win[1].z * cw[2].y + win[2].z * cw[2].z + win[3].z * cw[2].w + cb.z;
    y.w = win[0].w * cw[3].x + win[1].w * cw[3].y + win[2].w * cw[3].z + win[3].w * cw[3].w + cb.w;
    *reinterpret_cast<float4*>(a.xc + row * D + c0) = make_float4(silu_f(y.x), silu_f(y.y), silu_f(y.z), silu_f(y.w));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * 4 + k) * D + c0) = win[k];
  if (rs) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      *reinterpret_cast<float4*>(a.slstm_state + ((int64_t)s * a.state_B + b) * D + c0) = f4_zero();
  }
}

// sLSTM pointwise cell update for token t ([3P] slstm_pointwise: per-element n == 0 first-step rule).
__global__ __launch_bounds__(256) void slstm_pointwise_kernel(SlstmPointwiseArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  if (gid >= (int64_t)a.B * H) return;
  const int b = (int)(gid / H);
  const int c = (int)(gid - (int64_t)b * H);
  const int64_t row = (int64_t)b * a.T + a.t;
  const float* g = a.gates + row * 4 * H + c;
  const float* r = a.ry + (int64_t)b * 4 * H + c;
  const float iraw = g[0] + r[0] + a.bias[c];
  const float fraw = g[H] + r[H] + a.bias[H + c];
  const float zraw = g[2 * H] + r[2 * H] + a.bias[2 * H + c];
  const float oraw = g[3 * H] + r[3 * H] + a.bias[3 * H + c];
  const int64_t BH = (int64_t)a.state_B * H;
  float* st = a.state + (int64_t)b * H + c;
  const float cs = st[BH], ns = st[2 * BH], ms = st[3 * BH];
  const float logfplusm = ms + log_sigmoid(fraw);
  const float mnew = (ns == 0.f) ? iraw : fmaxf(iraw, logfplusm);
  const float ogate = sigmoid_f(oraw);
  const float igate = fminf(expf(iraw - mnew), 1.f);
  const float fgate = fminf(expf(logfplusm - mnew), 1.f);
  const float cnew = fgate * cs + igate * tanhf(zraw);
  const float nnew = fgate * ns + igate;
  const float ynew = ogate * cnew / nnew;
  st[0] = ynew;
  st[BH] = cnew;
  st[2 * BH] = nnew;
  st[3 * BH] = mnew;
  a.yout[row * H + c] = ynew;
}

// sLSTM token step for few env rows: the head's recurrent projection R_g h_{t-1} and the pointwise cell in ONE launch, where
// the generic path runs a batched per-head GEMM (a 128-row bf16x3 tile for, say, 64 rows: ~24 us of fixed latency) and the
// pointwise kernel after it.  One workgroup = 32 envs x 8 channels x 4 gates of one head: a single 32 x 32 tile of the exact
// fp32 matrix-core instruction (v_mfma_f32_32x32x2_f32), K = SDH split over the four waves.  Each lane requests its
// operands straight from global memory -- its env's h_{t-1} row and its output's R row over the wave's K range, contiguous
// 16-byte runs, all in flight together (one memory round trip, no LDS staging) -- the four partial tiles meet in LDS, then
// thread (env, channel) applies slstm_pointwise_kernel's cell update ([3P] slstm_pointwise) to its four gate sums.
// h_{t-1} is read from `hprev` (the state's h plane for the first token of a launch sequence, the previous token's rows of
// yout afterwards) and h_t goes to yout -- and to the state's h plane only when a.write_h is set (never in a launch that
// reads that plane: other workgroups still need it).
constexpr int kSlTokCh = 8, kSlTokEnv = 32;
typedef float sl_f32x16 __attribute__((ext_vector_type(16)));
template <int SDH>
__global__ __launch_bounds__(256) void slstm_token_kernel(SlstmTokenArgs a) {
  __shared__ float part[4][kSlTokEnv][33];
  const int H = a.H;
  constexpr int ncb = SDH / kSlTokCh;
  constexpr int S = SDH / 8;  // K steps per lane: wave w, lane half lh own k = w * SDH/4 + lh * SDH/8 + [0, S)
  static_assert(S % 4 == 0, "sLSTM token kernel: head dim must be a multiple of 32");
  const int cb = blockIdx.x % ncb, head = blockIdx.x / ncb;
  const int b0 = blockIdx.y * kSlTokEnv;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int kbase = w * (SDH / 4) + lh * S;
  // operands: A row = env li, B row = output n = li = 4 * channel + gate
  const int bl = min(b0 + li, a.B - 1);
  const float* ap = a.hprev + (int64_t)bl * a.hprev_ld + head * SDH + kbase;
  const float* bp = a.rt + (((int64_t)head * 4 + (li & 3)) * SDH + cb * kSlTokCh + (li >> 2)) * SDH + kbase;
  float4 av[S / 4], bv[S / 4];
#pragma unroll
  for (int i = 0; i < S / 4; ++i) av[i] = *reinterpret_cast<const float4*>(ap + 4 * i);
#pragma unroll
  for (int i = 0; i < S / 4; ++i) bv[i] = *reinterpret_cast<const float4*>(bp + 4 * i);
  // this thread's cell: env e, channel c (requested before the matrix products need their operands)
  const int e = tid >> 3, ch = tid & 7;
  const int c = head * SDH + cb * kSlTokCh + ch;
  const int b = b0 + e;
  const bool live = b < a.B;
  const int bb = live ? b : a.B - 1;
  const int64_t row = (int64_t)bb * a.T + a.t;
  const float* gp = a.gates + row * 4 * H + c;
  const float gi = gp[0] + a.bias[c], gf = gp[H] + a.bias[H + c], gz = gp[2 * H] + a.bias[2 * H + c],
              go = gp[3 * H] + a.bias[3 * H + c];
  const int64_t BH = (int64_t)a.state_B * H;
  float* st = a.state + (int64_t)bb * H + c;
  const float cs = st[BH], ns = st[2 * BH], ms = st[3 * BH];
  // every request above is issued before the first matrix product (left alone, the scheduler feeds the products with one
  // or two loads in flight: S / 4 dependent round trips instead of one)
  __builtin_amdgcn_sched_barrier(0);
  sl_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int i = 0; i < S / 4; ++i) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[i].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[i].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[i].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[i].w, acc, 0, 0, 0);
  }
  // lane (li, lh) holds column n = li of rows (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
  for (int r = 0; r < 16; ++r) part[w][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
  __syncthreads();
  if (!live) return;
  float sum[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
    sum[g] = (part[0][e][4 * ch + g] + part[1][e][4 * ch + g]) + (part[2][e][4 * ch + g] + part[3][e][4 * ch + g]);
  const float iraw = gi + sum[0], fraw = gf + sum[1], zraw = gz + sum[2], oraw = go + sum[3];
  const float logfplusm = ms + log_sigmoid(fraw);
  const float mnew = (ns == 0.f) ? iraw : fmaxf(iraw, logfplusm);
  const float ogate = sigmoid_f(oraw);
  const float igate = fminf(expf(iraw - mnew), 1.f);
  const float fgate = fminf(expf(logfplusm - mnew), 1.f);
  const float cnew = fgate * cs + igate * tanhf(zraw);
  const float nnew = fgate * ns + igate;
  const float ynew = ogate * cnew / nnew;
  if (a.write_h) st[0] = ynew;
  st[BH] = cnew;
  st[2 * BH] = nnew;
  st[3 * BH] = mnew;
  a.yout[row * H + c] = ynew;
}

// The same norm + skip + gate with the output written as proj_down's pre-split operand (GroupNormArgs::h2): one workgroup per row,
// wave = head; the row's largest magnitude over all heads meets in LDS, then every wave scales and splits its own segment.  Same
// values as group_norm_kernel's mode 0 (same order of operations); the planes hold exactly what gemm_f16x2.hip would make of the
// fp32 row with that maximum, so the projection's result is bit-identical to the fp32 + partial-maxima hand-over.
__global__ __launch_bounds__(512) void group_norm_planes_kernel(GroupNormArgs a) {
  __shared__ float s_max[8];
  const int row = blockIdx.x, h = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int DH = a.DH, D = a.NH * a.DH;
  const int nv = DH >> 2;
  const float* src = a.h + (int64_t)row * D + (int64_t)h * DH;
  float4 v[kGnMaxV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = lane + 64 * j;
    v[j] = i < nv ? *reinterpret_cast<const float4*>(src + 4 * i) : f4_zero();
    s += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  const float mean = wave_sum(s) / (float)DH;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
      const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
      q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
  }
  const float var = wave_sum(q) / (float)DH;
  const float rstd = 1.f / sqrtf(var + a.eps);
  float omax = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i >= nv) continue;
    const int hd = h * DH + 4 * i;
    const float4 g = *reinterpret_cast<const float4*>(a.gamma + hd);
    float4 o;
    o.x = (v[j].x - mean) * rstd * g.x;
    o.y = (v[j].y - mean) * rstd * g.y;
    o.z = (v[j].z - mean) * rstd * g.z;
    o.w = (v[j].w - mean) * rstd * g.w;
    if (a.beta != nullptr) {
      const float4 bb = *reinterpret_cast<const float4*>(a.beta + hd);
      o.x += bb.x;
      o.y += bb.y;
      o.z += bb.z;
      o.w += bb.w;
    }
    const float4 sk = *reinterpret_cast<const float4*>(a.skip + hd);
    const float4 xa = *reinterpret_cast<const float4*>(a.xa + (int64_t)row * D + hd);
    const float4 z = *reinterpret_cast<const float4*>(a.u + (int64_t)row * 2 * D + D + hd);
    o.x = (o.x + sk.x * xa.x) * silu_f(z.x);
    o.y = (o.y + sk.y * xa.y) * silu_f(z.y);
    o.z = (o.z + sk.z * xa.z) * silu_f(z.z);
    o.w = (o.w + sk.w * xa.w) * silu_f(z.w);
    v[j] = o;
    omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  omax = wave_max(omax);
  if (lane == 0) s_max[h] = omax;
  __syncthreads();
  float mx = s_max[0];
  for (int k = 1; k < a.NH; ++k) mx = fmaxf(mx, s_max[k]);
  const float sc = pow2_scale(mx);
  _Float16* h2 = reinterpret_cast<_Float16*>(a.h2);
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i >= nv) continue;
    const int k0 = h * DH + 4 * i;   // (DH a multiple of 4: the four elements stay inside one 32-deep K tile)
    split2_store4(v[j], sc, h2 + (int64_t)(k0 >> 5) * a.h2_kt + (int64_t)row * 32 + (k0 & 31), a.h2_plane);
  }
  if (threadIdx.x == 0) a.h2_inv[row] = 1.f / sc;
}

__global__ __launch_bounds__(256) void gelu_gate_kernel(const float* p, float* out, int64_t rows, int F) {
  const int nv = F >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * nv) return;
  const int64_t r = gid / nv;
  const int c = (int)(gid - r * nv) << 2;
  const float4 g = *reinterpret_cast<const float4*>(p + r * 2 * F + c);
  const float4 u = *reinterpret_cast<const float4*>(p + r * 2 * F + F + c);
  float4 o;
  o.x = gelu_f(g.x) * u.x;
  o.y = gelu_f(g.y) * u.y;
  o.z = gelu_f(g.z) * u.z;
  o.w = gelu_f(g.w) * u.w;
  *reinterpret_cast<float4*>(out + r * F + c) = o;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int T>
static void launch_pre_t(const MlstmPreArgs& a, hipStream_t s) {
  dim3 grid(a.B), block(kPreThreads);
  // one channel group per thread: the variant with one dependent memory round trip
  const bool single = (a.inner >> 2) <= kPreThreads;
  switch (a.NH) {
    case 1: hipLaunchKernelGGL((mlstm_pre_kernel<T, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((mlstm_pre_kernel<T, 2>), grid, block, 0, s, a); break;
    case 4:
      if (single)
        hipLaunchKernelGGL((mlstm_pre_kernel<T, 4, true>), grid, block, 0, s, a);
      else
        hipLaunchKernelGGL((mlstm_pre_kernel<T, 4>), grid, block, 0, s, a);
      break;
    case 8: hipLaunchKernelGGL((mlstm_pre_kernel<T, 8>), grid, block, 0, s, a); break;
    default: throw Error("lram: mLSTM num_heads must be 1, 2, 4 or 8");
  }
}

void launch_mlstm_pre(const MlstmPreArgs& a, hipStream_t stream) {
  if (a.T > kMaxTokens) return launch_mlstm_chunk_pre(a, stream);
  LRAM_REQUIRE(a.K >= 1 && a.K <= 4, "mLSTM conv1d_kernel_size must be in 1..4");
  LRAM_REQUIRE(a.inner % (4 * a.NH) == 0, "mLSTM inner dim must be a multiple of 4*num_heads");
  LRAM_REQUIRE(a.inner <= 4 * kPreThreads * kMaxGroups, "mLSTM inner dim too large");
  LRAM_REQUIRE(a.K == 4, "conv weights are packed with 4 taps per channel");
  switch (a.T) {
    case 1: launch_pre_t<1>(a, stream); break;
    case 2: launch_pre_t<2>(a, stream); break;
    case 3: launch_pre_t<3>(a, stream); break;
    case 4: launch_pre_t<4>(a, stream); break;
    default: {
      LRAM_REQUIRE(a.T >= 1 && a.T <= kMaxTokens, "tokens per launch out of range");
      dim3 grid(a.B), block(kPreThreads);
      switch (a.NH) {
        case 1: hipLaunchKernelGGL(mlstm_pre_seq_kernel<1>, grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL(mlstm_pre_seq_kernel<2>, grid, block, 0, stream, a); break;
        case 4: hipLaunchKernelGGL(mlstm_pre_seq_kernel<4>, grid, block, 0, stream, a); break;
        case 8: hipLaunchKernelGGL(mlstm_pre_seq_kernel<8>, grid, block, 0, stream, a); break;
        default: throw Error("lram: mLSTM num_heads must be 1, 2, 4 or 8");
      }
    }
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

template <int T, int LPR, int UNR>
static void launch_cell_tlu(const MlstmCellArgs& a, hipStream_t s) {
  constexpr int CW = 4 * LPR, RP = kCellThreads / LPR;
  dim3 grid(a.DH / CW, a.NH, a.B), block(kCellThreads);
  size_t shmem = sizeof(float) * (2 * T * a.DH + RP * T * CW);
  // Occupancy cap: `min_lds_bytes` of LDS per workgroup limits how many workgroups share a CU.  Fewer, longer
  // sequential streams per CU use HBM better, and the cap leaves LDS / registers for another slice's fp32-MFMA
  // GEMM workgroups (37 KB LDS each) when the engine overlaps the two.
  shmem = std::max(shmem, (size_t)a.min_lds_bytes);
  if (shmem > 48 * 1024) {
    static uint64_t raised = 0;
    if (first_use_on_device(raised)) {
      LRAM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlstm_cell_kernel<T, LPR, UNR>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
  }
  hipLaunchKernelGGL((mlstm_cell_kernel<T, LPR, UNR>), grid, block, shmem, s, a);
}

template <int T, int LPR>
static void launch_cell_tl(const MlstmCellArgs& a, hipStream_t s) {
  if (a.unroll >= 32)
    launch_cell_tlu<T, LPR, 32>(a, s);
  else if (a.unroll >= 16)
    launch_cell_tlu<T, LPR, 16>(a, s);
  else
    launch_cell_tlu<T, LPR, 8>(a, s);
}

template <int T>
static void launch_cell_t(const MlstmCellArgs& a, hipStream_t s) {
  // Few (env, head) pairs: cut the head's columns into narrower slices so that at least ~256 workgroups exist
  // (a single env at DH = 256 gets 16 workgroups of 64 columns instead of 4 of 256).
  const long pairs = (long)a.B * a.NH;
  if (a.DH % 256 == 0 && pairs * (a.DH / 256) >= 256)
    launch_cell_tl<T, 64>(a, s);
  else if (a.DH % 128 == 0 && pairs * (a.DH / 128) >= 256)
    launch_cell_tl<T, 32>(a, s);
  else if (a.DH % 64 == 0)
    launch_cell_tl<T, 16>(a, s);
  else if (a.DH % 16 == 0)  // the reference's *_half presets (head dims 352, 544, 720): 16-column slices
    launch_cell_tl<T, 4>(a, s);
  else
    throw Error("lram: mLSTM head dim must be a multiple of 16");
}

void launch_mlstm_cell(const MlstmCellArgs& a, hipStream_t stream) {
  if (a.T > kMaxTokens) return launch_mlstm_chunk_cell(a, stream);
  switch (a.T) {
    case 1: launch_cell_t<1>(a, stream); break;
    case 2: launch_cell_t<2>(a, stream); break;
    case 3: launch_cell_t<3>(a, stream); break;
    case 4: launch_cell_t<4>(a, stream); break;
    case 6: launch_cell_t<6>(a, stream); break;
    case 9: launch_cell_t<9>(a, stream); break;
    case 12: launch_cell_t<12>(a, stream); break;
    default: throw Error("lram: tokens per launch must be 1..4, 6, 9 or 12");
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_group_norm(const GroupNormArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(a.DH % 4 == 0 && a.DH <= 4 * 64 * kGnMaxV, "group norm: head dim must be a multiple of 4 and <= 1024");
  if (a.h2 != nullptr) {
    LRAM_REQUIRE(a.mode == 0 && a.NH >= 1 && a.NH <= 8 && a.h2_inv != nullptr && a.h2_kt >= 32 * (int64_t)a.rows && (a.NH * a.DH) % 32 == 0,
                 "group norm: operand planes need mode 0, <= 8 heads, the inverse-scale output and the K-tile pitch");
    hipLaunchKernelGGL(group_norm_planes_kernel, dim3(a.rows), dim3(64 * a.NH), 0, stream, a);
    LRAM_HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(group_norm_kernel, dim3(a.rows, a.NH), dim3(64), 0, stream, a);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_slstm_conv(const SlstmConvArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(a.K == 4 && a.D % 4 == 0, "sLSTM conv: K must be 4 and D a multiple of 4");
  const int64_t n = (int64_t)a.B * (a.D >> 2);
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (a.T) {
    case 1: hipLaunchKernelGGL(slstm_conv_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(slstm_conv_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(slstm_conv_kernel<3>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(slstm_conv_kernel<4>, grid, block, 0, stream, a); break;
    default:
      LRAM_REQUIRE(a.T >= 1 && a.T <= kChunkMaxTokens, "tokens per launch out of range");
      hipLaunchKernelGGL(slstm_conv_rt_kernel, grid, block, 0, stream, a);
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

bool slstm_token_supported(int H, int NH) {
  const int SDH = NH > 0 ? H / NH : 0;
  return NH > 0 && H % NH == 0 && (SDH == 32 || SDH == 64 || SDH == 128 || SDH == 192 || SDH == 256 || SDH == 320);
}

template <int SDH>
void launch_slstm_token_sdh(const SlstmTokenArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.NH * (SDH / kSlTokCh)), (unsigned)((a.B + kSlTokEnv - 1) / kSlTokEnv));
  hipLaunchKernelGGL(slstm_token_kernel<SDH>, grid, dim3(256), 0, stream, a);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_slstm_token(const SlstmTokenArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(slstm_token_supported(a.H, a.NH), "sLSTM token kernel: head dim must be 32, 64, 128, 192, 256 or 320");
  LRAM_REQUIRE(!(a.write_h && a.hprev == a.state), "sLSTM token kernel: a launch that reads the state's h plane must not write it");
  switch (a.H / a.NH) {
    case 32: return launch_slstm_token_sdh<32>(a, stream);
    case 64: return launch_slstm_token_sdh<64>(a, stream);
    case 128: return launch_slstm_token_sdh<128>(a, stream);
    case 192: return launch_slstm_token_sdh<192>(a, stream);
    case 256: return launch_slstm_token_sdh<256>(a, stream);
    default: return launch_slstm_token_sdh<320>(a, stream);
  }
}

void launch_slstm_pointwise(const SlstmPointwiseArgs& a, hipStream_t stream) {
  const int64_t n = (int64_t)a.B * a.H;
  hipLaunchKernelGGL(slstm_pointwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_gelu_gate(const float* p, float* out, int rows, int F, hipStream_t stream) {
  LRAM_REQUIRE(F % 4 == 0, "gelu gate: F must be a multiple of 4");
  const int64_t n = (int64_t)rows * (F >> 2);
  hipLaunchKernelGGL(gelu_gate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, out,
                     (int64_t)rows, F);
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
