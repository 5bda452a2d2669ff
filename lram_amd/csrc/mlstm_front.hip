// mLSTM step front end for the large-batch lazy path on gfx950: several env slots per workgroup.
//
// Same arithmetic as mlstm_pre_kernel (xlstm_kernels.hip) in its lean form -- conv1d_step (4 taps, depthwise) -> SiLU ->
// gate pre-activations Wi.[q,k,v] + bi, Wf.[q,k,v] + bf -> stabilised gate scalars f_t, i_t, m_t -> normaliser state n_t
// and denominators max(|q_t . n_t|, exp(-m_t)) + 1e-6 -- for T = 3 tokens of one env-step (conv1d_step,
// LinearHeadwiseExpand q / k / v, mLSTMCell igate / fgate, recurrent_step_stabilized_simple; reference call site
// src/algos/models/decision_xlstm.py:159-163).  What differs is how the work is laid out:
//
//   * mlstm_pre_kernel runs one workgroup per env.  Every workgroup pulls the block's 166 KB of conv / q / k / v / gate weights
//     out of L2 again, and walks three dependent block reductions and the serial gate chain with nothing else to do: one
//     workgroup alone takes 13.5 us, a 2048-env launch 182-215 us beside a read pass for 128 MB of HBM traffic (9 % of
//     peak, profiles/r03_step_timeline_xlstm16m_b4096.txt) -- the slowest link of the slice chain.
//   * Here a workgroup keeps the weights of its 256 channel groups in REGISTERS and loops over `epw` env slots; the next
//     env's operands (conv window, n, the x_m rows) are requested while the current env's gate sums are reduced and its
//     scalar chain runs.  The geometry it is built for -- inner = 1024, four heads of 256 channels -- makes wave w own head
//     w: the q . n reduction is wave-local, and one workgroup barrier per env (gate partial sums of the four waves) is all
//     that is left of the three.
//   * The gate projections act on [q, k, v] = [Wq xa, Wk xa, Wv x_m] with 4 x 4 block-diagonal Wq / Wk / Wv, so
//     Wi . [q, k, v] = a_i . xa + b_i . x_m with a_i = Wq^T wi_q + Wk^T wi_k, b_i = Wv^T wi_v folded once per weight upload
//     (gate_coef_kernel, products and sums in fp64, rounded once): 16 coefficients per channel instead of 24 weights plus
//     the 4 x 4 blocks, and v is not formed at all (the lazy read pass rebuilds q, k, v from xa / x_m: "lean" front end).
//     The pre-activations differ from the unfolded order by fp32 rounding only.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {

namespace {

// gc[cg][j][c]: j < 8: coefficient of xa[4 cg + c] for gate j (j < 4: input gate of head j, else forget gate of head j - 4);
// j >= 8: coefficient of x_m[4 cg + c] for gate j - 8.
__global__ __launch_bounds__(256) void gate_coef_kernel(const float* wq, const float* wk, const float* wv, const float* wi,
                                                        const float* wf, int inner, int NH, float* gc) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int ngroups = inner >> 2;
  if (gid >= ngroups * 2 * NH * 4) return;
  const int c = gid & 3, j = (gid >> 2) % (2 * NH), cg = gid / (8 * NH);
  const float* wg = (j < NH ? wi + (int64_t)j * 3 * inner : wf + (int64_t)(j - NH) * 3 * inner) + 4 * cg;
  double a = 0.0, b = 0.0;
  for (int o = 0; o < 4; ++o) {  // output channel o of the 4 x 4 block: q[o] = sum_c wq[cg][o][c] xa[c]
    a += (double)wg[o] * (double)wq[(int64_t)cg * 16 + o * 4 + c] + (double)wg[inner + o] * (double)wk[(int64_t)cg * 16 + o * 4 + c];
    b += (double)wg[2 * inner + o] * (double)wv[(int64_t)cg * 16 + o * 4 + c];
  }
  gc[((int64_t)cg * 4 * NH + j) * 4 + c] = (float)a;
  gc[((int64_t)cg * 4 * NH + 2 * NH + j) * 4 + c] = (float)b;
}

__device__ __forceinline__ float dot4(const float4& x, const float4& y) { return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }
__device__ __forceinline__ float4 and4(const float4& v, unsigned keep) {
  return make_float4(__uint_as_float(__float_as_uint(v.x) & keep), __uint_as_float(__float_as_uint(v.y) & keep),
                     __uint_as_float(__float_as_uint(v.z) & keep), __uint_as_float(__float_as_uint(v.w) & keep));
}

// One step of a recursive-halving wave reduction: v[0 .. 2 n) -> r[0 .. n), lanes with bit M set keep the upper half.
template <int N, int M>
__device__ __forceinline__ void halve(const float* v, float* r, int lane) {
  const bool up = (lane & M) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float send = up ? v[i] : v[i + N];
    const float keep = up ? v[i + N] : v[i];
    r[i] = keep + __shfl_xor(send, M, 64);
  }
}

constexpr int kNH = 4;
constexpr int kNG = 2 * kNH;  // gates per token

template <int T>
__global__ __launch_bounds__(256) void mlstm_front_kernel(MlstmFrontArgs a) {
  static_assert(T == 3, "the reduction tree below is laid out for 3 tokens x 8 gates");
  __shared__ float red[2][4][T * kNG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int inner = 4 * 256, c0 = tid << 2, h = wave;
  const int e_begin = blockIdx.x * a.epw, e_end = min(a.B, e_begin + a.epw);
  if (e_begin >= e_end) return;

  // ---- the workgroup's weights, once ----
  float4 cw[4], wq[4], wk[4], gc[2 * kNG];
#pragma unroll
  for (int c = 0; c < 4; ++c) cw[c] = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)(c0 + c) * 4);
  const float4 cb = *reinterpret_cast<const float4*>(a.conv_b + c0);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    wq[o] = *reinterpret_cast<const float4*>(a.wq + (int64_t)tid * 16 + o * 4);
    wk[o] = *reinterpret_cast<const float4*>(a.wk + (int64_t)tid * 16 + o * 4);
  }
#pragma unroll
  for (int j = 0; j < 2 * kNG; ++j) gc[j] = *reinterpret_cast<const float4*>(a.gc + ((int64_t)tid * 2 * kNG + j) * 4);
  const float bias_i = a.bi[h], bias_f = a.bf[h];
  const float sqrt_dh = 16.f;  // sqrt(256)

  // ---- operands of one env: conv window, n, the T x_m rows, m of this wave's head, the restart flag ----
  float4 win[4], nst, xm[T];
  float m_in;
  unsigned rsb;
  const uint8_t* rbase = a.reset != nullptr ? a.reset : reinterpret_cast<const uint8_t*>(a.conv_w);
  auto request = [&](int b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) win[k] = *reinterpret_cast<const float4*>(a.conv_state + ((int64_t)b * 4 + k) * inner + c0);
    nst = *reinterpret_cast<const float4*>(a.n_state + (int64_t)b * inner + c0);
#pragma unroll
    for (int t = 0; t < T; ++t) xm[t] = *reinterpret_cast<const float4*>(a.u + ((int64_t)b * T + t) * a.ldu + c0);
    m_in = a.m_state[(int64_t)b * kNH + h];
    rsb = rbase[a.reset != nullptr ? b : 0];
  };
  request(e_begin);

  for (int b = e_begin; b < e_end; ++b) {
    const int buf = (b - e_begin) & 1;
    // restart: state is requested whether or not the env restarts and zeroed by a bit mask (no control flow on the flag)
    const unsigned keep = (a.reset != nullptr && rsb != 0) ? 0u : 0xFFFFFFFFu;
    float4 w0 = and4(win[0], keep), w1 = and4(win[1], keep), w2 = and4(win[2], keep), w3 = and4(win[3], keep);
    float4 n = and4(nst, keep);
    float m = __uint_as_float(__float_as_uint(m_in) & keep);
    const float4 x0 = xm[0], x1 = xm[1], x2 = xm[2];

    // ---- conv -> SiLU, q / k, gate partial sums ----
    float4 q[T], k[T];
    float pg[T * kNG];
    auto token = [&](int t, const float4& p0, const float4& p1, const float4& p2, const float4& p3) {
      float4 y;
      y.x = p0.x * cw[0].x + p1.x * cw[0].y + p2.x * cw[0].z + p3.x * cw[0].w + cb.x;
      y.y = p0.y * cw[1].x + p1.y * cw[1].y + p2.y * cw[1].z + p3.y * cw[1].w + cb.y;
      y.z = p0.z * cw[2].x + p1.z * cw[2].y + p2.z * cw[2].z + p3.z * cw[2].w + cb.z;
      y.w = p0.w * cw[3].x + p1.w * cw[3].y + p2.w * cw[3].z + p3.w * cw[3].w + cb.w;
      const float4 xa = make_float4(silu_f(y.x), silu_f(y.y), silu_f(y.z), silu_f(y.w));
      *reinterpret_cast<float4*>(a.xa + ((int64_t)b * T + t) * inner + c0) = xa;
      q[t] = make_float4(dot4(wq[0], xa), dot4(wq[1], xa), dot4(wq[2], xa), dot4(wq[3], xa));
      k[t] = make_float4(dot4(wk[0], xa), dot4(wk[1], xa), dot4(wk[2], xa), dot4(wk[3], xa));
#pragma unroll
      for (int j = 0; j < kNG; ++j) pg[t * kNG + j] = dot4(gc[j], xa) + dot4(gc[kNG + j], p3);
    };
    token(0, w1, w2, w3, x0);
    token(1, w2, w3, x0, x1);
    token(2, w3, x0, x1, x2);
    // conv state after the T tokens (reference layout [B, K, inner], newest tap last)
    *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * 4 + 0) * inner + c0) = w3;
    *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * 4 + 1) * inner + c0) = x0;
    *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * 4 + 2) * inner + c0) = x1;
    *reinterpret_cast<float4*>(a.conv_state + ((int64_t)b * 4 + 3) * inner + c0) = x2;

    // ---- the next env's operands: in flight while this env's sums are reduced and its scalar chain runs ----
    // (unconditional: past the last env the last one is requested again and dropped -- a load under a condition makes
    // hipcc wait for every outstanding request at its first use)
    request(min(b + 1, e_end - 1));

    // ---- wave reduction of the 24 partial sums by recursive halving (30 shuffles instead of 144): 24 -> 12 -> 6 -> 3
    // values per lane over lane bits 5, 4, 3, then the last three over bits 2, 1, 0 ----
    float r12[12], r6[6], r3[3];
    halve<12, 32>(pg, r12, lane);
    halve<6, 16>(r12, r6, lane);
    halve<3, 8>(r6, r3, lane);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      r3[i] += __shfl_xor(r3[i], 4, 64);
      r3[i] += __shfl_xor(r3[i], 2, 64);
      r3[i] += __shfl_xor(r3[i], 1, 64);
    }
    if ((lane & 7) == 0) {
      const int idx = ((lane & 32) ? 12 : 0) + ((lane & 16) ? 6 : 0) + ((lane & 8) ? 3 : 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) red[buf][wave][idx + i] = r3[i];
    }
    __syncthreads();

    // ---- stabilised gate scalars of this wave's head (every lane computes the same chain) ----
    float f[T], ig[T], mt[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float gi = red[buf][0][t * kNG + h] + red[buf][1][t * kNG + h] + red[buf][2][t * kNG + h] + red[buf][3][t * kNG + h] + bias_i;
      const float gf = red[buf][0][t * kNG + kNH + h] + red[buf][1][t * kNG + kNH + h] + red[buf][2][t * kNG + kNH + h] +
                       red[buf][3][t * kNG + kNH + h] + bias_f;
      const float lf = log_sigmoid(gf);
      const float mn = fmaxf(lf + m, gi);
      f[t] = expf(lf + m - mn);
      ig[t] = expf(gi - mn);
      mt[t] = mn;
      m = mn;
    }
    // ---- normaliser state and q_t . n_t (wave-local: the wave's 256 channels are the head) ----
    float d[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      n.x = f[t] * n.x + ig[t] * (k[t].x / sqrt_dh);
      n.y = f[t] * n.y + ig[t] * (k[t].y / sqrt_dh);
      n.z = f[t] * n.z + ig[t] * (k[t].z / sqrt_dh);
      n.w = f[t] * n.w + ig[t] * (k[t].w / sqrt_dh);
      d[t] = wave_sum(dot4(q[t], n));
    }
    *reinterpret_cast<float4*>(a.n_state + (int64_t)b * inner + c0) = n;
    if (lane == 0) {
      a.m_state[(int64_t)b * kNH + h] = m;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float denom = fmaxf(fabsf(d[t]), expf(-mt[t])) + 1e-6f;
        *reinterpret_cast<float4*>(a.scal + (((int64_t)b * T + t) * kNH + h) * 4) = make_float4(f[t], ig[t], denom, mt[t]);
      }
    }
  }
}

}  // namespace

bool mlstm_front_supported(int inner, int NH, int K, int T) { return inner == 1024 && NH == kNH && K == 4 && T == 3; }

void launch_gate_coef(const float* wq, const float* wk, const float* wv, const float* wi, const float* wf, int inner, int NH,
                      float* gc, hipStream_t stream) {
  const int n = (inner >> 2) * 2 * NH * 4;
  hipLaunchKernelGGL(gate_coef_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, wq, wk, wv, wi, wf, inner, NH, gc);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_mlstm_front(const MlstmFrontArgs& a_in, hipStream_t stream) {
  MlstmFrontArgs a = a_in;
  LRAM_REQUIRE(mlstm_front_supported(a.inner, a.NH, a.K, a.T), "mLSTM multi-env front end: unsupported geometry");
  LRAM_REQUIRE(a.gc != nullptr && (a.ldu & 3) == 0, "mLSTM multi-env front end: missing gate coefficients / misaligned u");
  // LRAM_FRONT_EPW (measurement knob): env slots per workgroup
  static const int epw_env = [] {
    const char* v = std::getenv("LRAM_FRONT_EPW");
    return v ? std::atoi(v) : 0;
  }();
  // (same box, 4096 env slots: 2 envs per workgroup 426.2k env-steps/s, 4: 429.4k, 8: 431.1k / 429.7k, 16: 430.8k / 431.9k;
  // the one-workgroup-per-env kernel 420.3k / 419.0k -- profiles/r04_ab_front_kernel.txt)
  // fewer envs per workgroup for smaller slices, so that the launch still covers the chip: one workgroup per CU from 256 envs
  // (512-env slices at 8 envs per workgroup are 64 workgroups: 58.7 us per launch in the 1024-slot timeline)
  a.epw = epw_env > 0 ? epw_env : (a.epw > 0 ? a.epw : std::max(1, std::min(8, a.B / 256)));
  const int nwg = (a.B + a.epw - 1) / a.epw;
  hipLaunchKernelGGL(mlstm_front_kernel<3>, dim3((unsigned)nwg), dim3(256), 0, stream, a);
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
