// Mamba-1 recurrent step kernels for gfx950 (T tokens of one env-step per launch).
//
// Replace, on the reference path: causal_conv1d_update (CUDA) and
// selective_state_update (Triton) as called from [3P] mamba_ssm Mamba.step via
// src/algos/models/decision_mamba.py:130-147.  State layouts are the reference's:
// conv_state [B, d_inner, d_conv], ssm_state [B, d_inner, d_state], fp32.
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {
namespace {
typedef float v4f_t __attribute__((ext_vector_type(4)));

// conv_state.roll(-1); conv_state[..., -1] = x;  xc = silu(sum_k conv_state[.., k] * w[d, k] + b[d])
template <int T>
__global__ __launch_bounds__(256) void mamba_conv_kernel(MambaConvArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int di = a.d_inner;
  if (gid >= (int64_t)a.B * di) return;
  const int b = (int)(gid / di);
  const int d = (int)(gid - (int64_t)b * di);
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  float4 win = rs ? f4_zero() : *reinterpret_cast<const float4*>(a.conv_state + gid * 4);
  const float4 w = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)d * 4);
  const float bias = a.conv_b != nullptr ? a.conv_b[d] : 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int64_t row = (int64_t)b * T + t;
    const float x = a.xz[row * 2 * di + d];
    win.x = win.y;
    win.y = win.z;
    win.z = win.w;
    win.w = x;
    const float y = win.x * w.x + win.y * w.y + win.z * w.z + win.w * w.w + bias;
    const float o = silu_hw(y);
    a.xc[row * di + d] = o;
    if (a.amax != nullptr) {  // (d_inner is a multiple of 64: the lanes of a wave share the env, hence the row)
      const float m = wave_max_nonneg_lane63(fabsf(o));
      if ((threadIdx.x & 63) == 63) a.amax[row * (di >> 6) + (d >> 6)] = m;  // this wave's 64 channels of the row
    }
  }
  *reinterpret_cast<float4*>(a.conv_state + gid * 4) = win;
}

// selective_state_update for T tokens.  One workgroup = one env x 64 channels; 4 lanes per channel, each
// owning 4 of the N = 16 states (16 B per lane, consecutive lanes consecutive addresses).  The env's B_t, C_t
// vectors (shared by all its channels) are staged once in LDS and read back as broadcasts.
//   dt = softplus(dt_proj(dt_raw) + dt_bias);  s = s * exp(dt * A) + x * (dt * B);  y = s . C + D x;  y *= silu(z)
// same, runtime T (prefill chunks)
__global__ __launch_bounds__(256) void mamba_conv_rt_kernel(MambaConvArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int di = a.d_inner;
  if (gid >= (int64_t)a.B * di) return;
  const int b = (int)(gid / di);
  const int d = (int)(gid - (int64_t)b * di);
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  float4 win = rs ? f4_zero() : *reinterpret_cast<const float4*>(a.conv_state + gid * 4);
  const float4 w = *reinterpret_cast<const float4*>(a.conv_w + (int64_t)d * 4);
  const float bias = a.conv_b != nullptr ? a.conv_b[d] : 0.f;
  for (int t = 0; t < a.T; ++t) {
    const int64_t row = (int64_t)b * a.T + t;
    const float x = a.xz[row * 2 * di + d];
    win.x = win.y;
    win.y = win.z;
    win.z = win.w;
    win.w = x;
    a.xc[row * di + d] = silu_hw(win.x * w.x + win.y * w.y + win.z * w.z + win.w * w.w + bias);
  }
  *reinterpret_cast<float4*>(a.conv_state + gid * 4) = win;
}

// dst[c][r] = src[r][c] (finalize-time helper: small weight matrices only)
__global__ void transpose_f32_kernel(const float* src, int rows, int cols, float* dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  dst[(int64_t)c * rows + r] = src[i];
}

// kSsmEnvs envs per workgroup (4 for an env-step: amortises A = -exp(A_log) and keeps 4 state loads in flight
// per lane; 1 for the long prefill chunks, whose LDS staging grows with T)

// One workgroup = kSsmEnvs envs x 256/Q channels (64 for N = 16).  Per-(env, token, channel) scalars x, dt =
// softplus(.), silu(z) and the per-(env, token) vectors B, C are staged in LDS with coalesced loads (dt and the
// gate are evaluated once per channel there, not once per lane); outputs go back through LDS so the global
// stores are full 256-byte rows.
template <int T, int kSsmEnvs>
__global__ __launch_bounds__(256) void mamba_ssm_kernel(MambaSsmArgs a) {
  __shared__ __attribute__((aligned(16))) float bc[kSsmEnvs][T][2][64];   // [env][token][B|C][n], N <= 64
  __shared__ float sc[kSsmEnvs][T][3][64];                                // x | dt | silu(z) per channel
  __shared__ float yo[kSsmEnvs][T][64];
  const int di = a.d_inner, N = a.N;
  const int Q = N >> 2;               // float4 per channel
  const int cpb = 256 / Q;            // channels per block (<= 64)
  const int b0 = blockIdx.y * kSsmEnvs;
  const int ne = min(kSsmEnvs, a.B - b0);
  const int tid = threadIdx.x;
  const int ldx = a.R + 2 * N;
  const int dbase = blockIdx.x * cpb;
  // the recurrent state and the per-channel constants do not depend on the LDS staging below: request them first, so
  // their round trip overlaps the staging loads instead of following the barrier (one dependent memory phase fewer)
  const int qd = tid % Q;
  const int cl = tid / Q;
  const int d0 = dbase + cl;
  const bool active = d0 < di;
  const int d = active ? d0 : di - 1;
  const float4 al = *reinterpret_cast<const float4*>(a.A_log + (int64_t)d * N + 4 * qd);
  const float4 A = make_float4(-expf(al.x), -expf(al.y), -expf(al.z), -expf(al.w));
  const float Dd = a.Dp[d];
  float4 s[kSsmEnvs];
#pragma unroll
  for (int e = 0; e < kSsmEnvs; ++e) {
    const int b = b0 + e;
    const bool rs = e >= ne || (a.reset != nullptr && a.reset[b] != 0);
    if (rs) {
      s[e] = f4_zero();
    } else {  // streamed once per env-step: non-temporal, the projections' operands keep the caches
      const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(a.ssm_state + (((int64_t)b * di + d) * Q + qd) * 4));
      s[e] = make_float4(v.x, v.y, v.z, v.w);
    }
  }
  // dt_wt given (launcher: N = 16, so 64 channels = 64 lanes per wave): dt_proj is evaluated here, exact fp32 -- 64 channels
  // x <= 16 rows x dt_rank MACs per workgroup -- instead of a [rows, d_inner] GEMM launch plus a write and a read of its
  // output.  Wave g takes rows g, g + 4, ...: the raw dt row (x_proj's first dt_rank columns) has a wave-uniform address,
  // the weights are the transposed copy [dt_rank, d_inner] (lane = channel: 256-byte rows); nothing here waits for LDS, so
  // these loads are in flight with the state's.
  const bool fuse_dt = a.dt_wt != nullptr;
  if (fuse_dt) {
    constexpr int kRows = (kSsmEnvs * T + 3) / 4;
    const int c = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6), nrow = ne * T;
    const int dch = min(dbase + c, di - 1);
    const float* __restrict__ wt = a.dt_wt + dch;
    const float* __restrict__ xr[kRows];
    float acc[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      acc[i] = 0.f;
      xr[i] = a.xdb + ((int64_t)b0 * T + min(g + 4 * i, nrow - 1)) * ldx;
    }
#pragma unroll 8
    for (int r = 0; r < a.R; ++r) {
      const float wv = wt[(int64_t)r * di];
#pragma unroll
      for (int i = 0; i < kRows; ++i) acc[i] += wv * xr[i][r];
    }
    const float bias = a.dt_bias[dch];
#pragma unroll
    for (int i = 0; i < kRows; ++i) {
      const int et = g + 4 * i;
      if (et < nrow) sc[et / T][et % T][1][c] = softplus_hw(acc[i] + bias);
    }
  }
  for (int i = tid; i < ne * T * 2 * N; i += 256) {
    const int et = i / (2 * N), j = i - et * 2 * N;     // et = e * T + t
    bc[et / T][et % T][j / N][j % N] = a.xdb[((int64_t)b0 * T + et) * ldx + a.R + j];
  }
  for (int i = tid; i < ne * T * 3 * cpb; i += 256) {
    const int c = i % cpb, w = (i / cpb) % 3, et = i / (3 * cpb);
    if (fuse_dt && w == 1) continue;
    const int d = min(dbase + c, di - 1);
    const int64_t row = (int64_t)b0 * T + et;
    float v;
    if (w == 0)
      v = a.xc[row * di + d];
    else if (w == 1)
      v = softplus_hw(a.dtp[row * di + d] + a.dt_bias[d]);
    else
      v = silu_hw(a.xz[row * 2 * di + di + d]);
    sc[et / T][et % T][w][c] = v;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kSsmEnvs; ++e) {
    if (e >= ne) break;
    const int b = b0 + e;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float x = sc[e][t][0][cl];
      const float dt = sc[e][t][1][cl];
      const float4 Bm = *reinterpret_cast<const float4*>(&bc[e][t][0][4 * qd]);
      const float4 Cm = *reinterpret_cast<const float4*>(&bc[e][t][1][4 * qd]);
      // decay factors via the hardware exp2 (v_exp_f32): |dt * A| is O(1), relative error ~1e-7
      s[e].x = s[e].x * __expf(dt * A.x) + x * (dt * Bm.x);
      s[e].y = s[e].y * __expf(dt * A.y) + x * (dt * Bm.y);
      s[e].z = s[e].z * __expf(dt * A.z) + x * (dt * Bm.z);
      s[e].w = s[e].w * __expf(dt * A.w) + x * (dt * Bm.w);
      float y = s[e].x * Cm.x + s[e].y * Cm.y + s[e].z * Cm.z + s[e].w * Cm.w;
      // sum over the Q lanes of this channel (Q is a power of two, lanes are adjacent)
      for (int off = 1; off < Q; off <<= 1) y += __shfl_xor(y, off, 64);
      if (qd == 0) yo[e][t][cl] = (y + Dd * x) * sc[e][t][2][cl];
    }
    if (active) {
      v4f_t v;
      v.x = s[e].x, v.y = s[e].y, v.z = s[e].z, v.w = s[e].w;
      __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(a.ssm_state + (((int64_t)b * di + d) * Q + qd) * 4));
    }
  }
  __syncthreads();
  for (int i = tid; i < ne * T * cpb; i += 256) {
    const int c = i % cpb, et = i / cpb;
    float yabs = 0.f;
    if (dbase + c < di) {
      const int64_t off = ((int64_t)b0 * T + et) * di + dbase + c;
      const float yv = yo[et / T][et % T][c];
      yabs = fabsf(yv);
      if (a.y != nullptr) a.y[off] = yv;
    }
    if (a.amax != nullptr && cpb == 64) {  // one wave = one (env, token) row segment of 64 channels
      const float m = wave_max(yabs);
      if ((tid & 63) == 0) a.amax[((int64_t)b0 * T + et) * (di >> 6) + blockIdx.x] = m;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// selective_state_update, lane = channel form (d_state 16, dt_proj fused, T tokens of an env-step; round 4).
//
// mamba_ssm_kernel above spreads a channel's 16 states over 4 lanes and rebuilds everything per workgroup of 4 envs x 64
// channels: 32 KB of state per workgroup against two barriers, an LDS staging pass full of index arithmetic, a 4-lane
// shuffle reduction per output and 12 KB of dt_proj weights pulled in again for every 4 envs -- 168 us per 1024-env launch
// of Mamba-48M for 277 MB = 1.65 TB/s, 21 % of HBM peak (profiles/r03_kernel_stats_mamba48m_b2048.csv): latency-, not
// bandwidth- or ALU-bound.  Here one LANE owns one channel with all of its 16 states in registers (64 contiguous bytes per
// lane: a wave reads 4 KB runs), one WAVE owns 64 channels and loops over `epw` env slots with the next env's state / x / z
// already requested; the per-channel constants -- dt_proj's R weights, A = -exp(A_log), D, dt_bias -- are loaded once per
// wave, not once per 4 envs.  Everything that is per (env, token) and shared by the channels -- the raw dt row, B_t, C_t
// -- has a wave-uniform address: scalar loads, SGPR operands, no LDS, no barrier, no cross-lane reduction (y_t is a
// 16-term in-lane dot).  Same arithmetic as the 4-lane form to fp32 rounding: dt_proj's 48 products and y's 16 terms are summed
// as four interleaved partial chains (a 48-long dependent fma chain leaves the SIMD idle between issues), not in k order.  Reference: selective_state_update as called from Mamba.step
// (src/algos/models/decision_mamba.py:136-138, [3P] mamba_ssm 2.1.0).
// TR: the state block of a wave and an env -- 64 channels x 16 states, one contiguous 4 KB run -- moves between HBM and the
// registers COALESCED (instruction q of a lane: 16 bytes at 1 KB * q + 16 * lane) and is transposed to / from the lane = channel
// form through the wave's own 4 KB of LDS (in-order LDS pipeline of one wave: no barrier).  Without it each lane reads and writes
// its 64 contiguous bytes as four 16-byte pieces at a 64-byte lane stride: that shape alone (no arithmetic, same grid) runs at
// 2.5 TB/s, the coalesced one at 4.9 (scripts/rmw_pattern.cpp, profiles/r04_mamba_state_access_shape.txt).
template <int T, int R, bool ILP, int OCC, bool TR>
__global__ __launch_bounds__(256, OCC) void mamba_ssm_lane_kernel(float* __restrict__ ssm_state, const float* __restrict__ xc,
                                                             const float* __restrict__ xz, const float* __restrict__ xdb,
                                                             const float* __restrict__ dt_wt, const float* __restrict__ dt_bias,
                                                             const float* __restrict__ A_log, const float* __restrict__ Dp,
                                                             const uint8_t* __restrict__ reset, float* __restrict__ y,
                                                             float* __restrict__ amax, int B, int di, int epw) {
  constexpr int N = 16, LDX = R + 2 * N;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int ncb = di >> 6;
  const int cb = wv % ncb, eg = wv / ncb;
  const int e_begin = eg * epw, e_end = min(B, e_begin + epw);
  if (e_begin >= e_end) return;
  const int d = cb * 64 + lane;
  __shared__ v4f_t tb_all[TR ? 4 * 256 : 1];
  v4f_t* tb = tb_all + (TR ? (threadIdx.x >> 6) * 256 : 0);
  // per-channel constants, once per wave
  float wt[R];
#pragma unroll
  for (int r = 0; r < R; ++r) wt[r] = dt_wt[(int64_t)r * di + d];
  float A2[N];  // A * log2(e), A = -exp(A_log): the decay exp(dt * A) is one multiply and one v_exp_f32
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 al = *reinterpret_cast<const float4*>(A_log + (int64_t)d * N + 4 * q);
    constexpr float kLog2e = 1.4426950408889634f;
    A2[4 * q] = -kLog2e * expf(al.x), A2[4 * q + 1] = -kLog2e * expf(al.y), A2[4 * q + 2] = -kLog2e * expf(al.z),
            A2[4 * q + 3] = -kLog2e * expf(al.w);
  }
  const float Dd = Dp[d], bias = dt_bias[d];
  // operands of one env: its 16 states, x and z of the T tokens (requested one env ahead)
  v4f_t sn[4];
  float xn[T], zn[T];
  auto request = [&](int b) {
    const float* sp = TR ? ssm_state + ((int64_t)b * di + cb * 64) * N + 4 * lane : ssm_state + ((int64_t)b * di + d) * N;
#pragma unroll
    for (int q = 0; q < 4; ++q) sn[q] = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(sp + (TR ? 256 * q : 4 * q)));
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t row = (int64_t)b * T + t;
      xn[t] = xc[row * di + d];
      zn[t] = xz[row * 2 * di + di + d];
    }
  };
  request(e_begin);
  for (int b = e_begin; b < e_end; ++b) {
    float s[N], x[T], z[T];
    const bool rs = reset != nullptr && __builtin_amdgcn_readfirstlane((int)reset[b]) != 0;  // wave-uniform: a scalar branch
    if (TR) {  // piece q of lane l is float4 64 q + l of the block; channel c owns float4s 4 c .. 4 c + 3
#pragma unroll
      for (int q = 0; q < 4; ++q) tb[64 * q + lane] = sn[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) sn[q] = tb[4 * lane + q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s[4 * q] = sn[q].x, s[4 * q + 1] = sn[q].y, s[4 * q + 2] = sn[q].z, s[4 * q + 3] = sn[q].w;
    if (rs) {
#pragma unroll
      for (int n = 0; n < N; ++n) s[n] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) x[t] = xn[t], z[t] = zn[t];
    request(min(b + 1, e_end - 1));  // (unconditional: past the last env it re-requests the last one and drops it)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t row = (int64_t)b * T + t;
      const float* __restrict__ xr = xdb + row * LDX;  // wave-uniform: dt_raw[R] | B[16] | C[16]
      float acc;
      if (ILP) {  // four interleaved partial sums: a 48-long dependent fma chain leaves the SIMD idle between issues
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int r = 0; r < R; r += 4) p0 += wt[r] * xr[r], p1 += wt[r + 1] * xr[r + 1], p2 += wt[r + 2] * xr[r + 2], p3 += wt[r + 3] * xr[r + 3];
        acc = (p0 + p1) + (p2 + p3);
      } else {
        acc = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) acc += wt[r] * xr[r];
      }
      const float dt = softplus_hw(acc + bias);
      const float xv = x[t], dx = dt * xv;
      float yv = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
#pragma unroll
      for (int n = 0; n < N; ++n) {
        // decay via the hardware exp2 (v_exp_f32): |dt * A| is O(1), relative error ~1e-7
        s[n] = s[n] * __builtin_amdgcn_exp2f(dt * A2[n]) + dx * xr[R + n];
        const float term = s[n] * xr[R + N + n];
        if (!ILP || (n & 3) == 0) yv += term;
        else if ((n & 3) == 1) y1 += term;
        else if ((n & 3) == 2) y2 += term;
        else y3 += term;
      }
      if (ILP) yv = (yv + y1) + (y2 + y3);
      yv = (yv + Dd * xv) * silu_hw(z[t]);
      y[row * di + d] = yv;
      if (amax != nullptr) {
        const float m = wave_max_nonneg_lane63(fabsf(yv));
        if (lane == 63) amax[row * ncb + cb] = m;
      }
    }
    if (TR) {
      v4f_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v4f_t v;
        v.x = s[4 * q], v.y = s[4 * q + 1], v.z = s[4 * q + 2], v.w = s[4 * q + 3];
        tb[4 * lane + q] = v;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = tb[64 * q + lane];
      float* sp = ssm_state + ((int64_t)b * di + cb * 64) * N + 4 * lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) __builtin_nontemporal_store(o[q], reinterpret_cast<v4f_t*>(sp + 256 * q));
    } else {
      float* sp = ssm_state + ((int64_t)b * di + d) * N;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v4f_t v;
        v.x = s[4 * q], v.y = s[4 * q + 1], v.z = s[4 * q + 2], v.w = s[4 * q + 3];
        __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(sp + 4 * q));
      }
    }
  }
}

}  // namespace

void launch_transpose_f32(const float* src, int rows, int cols, float* dst, hipStream_t stream) {
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, rows, cols, dst);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_mamba_conv(const MambaConvArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(a.K == 4, "Mamba d_conv must be 4");
  const int64_t n = (int64_t)a.B * a.d_inner;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (a.T) {
    case 1: hipLaunchKernelGGL(mamba_conv_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(mamba_conv_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(mamba_conv_kernel<3>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(mamba_conv_kernel<4>, grid, block, 0, stream, a); break;
    default:
      LRAM_REQUIRE(a.T >= 1 && a.T <= kMaxTokens, "tokens per launch out of range");
      hipLaunchKernelGGL(mamba_conv_rt_kernel, grid, block, 0, stream, a);
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_mamba_ssm(const MambaSsmArgs& a, hipStream_t stream) {
  const int Q = a.N >> 2;
  LRAM_REQUIRE(a.N % 4 == 0 && Q >= 1 && Q <= 16 && (Q & (Q - 1)) == 0, "Mamba d_state must be 4 * 2^k, <= 64");
  const int cpb = 256 / Q;
  const unsigned gx = (unsigned)((a.d_inner + cpb - 1) / cpb);
  LRAM_REQUIRE(a.dt_wt == nullptr || mamba_ssm_dt_fusable(a.N, a.R), "fused dt_proj needs d_state 16 and dt_rank <= 128");
  LRAM_REQUIRE(a.dt_wt != nullptr || a.dtp != nullptr, "selective state update needs dtp or dt_w");
  dim3 block(256);
  // lane = channel form: env-steps of the d_state-16 geometries with dt_proj fused (Mamba-48M: dt_rank 48)
  // (variants measured and removed, profiles/EXPERIMENTS.md "Mamba state update": one dependent chain per dot product, four
  // waves per SIMD with spills, each lane moving its own 64 contiguous state bytes, other env counts per wave)
  if (a.T == 3 && a.N == 16 && a.R == 48 && a.dt_wt != nullptr && a.y != nullptr && a.d_inner % 64 == 0 && a.B >= 64) {
    const int epw = 8;  // env slots per wave
    const long waves = (long)(a.d_inner / 64) * ((a.B + epw - 1) / epw);
    const dim3 grid((unsigned)((waves + 3) / 4));
    hipLaunchKernelGGL((mamba_ssm_lane_kernel<3, 48, true, 3, true>), grid, block, 0, stream, a.ssm_state, a.xc, a.xz, a.xdb, a.dt_wt,
                       a.dt_bias, a.A_log, a.Dp, a.reset, a.y, a.amax, a.B, a.d_inner, epw);
    LRAM_HIP_CHECK(hipGetLastError());
    return;
  }
  dim3 g4(gx, (unsigned)((a.B + 3) / 4)), g1(gx, (unsigned)a.B);
  switch (a.T) {
    case 1: hipLaunchKernelGGL((mamba_ssm_kernel<1, 4>), g4, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((mamba_ssm_kernel<2, 4>), g4, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((mamba_ssm_kernel<3, 4>), g4, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((mamba_ssm_kernel<4, 4>), g4, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL((mamba_ssm_kernel<6, 1>), g1, block, 0, stream, a); break;
    case 9: hipLaunchKernelGGL((mamba_ssm_kernel<9, 1>), g1, block, 0, stream, a); break;
    case 12: hipLaunchKernelGGL((mamba_ssm_kernel<12, 1>), g1, block, 0, stream, a); break;
    default: throw Error("lram: tokens per launch must be 1..4, 6, 9 or 12");
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
