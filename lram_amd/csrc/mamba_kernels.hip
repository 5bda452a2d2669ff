// Mamba-1 recurrent step kernels for gfx950 (T tokens of one env-step per launch).
//
// Replace, on the reference path: causal_conv1d_update (CUDA) and
// selective_state_update (Triton) as called from [3P] mamba_ssm Mamba.step via
// src/algos/models/decision_mamba.py:130-147.  State layouts are the reference's:
// conv_state [B, d_inner, d_conv], ssm_state [B, d_inner, d_state], fp32.
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
    const float o = silu_f(y);
    a.xc[row * di + d] = o;
    if (a.amax != nullptr) {  // (d_inner is a multiple of 64: the lanes of a wave share the env, hence the row)
      const float m = wave_max(fabsf(o));
      if ((threadIdx.x & 63) == 0) a.amax[row * (di >> 6) + (d >> 6)] = m;  // this wave's 64 channels of the row
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
    a.xc[row * di + d] = silu_f(win.x * w.x + win.y * w.y + win.z * w.z + win.w * w.w + bias);
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
      if (et < nrow) sc[et / T][et % T][1][c] = softplus_f(acc[i] + bias);
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
      v = softplus_f(a.dtp[row * di + d] + a.dt_bias[d]);
    else
      v = silu_f(a.xz[row * 2 * di + di + d]);
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
      if (a.y3 != nullptr) split3_store(yv, a.y3 + off, a.y3_plane);
    }
    if (a.amax != nullptr && cpb == 64) {  // one wave = one (env, token) row segment of 64 channels
      const float m = wave_max(yabs);
      if ((tid & 63) == 0) a.amax[((int64_t)b0 * T + et) * (di >> 6) + blockIdx.x] = m;
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
