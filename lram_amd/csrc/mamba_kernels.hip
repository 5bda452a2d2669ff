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
    a.xc[row * di + d] = silu_f(y);
  }
  *reinterpret_cast<float4*>(a.conv_state + gid * 4) = win;
}

// selective_state_update for T tokens.  4 lanes per channel, each owning 4 of the N = 16 states
// (16 B per lane, consecutive lanes consecutive addresses).
//   dt = softplus(dt_proj(dt_raw) + dt_bias);  s = s * exp(dt * A) + x * (dt * B);  y = s . C + D x;  y *= silu(z)
template <int T>
__global__ __launch_bounds__(256) void mamba_ssm_kernel(MambaSsmArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int di = a.d_inner;
  const int Q = a.N >> 2;  // float4 per channel
  const int64_t total = (int64_t)a.B * di * Q;
  const bool active = gid < total;
  const int64_t g = active ? gid : total - 1;
  const int qd = (int)(g % Q);
  const int64_t bd = g / Q;
  const int b = (int)(bd / di);
  const int d = (int)(bd - (int64_t)b * di);
  const bool rs = a.reset != nullptr && a.reset[b] != 0;
  float4 s = rs ? f4_zero() : *reinterpret_cast<const float4*>(a.ssm_state + g * 4);
  const float4 al = *reinterpret_cast<const float4*>(a.A_log + (int64_t)d * a.N + 4 * qd);
  const float4 A = make_float4(-expf(al.x), -expf(al.y), -expf(al.z), -expf(al.w));
  const float Dd = a.Dp[d];
  const float dtb = a.dt_bias[d];
  const int ldx = a.R + 2 * a.N;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int64_t row = (int64_t)b * T + t;
    const float x = a.xc[row * di + d];
    const float dt = softplus_f(a.dtp[row * di + d] + dtb);
    const float4 Bm = *reinterpret_cast<const float4*>(a.xdb + row * ldx + a.R + 4 * qd);
    const float4 Cm = *reinterpret_cast<const float4*>(a.xdb + row * ldx + a.R + a.N + 4 * qd);
    s.x = s.x * expf(dt * A.x) + x * (dt * Bm.x);
    s.y = s.y * expf(dt * A.y) + x * (dt * Bm.y);
    s.z = s.z * expf(dt * A.z) + x * (dt * Bm.z);
    s.w = s.w * expf(dt * A.w) + x * (dt * Bm.w);
    float y = s.x * Cm.x + s.y * Cm.y + s.z * Cm.z + s.w * Cm.w;
    // sum over the Q lanes of this channel (Q is a power of two <= 64, lanes are adjacent)
    for (int off = 1; off < Q; off <<= 1) y += __shfl_xor(y, off, 64);
    if (active && qd == 0) {
      const float z = a.xz[row * 2 * di + di + d];
      a.y[row * di + d] = (y + Dd * x) * silu_f(z);
    }
  }
  if (active) *reinterpret_cast<float4*>(a.ssm_state + g * 4) = s;
}

}  // namespace

void launch_mamba_conv(const MambaConvArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(a.K == 4, "Mamba d_conv must be 4");
  const int64_t n = (int64_t)a.B * a.d_inner;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (a.T) {
    case 1: hipLaunchKernelGGL(mamba_conv_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(mamba_conv_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(mamba_conv_kernel<3>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(mamba_conv_kernel<4>, grid, block, 0, stream, a); break;
    default: throw Error("lram: tokens per step must be in 1..4");
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_mamba_ssm(const MambaSsmArgs& a, hipStream_t stream) {
  const int Q = a.N >> 2;
  LRAM_REQUIRE(a.N % 4 == 0 && Q >= 1 && Q <= 64 && (Q & (Q - 1)) == 0, "Mamba d_state must be 4 * 2^k, <= 256");
  LRAM_REQUIRE((a.R + 2 * a.N) % 4 == 0 && a.R % 4 == 0, "Mamba dt_rank must be a multiple of 4");
  const int64_t n = (int64_t)a.B * a.d_inner * Q;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (a.T) {
    case 1: hipLaunchKernelGGL(mamba_ssm_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(mamba_ssm_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(mamba_ssm_kernel<3>, grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL(mamba_ssm_kernel<4>, grid, block, 0, stream, a); break;
    default: throw Error("lram: tokens per step must be in 1..4");
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
