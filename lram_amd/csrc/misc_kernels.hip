// Row norms, token front end, action head argmax / de-tokenisation, utility copies.
//
// Reference functions replaced: nn.LayerNorm `embed_ln`, xlstm LayerNorm / LlamaRMSNorm
// (src/algos/models/rms_norm.py:17-22), embed_return / embed_rewards Linear(1, D)
// (src/algos/models/online_decision_transformer_model.py:522-530), torch.argmax +
// MinMaxTokenizer.inv_tokenize (src/algos/models/multi_domain_discrete_dt_model.py:83-94,
// src/tokenizers_custom/minmax_tokenizer.py:31-47).
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {
namespace {

constexpr int kNormMaxV = 8;  // float4 per lane: d <= 2048

// One wave per row.  LayerNorm: (x - mean) / sqrt(var + eps) * gamma + beta (biased variance, two pass);
// RMSNorm: gamma * (x * rsqrt(mean(x^2) + eps)).
// st.rtg given (the token front end of a single timestep): rows with token index 1 / 2 are not read but built here as
// Linear(1, D) of the env's return-to-go / reward (embed_rtg / embed_rewards, online_decision_transformer_model.py:509-521),
// so the scalar embeddings need no launch of their own.
__global__ __launch_bounds__(256) void row_norm_kernel(const float* in, int64_t in_stride, float* out,
                                                       int64_t out_stride, const float* gamma, const float* beta,
                                                       int rows, int d, float eps, int rms, float* out2, float* amax,
                                                       ScalarTokens st, _Float16* h2, int64_t h2_plane, float* h2_inv, int64_t h2_kt) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d >> 2;
  const float* src = in + (int64_t)row * in_stride;
  int tok = 0;
  float sx = 0.f;
  const float *sw = nullptr, *sb = nullptr;
  if (st.rtg != nullptr) {
    const int b = row / st.T;
    tok = row - b * st.T;
    if (tok == 1) sx = st.rtg[(int64_t)b * st.in_stride], sw = st.w_rtg, sb = st.b_rtg;
    if (tok == 2) sx = st.rew[(int64_t)b * st.in_stride], sw = st.w_rew, sb = st.b_rew;
  }
  float4 v[kNormMaxV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kNormMaxV; ++j) {
    const int i = lane + 64 * j;
    if (sw != nullptr && i < nv) {  // (wave-uniform branch: one row per wave)
      const float4 w4 = *reinterpret_cast<const float4*>(sw + 4 * i), b4 = *reinterpret_cast<const float4*>(sb + 4 * i);
      v[j] = make_float4(sx * w4.x + b4.x, sx * w4.y + b4.y, sx * w4.z + b4.z, sx * w4.w + b4.w);
    } else {
      v[j] = i < nv ? *reinterpret_cast<const float4*>(src + 4 * i) : f4_zero();
    }
    s += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  const float mean = rms ? 0.f : wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < kNormMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
      const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
      q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
  }
  const float var = wave_sum(q) / (float)d;
  const float rstd = rms ? rsqrtf(var + eps) : 1.f / sqrtf(var + eps);
  float* dst = out + (int64_t)row * out_stride;
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < kNormMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i >= nv) continue;
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * i);
    float4 o;
    o.x = (v[j].x - mean) * rstd * g.x;
    o.y = (v[j].y - mean) * rstd * g.y;
    o.z = (v[j].z - mean) * rstd * g.z;
    o.w = (v[j].w - mean) * rstd * g.w;
    if (beta != nullptr) {
      const float4 bb = *reinterpret_cast<const float4*>(beta + 4 * i);
      o.x += bb.x;
      o.y += bb.y;
      o.z += bb.z;
      o.w += bb.w;
    }
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    if (out != nullptr) *reinterpret_cast<float4*>(dst + 4 * i) = o;
    if (out2 != nullptr) *reinterpret_cast<float4*>(out2 + (int64_t)row * out_stride + 4 * i) = o;  // second copy (taps)
    v[j] = o;
  }
  if (amax != nullptr || h2 != nullptr) mx = wave_max(mx);  // (uniform) the row's largest magnitude
  if (amax != nullptr && lane == 0) amax[row] = mx;         // the f16x2 GEMM derives its operand scale from it
  if (h2 != nullptr) {  // ... or takes the operand already scaled and split: two f16 planes + the inverse scale (gemm_f16x2p.hip)
    const float sc = pow2_scale(mx);
#pragma unroll
    for (int j = 0; j < kNormMaxV; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) split2_store4(v[j], sc, h2 + (int64_t)(i >> 3) * h2_kt + (int64_t)row * 32 + 4 * (i & 7), h2_plane);
    }
    if (lane == 0) h2_inv[row] = 1.f / sc;
  }
}

// Mamba Block entry ([3P] mamba_ssm layer_norm_fn(prenorm=True, residual_in_fp32)):
// res_out = hidden + res_in ; normed = res_out * rsqrt(mean(res_out^2) + eps) * gamma
__global__ __launch_bounds__(256) void add_rms_norm_kernel(const float* hidden, const float* res_in, float* res_out,
                                                           float* normed, const float* gamma, int rows, int d,
                                                           float eps, float* amax, _Float16* h2, int64_t h2_plane, float* h2_inv, int64_t h2_kt) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d >> 2;
  const int64_t base = (int64_t)row * d;
  float4 v[kNormMaxV];
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < kNormMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i < nv) {
      float4 x = *reinterpret_cast<const float4*>(hidden + base + 4 * i);
      if (res_in != nullptr) {
        const float4 r = *reinterpret_cast<const float4*>(res_in + base + 4 * i);
        x.x += r.x;
        x.y += r.y;
        x.z += r.z;
        x.w += r.w;
      }
      v[j] = x;
      if (res_out != nullptr) *reinterpret_cast<float4*>(res_out + base + 4 * i) = x;
      q += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    } else {
      v[j] = f4_zero();
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < kNormMaxV; ++j) {
    const int i = lane + 64 * j;
    if (i >= nv) continue;
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * i);
    float4 o;
    o.x = v[j].x * rstd * g.x;
    o.y = v[j].y * rstd * g.y;
    o.z = v[j].z * rstd * g.z;
    o.w = v[j].w * rstd * g.w;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    if (normed != nullptr) *reinterpret_cast<float4*>(normed + base + 4 * i) = o;
    v[j] = o;
  }
  if (amax != nullptr || h2 != nullptr) mx = wave_max(mx);
  if (amax != nullptr && lane == 0) amax[row] = mx;
  if (h2 != nullptr) {  // operand planes of the pre-split f16x2 GEMM (see row_norm_kernel)
    const float sc = pow2_scale(mx);
#pragma unroll
    for (int j = 0; j < kNormMaxV; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) split2_store4(v[j], sc, h2 + (int64_t)(i >> 3) * h2_kt + (int64_t)row * 32 + 4 * (i & 7), h2_plane);
    }
    if (lane == 0) h2_inv[row] = 1.f / sc;
  }
}

__global__ __launch_bounds__(256) void embed_scalars_kernel(float* x, const float* rtg, const float* rew,
                                                            int64_t in_stride, const float* w_rtg, const float* b_rtg,
                                                            const float* w_rew, const float* b_rew, int B, int T,
                                                            int D) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * D) return;
  const int b = (int)(gid / D);
  const int d = (int)(gid - (int64_t)b * D);
  float* row = x + (int64_t)b * T * D;
  row[D + d] = rtg[b * in_stride] * w_rtg[d] + b_rtg[d];
  row[2 * D + d] = rew[b * in_stride] * w_rew[d] + b_rew[d];
}

__global__ __launch_bounds__(256) void scatter_token0_kernel(float* x, const float* emb, int64_t emb_stride, int B,
                                                             int T, int D) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * D) return;
  const int b = (int)(gid / D);
  const int d = (int)(gid - (int64_t)b * D);
  x[(int64_t)b * T * D + d] = emb[(int64_t)b * emb_stride + d];
}

// The (state, rtg, reward) token rows of `steps` consecutive timesteps of every env in one launch (stored contexts): token 0 from the
// state embeddings emb[b][j][:] (a GEMM over all timesteps, or the caller's own embeddings), tokens 1 / 2 as embed_scalars_kernel.
__global__ __launch_bounds__(256) void embed_chunk_kernel(float* x, const float* emb, int64_t emb_stride, const float* rtg,
                                                          const float* rew, int64_t in_stride, const float* w_rtg,
                                                          const float* b_rtg, const float* w_rew, const float* b_rew, int B,
                                                          int steps, int T, int D) {
  const int d4 = D >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * steps * d4) return;
  const int d = (int)(gid % d4) << 2;
  const int64_t bj = gid / d4;
  const int j = (int)(bj % steps), b = (int)(bj / steps);
  float* row = x + ((int64_t)b * T + 3 * j) * D + d;
  *reinterpret_cast<float4*>(row) = *reinterpret_cast<const float4*>(emb + (int64_t)b * emb_stride + (int64_t)j * D + d);
  const float g = rtg[b * in_stride + j], r = rew[b * in_stride + j];
  const float4 wg = *reinterpret_cast<const float4*>(w_rtg + d), bg = *reinterpret_cast<const float4*>(b_rtg + d);
  const float4 wr = *reinterpret_cast<const float4*>(w_rew + d), br = *reinterpret_cast<const float4*>(b_rew + d);
  *reinterpret_cast<float4*>(row + D) = make_float4(g * wg.x + bg.x, g * wg.y + bg.y, g * wg.z + bg.z, g * wg.w + bg.w);
  *reinterpret_cast<float4*>(row + 2 * D) = make_float4(r * wr.x + br.x, r * wr.y + br.y, r * wr.z + br.z, r * wr.w + br.w);
}

// One wave per (env, action dim): first index of the maximum (torch.argmax tie rule), then
// inv_tokenize: max(tok - shift, 0) * ((max - min) / channels) + min.
__global__ __launch_bounds__(256) void action_argmax_kernel(const float* logits, float* actions, int32_t* tokens,
                                                            int B, int act_dim, int n_vocab, int n_discrete,
                                                            int action_channels, float tok_min, float tok_max,
                                                            int discrete, int col_begin, int col_end) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ndim = discrete ? 1 : act_dim;
  if (item >= B * ndim) return;
  const int b = item / ndim, j = item - b * ndim;
  if (j < col_begin || j >= col_end) return;  // repeated-forward mode: pass p writes action dim p, the last pass
                                              // every dim from its own on (engine.hip::step_launches)
  const float* lg = logits + (int64_t)b * act_dim * n_vocab + (int64_t)j * n_vocab;
  const int n = discrete ? n_discrete : n_vocab;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < n; i += 64) {
    const float v = lg[i];
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) {
    if (tokens != nullptr) tokens[(int64_t)b * act_dim + j] = bi;
    float out;
    if (discrete) {
      out = (float)bi;
    } else {
      int t = bi - n_discrete;
      t = t < 0 ? 0 : t;
      const float bin_width = (tok_max - tok_min) / (float)action_channels;
      out = (float)t * bin_width + tok_min;
    }
    actions[(int64_t)b * act_dim + j] = out;
  }
}

// Observation front end: native obs [B, n_native] -> model input [B, state_dim]: scatter by an index table
// (identity = zero-pad, decision_xlstm.py:16-19; DMControl dict obs -> 204-dim full space,
// src/envs/dmcontrol_utils.py:35-59), then optional (x - mean) / std (decision_transformer_sb3.py:650-651).
__global__ __launch_bounds__(256) void pad_obs_kernel(const float* native, int n_native, const int32_t* index,
                                                      const float* mean, const float* stdv, float* out, int B,
                                                      int state_dim) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * state_dim) return;
  const int b = (int)(gid / state_dim);
  const int d = (int)(gid - (int64_t)b * state_dim);
  float v = 0.f;
  if (index == nullptr) {
    if (d < n_native) v = native[(int64_t)b * n_native + d];
  } else {
    const int src = index[d];  // inverse table: source column of output dim d, or -1
    if (src >= 0) v = native[(int64_t)b * n_native + src];
  }
  if (mean != nullptr) v = (v - mean[d]) / stdv[d];
  out[gid] = v;
}

__global__ __launch_bounds__(256) void stream_copy_kernel(float4* dst, const float4* src, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) dst[i] = src[i];
}

// STREAM-like copy: one workgroup per contiguous UNR x 4 KiB block, UNR independent 16-byte loads in flight per lane
// (each wave instruction touches whole 1 KiB rows), then UNR stores; NT = non-temporal loads and stores.
typedef float v4c_t __attribute__((ext_vector_type(4)));
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void stream_copy_blocked_kernel(v4c_t* __restrict__ dst, const v4c_t* __restrict__ src,
                                                                  size_t n4) {
  const size_t base = (size_t)blockIdx.x * (256 * UNR) + threadIdx.x;
  v4c_t v[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n4) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const size_t i = base + (size_t)u * 256;
    if (i < n4) {
      if (NT)
        __builtin_nontemporal_store(v[u], dst + i);
      else
        dst[i] = v[u];
    }
  }
}

// In-place read-modify-write stream with the access pattern of mlstm_cell_kernel and none of its arithmetic: one
// workgroup per contiguous 256 KiB block (256 rows of 1 KiB), a wave reads whole rows, 16 rows (16 x 16 B per lane)
// in flight per thread, non-temporal loads and stores, one workgroup per CU (LDS request).  Its rate is the
// practical ceiling for "read the state once, write it once" on this part.
typedef float v4f_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_rmw_kernel(float* buf, float scale) {
  extern __shared__ float pad_lds[];
  (void)pad_lds;
  float* blk = buf + (size_t)blockIdx.x * 65536;  // 256 rows x 256 floats
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  for (int r0 = rg; r0 < 256; r0 += 64) {
    v4f_t c[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
      c[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(blk + (size_t)(r0 + 4 * u) * 256 + 4 * cl));
#pragma unroll
    for (int u = 0; u < 16; ++u)
      __builtin_nontemporal_store(c[u] * scale, reinterpret_cast<v4f_t*>(blk + (size_t)(r0 + 4 * u) * 256 + 4 * cl));
  }
}

// buf viewed as [outer][B][row_elems] with outer stride `outer_stride`; zero row b where mask[b] != 0
// (mask == nullptr: every row).
__global__ __launch_bounds__(256) void zero_rows_kernel(float* buf, const uint8_t* mask, int B, int64_t row_elems,
                                                        int64_t outer_stride) {
  const int b = blockIdx.y;
  const int o = blockIdx.z;
  if (mask != nullptr && mask[b] == 0) return;
  float* row = buf + (int64_t)o * outer_stride + (int64_t)b * row_elems;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
  if ((row_elems & 3) == 0 && (outer_stride & 3) == 0) {
    for (int64_t i = tid0; i < (row_elems >> 2); i += nthr) reinterpret_cast<float4*>(row)[i] = f4_zero();
  } else {
    for (int64_t i = tid0; i < row_elems; i += nthr) row[i] = 0.f;
  }
}

}  // namespace

// out[0] = max over rows of the Euclidean norm of w[r][0 .. k)  (one workgroup; finalize-time helper)
void launch_row_norm(const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* gamma,
                     const float* beta, int rows, int d, float eps, int rms, hipStream_t stream, float* out2,
                     float* amax, const ScalarTokens* st, uint16_t* h2, int64_t h2_plane, float* h2_inv, int64_t h2_kt) {
  LRAM_REQUIRE(d % 4 == 0 && d <= 256 * kNormMaxV, "row norm: d must be a multiple of 4 and <= 2048");
  LRAM_REQUIRE(h2 == nullptr || (h2_inv != nullptr && h2_kt >= 32 * (int64_t)rows), "row norm: f16x2 operand planes need the inverse-scale output and the K-tile pitch");
  hipLaunchKernelGGL(row_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, in, in_stride, out, out_stride,
                     gamma, beta, rows, d, eps, rms, out2, amax, st ? *st : ScalarTokens(),
                     reinterpret_cast<_Float16*>(h2), h2_plane, h2_inv, h2_kt);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_add_rms_norm(const float* hidden, const float* res_in, float* res_out, float* normed,
                         const float* gamma, int rows, int d, float eps, hipStream_t stream, float* amax, uint16_t* h2,
                         int64_t h2_plane, float* h2_inv, int64_t h2_kt) {
  LRAM_REQUIRE(d % 4 == 0 && d <= 256 * kNormMaxV, "rms norm: d must be a multiple of 4 and <= 2048");
  LRAM_REQUIRE(h2 == nullptr || (h2_inv != nullptr && h2_kt >= 32 * (int64_t)rows), "rms norm: f16x2 operand planes need the inverse-scale output and the K-tile pitch");
  hipLaunchKernelGGL(add_rms_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, hidden, res_in, res_out,
                     normed, gamma, rows, d, eps, amax, reinterpret_cast<_Float16*>(h2), h2_plane, h2_inv, h2_kt);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_embed_scalars(float* x, const float* rtg, const float* rew, int64_t in_stride, const float* w_rtg,
                          const float* b_rtg, const float* w_rew, const float* b_rew, int B, int T, int D,
                          hipStream_t stream) {
  LRAM_REQUIRE(T >= 3, "embed: tokens_per_step must be >= 3 (state, rtg, reward)");
  const int64_t n = (int64_t)B * D;
  hipLaunchKernelGGL(embed_scalars_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, rtg, rew,
                     in_stride, w_rtg, b_rtg, w_rew, b_rew, B, T, D);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_scatter_token0(float* x, const float* emb, int64_t emb_stride, int B, int T, int D, hipStream_t stream) {
  const int64_t n = (int64_t)B * D;
  hipLaunchKernelGGL(scatter_token0_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, emb,
                     emb_stride, B, T, D);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_embed_chunk(float* x, const float* emb, int64_t emb_stride, const float* rtg, const float* rew, int64_t in_stride,
                        const float* w_rtg, const float* b_rtg, const float* w_rew, const float* b_rew, int B, int steps, int T,
                        int D, hipStream_t stream) {
  LRAM_REQUIRE(T >= 3 * steps && D % 4 == 0, "embed: 3 token rows per timestep, d_model a multiple of 4");
  const int64_t n = (int64_t)B * steps * (D >> 2);
  hipLaunchKernelGGL(embed_chunk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, emb, emb_stride, rtg, rew,
                     in_stride, w_rtg, b_rtg, w_rew, b_rew, B, steps, T, D);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_action_argmax(const float* logits, float* actions, int32_t* tokens, int B, int act_dim, int n_vocab,
                          int n_discrete, int action_channels, float tok_min, float tok_max, int discrete,
                          int col_begin, hipStream_t stream, int col_end) {
  const int items = B * (discrete ? 1 : act_dim);
  if (col_end < 0) col_end = act_dim;
  hipLaunchKernelGGL(action_argmax_kernel, dim3((items + 3) / 4), dim3(256), 0, stream, logits, actions, tokens, B,
                     act_dim, n_vocab, n_discrete, action_channels, tok_min, tok_max, discrete, col_begin, col_end);
  LRAM_HIP_CHECK(hipGetLastError());
}

// Read-only stream with the lazy read pass's access shape and none of its arithmetic: one workgroup per contiguous
// block of RB rows of 1 KiB (a wave reads whole rows), UNR rows (UNR x 16 B per lane) in flight per thread, non-temporal
// loads; the lane sums land in one float per workgroup so the loads cannot be dropped.  Its rate is the practical
// ceiling for "read the state once" on this part.
template <int UNR>
__global__ __launch_bounds__(256) void stream_read_kernel(const float* __restrict__ buf, float* __restrict__ sink, int rows_per_wg) {
  extern __shared__ float pad_lds[];
  (void)pad_lds;
  const float* blk = buf + (size_t)blockIdx.x * rows_per_wg * 256;
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  v4f_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = rg; r0 < rows_per_wg; r0 += 4 * UNR) {
    v4f_t c[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      c[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(blk + (size_t)(r0 + 4 * u) * 256 + 4 * cl));
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += c[u];
  }
  float s = acc.x + acc.y + acc.z + acc.w;
  s = wave_sum(s);
  if (cl == 0) atomicAdd(sink + (blockIdx.x & 1023), s);
}

void launch_pad_obs(const float* native, int n_native, const int32_t* inv_index, const float* mean, const float* stdv,
                    float* out, int B, int state_dim, hipStream_t stream) {
  const int64_t n = (int64_t)B * state_dim;
  hipLaunchKernelGGL(pad_obs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, native, n_native,
                     inv_index, mean, stdv, out, B, state_dim);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_stream_copy(float* dst, const float* src, size_t numel, hipStream_t stream) {
  LRAM_REQUIRE(numel % 4 == 0, "stream copy: numel must be a multiple of 4");
  // LRAM_COPY_VARIANT (measurement knob): 0 grid-stride loop, 1 blocked x8, 2 blocked x8 non-temporal,
  // 3 blocked x16 non-temporal, 4 blocked x4 non-temporal (default: measured 6.48 TB/s on MI355X against 4.4-5.5 for
  // the deeper variants and 5.0 for the grid-stride loop), 5 blocked x2 nt, 6 blocked x1 nt
  static const int variant = [] {
    const char* v = std::getenv("LRAM_COPY_VARIANT");
    return v ? std::atoi(v) : 4;
  }();
  const size_t n4 = numel / 4;
  v4c_t* d = reinterpret_cast<v4c_t*>(dst);
  const v4c_t* sp = reinterpret_cast<const v4c_t*>(src);
  auto blocks = [&](int unr) { return dim3((unsigned)((n4 + 256 * (size_t)unr - 1) / (256 * (size_t)unr))); };
  switch (variant) {
    case 0:
      hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, stream, reinterpret_cast<float4*>(dst),
                         reinterpret_cast<const float4*>(src), n4);
      break;
    case 1: hipLaunchKernelGGL((stream_copy_blocked_kernel<8, false>), blocks(8), dim3(256), 0, stream, d, sp, n4); break;
    case 3: hipLaunchKernelGGL((stream_copy_blocked_kernel<16, true>), blocks(16), dim3(256), 0, stream, d, sp, n4); break;
    case 2: hipLaunchKernelGGL((stream_copy_blocked_kernel<8, true>), blocks(8), dim3(256), 0, stream, d, sp, n4); break;
    case 5: hipLaunchKernelGGL((stream_copy_blocked_kernel<2, true>), blocks(2), dim3(256), 0, stream, d, sp, n4); break;
    case 6: hipLaunchKernelGGL((stream_copy_blocked_kernel<1, true>), blocks(1), dim3(256), 0, stream, d, sp, n4); break;
    default: hipLaunchKernelGGL((stream_copy_blocked_kernel<4, true>), blocks(4), dim3(256), 0, stream, d, sp, n4); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_stream_rmw(float* buf, size_t numel, hipStream_t stream) {
  LRAM_REQUIRE(numel % 65536 == 0 && numel > 0, "stream rmw: numel must be a positive multiple of 65536");
  static uint64_t raised = 0;
  if (first_use_on_device(raised)) {
    LRAM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_rmw_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));
  }
  hipLaunchKernelGGL(stream_rmw_kernel, dim3((unsigned)(numel / 65536)), dim3(256), 84 * 1024, stream, buf, 1.0f);
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_stream_read(const float* buf, size_t numel, float* sink, hipStream_t stream) {
  // LRAM_READ_VARIANT = 100 * rows-in-flight per thread (4 / 8 / 16) + log2(rows per workgroup / 64) (0..4), + 1000 to
  // request 48 KiB of LDS per workgroup (three workgroups per CU, the read pass's residency); default 1804: 8 rows in
  // flight, 1 MiB blocks, three workgroups per CU (6.0-6.6 TB/s over the variants on MI355X, profiles/r03_read_ceiling.txt)
  static const int variant = [] {
    const char* v = std::getenv("LRAM_READ_VARIANT");
    return v ? std::atoi(v) : 1804;
  }();
  const int unr = (variant % 1000) / 100, lg = variant % 100;
  const int rows = 64 << std::max(0, std::min(lg, 4));
  const size_t lds = variant >= 1000 ? 48 * 1024 : 0;
  LRAM_REQUIRE(numel % ((size_t)rows * 256) == 0 && numel > 0, "stream read: numel must be a positive multiple of the block size");
  const dim3 grid((unsigned)(numel / ((size_t)rows * 256)));
  switch (unr) {
    case 4: hipLaunchKernelGGL(stream_read_kernel<4>, grid, dim3(256), lds, stream, buf, sink, rows); break;
    case 16: hipLaunchKernelGGL(stream_read_kernel<16>, grid, dim3(256), lds, stream, buf, sink, rows); break;
    default: hipLaunchKernelGGL(stream_read_kernel<8>, grid, dim3(256), lds, stream, buf, sink, rows); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_zero_rows(float* buf, const uint8_t* mask, int B, int64_t row_elems, int64_t outer, int64_t outer_stride,
                      hipStream_t stream) {
  const int64_t n4 = row_elems >> 2;
  int gx = (int)((n4 + 255) / 256);
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(zero_rows_kernel, dim3(gx, B, (unsigned)outer), dim3(256), 0, stream, buf, mask, B, row_elems,
                     outer_stride);
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
