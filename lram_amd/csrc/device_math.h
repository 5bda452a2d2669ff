// Device-side scalar helpers shared by the recurrent kernels.  The formulas follow the order of
// operations of the PyTorch CPU ops the oracle uses, so that states stay within a few ulp of it.
#pragma once
#include <hip/hip_runtime.h>

namespace lram {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// torch.nn.functional.logsigmoid: min(x, 0) - log1p(exp(-|x|))
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// torch silu: x * sigmoid(x)
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + expf(-x)); }

// torch softplus (beta 1, threshold 20)
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// The same two on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each; results within ~3 ulp of the libm
// forms above), for kernels that evaluate them per (channel, token): the libm softplus is a 145-instruction double-float log1p and
// was 39 % of the Mamba state-update kernel's instruction stream.
__device__ __forceinline__ float silu_hw(float x) {
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// softplus(x) = max(x, 0) + log1p(e), e = exp(-|x|); log1p(e) = log(u) * e / (u - 1) with u = fl(1 + e) cancels the rounding of
// 1 + e (for u == 1 the result is e itself); above the reference's threshold 20 the second term is below half an ulp of x
__device__ __forceinline__ float softplus_hw(float x) {
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(x));
  const float u = 1.f + e, d = u - 1.f;
  const float t = (__builtin_amdgcn_logf(u) * 0.69314718055994531f) * (e * __builtin_amdgcn_rcpf(d));
  return fmaxf(x, 0.f) + (d == 0.f ? e : t);
}

// exact (erf) GELU, torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Exact 3-way bf16 split of an fp32 value, x = hi + mid + lo (the operand format of gemm_bf16x3.hip): producers that
// feed a projection write these three planes instead of (or next to) the fp32 value, so the GEMM stages its A operand
// with plain 16-byte copies.  p points at the element in plane 0; the planes are `plane` elements apart.
__device__ __forceinline__ void split3_store(float x, uint16_t* p, int64_t plane) {
  const __bf16 hi = (__bf16)x;
  const float r1 = x - (float)hi;
  const __bf16 mid = (__bf16)r1;
  const __bf16 lo = (__bf16)(r1 - (float)mid);
  p[0] = __builtin_bit_cast(uint16_t, hi);
  p[plane] = __builtin_bit_cast(uint16_t, mid);
  p[2 * plane] = __builtin_bit_cast(uint16_t, lo);
}
__device__ __forceinline__ void split3_store4(const float4& v, uint16_t* p, int64_t plane) {
  const float xs[4] = {v.x, v.y, v.z, v.w};
  uint16_t h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hi = (__bf16)xs[e];
    const float r1 = xs[e] - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    h[e] = __builtin_bit_cast(uint16_t, hi), m[e] = __builtin_bit_cast(uint16_t, mid), l[e] = __builtin_bit_cast(uint16_t, lo);
  }
  *reinterpret_cast<uint2*>(p) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
  *reinterpret_cast<uint2*>(p + plane) = make_uint2(m[0] | ((uint32_t)m[1] << 16), m[2] | ((uint32_t)m[3] << 16));
  *reinterpret_cast<uint2*>(p + 2 * plane) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
}

// ---- row maxima for the f16x2 GEMM (gemm_f16x2.hip) ----
// power-of-two scale that puts a row whose largest magnitude is `mx` into [2^14, 2^15); 1 for an all-zero row
__device__ __forceinline__ float pow2_scale(float mx) {
  if (!(mx > 0.f)) return 1.f;
  int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);  // biased exponent (0: subnormal)
  if (e == 0) e = 1;
  int se = 268 - e;                                    // biased exponent of 2^(14 - (e - 127))
  se = se > 254 ? 254 : (se < 1 ? 1 : se);
  return __uint_as_float((unsigned)se << 23);
}
// Exact 2-way f16 split of s * v (the operand format of gemm_f16x2p.hip): p points at the element in the hi plane, the lo
// plane is `plane` elements on.  s is the row's power-of-two scale (pow2_scale of its largest magnitude).
__device__ __forceinline__ void split2_store4(const float4& v, float s, _Float16* p, int64_t plane) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const float xs[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
  h4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const _Float16 h = (_Float16)xs[e];
    hi[e] = h;
    lo[e] = (_Float16)(xs[e] - (float)h);
  }
  *reinterpret_cast<h4*>(p) = hi;
  *reinterpret_cast<h4*>(p + plane) = lo;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// The wave's sum as a wave-uniform value, on the DPP cross-lane paths (no LDS crossbar round trips): quad permutes and row
// mirrors leave every lane of a 16-lane row with the row's sum, the two row broadcasts assemble the four rows in row 3, and
// lane 63 is read back into a scalar register.  Fixed summation order (deterministic).
__device__ __forceinline__ float wave_sum_bcast(float x) {
#define LRAM_DPP_ADD(ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false))
  LRAM_DPP_ADD(0xB1);   // quad_perm [1, 0, 3, 2]
  LRAM_DPP_ADD(0x4E);   // quad_perm [2, 3, 0, 1]
  LRAM_DPP_ADD(0x141);  // row_half_mirror
  LRAM_DPP_ADD(0x140);  // row_mirror
  LRAM_DPP_ADD(0x142);  // row_bcast15: rows 1, 3 (and 2) take the sum of the row before
  LRAM_DPP_ADD(0x143);  // row_bcast31: row 3 takes rows 0 + 1
#undef LRAM_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// The wave's maximum of NON-NEGATIVE values, in lane 63 only, on the DPP cross-lane paths (two quad permutes, two row mirrors, two
// row broadcasts) instead of six ds_bpermute round trips through the LDS crossbar: for kernels that take a maximum per (row,
// channel block) inside their inner loop.  Non-negative floats order like their bit patterns, so the maximum is taken on unsigned
// integers with 0 for lanes a step has no source for -- the form hipcc folds into one v_max_u32_dpp per step.
__device__ __forceinline__ float wave_max_nonneg_lane63(float x) {
  unsigned v = __builtin_bit_cast(unsigned, x);
#define LRAM_DPP_MAX(ctrl) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false))
  LRAM_DPP_MAX(0xB1);   // quad_perm [1, 0, 3, 2]
  LRAM_DPP_MAX(0x4E);   // quad_perm [2, 3, 0, 1]
  LRAM_DPP_MAX(0x141);  // row_half_mirror
  LRAM_DPP_MAX(0x140);  // row_mirror: every lane of a 16-lane row holds the row's maximum
  LRAM_DPP_MAX(0x142);  // row_bcast15: row r takes row r - 1's
  LRAM_DPP_MAX(0x143);  // row_bcast31: rows 2, 3 take row 1's (= rows 0, 1)
#undef LRAM_DPP_MAX
  return __builtin_bit_cast(float, v);
}

}  // namespace lram
