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

// exact (erf) GELU, torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

}  // namespace lram
