// RMW bandwidth of the Mamba state-update access shape: state [B][di][16] fp32, wave = 64 channels x 8 envs.
// variants: lane-contiguous 64 B (4 x dwordx4 at 64 B lane stride) vs coalesced (instr q: base + 1 KB q + 16 lane), nt or not,
// prefetch depth 1 or 2 envs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool COAL, bool NT, int DEPTH, int OCC>
__global__ __launch_bounds__(256, OCC) void rmw(float* st, int B, int di, int epw, float mul) {
  const int lane = threadIdx.x & 63;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ncb = di >> 6;
  const int cb = wv % ncb, eg = wv / ncb;
  const int e0 = eg * epw, e1 = min(B, e0 + epw);
  if (e0 >= e1) return;
  auto addr = [&](int b, int q) -> v4* {
    float* base = st + ((int64_t)b * di + cb * 64) * 16;
    return reinterpret_cast<v4*>(COAL ? base + q * 256 + lane * 4 : base + lane * 16 + q * 4);
  };
  v4 buf[DEPTH][4];
  auto req = [&](int b, int slot) {
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[slot][q] = NT ? __builtin_nontemporal_load(addr(b, q)) : *addr(b, q);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) req(min(e0 + d, e1 - 1), d);
  for (int b0 = e0; b0 < e1; b0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int b = b0 + d;
      if (b >= e1) break;
      v4 cur[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cur[q] = buf[d][q] * mul + 1.0f;
      req(min(b + DEPTH, e1 - 1), d);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (NT) __builtin_nontemporal_store(cur[q], addr(b, q)); else *addr(b, q) = cur[q];
      }
    }
  }
}

int main() {
  const int B = 1024, di = 1536;
  const size_t n = (size_t)B * di * 16;
  float* st; CK(hipMalloc(&st, n * 4)); CK(hipMemset(st, 0, n * 4));
  float* other; CK(hipMalloc(&other, 512u << 20)); 
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, auto kern, int epw) {
    const long waves = (long)(di / 64) * ((B + epw - 1) / epw);
    dim3 grid((waves + 3) / 4), block(256);
    float best = 1e9;
    for (int it = 0; it < 12; ++it) {
      CK(hipMemsetAsync(other, 0, 512u << 20, 0));  // evict L2 / MALL
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(kern, grid, block, 0, 0, st, B, di, epw, 0.5f);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms;
    }
    printf("%-40s epw %2d  %7.1f us  %6.0f GB/s\n", name, epw, best * 1e3, 2.0 * n * 4 / best / 1e6);
  };
  for (int epw : {4, 8, 16}) {
    run("lane64B nt depth1 occ3", rmw<false, true, 1, 3>, epw);
    run("lane64B    depth1 occ3", rmw<false, false, 1, 3>, epw);
    run("coalesced nt depth1 occ3", rmw<true, true, 1, 3>, epw);
    run("coalesced    depth1 occ3", rmw<true, false, 1, 3>, epw);
    run("lane64B nt depth2 occ3", rmw<false, true, 2, 3>, epw);
    run("coalesced nt depth2 occ3", rmw<true, true, 2, 3>, epw);
    run("lane64B nt depth1 occ8", rmw<false, true, 1, 8>, epw);
    run("coalesced nt depth1 occ8", rmw<true, true, 1, 8>, epw);
    run("lane64B nt depth2 occ8", rmw<false, true, 2, 8>, epw);
    run("coalesced nt depth2 occ8", rmw<true, true, 2, 8>, epw);
  }
  return 0;
}
