// Ablation / phase-share harness of the 8-phase f16x2 GEMM (lram_amd/csrc/gemm_f16x2_8p.hip).  Builds on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -I lram_amd/csrc -I include \
//         scripts/gemm8p_ablate.cpp -o /tmp/gemm8p_ablate && /tmp/gemm8p_ablate
// Variants (template ABL): 0 product, 1 zero-record DMA descriptors (instruction stream intact, no bytes), 2 no DMA instructions,
// 3 fragments read once, 4 stamps.  Interleaved rounds in ONE process (median of 15), random f16 planes.  Outputs of 1-4 are garbage.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../lram_amd/csrc/gemm_f16x2_8p.hip"

namespace lram {   // the pieces of the library the kernel file refers to (only the supported() predicate and launcher use them)
bool gemm_f16x2p_supported(const GemmArgs&) { return true; }
void gemm_choose_xcd_split(GemmArgs& g, int, int, int) { g.xcd_gm = g.xcd_gn = 0, g.panel_w = 0; }
void launch_splitk_reduce(const GemmArgs&, hipStream_t) {}
}  // namespace lram

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ABL>
static void launch(const lram::GemmArgs& g, int tiles, hipStream_t s) {
  hipLaunchKernelGGL((lram::gemm_f16x2_8p_kernel<false, false, ABL>), dim3(tiles), dim3(512), 0, s, g);
}

int main() {
  struct Shape { const char* tag; int m, n, k; };
  const Shape shapes[] = {{"prefill_up", 24576, 2048, 512}, {"c5_up", 16128, 5120, 1280}, {"mamba_in_s", 3072, 3072, 768},
                          {"mamba_in", 6144, 3072, 768}, {"one_round", 4096, 4096, 1024}};
  std::mt19937 rng(1);
  for (const Shape& sh : shapes) {
    const size_t an = (size_t)sh.m * sh.k, wn = (size_t)sh.n * sh.k;
    std::vector<uint16_t> ha(2 * an), hw(2 * wn);
    for (auto& v : ha) v = (uint16_t)((rng() & 0x8000) | (0x2c00 + (rng() & 0x0fff)));   // finite, magnitude ~0.06..0.25
    for (auto& v : hw) v = (uint16_t)((rng() & 0x8000) | (0x2c00 + (rng() & 0x0fff)));
    uint16_t *da, *dw;
    float *dc, *dinv, *dws;
    CK(hipMalloc((void**)&da, ha.size() * 2)); CK(hipMalloc((void**)&dw, hw.size() * 2));
    CK(hipMalloc((void**)&dc, (size_t)sh.m * sh.n * 4)); CK(hipMalloc((void**)&dinv, (size_t)(sh.m + sh.n) * 4));
    CK(hipMalloc((void**)&dws, 1 << 20));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> ones((size_t)sh.m + sh.n, 1.f);
    CK(hipMemcpy(dinv, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
    lram::GemmArgs g;
    g.c = dc, g.ldc = sh.n, g.m = sh.m, g.n = sh.n, g.k = sh.k;
    g.a2 = da, g.a2_plane = (int64_t)an, g.a2_kt = 32 * (int64_t)sh.m, g.a2_inv = dinv;
    g.w2 = dw, g.w2_plane = (int64_t)wn, g.w2_kt = 32 * (int64_t)sh.n, g.w_inv = dinv + sh.m;
    g.splitk_ws = dws;   // (stamps land here; split_k stays 1)
    const int tiles = ((sh.m + 255) / 256) * ((sh.n + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t[5];
    for (int rep = 0; rep < 16; ++rep)
      for (int v = 0; v < 5; ++v) {
        CK(hipEventRecord(e0, 0));
        switch (v) { case 0: launch<0>(g, tiles, 0); break; case 1: launch<1>(g, tiles, 0); break; case 2: launch<2>(g, tiles, 0); break; case 3: launch<3>(g, tiles, 0); break; default: launch<5>(g, tiles, 0); }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) t[v].push_back(ms * 1e3f);
      }
    const double flop = 6.0 * sh.m * sh.n * sh.k;
    printf("%-11s M=%6d N=%5d K=%5d  %4d workgroups, %2d K tiles |", sh.tag, sh.m, sh.n, sh.k, tiles, sh.k / 32);
    const char* nm[5] = {"product", "zero-record DMA", "no DMA", "no ds_read", "ring staged once (random operands, no DMA in the loop)"};
    for (int v = 0; v < 5; ++v) { std::sort(t[v].begin(), t[v].end()); printf("  %s %.1f us (%.0f TF issued)", nm[v], t[v][7], flop / t[v][7] / 1e6); }
    printf("\n");
    CK(hipMemset(dws, 0, 1 << 20));
    CK(hipEventRecord(e0, 0));
    launch<4>(g, tiles, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms4; CK(hipEventElapsedTime(&ms4, e0, e1));
    unsigned long long st[34];
    CK(hipMemcpy(st, dws, sizeof(st), hipMemcpyDeviceToHost));
    printf("   K loop of workgroup 0: %.1f us (s_memrealtime, 100 MHz), %llu s_memtime ticks => %.0f MHz per tick\n", st[32] / 100.0, st[33], st[33] / (st[32] / 100.0));
    {
      std::vector<unsigned long long> tl(4 * (size_t)tiles);
      CK(hipMemcpy(tl.data(), reinterpret_cast<unsigned long long*>(dws) + 64, tl.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t1 = 0;
      std::vector<double> pro, loop, epi, start;
      for (int b = 0; b < tiles; ++b) { t0 = std::min(t0, tl[4 * b]); t1 = std::max(t1, tl[4 * b + 3]); }
      for (int b = 0; b < tiles; ++b) {
        start.push_back((double)(tl[4 * b] - t0)); pro.push_back((double)(tl[4 * b + 1] - tl[4 * b]));
        loop.push_back((double)(tl[4 * b + 2] - tl[4 * b + 1])); epi.push_back((double)(tl[4 * b + 3] - tl[4 * b + 2]));
      }
      auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
      const double us_per_tick = ms4 * 1e3 / (double)(t1 - t0);   // (the stamped kernel's own event time over its first-entry .. last-exit span)
      printf("   stamped kernel %.1f us by events; s_memrealtime span %llu ticks => %.4f us per tick (%.1f MHz)\n", ms4 * 1e3, (unsigned long long)(t1 - t0), us_per_tick, 1.0 / us_per_tick);
      printf("   per workgroup, us (median / max): entry after first %.1f / %.1f | prologue %.1f / %.1f | K loop %.1f / %.1f | epilogue %.1f / %.1f\n",
             med(start) * us_per_tick, mx(start) * us_per_tick, med(pro) * us_per_tick, mx(pro) * us_per_tick, med(loop) * us_per_tick, mx(loop) * us_per_tick,
             med(epi) * us_per_tick, mx(epi) * us_per_tick);
    }
    for (int w = 0; w < 2; ++w) {
      printf("   stamps wave %d (cycles per K tile: reads+DMA issue(+VM wait) | barrier 1 | MFMA issue | barrier 2):", w * 4);
      const double nkt = sh.k / 32;
      for (int p = 0; p < 4; ++p) printf("  ph%d %5.0f %5.0f %5.0f %5.0f", p, st[16 * w + 4 * p] / nkt, st[16 * w + 4 * p + 1] / nkt, st[16 * w + 4 * p + 2] / nkt, st[16 * w + 4 * p + 3] / nkt);
      printf("\n");
    }
    (void)hipFree(da); (void)hipFree(dw); (void)hipFree(dc); (void)hipFree(dinv); (void)hipFree(dws);
  }
  return 0;
}
