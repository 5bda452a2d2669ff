// Hardware-hazard self-test (diagnostic entry point, not on the hot path).
//
// Found while bringing up the micro-batch pipeline on MI355X (gfx950, ROCm 7.2): packed-fp32 VALU instructions
// (`v_pk_fma_f32` / `v_pk_mul_f32` / `v_pk_add_f32`) of one wave return wrong results for a quarter of the wave
// (16 lanes, one instruction) when a co-resident wave of ANOTHER dispatch is issuing `v_mfma_f32_32x32x16_bf16`.
// Symptom in the engine: with the bf16x3 projections of one env slice running beside `mlstm_pre_kernel` of the
// other slice, single gate pre-activations came out a few percent off for a few dozen of 4096 envs, differently on
// every run; every kernel alone, and the same pair with the fp32-input MFMA GEMM, were bit-exact.  LDS, barrier,
// VGPR and global-load canaries beside the same GEMM were clean, forcing `s_waitcnt 0` everywhere did not help,
// removing the packed-fp32 instructions did.  The library is therefore built with `-packed-fp32-ops` removed
// from the device target features (lram_amd/build.py); this routine reproduces the pairing and must report 0.
#include <vector>

#include "../../include/lram_hip.h"
#include "common.h"

using namespace lram;

extern "C" int32_t lram_selftest_concurrent(int32_t iters, int64_t* n_diff) {
  try {
    if (n_diff == nullptr || iters < 1) throw Error("lram_selftest_concurrent: bad argument");
    const int B = 2048, T = 3, inner = 1024, NH = 4, K = 4;
    const int m = 6144, n = 2048, k = 512;
    uint32_t st = 777u;
    auto rnd = [&]() {
      st = st * 1664525u + 1013904223u;
      return ((st >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    std::vector<float*> owned;
    auto dev = [&](size_t cnt, float scale) {
      std::vector<float> h(cnt);
      for (auto& v : h) v = rnd() * scale;
      float* d = nullptr;
      LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), cnt * 4));
      owned.push_back(d);
      LRAM_HIP_CHECK(hipMemcpy(d, h.data(), cnt * 4, hipMemcpyHostToDevice));
      return d;
    };
    MlstmPreArgs pa;
    pa.u = dev((size_t)B * T * 2 * inner, 1.f);
    float* conv0 = dev((size_t)B * K * inner, 1.f);
    pa.conv_state = dev((size_t)B * K * inner, 1.f);
    pa.n_state = dev((size_t)B * inner, 0.f);
    pa.m_state = dev((size_t)B * NH, 0.f);
    pa.conv_w = dev((size_t)inner * 4, 0.5f), pa.conv_b = dev(inner, 0.1f);
    pa.wq = dev((size_t)inner * 4, 0.5f), pa.wk = dev((size_t)inner * 4, 0.5f), pa.wv = dev((size_t)inner * 4, 0.5f);
    pa.wi = dev((size_t)NH * 3 * inner, 0.02f), pa.wf = dev((size_t)NH * 3 * inner, 0.02f);
    pa.bi = dev(NH, 0.1f), pa.bf = dev(NH, 1.f);
    pa.q = dev((size_t)B * T * inner, 0.f), pa.k = dev((size_t)B * T * inner, 0.f), pa.v = dev((size_t)B * T * inner, 0.f);
    pa.xa = dev((size_t)B * T * inner, 0.f), pa.scal = dev((size_t)B * T * NH * 4, 0.f);
    pa.reset = nullptr, pa.B = B, pa.T = T, pa.inner = inner, pa.NH = NH, pa.K = K;
    GemmArgs g;
    float* gw = dev((size_t)n * k, 1.f);
    g.a = dev((size_t)m * k, 1.f), g.lda = k, g.w = gw, g.ldw = k, g.c = dev((size_t)m * n, 0.f), g.ldc = n;
    g.m = m, g.n = n, g.k = k;
    uint16_t* planes = nullptr;
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&planes), 3 * (size_t)n * k * 2));
    g.w3 = planes, g.w3_plane = (int64_t)n * k;
    launch_split_bf16x3(gw, planes, (size_t)n * k, nullptr);
    LRAM_HIP_CHECK(hipDeviceSynchronize());
    hipStream_t s1, s2;
    LRAM_HIP_CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    LRAM_HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto reset_state = [&]() {
      LRAM_HIP_CHECK(hipMemcpy(pa.conv_state, conv0, (size_t)B * K * inner * 4, hipMemcpyDeviceToDevice));
      LRAM_HIP_CHECK(hipMemset(pa.n_state, 0, (size_t)B * inner * 4));
      LRAM_HIP_CHECK(hipMemset(pa.m_state, 0, (size_t)B * NH * 4));
      LRAM_HIP_CHECK(hipDeviceSynchronize());
    };
    struct Out {
      float* p;
      size_t n;
      std::vector<float> ref, got;
    };
    std::vector<Out> outs = {{pa.scal, (size_t)B * T * NH * 4, {}, {}}, {pa.n_state, (size_t)B * inner, {}, {}},
                             {pa.m_state, (size_t)B * NH, {}, {}}, {pa.q, (size_t)B * T * inner, {}, {}}};
    reset_state();
    launch_mlstm_pre(pa, s1);  // solo reference
    LRAM_HIP_CHECK(hipDeviceSynchronize());
    for (auto& o : outs) {
      o.ref.resize(o.n), o.got.resize(o.n);
      LRAM_HIP_CHECK(hipMemcpy(o.ref.data(), o.p, o.n * 4, hipMemcpyDeviceToHost));
    }
    int64_t diff = 0;
    for (int it = 0; it < iters; ++it) {
      reset_state();
      launch_mlstm_pre(pa, s1);
      launch_gemm_bf16x3(g, s2);  // the other slice's projection, concurrently
      LRAM_HIP_CHECK(hipDeviceSynchronize());
      for (auto& o : outs) {
        LRAM_HIP_CHECK(hipMemcpy(o.got.data(), o.p, o.n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < o.n; ++i) diff += o.got[i] != o.ref[i];
      }
    }
    *n_diff = diff;
    (void)hipStreamDestroy(s1);
    (void)hipStreamDestroy(s2);
    for (float* p : owned) (void)hipFree(p);
    (void)hipFree(planes);
    return 0;
  } catch (const std::exception&) {
    return 1;
  }
}
