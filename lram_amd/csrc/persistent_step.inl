// Whole env-step of the xLSTM stack as ONE cooperative launch, for small batches (B <= kPersistMaxBatch env slots).
// Included by xlstm_kernels.hip inside namespace lram::{anonymous} (it calls the mLSTM front-end and cell bodies of
// that file).
//
// The reference's own operating point is one env (src/callbacks/evaluation.py:80).  There the launch-per-kernel path is
// bound by its ~70 dependent launches (0.4 ms per env-step, hipGraph replay included): every kernel moves a few
// kilobytes.  Here a fixed grid of workgroups walks through the step's phases and meets at a device-wide barrier
// (one atomic counter, agent-scope release / acquire) wherever a phase needs another workgroup's results:
//
//   front end   one workgroup per token row: state / return-to-go / reward embedding + embed_ln
//   mLSTM block A  proj_up: every workgroup normalises the <= 24 rows itself (LDS) and owns a range of output columns
//               B  conv / q,k,v / gates / normaliser state: mlstm_pre_body, one workgroup per env
//               C  matrix-memory update + readout: mlstm_cell_body over (env, head, 64-column slice)
//               D  proj_down (+ residual): every workgroup rebuilds the gated, group-normalised rows itself
//   sLSTM block S1 norm + conv (per env)   S2 the four headwise gate projections   S3 recurrent cell, the step's three
//               tokens in sequence inside one workgroup per (env, head)   S4 group norm + residual + FFN norm (per row)
//               S5 FFN up   S6 GELU gate + FFN down (+ residual)
//   head        post-blocks norm + action logits, then argmax / de-tokenisation
//
// 36 barriers for the 16M stack instead of ~70 launch boundaries, no intermediate kernel re-launch latency, weights
// streamed once per step from L2 / the memory-side cache.  All arithmetic is fp32 fma (the M <= 8 path of the
// launch-per-kernel engine is too), the summation order differs from the tile kernels' within the parity tolerances.
// Recurrent state stays in the engine's reference-layout buffers, so every other entry point (export / import, prefill,
// a later larger batch) sees it unchanged.

struct PsBar {
  unsigned long long* trace;   // optional [n]: wall clock (100 MHz) of workgroup 0 after each barrier + its own arrival
  int n_trace;
  unsigned long long* counter;
  unsigned long long target;
  unsigned int nwg;
  int* abort_dev;   // device word next to the counter: set when a barrier timed out, polled by the others
  int* err_host;    // host-mapped word the engine checks before the next call
};

// Bounded spin: a barrier that cannot complete (a workgroup that never arrives) sets the error words and lets every
// workgroup run to the end instead of hanging the device.
__device__ __forceinline__ void ps_grid_sync(PsBar& g) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (g.trace != nullptr && blockIdx.x == 0) g.trace[g.n_trace++] = wall_clock64();
    g.target += g.nwg;
    __threadfence();  // release this workgroup's writes at device scope
    atomicAdd(g.counter, 1ull);
    long spins = 0;
    while (__hip_atomic_load(g.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.target) {
      __builtin_amdgcn_s_sleep(2);
      ++spins;
      if ((spins & 1023) == 0 && (spins > (1l << 22) || __hip_atomic_load(g.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        atomicExch(g.abort_dev, 1);
        *g.err_host = 1;
        break;
      }
    }
    __threadfence();
    if (g.trace != nullptr && blockIdx.x == 0) g.trace[g.n_trace++] = wall_clock64();
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: no stale lines from before the barrier
}

constexpr int kPsRowGroup = 8;  // rows per GEMV pass (accumulators per lane: kPsRowGroup x kPsCols)
constexpr int kPsCols = 4;      // output columns per wave and pass

// rows of width d: dst[r][:] = norm(src[r][:]) * gamma (+ beta); one wave per row (same arithmetic as row_norm_kernel)
__device__ __forceinline__ void ps_norm_row(const float* src, float* dst, int d, const float* gamma, const float* beta,
                                            float eps, int rms, int lane, float* dst2 = nullptr) {
  float s = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    s += v.x + v.y + v.z + v.w;
  }
  const float mean = rms ? 0.f : wave_sum(s) / (float)d;
  float q = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    q += dx * dx + dy * dy + dz * dz + dw * dw;
  }
  const float var = wave_sum(q) / (float)d;
  const float rstd = rms ? rsqrtf(var + eps) : 1.f / sqrtf(var + eps);
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    const float4 g = *reinterpret_cast<const float4*>(gamma + i);
    float4 o = make_float4((v.x - mean) * rstd * g.x, (v.y - mean) * rstd * g.y, (v.z - mean) * rstd * g.z,
                           (v.w - mean) * rstd * g.w);
    if (beta != nullptr) {
      const float4 bb = *reinterpret_cast<const float4*>(beta + i);
      o.x += bb.x, o.y += bb.y, o.z += bb.z, o.w += bb.w;
    }
    *reinterpret_cast<float4*>(dst + i) = o;
    if (dst2 != nullptr) *reinterpret_cast<float4*>(dst2 + i) = o;
  }
}

// C[r][n] = sum_k A[r][k] W[n][k] (+ bias[n]) (+ C[r][n]) for r < rows (<= kPsRowGroup), n in [n_begin, n_end):
// A in LDS or global (row stride lda), W global rows (stride ldw), K a multiple of 4.  Every wave takes kPsCols columns
// at a time, its lanes stride over K with 16-byte loads, a wave reduction finishes the dot products.
__device__ __forceinline__ void ps_gemv(const float* A, int lda, int rows, const float* W, int64_t ldw, int K,
                                        int n_begin, int n_end, const float* bias, float* C, int64_t ldc, bool accumulate,
                                        int lane, int wave) {
  for (int n0 = n_begin + wave * kPsCols; n0 < n_end; n0 += 4 * kPsCols) {
    float acc[kPsRowGroup][kPsCols];
#pragma unroll
    for (int m = 0; m < kPsRowGroup; ++m)
#pragma unroll
      for (int c = 0; c < kPsCols; ++c) acc[m][c] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float4 w[kPsCols];
#pragma unroll
      for (int c = 0; c < kPsCols; ++c)
        w[c] = (n0 + c < n_end) ? *reinterpret_cast<const float4*>(W + (int64_t)(n0 + c) * ldw + k) : f4_zero();
#pragma unroll
      for (int m = 0; m < kPsRowGroup; ++m) {
        if (m < rows) {
          const float4 a = *reinterpret_cast<const float4*>(A + (int64_t)m * lda + k);
#pragma unroll
          for (int c = 0; c < kPsCols; ++c) {
            acc[m][c] = fmaf(a.x, w[c].x, acc[m][c]);
            acc[m][c] = fmaf(a.y, w[c].y, acc[m][c]);
            acc[m][c] = fmaf(a.z, w[c].z, acc[m][c]);
            acc[m][c] = fmaf(a.w, w[c].w, acc[m][c]);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < kPsRowGroup; ++m) {
      if (m < rows) {
#pragma unroll
        for (int c = 0; c < kPsCols; ++c) {
          const float v = wave_sum(acc[m][c]);
          if (lane == 0 && n0 + c < n_end) {
            float o = v;
            if (bias != nullptr) o += bias[n0 + c];
            float* dst = C + (int64_t)m * ldc + n0 + c;
            *dst = accumulate ? *dst + o : o;
          }
        }
      }
    }
  }
}

// contiguous share [lo, hi) of n items (in units of `unit`) for workgroup wg of nwg
__device__ __forceinline__ void ps_share(int n, int unit, int wg, int nwg, int& lo, int& hi) {
  const int units = (n + unit - 1) / unit;
  const int per = (units + nwg - 1) / nwg;
  lo = min(n, wg * per * unit);
  hi = min(n, (wg + 1) * per * unit);
}

template <int T>
__device__ __forceinline__ void ps_front(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                       float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
// ================= front end: token rows (s, rtg, r) + embed_ln =================
for (int r = wg; r < R; r += nwg) {
  const int b = r / T, t = r - b * T;
  float* xr = smem;  // [D]
  if (t == 0) {
    if (p.emb) {
      for (int i = tid; i < D; i += 256) xr[i] = p.obs[(int64_t)b * D + i];
    } else {
      ps_gemv(p.obs + (int64_t)b * p.state_dim, p.state_dim, 1, p.w_state, p.state_dim, p.state_dim, 0, D, p.b_state,
              xr, D, false, lane, wave);
    }
  } else {
    const float s = t == 1 ? p.rtg[b] : p.rew[b];
    const float* w = t == 1 ? p.w_rtg : p.w_rew;
    const float* bb = t == 1 ? p.b_rtg : p.b_rew;
    for (int i = tid; i < D; i += 256) xr[i] = s * w[i] + bb[i];
  }
  __syncthreads();
  if (wave == 0) ps_norm_row(xr, p.X + (int64_t)r * D, D, p.eln_g, p.eln_b, 1e-5f, 0, lane, p.TOK + (int64_t)r * D);
  __syncthreads();
}
}

template <int T>
__device__ __forceinline__ void ps_mlstm_A(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                         float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int inner = p.inner, NH = p.NH, DH = p.DH;
  (void)inner, (void)NH, (void)DH;
  // ---- A: U = LN(X) proj_up^T ----
  {
    int lo, hi;
    ps_share(2 * inner, kPsCols, wg, nwg, lo, hi);
    if (lo < hi) {
      for (int r0 = 0; r0 < R; r0 += kPsRowGroup) {
        const int rows = min(kPsRowGroup, R - r0);
        for (int r = wave; r < rows; r += 4)
          ps_norm_row(p.X + (int64_t)(r0 + r) * D, smem + r * D, D, w.norm_g, w.norm_b, p.ln_eps, p.norm_is_rms, lane);
        __syncthreads();
        ps_gemv(smem, D, rows, w.proj_up, D, D, lo, hi, nullptr, p.U + (int64_t)r0 * 2 * inner, 2 * inner, false, lane,
                wave);
        __syncthreads();
      }
    }
  }
}

template <int T>
__device__ __forceinline__ void ps_mlstm_B(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                         float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int inner = p.inner, NH = p.NH, DH = p.DH;
  (void)inner, (void)NH, (void)DH;
  // ---- B: conv / q,k,v / gates / n, m state (one workgroup per env) ----
  {
    MlstmPreArgs a;
    a.u = p.U, a.conv_state = w.conv, a.n_state = w.n, a.m_state = w.m, a.conv_w = w.conv_w, a.conv_b = w.conv_b;
    a.wq = w.wq, a.wk = w.wk, a.wv = w.wv, a.wi = w.wi, a.bi = w.bi, a.wf = w.wf, a.bf = w.bf;
    a.q = p.Q, a.k = p.K, a.v = p.V, a.xa = p.XA, a.scal = p.SCAL, a.reset = p.reset;
    a.B = B, a.T = T, a.inner = inner, a.NH = NH, a.K = 4;
    for (int b = wg; b < B; b += nwg) {
      switch (NH) {
        case 1: mlstm_pre_body<T, 1>(a, b); break;
        case 2: mlstm_pre_body<T, 2>(a, b); break;
        case 4: mlstm_pre_body<T, 4>(a, b); break;
        default: mlstm_pre_body<T, 8>(a, b); break;
      }
      __syncthreads();
    }
  }
}

template <int T>
__device__ __forceinline__ void ps_mlstm_C(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                         float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int inner = p.inner, NH = p.NH, DH = p.DH;
  (void)inner, (void)NH, (void)DH;
  // ---- C: matrix memory update + readout over (env, head, 64-column slice) ----
  {
    MlstmCellArgs a;
    a.C = w.s0, a.q = p.Q, a.k = p.K, a.v = p.V, a.scal = p.SCAL, a.h = p.H, a.reset = p.reset;
    a.B = B, a.T = T, a.NH = NH, a.DH = DH;
    const int nsl = DH / 64, items = B * NH * nsl;
    for (int it = wg; it < items; it += nwg) {
      const int slice = it % nsl, h = (it / nsl) % NH, b = it / (nsl * NH);
      mlstm_cell_body<T, 16, 8>(a, slice, h, b, smem);
      __syncthreads();
    }
  }
}

template <int T>
__device__ __forceinline__ void ps_mlstm_D(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                         float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int inner = p.inner, NH = p.NH, DH = p.DH;
  (void)inner, (void)NH, (void)DH;
  // ---- D: X += ((GN(h) + skip xa) silu(z)) proj_down^T ----
  {
    int lo, hi;
    ps_share(D, kPsCols, wg, nwg, lo, hi);
    if (lo < hi) {
      for (int r0 = 0; r0 < R; r0 += kPsRowGroup) {
        const int rows = min(kPsRowGroup, R - r0);
        // gated rows into LDS: one wave per (row, head)
        for (int it = wave; it < rows * NH; it += 4) {
          const int r = it / NH, h = it - r * NH;
          const float* src = p.H + (int64_t)(r0 + r) * inner + (int64_t)h * DH;
          float s = 0.f;
          for (int i = lane * 4; i < DH; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            s += v.x + v.y + v.z + v.w;
          }
          const float mean = wave_sum(s) / (float)DH;
          float q = 0.f;
          for (int i = lane * 4; i < DH; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            q += dx * dx + dy * dy + dz * dz + dw * dw;
          }
          const float rstd = 1.f / sqrtf(wave_sum(q) / (float)DH + p.ln_eps);
          for (int i = lane * 4; i < DH; i += 256) {
            const int hd = h * DH + i;
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            const float4 g = *reinterpret_cast<const float4*>(w.on_g + hd);
            float4 o = make_float4((v.x - mean) * rstd * g.x, (v.y - mean) * rstd * g.y, (v.z - mean) * rstd * g.z,
                                   (v.w - mean) * rstd * g.w);
            if (w.on_b != nullptr) {
              const float4 bb = *reinterpret_cast<const float4*>(w.on_b + hd);
              o.x += bb.x, o.y += bb.y, o.z += bb.z, o.w += bb.w;
            }
            const float4 sk = *reinterpret_cast<const float4*>(w.skip + hd);
            const float4 xa = *reinterpret_cast<const float4*>(p.XA + (int64_t)(r0 + r) * inner + hd);
            const float4 z = *reinterpret_cast<const float4*>(p.U + (int64_t)(r0 + r) * 2 * inner + inner + hd);
            o.x = (o.x + sk.x * xa.x) * silu_f(z.x);
            o.y = (o.y + sk.y * xa.y) * silu_f(z.y);
            o.z = (o.z + sk.z * xa.z) * silu_f(z.z);
            o.w = (o.w + sk.w * xa.w) * silu_f(z.w);
            *reinterpret_cast<float4*>(smem + r * inner + hd) = o;
          }
        }
        __syncthreads();
        ps_gemv(smem, inner, rows, w.proj_down, inner, inner, lo, hi, nullptr, p.X + (int64_t)r0 * D, D, true, lane,
                wave);
        __syncthreads();
      }
    }
  }
}

template <int T>
__device__ __forceinline__ void ps_slstm_S1(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                          float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int NH = p.NH, SDH = p.SDH, F = p.F;
  (void)NH, (void)SDH, (void)F;
  // ---- S1: XN = LN(X); XC = silu(conv(XN)); conv state; state reset (one workgroup per env) ----
  for (int b = wg; b < B; b += nwg) {
    if (wave < T) ps_norm_row(p.X + (int64_t)(b * T + wave) * D, p.XN + (int64_t)(b * T + wave) * D, D, w.norm_g,
                              w.norm_b, p.ln_eps, p.norm_is_rms, lane);
    if (T > 4 && wave == 0)
      for (int t = 4; t < T; ++t)
        ps_norm_row(p.X + (int64_t)(b * T + t) * D, p.XN + (int64_t)(b * T + t) * D, D, w.norm_g, w.norm_b, p.ln_eps,
                    p.norm_is_rms, lane);
    __syncthreads();
    const bool rs = p.reset != nullptr && p.reset[b] != 0;
    for (int c0 = tid * 4; c0 < D; c0 += 1024) {
      float4 win[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        win[k] = rs ? f4_zero() : *reinterpret_cast<const float4*>(w.conv + ((int64_t)b * 4 + k) * D + c0);
      float4 cw[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) cw[c] = *reinterpret_cast<const float4*>(w.conv_w + (int64_t)(c0 + c) * 4);
      const float4 cb = *reinterpret_cast<const float4*>(w.conv_b + c0);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int64_t row = (int64_t)b * T + t;
        const float4 x = *reinterpret_cast<const float4*>(p.XN + row * D + c0);
        win[0] = win[1], win[1] = win[2], win[2] = win[3], win[3] = x;
        float4 y;
        y.x = win[0].x * cw[0].x + win[1].x * cw[0].y + win[2].x * cw[0].z + win[3].x * cw[0].w + cb.x;
        y.y = win[0].y * cw[1].x + win[1].y * cw[1].y + win[2].y * cw[1].z + win[3].y * cw[1].w + cb.y;
        y.z = win[0].z * cw[2].x + win[1].z * cw[2].y + win[2].z * cw[2].z + win[3].z * cw[2].w + cb.z;
        y.w = win[0].w * cw[3].x + win[1].w * cw[3].y + win[2].w * cw[3].z + win[3].w * cw[3].w + cb.w;
        *reinterpret_cast<float4*>(p.Q + row * D + c0) = make_float4(silu_f(y.x), silu_f(y.y), silu_f(y.z), silu_f(y.w));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(w.conv + ((int64_t)b * 4 + k) * D + c0) = win[k];
      if (rs) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<float4*>(w.s0 + ((int64_t)s * p.state_B + b) * D + c0) = f4_zero();
      }
    }
    __syncthreads();
  }
}

template <int T>
__device__ __forceinline__ void ps_slstm_S2(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                          float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int NH = p.NH, SDH = p.SDH, F = p.F;
  (void)NH, (void)SDH, (void)F;
  // ---- S2: gate pre-activations, four headwise (block-diagonal) projections: U[r][g][h*SDH + o] ----
  {
    int lo, hi;  // share of the 4 * D flattened (gate, head, out) columns; a (gate, head) segment has SDH of them
    ps_share(4 * D, kPsCols, wg, nwg, lo, hi);
    for (int seg = lo / SDH; seg * SDH < hi; ++seg) {
      const int g = seg / NH, h = seg - g * NH;
      const int n_lo = max(lo, seg * SDH) - seg * SDH, n_hi = min(hi, (seg + 1) * SDH) - seg * SDH;
      const float* A = (g < 2 ? p.Q : p.XN) + (int64_t)h * SDH;
      for (int r0 = 0; r0 < R; r0 += kPsRowGroup)
        ps_gemv(A + (int64_t)r0 * D, D, min(kPsRowGroup, R - r0), w.gate_w[g] + (int64_t)h * SDH * SDH, SDH, SDH, n_lo,
                n_hi, nullptr, p.U + (int64_t)r0 * 4 * D + (int64_t)g * D + (int64_t)h * SDH, 4 * D, false, lane, wave);
    }
  }
}

template <int T>
__device__ __forceinline__ void ps_slstm_S3(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                          float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int NH = p.NH, SDH = p.SDH, F = p.F;
  (void)NH, (void)SDH, (void)F;
  // ---- S3: recurrent cell, T tokens in sequence, one workgroup per (env, head) ----
  for (int it = wg; it < B * NH; it += nwg) {
    const int b = it / NH, h = it - b * NH;
    float* ys = smem;            // [SDH] y_{t-1} of this head
    float* raw = smem + SDH;     // [4][SDH]
    const int64_t BH = (int64_t)p.state_B * D;
    float* st = w.s0 + (int64_t)b * D + (int64_t)h * SDH;
    for (int i = tid; i < SDH; i += 256) ys[i] = st[i];
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const int64_t row = (int64_t)b * T + t;
      // raw[g][o] = sum_i y[i] R[h][g][o][i]   (rt is [head, gate, out, in], K-contiguous)
      ps_gemv(ys, SDH, 1, w.rt + (int64_t)h * 4 * SDH * SDH, SDH, SDH, 0, 4 * SDH, nullptr, raw, 4 * SDH, false, lane,
              wave);
      __syncthreads();
      for (int o = tid; o < SDH; o += 256) {
        const int c = h * SDH + o;
        const float* gt = p.U + row * 4 * D + c;
        const float iraw = gt[0] + raw[o] + w.rbias[c];
        const float fraw = gt[D] + raw[SDH + o] + w.rbias[D + c];
        const float zraw = gt[2 * D] + raw[2 * SDH + o] + w.rbias[2 * D + c];
        const float oraw = gt[3 * D] + raw[3 * SDH + o] + w.rbias[3 * D + c];
        const float cs = st[BH + o], ns = st[2 * BH + o], ms = st[3 * BH + o];
        const float logfplusm = ms + log_sigmoid(fraw);
        const float mnew = (ns == 0.f) ? iraw : fmaxf(iraw, logfplusm);
        const float ogate = sigmoid_f(oraw);
        const float igate = fminf(expf(iraw - mnew), 1.f);
        const float fgate = fminf(expf(logfplusm - mnew), 1.f);
        const float cnew = fgate * cs + igate * tanhf(zraw);
        const float nnew = fgate * ns + igate;
        const float ynew = ogate * cnew / nnew;
        st[o] = ynew, st[BH + o] = cnew, st[2 * BH + o] = nnew, st[3 * BH + o] = mnew;
        ys[o] = ynew;
        p.H[row * D + c] = ynew;
      }
      __syncthreads();
    }
  }
}

template <int T>
__device__ __forceinline__ void ps_slstm_S4(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                          float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int NH = p.NH, SDH = p.SDH, F = p.F;
  (void)NH, (void)SDH, (void)F;
  // ---- S4: X += GN(y) gamma; XN = LN_ffn(X)   (one workgroup per row) ----
  for (int r = wg; r < R; r += nwg) {
    float* xr = smem;  // [D]
    for (int h = wave; h < NH; h += 4) {
      const float* src = p.H + (int64_t)r * D + (int64_t)h * SDH;
      float s = 0.f;
      for (int i = lane * 4; i < SDH; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        s += v.x + v.y + v.z + v.w;
      }
      const float mean = wave_sum(s) / (float)SDH;
      float q = 0.f;
      for (int i = lane * 4; i < SDH; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
      }
      const float rstd = 1.f / sqrtf(wave_sum(q) / (float)SDH + p.ln_eps);
      for (int i = lane * 4; i < SDH; i += 256) {
        const int hd = h * SDH + i;
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        const float4 g = *reinterpret_cast<const float4*>(w.gn_g + hd);
        float4 o = make_float4((v.x - mean) * rstd * g.x, (v.y - mean) * rstd * g.y, (v.z - mean) * rstd * g.z,
                               (v.w - mean) * rstd * g.w);
        if (w.gn_b != nullptr) {
          const float4 bb = *reinterpret_cast<const float4*>(w.gn_b + hd);
          o.x += bb.x, o.y += bb.y, o.z += bb.z, o.w += bb.w;
        }
        float4 x = *reinterpret_cast<const float4*>(p.X + (int64_t)r * D + hd);
        x.x += o.x, x.y += o.y, x.z += o.z, x.w += o.w;
        *reinterpret_cast<float4*>(p.X + (int64_t)r * D + hd) = x;
        *reinterpret_cast<float4*>(xr + hd) = x;
      }
    }
    __syncthreads();
    if (wave == 0) ps_norm_row(xr, p.XN + (int64_t)r * D, D, w.ffn_norm_g, w.ffn_norm_b, p.ln_eps, p.norm_is_rms, lane);
    __syncthreads();
  }
}

template <int T>
__device__ __forceinline__ void ps_slstm_S5(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                          float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int NH = p.NH, SDH = p.SDH, F = p.F;
  (void)NH, (void)SDH, (void)F;
  // ---- S5: U = XN ffn_up^T ----
  {
    int lo, hi;
    ps_share(2 * F, kPsCols, wg, nwg, lo, hi);
    if (lo < hi)
      for (int r0 = 0; r0 < R; r0 += kPsRowGroup)
        ps_gemv(p.XN + (int64_t)r0 * D, D, min(kPsRowGroup, R - r0), w.ffn_up, D, D, lo, hi, nullptr,
                p.U + (int64_t)r0 * 2 * F, 2 * F, false, lane, wave);
  }
}

template <int T>
__device__ __forceinline__ void ps_slstm_S6(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                          float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
  const int NH = p.NH, SDH = p.SDH, F = p.F;
  (void)NH, (void)SDH, (void)F;
  // ---- S6: X += (gelu(g) * u) ffn_down^T ----
  {
    int lo, hi;
    ps_share(D, kPsCols, wg, nwg, lo, hi);
    if (lo < hi) {
      for (int r0 = 0; r0 < R; r0 += kPsRowGroup) {
        const int rows = min(kPsRowGroup, R - r0);
        for (int idx = tid * 4; idx < rows * F; idx += 1024) {
          const int r = idx / F, c = idx - r * F;
          const float4 g = *reinterpret_cast<const float4*>(p.U + (int64_t)(r0 + r) * 2 * F + c);
          const float4 u = *reinterpret_cast<const float4*>(p.U + (int64_t)(r0 + r) * 2 * F + F + c);
          *reinterpret_cast<float4*>(smem + r * F + c) =
              make_float4(gelu_f(g.x) * u.x, gelu_f(g.y) * u.y, gelu_f(g.z) * u.z, gelu_f(g.w) * u.w);
        }
        __syncthreads();
        ps_gemv(smem, F, rows, w.ffn_down, F, F, lo, hi, nullptr, p.X + (int64_t)r0 * D, D, true, lane, wave);
        __syncthreads();
      }
    }
  }
}

template <int T>
__device__ __forceinline__ void ps_head(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                      float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
// ================= head: post-blocks norm, action logits =================
{
  const int nlog = p.act_dim * p.n_vocab;
  int lo, hi;
  ps_share(nlog, kPsCols, wg, nwg, lo, hi);
  // every row's hidden state (taps) -- row r by workgroup r % nwg; the prediction rows are normalised again below
  for (int r = wg; r < R; r += nwg) {
    if (wave == 0) ps_norm_row(p.X + (int64_t)r * D, p.HID + (int64_t)r * D, D, p.post_g, p.post_b, p.ln_eps, p.norm_is_rms, lane);
  }
  if (lo < hi) {
    for (int b = wave; b < B; b += 4)
      ps_norm_row(p.X + (int64_t)(b * T + p.pred_token) * D, smem + b * D, D, p.post_g, p.post_b, p.ln_eps, p.norm_is_rms,
                  lane);
    __syncthreads();
    ps_gemv(smem, D, B, p.w_head, D, D, lo, hi, p.b_head, p.LOGITS, nlog, false, lane, wave);
  }
}
}

template <int T>
__device__ __forceinline__ void ps_argmax(const PersistArgs& p, const PersistBlock& w, const int wg, const int nwg,
                                        float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D, R = B * T;
  (void)tid, (void)lane, (void)wave, (void)B, (void)D, (void)R;
// ================= argmax / de-tokenise: one wave per (env, action dim) =================
{
  const int ndim = p.discrete ? 1 : p.act_dim;
  // a device-wide barrier of the whole-step kernel timed out somewhere before this phase: the logits are not to be
  // trusted, and callers enqueue asynchronously (the error code only reaches them with the NEXT lram_step) -- poison
  // what this call hands back (NaN actions, token -1) so that nothing downstream can consume it as a result
  const bool aborted = p.abort_dev != nullptr && __hip_atomic_load(p.abort_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  for (int item = wg * 4 + wave; item < B * ndim; item += nwg * 4) {
    const int b = item / ndim, j = item - b * ndim;
    if (aborted) {
      if (lane == 0) {
        if (p.tokens != nullptr) p.tokens[(int64_t)b * p.act_dim + j] = -1;
        p.actions[(int64_t)b * p.act_dim + j] = __builtin_nanf("");
      }
      continue;
    }
    const float* lg = p.LOGITS + (int64_t)b * p.act_dim * p.n_vocab + (int64_t)j * p.n_vocab;
    const int n = p.discrete ? p.n_discrete : p.n_vocab;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
      const float v = lg[i];
      if (v > best || (v == best && i < bi)) best = v, bi = i;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
    }
    if (lane == 0) {
      if (p.tokens != nullptr) p.tokens[(int64_t)b * p.act_dim + j] = bi;
      float out;
      if (p.discrete) {
        out = (float)bi;
      } else {
        int tk = bi - p.n_discrete;
        tk = tk < 0 ? 0 : tk;
        out = (float)tk * ((p.tok_max - p.tok_min) / (float)p.action_channels) + p.tok_min;
      }
      p.actions[(int64_t)b * p.act_dim + j] = out;
    }
  }
}
}

// ---- the cooperative whole-step kernel: every phase, a device-wide barrier after each ----
template <int T>
__global__ __launch_bounds__(256) void xlstm_persistent_step_kernel(PersistArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wg = blockIdx.x, nwg = gridDim.x;
  PsBar bar{p.trace, 0, p.counter, p.base, (unsigned)nwg, p.abort_dev, p.err_host};
  if (p.trace != nullptr && wg == 0 && threadIdx.x == 0) p.trace[bar.n_trace++] = wall_clock64();
  ps_front<T>(p, p.blocks[0], wg, nwg, smem);
  ps_grid_sync(bar);
  for (int blk = 0; blk < p.n_blocks; ++blk) {
    const PersistBlock& w = p.blocks[blk];
    if (!w.is_slstm) {
      ps_mlstm_A<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_mlstm_B<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_mlstm_C<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_mlstm_D<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
    } else {
      ps_slstm_S1<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_slstm_S2<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_slstm_S3<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_slstm_S4<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_slstm_S5<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
      ps_slstm_S6<T>(p, w, wg, nwg, smem);
      ps_grid_sync(bar);
    }
  }
  ps_head<T>(p, p.blocks[0], wg, nwg, smem);
  ps_grid_sync(bar);
  ps_argmax<T>(p, p.blocks[0], wg, nwg, smem);
}

// =============================================================================================================
// The same phases as kernels of their own ("small-batch path"): launch boundaries instead of device-wide barriers, each
// phase with as many workgroups as it has independent work.  Against the generic launch-per-kernel path this fuses
// the norms and gates into the GEMV that consumes them, the four sLSTM gate projections into one launch and each
// token's recurrent GEMV with its pointwise cell: 40 launches per env-step of the 16M stack instead of ~70.
// =============================================================================================================
// front end, part 1: raw token rows (state embedding columns spread over workgroups; rtg / reward rows)
template <int T>
__device__ __forceinline__ void ps_front_embed(const PersistArgs& p, const int wg, const int nwg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, D = p.D;
  const int chunks = (D + 63) / 64;  // 64 columns per work item
  for (int it = wg; it < B * T * chunks; it += nwg) {
    const int r = it / chunks, c0 = (it - r * chunks) * 64, c1 = min(D, c0 + 64);
    const int b = r / T, t = r - b * T;
    float* xr = p.X + (int64_t)r * D;
    if (t == 0) {
      if (p.emb) {
        for (int i = c0 + tid; i < c1; i += 256) xr[i] = p.obs[(int64_t)b * D + i];
      } else {
        ps_gemv(p.obs + (int64_t)b * p.state_dim, p.state_dim, 1, p.w_state, p.state_dim, p.state_dim, c0, c1, p.b_state, xr,
                D, false, lane, wave);
      }
    } else {
      const float s = t == 1 ? p.rtg[b] : p.rew[b];
      const float* wv = t == 1 ? p.w_rtg : p.w_rew;
      const float* bb = t == 1 ? p.b_rtg : p.b_rew;
      for (int i = c0 + tid; i < c1; i += 256) xr[i] = s * wv[i] + bb[i];
    }
  }
}
// front end, part 2: embed_ln in place (+ the token tap)
template <int T>
__device__ __forceinline__ void ps_front_norm(const PersistArgs& p, const int wg, const int nwg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wg * 4 + wave; r < p.B * T; r += nwg * 4)
    ps_norm_row(p.X + (int64_t)r * p.D, p.X + (int64_t)r * p.D, p.D, p.eln_g, p.eln_b, 1e-5f, 0, lane, p.TOK + (int64_t)r * p.D);
}

// sLSTM recurrent cell for token t over (env, head, 16-unit chunk): raw = gates + R y_{t-1} + b, pointwise update.
// y_{t-1} comes from the state (t == 0) or from the previous token's output row, y_t goes to the output row (and, after
// the last token, to the state), so no workgroup reads what another one writes in the same launch.
template <int T>
__device__ __forceinline__ void ps_slstm_token(const PersistArgs& p, const PersistBlock& w, const int t, const int wg,
                                               const int nwg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = p.B, D = p.D, NH = p.NH, SDH = p.SDH;
  const int chunks = SDH / 16;  // 16 units per workgroup, 4 per wave
  const int64_t BH = (int64_t)p.state_B * D;
  for (int it = wg; it < B * NH * chunks; it += nwg) {
    const int b = it / (NH * chunks), h = (it / chunks) % NH, o0 = (it % chunks) * 16 + wave * 4;
    const int64_t row = (int64_t)b * T + t;
    float* st = w.s0 + (int64_t)b * D + (int64_t)h * SDH;
    const float* yprev = t == 0 ? st : p.H + (row - 1) * D + (int64_t)h * SDH;
    const float* Rh = w.rt + (int64_t)h * 4 * SDH * SDH;
    float acc[4][4];  // [unit][gate]
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[u][g] = 0.f;
    for (int k = lane * 4; k < SDH; k += 256) {
      const float4 y = *reinterpret_cast<const float4*>(yprev + k);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 r = *reinterpret_cast<const float4*>(Rh + ((int64_t)g * SDH + o0 + u) * SDH + k);
          acc[u][g] = fmaf(y.x, r.x, acc[u][g]);
          acc[u][g] = fmaf(y.y, r.y, acc[u][g]);
          acc[u][g] = fmaf(y.z, r.z, acc[u][g]);
          acc[u][g] = fmaf(y.w, r.w, acc[u][g]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[u][g] = wave_sum(acc[u][g]);
    if (lane < 4) {
      float raw[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = acc[0][g];
#pragma unroll
        for (int u = 1; u < 4; ++u) v = lane == u ? acc[u][g] : v;
        raw[g] = v;
      }
      const int o = o0 + lane, c = h * SDH + o;
      const float* gt = p.U + row * 4 * D + c;
      const float iraw = gt[0] + raw[0] + w.rbias[c];
      const float fraw = gt[D] + raw[1] + w.rbias[D + c];
      const float zraw = gt[2 * D] + raw[2] + w.rbias[2 * D + c];
      const float oraw = gt[3 * D] + raw[3] + w.rbias[3 * D + c];
      const float cs = st[BH + o], ns = st[2 * BH + o], ms = st[3 * BH + o];
      const float logfplusm = ms + log_sigmoid(fraw);
      const float mnew = (ns == 0.f) ? iraw : fmaxf(iraw, logfplusm);
      const float ogate = sigmoid_f(oraw);
      const float igate = fminf(expf(iraw - mnew), 1.f);
      const float fgate = fminf(expf(logfplusm - mnew), 1.f);
      const float cnew = fgate * cs + igate * tanhf(zraw);
      const float nnew = fgate * ns + igate;
      const float ynew = ogate * cnew / nnew;
      st[BH + o] = cnew, st[2 * BH + o] = nnew, st[3 * BH + o] = mnew;
      if (t == T - 1) st[o] = ynew;
      p.H[row * D + c] = ynew;
    }
  }
}

enum PsPhase { kPhFrontEmbed, kPhFrontNorm, kPhMA, kPhMB, kPhMC, kPhMD, kPhS1, kPhS2, kPhS3Tok, kPhS4, kPhS5, kPhS6, kPhHead,
               kPhArgmax };

template <int T, int PH>
__global__ __launch_bounds__(256) void xlstm_phase_kernel(PersistArgs p, int blk, int tok) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wg = blockIdx.x, nwg = gridDim.x;
  const PersistBlock& w = p.blocks[blk];
  if (PH == kPhFrontEmbed) ps_front_embed<T>(p, wg, nwg);
  if (PH == kPhFrontNorm) ps_front_norm<T>(p, wg, nwg);
  if (PH == kPhMA) ps_mlstm_A<T>(p, w, wg, nwg, smem);
  if (PH == kPhMB) ps_mlstm_B<T>(p, w, wg, nwg, smem);
  if (PH == kPhMC) ps_mlstm_C<T>(p, w, wg, nwg, smem);
  if (PH == kPhMD) ps_mlstm_D<T>(p, w, wg, nwg, smem);
  if (PH == kPhS1) ps_slstm_S1<T>(p, w, wg, nwg, smem);
  if (PH == kPhS2) ps_slstm_S2<T>(p, w, wg, nwg, smem);
  if (PH == kPhS3Tok) ps_slstm_token<T>(p, w, tok, wg, nwg);
  if (PH == kPhS4) ps_slstm_S4<T>(p, w, wg, nwg, smem);
  if (PH == kPhS5) ps_slstm_S5<T>(p, w, wg, nwg, smem);
  if (PH == kPhS6) ps_slstm_S6<T>(p, w, wg, nwg, smem);
  if (PH == kPhHead) ps_head<T>(p, w, wg, nwg, smem);
  if (PH == kPhArgmax) ps_argmax<T>(p, w, wg, nwg, smem);
}
