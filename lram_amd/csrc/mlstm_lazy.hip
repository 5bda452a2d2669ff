// Lazy matrix memory for the mLSTM step on gfx950.
//
// The step kernel of xlstm_kernels.hip reads and rewrites every C element once per env-step -- the algorithmic
// minimum while C_t is kept materialised.  It does not have to be: with the stabilised per-token factors
// (f_t, i_t <= 1; [3P] recurrent_step_stabilized_simple)
//     C_t = g_t C_base + sum_j c_{t,j} khat_j v_j^T ,      g_t = g_{t-1} f_t ,  c_{t,j} = c_{t-1,j} f_t ,  c_{t,t} = i_t
//     q_t^T C_t = g_t (q_t^T C_base) + sum_j c_{t,j} (q_t . khat_j) v_j^T
// so a step only has to READ C_base (one pass instead of two) and attend over the short window of tokens that
// have not been folded in yet; the window's (khat_j, v_j) rows are appended, not merged.  Every `period` steps
// (staggered over the envs so the extra pass is spread evenly) `mlstm_lazy_fold_kernel` rewrites
// C_base <- g C_base + (c Khat)^T V on the fp32 matrix cores and empties the window.  HBM bytes per env-step and
// block: DH^2 * 4 (read) + 2 * DH^2 * 4 / period (fold) + the window rows, e.g. 1.31 MB instead of 2.05 MB at
// period 13 on the 16M geometry.  Same mathematics as the materialised update; the fp32 rounding differs
// (fewer roundings: C_base is touched once per period), results stay within the parity bars.
//
// Bookkeeping (per env: pending token count; per (env, head): scale g and the coefficients c_j) is ping-ponged
// between two buffers by step parity: the kernels of one step read the "in" side, the cell kernel's column slice 0
// writes the "out" side.  Whether an env folds / restarts this step is a pure function of (count_in, reset, phase), which
// every kernel evaluates identically -- no flags, no inter-workgroup hand-off.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_math.h"

namespace lram {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

constexpr int W = kLazyWindow;

struct LazyView {
  int n_in;   // tokens pending at the start of the step (what a fold merges)
  int n;      // tokens pending that this step's readout attends over
  bool fold;  // C_base is rewritten before this step
  bool rs;    // env restarts
  bool zero_in;  // C_base held no information at the start of the step (restarted since its last fold)
  bool zero;     // ... for this step's readout: skip the pass over C_base
};

// count word: bits 0..15 pending tokens, bit 16 "C_base is logically zero" (an env restart drops the old memory
// without touching HBM: the next fold simply writes the window instead of accumulating into the stale C_base)
constexpr int kZeroBit = 1 << 16;

// the two words an env's view derives from, requested apart from their use: a kernel can issue them first and look at them
// after its other requests (lazy_view_of), instead of starting with a memory round trip for five bytes
struct LazyRaw {
  unsigned rb;
  int word;
};
__device__ __forceinline__ LazyRaw lazy_raw(const MlstmLazyArgs& a, int b) {
  // (flag read through a pointer select: a load under a branch on a.reset is waited for on the spot)
  const uint8_t* rp = a.reset != nullptr ? a.reset + b : reinterpret_cast<const uint8_t*>(a.count_in);
  LazyRaw r;
  r.rb = *rp;
  r.word = a.count_in[b];
  return r;
}
__device__ __forceinline__ LazyView lazy_view_of(const MlstmLazyArgs& a, int b, const LazyRaw& raw);
__device__ __forceinline__ LazyView lazy_view(const MlstmLazyArgs& a, int b) { return lazy_view_of(a, b, lazy_raw(a, b)); }
__device__ __forceinline__ LazyView lazy_view_of(const MlstmLazyArgs& a, int b, const LazyRaw& raw) {
  LazyView v;
  const unsigned rb = raw.rb;
  const int word = raw.word;
  v.rs = a.reset != nullptr && rb != 0;
  v.n_in = word & 0xFFFF;
  v.zero_in = (word & kZeroBit) != 0;
  v.fold = !v.rs && v.n_in > 0 && (a.force != 0 || ((a.phase + b) % a.period) == 0 || v.n_in + a.T > W);
  v.n = (v.rs || v.fold) ? 0 : v.n_in;
  v.zero = v.rs ? true : (v.fold ? false : v.zero_in);
  return v;
}

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// =============================================================================================
// fold: one workgroup per (env, head, 128 columns); exits at once unless the env folds or restarts this step.
//   C[r][c] <- g C[r][c] + sum_j (coef_j khat_j[r]) v_j[c]     (j < n_in, padded to a multiple of 8 with zeros)
// =============================================================================================
constexpr int kFC = 128;   // columns per workgroup
constexpr int kFPV = 136;  // LDS pitch of the V rows
constexpr int kFPK = 40;   // LDS pitch of the scaled khat tile (32 rows of DH)
constexpr int kFR = 64;    // rows of C per workgroup (two 32-row MFMA tiles): many short workgroups hide the
                           // load -> MFMA -> store latency of a fold better than few long ones
constexpr int WF = W;      // window rows held in LDS
// (Removed after measurement, profiles/EXPERIMENTS.md: a 40-row LDS window for four workgroups per CU -- the fold got 25 %
// shorter, the step did not; a fold + readout form that handed the read pass q . C_new for the strips it had just rewritten --
// 2.3 GB of 48 fewer per step and 4-5 % SLOWER, because a fold that needs this step's q sits on the slice's critical chain.)
__global__ __launch_bounds__(256) void mlstm_lazy_fold_kernel(MlstmLazyArgs a) {
  __shared__ __attribute__((aligned(16))) float Vs[WF * kFPV];
  __shared__ __attribute__((aligned(16))) float Ks[(kFR / 32) * WF * kFPK];
  const int DH = a.DH, NH = a.NH;
  const int nsl = DH / kFC;
  int wid = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wid = (wid & 7) * (nwg >> 3) + (wid >> 3);
  const int nrs = DH / kFR;
  const int rsplit = wid % nrs;
  wid /= nrs;
  const int slice = wid % nsl, bh = wid / nsl, h = bh % NH;
  // compact grid: only the envs whose fold phase comes up this step (b = first, first + period, ...)
  const int b = a.compact ? a.first + (bh / NH) * a.period : bh / NH;
  if (b >= a.B) return;
  // Compact launches (every workgroup's env folds): the env's count / restart words, its scale and the workgroup's tile of
  // C_base are requested together and the view is looked at afterwards -- the tile's 32 KB no longer wait behind a round trip
  // for five bytes.  (Other launches look first: most of their workgroups leave at once.)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int row0 = rsplit * kFR;
  float* Cg = a.C + (((int64_t)b * NH + h) * DH) * DH + slice * kFC;
  float* cp0 = Cg + (int64_t)(row0 + 4 * lh) * DH + 32 * w + li;
  float cold[kFR / 32][16];
  LazyRaw raw = lazy_raw(a, b);
  const float g_raw = a.g_in[(int64_t)b * NH + h];
  const bool early = a.compact != 0;
  if (early) {
#pragma unroll
    for (int t = 0; t < kFR / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) cold[t][r] = cp0[(int64_t)(32 * t + (r & 3) + 8 * (r >> 2)) * DH];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(raw.rb), "+v"(raw.word));  // opaque: the view (and its wait) stays behind the requests above
  }
  const LazyView lv = lazy_view_of(a, b, raw);
  if (!lv.fold) return;
  const int n = lv.n_in;
  const int kt8 = (n + 7) >> 3;
  const float g = lv.zero_in ? 0.f : g_raw;
  const float* wkb = a.wk + (((int64_t)b * NH + h) * W) * DH + row0;
  const float* wvb = a.wv + (((int64_t)b * NH + h) * W) * DH + slice * kFC;
  const float* cfb = a.coef_in + ((int64_t)b * NH + h) * W;
  // every global load of the workgroup is issued before the first use: one memory round trip, not one per phase
  if (!early) {
#pragma unroll
    for (int t = 0; t < kFR / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        cold[t][r] = lv.zero_in ? 0.f : cp0[(int64_t)(32 * t + (r & 3) + 8 * (r >> 2)) * DH];
  } else if (lv.zero_in) {
#pragma unroll
    for (int t = 0; t < kFR / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) cold[t][r] = 0.f;
  }
  float4 rv[(WF * (kFC / 4) + 255) / 256];
#pragma unroll
  for (int i = 0; i < (WF * (kFC / 4) + 255) / 256; ++i) {
    const int idx = tid + 256 * i;
    const int j = idx / (kFC / 4), c4 = (idx % (kFC / 4)) << 2;
    rv[i] = (idx < WF * (kFC / 4) && j < n) ? *reinterpret_cast<const float4*>(wvb + (int64_t)j * DH + c4) : f4_zero();
  }
  float4 rk[(WF * (kFR / 4) + 255) / 256];
  float rc[(WF * (kFR / 4) + 255) / 256];
#pragma unroll
  for (int i = 0; i < (WF * (kFR / 4) + 255) / 256; ++i) {
    const int idx = tid + 256 * i;
    const int j = idx / (kFR / 4), r4 = (idx % (kFR / 4)) << 2;
    const bool ok = idx < WF * (kFR / 4) && j < n;
    rk[i] = ok ? *reinterpret_cast<const float4*>(wkb + (int64_t)j * DH + r4) : f4_zero();
    rc[i] = ok ? cfb[j] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < (WF * (kFC / 4) + 255) / 256; ++i) {
    const int idx = tid + 256 * i;
    if (idx < WF * (kFC / 4)) {
      const int j = idx / (kFC / 4), c4 = (idx % (kFC / 4)) << 2;
      *reinterpret_cast<float4*>(Vs + j * kFPV + c4) = rv[i];
    }
  }
#pragma unroll
  for (int i = 0; i < (WF * (kFR / 4) + 255) / 256; ++i) {
    const int idx = tid + 256 * i;
    if (idx < WF * (kFR / 4)) {
      const int j = idx / (kFR / 4), r4 = (idx % (kFR / 4)) << 2;  // row r4 .. r4+3 of this workgroup's kFR rows
      const float cj = rc[i];
      // scaled khat, tile (r4 / 32): Ks[tile][j][r4 % 32]
      *reinterpret_cast<float4*>(Ks + (r4 >> 5) * WF * kFPK + j * kFPK + (r4 & 31)) =
          make_float4(cj * rk[i].x, cj * rk[i].y, cj * rk[i].z, cj * rk[i].w);
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < kFR / 32; ++t) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = g * cold[t][r];
    const float* Kt = Ks + t * WF * kFPK;
    for (int j8 = 0; j8 < kt8; ++j8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 8 * j8 + 4 * lh + i;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[j * kFPK + li], Vs[j * kFPV + 32 * w + li], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) cp0[(int64_t)(32 * t + (r & 3) + 8 * (r >> 2)) * DH] = acc[r];
  }
}

constexpr int WT = kLazyWT;  // window + this step's tokens (row pitch of the score rows)

// =============================================================================================
// score (geometries with several column slices per head, where the cell kernel's own score computation would be
// repeated by every slice): one workgroup per (env, head), on the slice's stream beside the front end:
//   * bookkeeping after this step's T tokens (coefficients, scale, count) into the "out" side;
//   * the window attention weights of this step, p[t][j] = c_{t,j} (q_t . khat_j), j over the pending window and
//     this step's own tokens, written to `pw` [B, NH, T, kLazyWT] for the cell kernel.
// =============================================================================================

template <int T>
__global__ __launch_bounds__(256) void mlstm_lazy_score_kernel(MlstmLazyArgs a) {
  extern __shared__ float qk[];  // qs [T][DH], ks [T][DH]
  __shared__ float s_coef[W];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NH = a.NH, DH = a.DH, inner = NH * DH;
  float* qs = qk;
  float* ks = qk + T * DH;
  // Requests first, looks later (as in the read pass): the env's count / restart words, the gate scalars, the window
  // coefficient and up to three channels' q / k operands per thread are all issued before anything is waited for or
  // stored; the staging of head dims <= 768 has no loop (a loop header drains every outstanding request).
  LazyRaw raw = lazy_raw(a, b);
  const int64_t base = ((int64_t)b * NH + h) * W;
  const float coef_raw = tid < W ? a.coef_in[base + tid] : 0.f;
  float f[T], ig[T], F[T];
  float Fc = 1.f;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float4 s = *reinterpret_cast<const float4*>(a.scal + (((int64_t)b * T + t) * NH + h) * 4);
    f[t] = s.x;
    ig[t] = s.y;
    Fc *= s.x;
    F[t] = Fc;
  }
  const float sqrt_dh = sqrtf((float)DH);
  {  // q / k of the step's tokens -> LDS: one float4 of four channels per thread (head dims up to 1024 without a loop)
    const int r4 = min(4 * tid, DH - 4);
    float4 qv[T], kv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int64_t off = ((int64_t)b * T + t) * inner + (int64_t)h * DH + r4;
      qv[t] = *reinterpret_cast<const float4*>(a.q + off);
      kv[t] = *reinterpret_cast<const float4*>(a.k + off);
    }
    if (4 * tid < DH) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        *reinterpret_cast<float4*>(qs + t * DH + r4) = qv[t];
        *reinterpret_cast<float4*>(ks + t * DH + r4) = make_float4(kv[t].x / sqrt_dh, kv[t].y / sqrt_dh, kv[t].z / sqrt_dh, kv[t].w / sqrt_dh);
      }
    }
    for (int r = 4 * (tid + 256); r < DH; r += 1024) {   // (head dims beyond 1024)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int64_t off = ((int64_t)b * T + t) * inner + (int64_t)h * DH + r;
        const float4 q4 = *reinterpret_cast<const float4*>(a.q + off), k4 = *reinterpret_cast<const float4*>(a.k + off);
        *reinterpret_cast<float4*>(qs + t * DH + r) = q4;
        *reinterpret_cast<float4*>(ks + t * DH + r) = make_float4(k4.x / sqrt_dh, k4.y / sqrt_dh, k4.z / sqrt_dh, k4.w / sqrt_dh);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(raw.rb), "+v"(raw.word));  // opaque: the view (and its wait) stays behind the requests above
  const LazyView lv = lazy_view_of(a, b, raw);
  const int n = lv.n;
  if (tid < W) s_coef[tid] = tid < n ? coef_raw : 0.f;
  __syncthreads();
  // ---- bookkeeping for the next step ----
  if (tid < n) a.coef_out[base + tid] = s_coef[tid] * F[T - 1];
  if (tid < T) {
    float c = 1.f;
#pragma unroll
    for (int x = 0; x < T; ++x) {
      if (x == tid) c *= ig[x];
      if (x > tid) c *= f[x];
    }
    a.coef_out[base + n + tid] = c;
  }
  if (tid == 0) {
    const float g = (lv.rs || lv.fold) ? 1.f : a.g_in[(int64_t)b * NH + h];
    a.g_out[(int64_t)b * NH + h] = g * F[T - 1];
    if (h == 0) a.count_out[b] = (n + T) | (lv.zero ? kZeroBit : 0);
  }
  // ---- p[t][j]: each wave takes four window rows at a time (their loads are issued together) ----
  const float* wkb = a.wk + base * DH;
  float* pwo = a.pw + (((int64_t)b * NH + h) * T) * WT;
  for (int j0 = 4 * wave; j0 < n + T; j0 += 16) {
    float p[4][T];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < T; ++t) p[i][t] = 0.f;
    // all of the four rows' window values of this lane -- one float4 of four channels per 256 channels (head dims <= 768: three
    // per row; round 5: they were twelve four-byte requests per row, and a wave's request costs the memory pipeline the same
    // whether its lanes ask for 4 or 16 bytes) -- are requested before the first fma
    constexpr int kSIt = 3;
    float4 kv[4][kSIt];
#pragma unroll
    for (int it = 0; it < kSIt; ++it) {
      const int r = 4 * (lane + 64 * it);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = j0 + i;
        kv[i][it] = (r < DH && j < n) ? *reinterpret_cast<const float4*>(wkb + (int64_t)j * DH + r) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int it = 0; it < kSIt; ++it) {
      const int r = 4 * (lane + 64 * it);
      if (r < DH) {
        float4 q4[T];
#pragma unroll
        for (int t = 0; t < T; ++t) q4[t] = *reinterpret_cast<const float4*>(qs + t * DH + r);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = j0 + i;
          const float4 kx = j < n ? kv[i][it] : (j < n + T ? *reinterpret_cast<const float4*>(ks + (j - n) * DH + r) : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
          for (int t = 0; t < T; ++t) p[i][t] += q4[t].x * kx.x + q4[t].y * kx.y + q4[t].z * kx.z + q4[t].w * kx.w;
        }
      }
    }
    for (int r = 4 * (lane + 64 * kSIt); r < DH; r += 256) {  // (head dims beyond 768)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = j0 + i;
        const float4 kx = j < n ? *reinterpret_cast<const float4*>(wkb + (int64_t)j * DH + r)
                                : (j < n + T ? *reinterpret_cast<const float4*>(ks + (j - n) * DH + r) : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float4 q4 = *reinterpret_cast<const float4*>(qs + t * DH + r);
          p[i][t] += q4.x * kx.x + q4.y * kx.y + q4.z * kx.z + q4.w * kx.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = j0 + i;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float s = wave_sum(p[i][t]);
        if (lane == 0 && j < n + T) {
          float c;
          if (j < n) {
            c = s_coef[j] * F[t];
          } else {
            const int u = j - n;  // this step's token u reaches t >= u with i_u f_{u+1} .. f_t
            c = 0.f;
            if (u <= t) {
              c = 1.f;
#pragma unroll
              for (int x = 0; x < T; ++x) {
                if (x == u) c *= ig[x];
                if (x > u && x <= t) c *= f[x];
              }
            }
          }
          pwo[t * WT + j] = c * s;
        }
      }
    }
  }
}

// =============================================================================================
// cell: read-only pass over C_base + the window terms + the step's bookkeeping.  One workgroup per (env, head, column
// slice); thread c of the slice owns column c of the window's V rows, fetched into registers before the pass over
// C_base starts.
//   y_t = q_t^T C_base ;   p[t][j] = c_{t,j} (q_t . khat_j) ;   h_t = ( G_t y_t + sum_j p[t][j] v_j ) / den_t
// The window scores p (formerly a kernel of their own on the slice's stream, with a round trip through HBM) are
// computed here: the window's khat rows are fetched into registers before the pass starts (KPL > 0: DH == 64 KPL,
// each wave keeps every fourth row) or loaded four rows per wave at a time after it (KPL == 0: any head dim), the
// dot products are reduced per wave and land in LDS while the row-group partial sums of the pass are being combined.
// Column slice 0 also appends the step's T tokens to the window and writes the bookkeeping of the next step
// (coefficients, scale, pending count) into the "out" side.
// =============================================================================================
// KPL selects how the window scores are formed:
//   4  (256-wide heads, LPR == 64; "W4" below): the window's khat AND v rows are fetched as one float4 per lane and row -- wave
//      w keeps rows w, w + 4, ... of both, lane l their columns 4 l .. 4 l + 3 -- and the scores are reduced BEFORE the pass over
//      C_base (the khat registers are dead while it runs and hold its rows in flight instead).  The memory pipeline handles a
//      wave's request in the same time whether its lanes ask for 4 or for 16 bytes, and the narrow per-lane form this replaced
//      cost the pass as much request time as the whole stream over C_base (with the stream switched off the pass took 0.32 of
//      its 0.92 ms).  A wave reduces its rows' scores itself (DPP adds, broadcast by readlane), multiplies them onto its v rows
//      right away and hands the sums to the row-group reduction of the pass (red = G y_partial + window partial).
//   0  (128-wide heads): scores after the pass, window rows loaded late, four at a time per wave.
//  -1  (several column slices per head: 384 .. 896): scores and bookkeeping come from mlstm_lazy_score_kernel.
// WP: window rows the register prefetch covers (W, or fewer when the fold period bounds the pending count: with the
// default period of 13 and 3 tokens per step at most 36 rows are pending when a read pass starts; rows beyond it take a
// late-load path).  Variants that were measured and removed (narrow requests, scores after the pass, first rows of C_base
// requested before the scores, other row counts in flight): profiles/EXPERIMENTS.md.
template <int T, int LPR, int UNR, int KPL, int WP = W>
__global__ __launch_bounds__(256) void mlstm_lazy_cell_kernel(MlstmLazyArgs a) {
  static_assert(KPL == 4 || KPL == 0 || KPL == -1, "KPL: 4 (wide window rows, scores first), 0 (late loads), -1 (score kernel)");
  static_assert(KPL != 4 || LPR == 64, "wide window rows: 256-wide heads");
  constexpr bool W4 = KPL == 4, SF = W4;
  constexpr int kRowsPerWave = (WP + T + 3) / 4;
  constexpr int CW = 4 * LPR;
  constexpr int RP = 256 / LPR;
  // WV (scores from the score kernel, i.e. known before the pass): the window's v rows as one float4 per lane and row too -- row
  // group rg keeps rows rg, rg + RP, ... at its own four columns, multiplies the scores on after the pass and adds the sums to its
  // partial of the pass, as W4 does (the per-column form: WP four-byte requests per lane, half the workgroup idle)
  constexpr bool WV = KPL < 0;
  constexpr int kVRows = (WP + RP - 1) / RP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int DH = a.DH;
  float* qs = smem;                 // [T][DH]
  float* ks = qs + T * DH;          // [T][DH] khat of this step's tokens
  float* red = ks + T * DH;         // [RP][T][CW]
  float* pw = red + RP * T * CW;    // [T][WT]
  float* s_coef = pw + T * WT;      // [W]
  float* gnred = s_coef + W;        // [2][4][T] group-norm partial sums (mean, then variance) per wave

  const int b = blockIdx.z, h = blockIdx.y, slice = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = tid % LPR, rg = tid / LPR;
  const int NH = a.NH, inner = NH * DH;
  // the env's count / restart words and its scale are requested here and looked at after the q / k / v operands below
  // are under way (lv_now): the workgroup otherwise starts with two dependent round trips for a few bytes
  LazyRaw raw = lazy_raw(a, b);
  const float g_raw = a.g_in[(int64_t)b * NH + h];

  float den[T], G[T], f[T], ig[T], F[T];
  {
    float Fc = 1.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float4 s = *reinterpret_cast<const float4*>(a.scal + (((int64_t)b * T + t) * NH + h) * 4);
      f[t] = s.x;
      ig[t] = s.y;
      den[t] = s.z;
      Fc *= s.x;
      F[t] = Fc;          // f_1 .. f_t
    }
  }
  const float sqrt_dh = sqrtf((float)DH);
  const bool lean = KPL >= 0 && a.lean_wq != nullptr;
  // (the window coefficient is requested with everything above and kept or dropped once the pending count is known)
  const float coef_raw = tid < W ? a.coef_in[((int64_t)b * NH + h) * W + tid] : 0.f;
  // (head dims up to 256: one channel per thread, written without a loop -- at a loop header the compiler drains every
  // request issued before it, the env's count / restart words included)
  auto stage_qk = [&](int r) {
    const int ch = h * DH + r;  // channel; its 4 x 4 block is ch / 4, its row in the block ch % 4
    // every request of this channel -- the 4 x 4 block rows of wq / wk once, the T tokens' operands -- before the first LDS
    // store (token by token the compiler emitted load -> wait -> store: 3 T dependent round trips at the head of the workgroup)
    if (lean) {
      const float4 wq4 = *reinterpret_cast<const float4*>(a.lean_wq + (int64_t)ch * 4);
      const float4 wk4 = *reinterpret_cast<const float4*>(a.lean_wk + (int64_t)ch * 4);
      float4 xa[T];
#pragma unroll
      for (int t = 0; t < T; ++t)
        xa[t] = *reinterpret_cast<const float4*>(a.lean_xa + ((int64_t)b * T + t) * inner + (ch & ~3));
#pragma unroll
      for (int t = 0; t < T; ++t) {
        qs[t * DH + r] = wq4.x * xa[t].x + wq4.y * xa[t].y + wq4.z * xa[t].z + wq4.w * xa[t].w;
        ks[t * DH + r] = (wk4.x * xa[t].x + wk4.y * xa[t].y + wk4.z * xa[t].z + wk4.w * xa[t].w) / sqrt_dh;
      }
    } else {
      float qv[T], kv[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int64_t off = ((int64_t)b * T + t) * inner + ch;
        qv[t] = a.q[off];
        kv[t] = a.k[off];
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        qs[t * DH + r] = qv[t];
        ks[t * DH + r] = kv[t] / sqrt_dh;
      }
    }
  };
  if (tid < DH) stage_qk(tid);
  for (int r = tid + 256; r < DH; r += 256) stage_qk(r);
  // the view is resolved only here, with the q / k operands requested and stored
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(raw.rb), "+v"(raw.word));  // opaque: its consumers (and their wait) stay behind the requests above
  const LazyView lv = lazy_view_of(a, b, raw);
  const int n = lv.n;
  const float g0 = (lv.rs || lv.fold) ? 1.f : g_raw;
#pragma unroll
  for (int t = 0; t < T; ++t) G[t] = g0 * F[t];  // g f_1 .. f_t
  const int64_t base = ((int64_t)b * NH + h) * W;
  if (tid < W) s_coef[tid] = tid < n ? coef_raw : 0.f;
  if (KPL < 0) {  // scores and bookkeeping come from mlstm_lazy_score_kernel
    const float* pwi = a.pw + (((int64_t)b * NH + h) * T) * WT;
    for (int idx = tid; idx < T * WT; idx += 256) pw[idx] = pwi[idx];
  } else {
    for (int idx = tid; idx < T * WT; idx += 256) pw[idx] = 0.f;
  }
  // window V column of this thread (threads >= CW idle here): in flight during the pass over C_base
  const float* wvb = a.wv + (base * DH) + slice * CW + (tid < CW ? tid : 0);
  float vw[W4 || WV ? 1 : WP];
  if (!W4 && !WV) {
#pragma unroll
    for (int j = 0; j < WP; ++j) vw[W4 || WV ? 0 : j] = (tid < CW && j < n) ? wvb[(int64_t)j * DH] : 0.f;
  }
  v4f v4r[WV ? kVRows : 1];
  (void)v4r;
  if (WV) {
    const float* wvs = a.wv + base * DH + slice * CW + 4 * cl;
#pragma unroll
    for (int i = 0; i < kVRows; ++i) {
      const int j = rg + RP * i;
      v4r[WV ? i : 0] = (j < n && j < WP) ? *reinterpret_cast<const v4f*>(wvs + (int64_t)j * DH) : (v4f)(0.f);
    }
  }
  // window khat rows of this wave (rows wave, wave + 4, ...), KPL values per lane and row
  const float* wkb = a.wk + base * DH;
  v4f k4[W4 ? kRowsPerWave : 1], v4w[W4 ? kRowsPerWave : 1];
  (void)k4, (void)v4w;
  if (W4) {
    const float* wvr = a.wv + base * DH;
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
      const int j = wave + 4 * i;
      k4[W4 ? i : 0] = j < n ? *reinterpret_cast<const v4f*>(wkb + (int64_t)j * DH + 4 * lane) : (v4f)(0.f);
      v4w[W4 ? i : 0] = (j < n && j < WP) ? *reinterpret_cast<const v4f*>(wvr + (int64_t)j * DH + 4 * lane) : (v4f)(0.f);
    }
  }
  // (this step's v operands last: their combination waits for everything requested above, the window rows included)
  float vcur[T];
  {
    const int ch = h * DH + slice * CW + (tid < CW ? tid : 0);  // (threads >= CW request channel 0's operands and drop them)
    if (lean) {  // v from the pre-conv branch (x half of u)
      const float4 wv4 = *reinterpret_cast<const float4*>(a.lean_wv + (int64_t)ch * 4);
      float4 xm[T];
#pragma unroll
      for (int t = 0; t < T; ++t) xm[t] = *reinterpret_cast<const float4*>(a.lean_u + ((int64_t)b * T + t) * 2 * inner + (ch & ~3));
#pragma unroll
      for (int t = 0; t < T; ++t)
        vcur[t] = tid < CW ? wv4.x * xm[t].x + wv4.y * xm[t].y + wv4.z * xm[t].z + wv4.w * xm[t].w : 0.f;
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float v = a.v[((int64_t)b * T + t) * inner + ch];
        vcur[t] = tid < CW ? v : 0.f;
      }
    }
  }
  __syncthreads();

  // ---- window scores p[t][j] = c_{t,j} (q_t . khat_j), j over the pending window and this step's own tokens ----
  auto coefficient = [&](int j, int t) -> float {
    if (j < n) return s_coef[j] * F[t];
    const int u = j - n;  // this step's token u reaches t >= u with i_u f_{u+1} .. f_t
    if (u > t) return 0.f;
    float c = 1.f;
#pragma unroll
    for (int x = 0; x < T; ++x) {
      if (x == u) c *= ig[x];
      if (x > u && x <= t) c *= f[x];
    }
    return c;
  };
  v4f accw[W4 || WV ? T : 1];  // W4 / WV: this wave's (row group's) rows of sum_j p[t][j] v_j at its four columns
#pragma unroll
  for (int t = 0; t < (W4 || WV ? T : 1); ++t) accw[t] = (v4f)(0.f);
  auto window_scores = [&]() {
  if (KPL < 0) {
  } else if (W4) {
#pragma unroll
    for (int i = 0; i < kRowsPerWave; ++i) {
      const int j = wave + 4 * i;
      if (j < n + T) {  // (uniform over the wave)
        const v4f kv = j < n ? k4[W4 ? i : 0] : *reinterpret_cast<const v4f*>(ks + (j - n) * DH + 4 * lane);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const v4f q4 = *reinterpret_cast<const v4f*>(qs + t * DH + 4 * lane);
          const float sc = coefficient(j, t) * wave_sum_bcast(q4.x * kv.x + q4.y * kv.y + q4.z * kv.z + q4.w * kv.w);
          if (lane == 0) pw[t * WT + j] = sc;
          if (j < n && j < WP) accw[W4 ? t : 0] += sc * v4w[W4 ? i : 0];
        }
      }
    }
    if (WP < W) {  // rows beyond the register prefetch (only when more than WP tokens are pending)
      for (int j = wave + 4 * kRowsPerWave; j < n + T; j += 4) {
        float p[T];
#pragma unroll
        for (int t = 0; t < T; ++t) p[t] = 0.f;
        for (int r = lane; r < DH; r += 64) {
          const float kv = j < n ? wkb[(int64_t)j * DH + r] : ks[(j - n) * DH + r];
#pragma unroll
          for (int t = 0; t < T; ++t) p[t] += qs[t * DH + r] * kv;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float sm = wave_sum(p[t]);
          if (lane == 0) pw[t * WT + j] = coefficient(j, t) * sm;
        }
      }
    }
  } else {
    for (int j0 = 4 * wave; j0 < n + T; j0 += 16) {  // four rows per wave at a time (their loads are issued together)
      float p[4][T];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) p[i][t] = 0.f;
      for (int r = lane; r < DH; r += 64) {
        float kv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = j0 + i;
          kv[i] = j < n ? wkb[(int64_t)j * DH + r] : (j < n + T ? ks[(j - n) * DH + r] : 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < T; ++t) p[i][t] += qs[t * DH + r] * kv[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = j0 + i;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float sm = wave_sum(p[i][t]);
          if (lane == 0 && j < n + T) pw[t * WT + j] = coefficient(j, t) * sm;
        }
      }
    }
  }
  };
  const int col0 = slice * CW + 4 * cl;
  v4f acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = (v4f)(0.f);
  const float* Cb = a.C + (((int64_t)b * NH + h) * DH) * DH + col0;
  const bool stream = !lv.zero;  // (after a restart C_base holds nothing until the env's next fold)
  if (SF) window_scores();
  // Occupancy cap by REGISTERS for the sliced-head instances (WV): twelve registers held across the pass put the kernel at 174
  // VGPRs, i.e. two workgroups per CU instead of three, which leaves 164 registers and ~100 KB of LDS per SIMD lane / CU free: the
  // other slice's projection workgroups (118 / 155 VGPRs, 48-65 KB) then START beside the read pass instead of waiting for one
  // of its workgroups to retire.  The pass itself gets slower (0.32 -> 0.345 ms at 206M / 256-env slices), the step faster:
  // 33,008 / 33,220 -> 33,553 / 33,377 env-steps/s (+1 %; profiles/r05_ab_206m_chain.txt).  An LDS cap cannot do this here --
  // it takes the LDS those workgroups need -- and more rows in flight instead of idle registers (20 / 24: 186 / 211 VGPRs) lose.
  constexpr int kRegPad = WV ? 12 : 0;
  float rpad[kRegPad > 0 ? kRegPad : 1];
  (void)rpad;
#pragma unroll
  for (int i = 0; i < kRegPad; ++i) asm volatile("v_mov_b32 %0, 0" : "=v"(rpad[i]));
  if (stream) {
    for (int r0 = rg; r0 < DH; r0 += RP * UNR) {
      v4f c[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = r0 + u * RP;
        c[u] = r < DH ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(Cb + (int64_t)r * DH)) : (v4f)(0.f);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int r = r0 + u * RP;
        if (r < DH) {
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] += qs[t * DH + r] * c[u];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kRegPad; ++i) asm volatile("" ::"v"(rpad[i]));   // (the pad registers' live range ends here)
  if (WV) {  // (pw: the score kernel's rows, in LDS since the barrier above; zero v rows beyond the pending window)
#pragma unroll
    for (int i = 0; i < kVRows; ++i) {
      const int j = rg + RP * i;
      if (j < WP) {
#pragma unroll
        for (int t = 0; t < T; ++t) accw[W4 || WV ? t : 0] += pw[t * WT + j] * v4r[WV ? i : 0];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
    *reinterpret_cast<v4f*>(red + ((rg * T + t) * CW) + 4 * cl) = (W4 || WV) ? G[t] * acc[t] + accw[W4 || WV ? t : 0] : acc[t];

  if (!SF) window_scores();
  __syncthreads();
  const bool gn = a.gn_g != nullptr && CW == DH;  // the workgroup holds the head's whole output row
  float hv[T];
#pragma unroll
  for (int t = 0; t < T; ++t) hv[t] = 0.f;
  if (tid < CW) {
    const int c = tid;
    float hn[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float y = 0.f;
#pragma unroll
      for (int g = 0; g < RP; ++g) y += red[(g * T + t) * CW + c];
      hn[t] = (W4 || WV) ? y : G[t] * y;  // (W4 / WV: the row groups' sums hold G y + the window rows' part already)
    }
    if (!W4 && !WV) {
#pragma unroll
      for (int j = 0; j < WP; ++j) {
#pragma unroll
        for (int t = 0; t < T; ++t) hn[t] += pw[t * WT + j] * vw[W4 || WV ? 0 : j];  // vw == 0 beyond the window, pw zero-filled
      }
    }
    if (WP < W) {
      for (int j = WP; j < n; ++j) {
        const float vj = wvb[(int64_t)j * DH];
#pragma unroll
        for (int t = 0; t < T; ++t) hn[t] += pw[t * WT + j] * vj;
      }
    }
#pragma unroll
    for (int u = 0; u < T; ++u)
#pragma unroll
      for (int t = u; t < T; ++t) hn[t] += pw[t * WT + n + u] * vcur[u];
#pragma unroll
    for (int t = 0; t < T; ++t) hv[t] = hn[t] / den[t];
    if (!gn) {
#pragma unroll
      for (int t = 0; t < T; ++t) a.h[((int64_t)b * T + t) * inner + (int64_t)h * DH + slice * CW + c] = hv[t];
    }
    // append this step's v rows; khat rows and the bookkeeping are written by slice 0 below
    float* wvo = a.wv + (base + n) * DH + slice * CW + c;
#pragma unroll
    for (int t = 0; t < T; ++t) wvo[(int64_t)t * DH] = vcur[t];
  }
  if (gn) {  // (uniform over the workgroup) two-pass group norm over the head's DH outputs, + skip * xa
    float mean[T], dv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float sm = wave_sum(hv[t]);
      if (lane == 0) gnred[wave * T + t] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t) {
      mean[t] = (gnred[t] + gnred[T + t] + gnred[2 * T + t] + gnred[3 * T + t]) / (float)DH;
      dv[t] = tid < CW ? hv[t] - mean[t] : 0.f;
      const float sq = wave_sum(dv[t] * dv[t]);
      if (lane == 0) gnred[4 * T + wave * T + t] = sq;
    }
    __syncthreads();
    if (tid < CW) {
      const int ch = h * DH + tid;
      const float gg = a.gn_g[ch], bb = a.gn_b != nullptr ? a.gn_b[ch] : 0.f, sk = a.gn_skip[ch];
      // (all of the epilogue's operands requested before the first store)
      float xs[T];
#pragma unroll
      for (int t = 0; t < T; ++t) xs[t] = a.lean_xa[((int64_t)b * T + t) * inner + ch];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float* q4 = gnred + 4 * T;
        const float var = (q4[t] + q4[T + t] + q4[2 * T + t] + q4[3 * T + t]) / (float)DH;
        const float rstd = 1.f / sqrtf(var + a.gn_eps);
        const int64_t off = ((int64_t)b * T + t) * inner + ch;
        a.h[off] = dv[t] * rstd * gg + bb + sk * xs[t];
      }
    }
  }
  if (slice == 0) {
    float* wko = a.wk + (base + n) * DH;
    for (int idx = tid; idx < T * DH; idx += 256) wko[idx] = ks[idx];
    if (KPL < 0) return;
    // ---- bookkeeping for the next step ("out" side of the ping-pong) ----
    if (tid < n) a.coef_out[base + tid] = s_coef[tid] * F[T - 1];
    if (tid < T) {
      float c = 1.f;
#pragma unroll
      for (int x = 0; x < T; ++x) {
        if (x == tid) c *= ig[x];
        if (x > tid) c *= f[x];
      }
      a.coef_out[base + n + tid] = c;
    }
    if (tid == 0) {
      a.g_out[(int64_t)b * NH + h] = g0 * F[T - 1];
      if (h == 0) a.count_out[b] = (n + T) | (lv.zero ? kZeroBit : 0);
    }
  }
}

__global__ __launch_bounds__(256) void mlstm_lazy_clear_kernel(int32_t* count, float* g, const uint8_t* mask, int B,
                                                               int NH) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= B * NH) return;
  const int b = gid / NH;
  if (mask != nullptr && mask[b] == 0) return;
  g[gid] = 1.f;
  if (gid == b * NH) count[b] = 0;
}

template <int T, int LPR, int UNR, int KPL, int WP = W>
void launch_cell_tluk(const MlstmLazyArgs& a, hipStream_t s) {
  constexpr int CW = 4 * LPR, RP = 256 / LPR;
  dim3 grid(a.DH / CW, a.NH, a.B), block(256);
  size_t shmem = sizeof(float) * (2 * T * a.DH + RP * T * CW + T * kLazyWT + W + 8 * T);
  shmem = std::max(shmem, (size_t)a.min_lds_bytes);
  // 116 VGPRs would let four workgroups share a CU; three (41 KB of LDS each) leave 152 registers per SIMD lane free,
  // so the slice streams' front-end and 64-row GEMM workgroups start beside them at once and a 128-row GEMM
  // workgroup (224 registers) after ONE read-pass workgroup retires: 386k vs 381k env-steps/s (two: 56 KB, 381k)
  if (KPL == 4 && a.min_lds_bytes == 0) shmem = std::max(shmem, (size_t)41 * 1024);
  if (shmem > 48 * 1024) {
    static uint64_t raised = 0;
    if (first_use_on_device(raised)) {
      LRAM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlstm_lazy_cell_kernel<T, LPR, UNR, KPL, WP>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
  }
  hipLaunchKernelGGL((mlstm_lazy_cell_kernel<T, LPR, UNR, KPL, WP>), grid, block, shmem, s, a);
}

template <int T>
void launch_cell_t(const MlstmLazyArgs& a, hipStream_t s) {
  // 256-wide heads (the 16M geometry): wide window rows, scores first, 8 rows of C_base in flight per lane, 36-row window
  // prefetch (pending rows when a read pass starts: (period - 1) * T = 36 at most with the default period -- the env's fold
  // empties the window first; longer windows take the kernel's late-load path for the rows beyond it).  Measured at 4096 env
  // slots: 4 rows in flight 381k env-steps/s, 8: 386k, 16: 363k.
  if (a.DH == 256) return launch_cell_tluk<T, 64, 8, 4, 36>(a, s);
  // one column slice per head: fused scores (mlstm_lazy_fused_scores: the engine allocates no score buffer and runs the
  // lean front end for these head dims, so every branch for them must be a fused-score instance)
  if (a.DH == 128) return launch_cell_tluk<T, 32, 16, 0>(a, s);
  // several column slices per head: scores from mlstm_lazy_score_kernel (launch_mlstm_lazy_book).  36-row prefetch, 16 rows in
  // flight (174 VGPRs; 206M at 512 env slots 27.7k env-steps/s; 48-row prefetch 27.5k; 8 rows in flight -- 113 VGPRs, the pass
  // itself 0.41 -> 0.31 ms -- 27.1k: the projections of that model are the longer side of the pipeline and lose what the pass gains)
  LRAM_REQUIRE(a.pw != nullptr, "lazy mLSTM: missing score buffer");
  // (round 5, 206M at 512 slots: the window's v rows as float4 per lane and row -- WV in the kernel, 177 -> 162 VGPRs, the pass
  // 0.36 -> 0.32 ms -- 32.1k -> 32.4k env-steps/s; 8 / 12 rows in flight: 29.2-31.7k / 30.8k, with an LDS cap of three workgroups per
  // CU 32.3-32.6k / 31.5k: profiles/r05_ab_206m_chain.txt)
  if (a.DH % 256 == 0) return launch_cell_tluk<T, 64, 16, -1, 36>(a, s);
  launch_cell_tluk<T, 32, 16, -1, 36>(a, s);
}

}  // namespace

bool mlstm_lazy_supported(int DH, int T) { return DH % 128 == 0 && T >= 1 && T <= 4; }

void launch_mlstm_lazy_fold(const MlstmLazyArgs& a_in, hipStream_t stream) {
  MlstmLazyArgs a = a_in;
  LRAM_REQUIRE(a.DH % kFC == 0 && a.DH % kFR == 0, "lazy mLSTM: head dim must be a multiple of 128");
  long envs = a.B;
  if (a.compact) {
    a.first = (a.period - a.phase % a.period) % a.period;  // smallest b with (phase + b) % period == 0
    if (a.first >= a.B) return;
    envs = (a.B - a.first + a.period - 1) / a.period;
  }
  const long nwg = envs * a.NH * (a.DH / kFC) * (a.DH / kFR);
  hipLaunchKernelGGL(mlstm_lazy_fold_kernel, dim3((unsigned)nwg), dim3(256), 0, stream, a);
  LRAM_HIP_CHECK(hipGetLastError());
}

bool mlstm_lazy_fused_scores(int DH) { return DH == 256 || DH == 128; }

void launch_mlstm_lazy_book(const MlstmLazyArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(a.T >= 1 && a.T <= 4, "lazy mLSTM: 1..4 tokens per step");
  LRAM_REQUIRE(a.pw != nullptr, "lazy mLSTM: missing score buffer");
  dim3 grid(a.NH, a.B), block(256);
  const size_t shmem = sizeof(float) * 2 * a.T * a.DH;
  switch (a.T) {
    case 1: hipLaunchKernelGGL(mlstm_lazy_score_kernel<1>, grid, block, shmem, stream, a); break;
    case 2: hipLaunchKernelGGL(mlstm_lazy_score_kernel<2>, grid, block, shmem, stream, a); break;
    case 3: hipLaunchKernelGGL(mlstm_lazy_score_kernel<3>, grid, block, shmem, stream, a); break;
    default: hipLaunchKernelGGL(mlstm_lazy_score_kernel<4>, grid, block, shmem, stream, a); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_mlstm_lazy_cell(const MlstmLazyArgs& a, hipStream_t stream) {
  LRAM_REQUIRE(mlstm_lazy_supported(a.DH, a.T), "lazy mLSTM: unsupported geometry");
  switch (a.T) {
    case 1: launch_cell_t<1>(a, stream); break;
    case 2: launch_cell_t<2>(a, stream); break;
    case 3: launch_cell_t<3>(a, stream); break;
    default: launch_cell_t<4>(a, stream); break;
  }
  LRAM_HIP_CHECK(hipGetLastError());
}

void launch_mlstm_lazy_clear(int32_t* count, float* g, const uint8_t* mask, int B, int NH, hipStream_t stream) {
  hipLaunchKernelGGL(mlstm_lazy_clear_kernel, dim3((unsigned)((B * NH + 255) / 256)), dim3(256), 0, stream, count, g,
                     mask, B, NH);
  LRAM_HIP_CHECK(hipGetLastError());
}

namespace {
__global__ void lazy_counts_as_float_kernel(const int32_t* count, float* out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = (float)(count[b] & 0xFFFF);
}
}  // namespace

// pending window tokens per env (low 16 bits of the count word) as floats: lram_lazy_peek
void launch_lazy_counts_as_float(const int32_t* count, float* out, int B, hipStream_t stream) {
  hipLaunchKernelGGL(lazy_counts_as_float_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, count, out, B);
  LRAM_HIP_CHECK(hipGetLastError());
}

}  // namespace lram
