// Shared declarations of the engine's HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

namespace lram {

// True the first time a call site sees the current HIP device: per-device one-time setup such as raising a kernel's
// dynamic-LDS limit (hipFuncSetAttribute applies to the current device only).  `mask` is the call site's static.
inline bool first_use_on_device(uint64_t& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define LRAM_HIP_CHECK(expr)                                                                         \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      throw ::lram::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + \
                          ":" + std::to_string(__LINE__) + ")");                                     \
  } while (0)

#define LRAM_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) throw ::lram::Error(std::string("lram: ") + msg); \
  } while (0)

constexpr int kMaxTokens = 12;  // tokens per launch of the fused recurrent kernels: 3 per env-step, up to 4
                                // timesteps (12 tokens) per state pass when a stored context is prefilled
constexpr int kChunkMaxTokens = 64;  // tokens per state pass of the chunkwise (matrix-core) mLSTM prefill kernels

// ---------------------------------------------------------------------------------------------
// GEMM  C[M,N] = A[M,K] * W[N,K]^T  (fp32 in, fp32 MFMA accumulate), optional bias / residual.
// Two-level batching: z = z1 * nb2 + z2 ; pointer = base + z1 * s?1 + z2 * s?2.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* a = nullptr;
  const float* w = nullptr;
  float* c = nullptr;
  const float* bias = nullptr;  // [N] (per batch: bias + z1*sBias1 + z2*sBias2)
  const float* residual = nullptr;  // same layout as c (may alias c)
  int64_t lda = 0, ldw = 0, ldc = 0;
  int m = 0, n = 0, k = 0;
  int nb1 = 1, nb2 = 1;
  int64_t sA1 = 0, sA2 = 0, sW1 = 0, sW2 = 0, sC1 = 0, sC2 = 0, sBias1 = 0, sBias2 = 0;
  // optional: W pre-split into three bf16 planes (hi, mid, lo), each laid out like w, `w3_plane` elements apart
  const uint16_t* w3 = nullptr;
  int64_t w3_plane = 0;
  // optional (bf16x3 kernel, fp32 A, no batch, no split-K): A[r][k] is multiplied by gate[r * ldg + k] while it is staged
  // (the mLSTM output gate: gate holds silu(z), written by proj_up's epilogue via act_silu_from)
  const float* gate = nullptr;
  int64_t ldg = 0;
  // optional (bf16x3 kernel, no split-K): output columns >= act_silu_from are stored as silu(value); -1 = none
  int act_silu_from = -1;
  // optional (f16x2 kernel): W pre-split into two f16 planes (hi, lo) of the row-scaled weight, `w2_plane` elements apart,
  // K-TILE-MAJOR: element (n, k) of a plane lives at (k / 32) * w2_kt + n * 32 + k % 32, w2_kt = 32 x the rows of the whole
  // weight (K padded to a multiple of 32) -- the 128 rows x 32 columns a workgroup stages per K tile are ONE contiguous 8 KB run
  // instead of 128 pieces of 64 bytes at the row pitch (half cache lines: 1.4 x slower to deliver, scripts/tile_delivery.cpp);
  // with the exact inverse of each weight row's power-of-two scale; a_amax[r] = largest magnitude of A's row r (of the
  // gated row with `gate`), from which the kernel derives the row's power-of-two scale
  const uint16_t* w2 = nullptr;
  int64_t w2_plane = 0, w2_kt = 0;
  const float* w_inv = nullptr;
  const float* a_amax = nullptr;
  int amax_parts = 1;  // row r's maximum = max of a_amax[r * amax_parts + 0 .. amax_parts - 1]
  // optional (f16x2 kernel with pre-split operands, gemm_f16x2p.hip): A pre-split by its producer into two f16 planes (hi, lo)
  // of the row-scaled value, K-tile-major like w2 (element (r, k) at (k / 32) * a2_kt + r * 32 + k % 32), `a2_plane` elements
  // apart, with a2_inv[r] = the exact inverse of row r's power-of-two scale (`a` may then be null; K a multiple of 32)
  const uint16_t* a2 = nullptr;
  int64_t a2_plane = 0, a2_kt = 0;
  const float* a2_inv = nullptr;
  // optional split-K workspace (un-batched GEMMs with few output tiles: skinny N or small M): partial [S][M][N]
  // slabs are written by S x tiles workgroups and summed, in fixed order, by a second tiny kernel (deterministic)
  float* splitk_ws = nullptr;
  int64_t splitk_ws_elems = 0;
  // optional (few-row kernel, K <= 512): A is the UN-normalised input; every workgroup normalises its 32 rows in registers
  // before the products -- LayerNorm (x - mean) / sqrt(var + eps) * norm_g (+ norm_b), biased variance, two pass; RMSNorm
  // x * rsqrt(mean(x^2) + eps) * norm_g -- as row_norm_kernel does: no norm launch, no [rows, K] round trip
  // optional (few-row kernel; bf16x3 kernel with w3_tab): the outer batch index z2 (< nb2 <= 4) picks its operands from these tables instead of
  // a / w / c + z2 * stride -- several projections with unrelated base pointers in one launch (the four sLSTM gates)
  const float* a_tab[4] = {nullptr, nullptr, nullptr, nullptr};
  const float* w_tab[4] = {nullptr, nullptr, nullptr, nullptr};
  float* c_tab[4] = {nullptr, nullptr, nullptr, nullptr};
  // (bf16x3 kernel with a_tab / c_tab: the split planes of w_tab[z2], `w3_plane` elements apart like w3's)
  const uint16_t* w3_tab[4] = {nullptr, nullptr, nullptr, nullptr};
  const float* norm_g = nullptr;
  const float* norm_b = nullptr;
  float norm_eps = 0.f;
  int norm_rms = 0;
  // set by the launcher (gemm_choose_xcd_split): the 8 XCDs (each with its own 4 MB L2) take an xcd_gm x xcd_gn grid of
  // output blocks, xcd_gm * xcd_gn == 8; 0 = the one-dimensional map (each XCD a contiguous run of tiles, M-major)
  int xcd_gm = 0, xcd_gn = 0;
  // ... or (panel_w > 0) the panel order: the tile grid is cut into column panels of panel_w tiles, walked row by row (odd
  // panels bottom-up), and every XCD takes one contiguous eighth of that walk
  int panel_w = 0;
  // caller's hint (pre-split kernel): this projection runs BESIDE another env slice's memory-bound kernels -- where the launch is
  // 64 .. 160 tiles of 256 x 256 the 8-phase kernel (one workgroup per CU on part of the chip, half the L2 -> CU traffic) is slower
  // alone and faster for the step (Mamba-48M in_proj at 2048 slots: 65 vs 53 us alone, step +2.8 %: profiles/r06_ab_mamba_in_proj_8phase.txt)
  int beside_memory_bound = 0;
  int mfma_prio = 0;      // set by the launcher: raise the wave's issue priority around the MFMA block
  int split_k = 1;        // set by the launcher
  int k_tiles_per_split = 0;
};
// Measurement knobs of the projection launchers (LRAM_GEMM_TILE / LRAM_F16P_STAGES / LRAM_GEMM_PANEL / LRAM_SPLITK_TILES), read
// from the environment ONCE -- no getenv on the step path (an environ scan per launch, and a race with a concurrent setenv from
// another host thread) -- and again only where an engine is created (lram_create, like every other knob) or a standalone test /
// micro-benchmark entry (lram_gemm_*) starts: that is how the bit-identity tests walk through tiles, stages and tile orders.
struct GemmKnobs {
  int tile = 0;          // 64 / 128: force the workgroup tile height of the f16x2 kernels; 256: the 8-phase 256 x 256 kernel; -1: never it
  int stages = 0;        // 1 / 2: force the LDS stage count of the pre-split kernel
  int panel = 0;         // > 0: force the column-panel tile order with that width
  int splitk_tiles = 56; // outputs with fewer 128 x 128 tiles than this are split along K
};
const GemmKnobs& gemm_knobs();
void gemm_knobs_reload();
int gemm_choose_split_k(GemmArgs& g);                             // fills split_k / k_tiles_per_split, returns S
// Output-tile -> XCD map of the 128-column-tile projection kernels (bm = their tile height, bytes_per_elem = operand bytes
// per element as staged: 4 for fp32 / two f16 planes).  Hardware hands consecutive workgroup ids to the 8 XCDs in turn, so
// workgroups b, b + 8, ... share an L2.  Every XCD streams its M band of A once and wants its N band of W resident in
// its 4 MB L2: the split (gm, gn) minimises M * gn + N * gm over the splits whose W band fits and that divide the tile grid.
void gemm_choose_xcd_split(GemmArgs& g, int bm, int bn, int bytes_per_elem);
// the tile a workgroup id maps to under that split (device side)
__device__ __forceinline__ void gemm_tile_of(const GemmArgs& g, int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  if (g.panel_w > 0) {
    const int x = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int pos = x * q + min(x, r) + (bid >> 3);           // XCD x owns walk positions [x q + min(x, r), ...)
    const int per_panel = tiles_m * g.panel_w;
    const int p = pos / per_panel, rem = pos - p * per_panel;
    const int w = min(g.panel_w, tiles_n - p * g.panel_w);
    const int row = rem / w;
    tm = (p & 1) ? tiles_m - 1 - row : row;
    tn = p * g.panel_w + rem - row * w;
    return;
  }
  if (g.xcd_gm > 0) {
    const int x = bid & 7, idx = bid >> 3;                    // XCD, position in that XCD's stream of workgroups
    const int bm_t = tiles_m / g.xcd_gm, bn_t = tiles_n / g.xcd_gn;   // tiles per band
    const int xm = x / g.xcd_gn, xn = x - xm * g.xcd_gn;
    tm = xm * bm_t + idx / bn_t;                              // M outer, N inner: the band's N tiles of one M tile run together
    tn = xn * bn_t + idx % bn_t;
    return;
  }
  if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
  tm = bid / tiles_n;
  tn = bid - tm * tiles_n;
}
void launch_splitk_reduce(const GemmArgs& g, hipStream_t stream);  // C = sum_s ws[s] (+ bias) (+ residual)
void launch_gemm_f32(const GemmArgs& g, hipStream_t stream);      // exact fp32 MFMA (k-ordered fma chain)
bool gemm_small_m(const GemmArgs& g);                             // M <= 8: launch_gemm_f32 takes the GEMV path
bool gemm_skinny_supported(const GemmArgs& g);                    // few-row kernel: 32 x 32 fp32 MFMA tile per workgroup, K over the waves
bool gemm_skinny_norm_supported(const GemmArgs& g);               // ... with the row norm of A in its prologue (K <= 512)
void launch_gemm_skinny(const GemmArgs& g, hipStream_t stream);
bool gemm_bf16x3_supported(const GemmArgs& g);
void launch_gemm_bf16x3(const GemmArgs& g, hipStream_t stream);   // fp32-accurate, 3 x bf16 split operands
void launch_split_bf16x3(const float* w, uint16_t* out, size_t n, hipStream_t stream);
bool gemm_f16x2_supported(const GemmArgs& g);
void launch_gemm_f16x2(const GemmArgs& g, hipStream_t stream);    // fp32-accurate, 2 x f16 split operands, row-scaled
bool gemm_f16x2p_supported(const GemmArgs& g);
void launch_gemm_f16x2p(const GemmArgs& g, hipStream_t stream);   // the same with A pre-split too: DMA staging, MFMA-only loop
// narrow-output projection (N <= 96, K a multiple of 64; gemm_narrow.hip): exact fp32 MFMA, 16 rows per workgroup, W packed at upload
bool gemm_narrow_shape(int n, int k);
bool gemm_narrow_supported(const GemmArgs& g);
size_t gemm_narrow_pack_elems(int n, int k);
void launch_gemm_narrow_pack(const float* w, int n, int k, float* packed, hipStream_t stream);
void launch_gemm_narrow(const GemmArgs& g, const float* packed, hipStream_t stream);
// ... as f16x2 split products (g.w2 / w_inv = the K-tile-major planes of the f16x2 tile GEMMs, g.a_amax / amax_parts = A's row maxima)
bool gemm_narrow16_supported(const GemmArgs& g);
void launch_gemm_narrow16(const GemmArgs& g, hipStream_t stream);
bool gemm_f16x2_8p_supported(const GemmArgs& g);
void launch_gemm_f16x2_8p(const GemmArgs& g, hipStream_t stream); // ... as 256 x 256 tiles, 8 staggered waves, counted DMA waits (gemm_f16x2_8p.hip)
// planes[0 / 1] (K-tile-major: (r, k) at (k / 32) * kt + r * 32 + k % 32; `plane` elements apart) = hi / lo of
// scale_r * a[r][k] (* gate[r][k]), inv[r] = 1 / scale_r
void launch_row_split_f16x2(const float* a, int64_t lda, const float* gate, int64_t ldg, int rows, int k, uint16_t* planes,
                            int64_t kt, int64_t plane, float* inv, hipStream_t stream);
// amax[r] = max_k |a[r][k] (* gate[r][k])|
void launch_row_amax(const float* a, int64_t lda, const float* gate, int64_t ldg, int rows, int k, float* amax,
                     hipStream_t stream);
// planes [2][K tiles][rows][32] f16 (hi, lo of the row-scaled weight, K-tile-major, split_f16x2_plane_elems(rows, k) each),
// inv[rows] = 1 / scale
inline size_t split_f16x2_plane_elems(size_t rows, size_t k) { return rows * ((k + 31) / 32 * 32); }
void launch_split_f16x2(const float* w, int rows, int k, uint16_t* planes, float* inv, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// normalisation / elementwise
// ---------------------------------------------------------------------------------------------
// out[r, :] = norm(in[r, :]) * gamma (+ beta);  rms != 0 -> RMSNorm (no mean subtraction).
// (out2: optional second copy of the result, same row stride as out)
// (amax: optional [rows] largest output magnitude per row, the f16x2 GEMM's a_amax)
// scalar tokens built inside the norm launch (token front end of one timestep): rows b * T + 1 / + 2 = Linear(1, D) of rtg[b] / rew[b]
struct ScalarTokens {
  const float *rtg = nullptr, *rew = nullptr, *w_rtg = nullptr, *b_rtg = nullptr, *w_rew = nullptr, *b_rew = nullptr;
  int64_t in_stride = 1;
  int T = 3;
};
// (h2: optional f16x2 operand planes (hi, lo) of the row-scaled result, K-tile-major with h2_kt elements per K tile, h2_plane
// elements apart, + h2_inv[rows] inverse row scales -- the pre-split projection kernel's A operand; out may then be null)
void launch_row_norm(const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* gamma,
                     const float* beta, int rows, int d, float eps, int rms, hipStream_t stream, float* out2 = nullptr,
                     float* amax = nullptr, const ScalarTokens* st = nullptr, uint16_t* h2 = nullptr, int64_t h2_plane = 0,
                     float* h2_inv = nullptr, int64_t h2_kt = 0);
// Mamba block entry: res_out = hidden (+ res_in);  normed = RMSNorm(res_out) * gamma.
void launch_add_rms_norm(const float* hidden, const float* res_in, float* res_out, float* normed,
                         const float* gamma, int rows, int d, float eps, hipStream_t stream,
                         float* amax = nullptr,     // [rows] largest |normed| per row (f16x2 GEMM's a_amax)
                         uint16_t* h2 = nullptr, int64_t h2_plane = 0, float* h2_inv = nullptr, int64_t h2_kt = 0);

// ---------------------------------------------------------------------------------------------
// front end / head
// ---------------------------------------------------------------------------------------------
// x[b,1,:] = rtg[b]*w_rtg + b_rtg ; x[b,2,:] = rew[b]*w_rew + b_rew   (x: [B,T,D], tokens 1 and 2)
// rtg / rew of env b are read at index b * in_stride (in_stride = L when they come from a [B, L] sequence)
void launch_embed_scalars(float* x, const float* rtg, const float* rew, int64_t in_stride, const float* w_rtg,
                          const float* b_rtg, const float* w_rew, const float* b_rew, int B, int T, int D,
                          hipStream_t stream);
// copy caller-provided state embeddings [B,D] into token slot 0 of x [B,T,D]
void launch_scatter_token0(float* x, const float* emb, int64_t emb_stride, int B, int T, int D, hipStream_t stream);
void launch_embed_chunk(float* x, const float* emb, int64_t emb_stride, const float* rtg, const float* rew, int64_t in_stride,
                        const float* w_rtg, const float* b_rtg, const float* w_rew, const float* b_rew, int B, int steps, int T,
                        int D, hipStream_t stream);
// argmax over logits [B, act_dim*n_vocab] (+ de-tokenise)
void launch_action_argmax(const float* logits, float* actions, int32_t* tokens, int B, int act_dim, int n_vocab,
                          int n_discrete, int action_channels, float tok_min, float tok_max, int discrete,
                          int col_begin, hipStream_t stream, int col_end = -1);

// ---------------------------------------------------------------------------------------------
// xLSTM
// ---------------------------------------------------------------------------------------------
struct MlstmPreArgs {
  const float* u;        // [B*T, 2*inner]  proj_up output: x_mlstm | z
  float* conv_state;     // [B, K, inner]   in/out
  float* n_state;        // [B, NH, DH]     in/out
  float* m_state;        // [B, NH]         in/out
  const float* conv_w;   // [inner, K]      (nn.Conv1d weight [inner,1,K])
  const float* conv_b;   // [inner]
  const float* wq;       // [inner/4, 4, 4]
  const float* wk;
  const float* wv;
  const float* wi;       // [NH, 3*inner]
  const float* bi;       // [NH]
  const float* wf;
  const float* bf;
  float* q;              // [B*T, inner] out
  float* k;
  float* v;
  float* xa;             // [B*T, inner] out  silu(conv)
  float* scal;           // [B*T, NH, 4] out  (f_t, i_t, denom_t, m_t)
  const uint8_t* reset;  // [B] or null
  int B, T, inner, NH, K;
  int lean = 0;          // 1: q, k, v are not written (the lazy read pass rebuilds them from xa / u); T <= 4 kernel only
  // chunkwise prefill (T > kMaxTokens) only, see mlstm_chunk.hip
  float* gates = nullptr;  // [B*T, NH, 2] out  raw (i~, f~) gate pre-activations
  float* amat = nullptr;   // [B, NH, 64, 64] out  intra-chunk weights A[t][s]
  float* vec = nullptr;    // [B, NH, 3, 64] out   fcum_t | w_s | denom_t
};
void launch_mlstm_pre(const MlstmPreArgs& a, hipStream_t stream);

// The same front end for large launches of the lazy ("lean") path, several env slots per workgroup with the block's
// weights held in registers (mlstm_front.hip): inner = 1024, 4 heads, 3 tokens.  q / k / v are not written.
struct MlstmFrontArgs {
  const float* u = nullptr;   // [B*T, ldu]  x_m half of proj_up's output
  int64_t ldu = 0;
  float* conv_state = nullptr;  // [B, 4, inner] in/out
  float* n_state = nullptr;     // [B, NH, DH]   in/out
  float* m_state = nullptr;     // [B, NH]       in/out
  const float *conv_w = nullptr, *conv_b = nullptr, *wq = nullptr, *wk = nullptr;
  const float* gc = nullptr;    // [inner/4, 4*NH, 4] folded gate coefficients (launch_gate_coef)
  const float *bi = nullptr, *bf = nullptr;
  float* xa = nullptr;          // [B*T, inner] out  silu(conv)
  float* scal = nullptr;        // [B*T, NH, 4] out  (f_t, i_t, denom_t, m_t)
  const uint8_t* reset = nullptr;
  int B = 0, T = 0, inner = 0, NH = 0, K = 0;
  int epw = 0;                  // env slots per workgroup (0 = default)
};
bool mlstm_front_supported(int inner, int NH, int K, int T);
void launch_mlstm_front(const MlstmFrontArgs& a, hipStream_t stream);
// gc[cg][j][c] (j < 2 NH: coefficient of xa[4 cg + c], else of x_m[4 cg + c]) for the i / f gates of every head
void launch_gate_coef(const float* wq, const float* wk, const float* wv, const float* wi, const float* wf, int inner, int NH,
                      float* gc, hipStream_t stream);
// chunkwise path: front end + gate scan / A matrix (two launches), then the cell contraction
bool mlstm_chunk_supported(int inner, int NH, int K);
void launch_mlstm_chunk_pre(const MlstmPreArgs& a, hipStream_t stream);

struct MlstmCellArgs {
  float* C;            // [B, NH, DH, DH] in/out
  const float* q;      // [B*T, inner]
  const float* k;
  const float* v;
  const float* scal;   // [B*T, NH, 4]
  float* h;            // [B*T, inner] out: (q^T C_t) / denom_t
  const uint8_t* reset;
  int B, T, NH, DH;
  int unroll = 8;         // rows in flight per thread (8 or 16)
  int min_lds_bytes = 0;  // > 0: request at least this much LDS per workgroup (occupancy cap, see launcher)
  const float* amat = nullptr;  // chunkwise prefill only (written by launch_mlstm_chunk_pre)
  const float* vec = nullptr;
  int chunk_exact_fp32 = 0;     // chunkwise cell on the fp32-input matrix cores (LRAM_CHUNK_CELL=0) instead of the bf16x3 form
};
void launch_mlstm_cell(const MlstmCellArgs& a, hipStream_t stream);
void launch_mlstm_chunk_cell(const MlstmCellArgs& a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Lazy matrix memory (mlstm_lazy.hip): C_t = g * C_base + sum_j coef_j khat_j v_j^T.  A step reads C_base once
// and appends its tokens to a window; C_base is rewritten (folded) once every `period` steps per env.
// ---------------------------------------------------------------------------------------------
constexpr int kLazyWindow = 48;  // window capacity in tokens
constexpr int kLazyWT = kLazyWindow + 4;  // row pitch of the per-step window scores (LDS): window + this step's tokens
struct MlstmLazyArgs {
  float* C;               // [B, NH, DH, DH] C_base (fold: in/out; cell: in)
  float* wk;              // [B, NH, W, DH] window khat_j = k_j / sqrt(DH)
  float* wv;              // [B, NH, W, DH] window v_j
  const float* coef_in;   // [B, NH, W]   coefficients at the start of this step
  float* coef_out;        // book: coefficients after this step
  const float* g_in;      // [B, NH]      scale of C_base at the start of this step
  float* g_out;
  const int32_t* count_in;  // [B] tokens pending at the start of this step (bit 16: C_base logically zero)
  int32_t* count_out;
  const float* q;         // [B*T, inner]
  const float* k;
  const float* v;
  const float* scal;      // [B*T, NH, 4] (f_t, i_t, denom_t, m_t) from mlstm_pre_kernel
  float* h;               // [B*T, inner] out
  // lean front end (fused-score geometries): q, k, v rebuilt here from xa (conv branch) and u's x half with the
  // block-diagonal 4 x 4 weights, instead of being read back from HBM
  const float *lean_xa = nullptr, *lean_u = nullptr, *lean_wq = nullptr, *lean_wk = nullptr, *lean_wv = nullptr;
  // output group norm + learnable skip in the read pass's epilogue (one column slice per head only): h is then stored as
  // GN(h) * gn_g (+ gn_b) + gn_skip * xa, the output gate silu(z) is applied by proj_down while it stages its operand
  const float *gn_g = nullptr, *gn_b = nullptr, *gn_skip = nullptr;
  float gn_eps = 0.f;
  float* pw = nullptr;    // [B, NH, T, kLazyWT] window scores: only for geometries with several column slices per head
  const uint8_t* reset;   // [B] or null
  int B, T, NH, DH;
  int phase, period;      // env b folds when (phase + b) % period == 0 (staggered), or when the window would overflow
  int force;              // fold kernel: fold every env that has pending tokens (materialise)
  int compact = 0;        // fold kernel: launch only over the envs whose phase comes up (no window can overflow)
  int first = 0;          // set by the launcher
  int min_lds_bytes = 0;
};
void launch_mlstm_lazy_fold(const MlstmLazyArgs& a, hipStream_t stream);
void launch_mlstm_lazy_cell(const MlstmLazyArgs& a, hipStream_t stream);
// one column slice per head (DH 128 / 256): the cell kernel computes the window scores and the bookkeeping itself;
// otherwise launch_mlstm_lazy_book must run before it (fills a.pw, coef_out, g_out, count_out)
bool mlstm_lazy_fused_scores(int DH);
void launch_mlstm_lazy_book(const MlstmLazyArgs& a, hipStream_t stream);
bool mlstm_lazy_supported(int DH, int T);
// count[b] = 0, g[b, :] = 1 for masked envs (mask == nullptr: all), both parities handled by the caller
void launch_mlstm_lazy_clear(int32_t* count, float* g, const uint8_t* mask, int B, int NH, hipStream_t stream);
void launch_lazy_counts_as_float(const int32_t* count, float* out, int B, hipStream_t stream);


// mode 0 (mLSTM): out[r, hd] = (GN(h)[r,hd] * gamma + skip*xa) * silu(z)      z = u[r, inner + hd]
// mode 1 (sLSTM): x[r, hd] += GN(h)[r,hd] * gamma
struct GroupNormArgs {
  const float* h;      // [rows, NH*DH]
  const float* gamma;  // [NH*DH]  (already 1 + w)
  const float* beta;   // optional
  const float* skip;   // mode 0
  const float* xa;     // mode 0 [rows, NH*DH]
  const float* u;      // mode 0 [rows, 2*NH*DH]
  float* out;          // mode 0: g [rows, NH*DH]; mode 1: x [rows, NH*DH] (+=)
  int rows, NH, DH, mode;
  float eps;
  float* amax = nullptr;  // mode 0, optional: [rows, NH] max |out| of the (row, head) segment (partial row maxima for the f16x2 GEMM)  // mode 0, optional: the output as the pre-split f16x2 GEMM's A operand instead of fp32 -- two f16 planes, K-tile-major
  // [inner / 32][rows][32] with K-tile pitch h2_kt elements, the lo plane h2_plane elements after the hi plane, of the row scaled by
  // pow2_scale(row maximum over ALL heads); h2_inv[row] = the exact inverse scale.  One workgroup per row (NH waves).
  uint16_t* h2 = nullptr;
  int64_t h2_plane = 0, h2_kt = 0;
  float* h2_inv = nullptr;
};
void launch_group_norm(const GroupNormArgs& a, hipStream_t stream);

struct SlstmConvArgs {
  const float* xn;      // [B*T, D]
  float* conv_state;    // [B, K, D]
  float* slstm_state;   // [4, B, D]  (zeroed for reset envs)
  const float* conv_w;  // [D, K]
  const float* conv_b;
  float* xc;            // [B*T, D] out silu(conv)
  const uint8_t* reset;
  int B, T, D, K;
  int state_B;          // env count of the full state tensor (stride of its leading [4] axis)
};
void launch_slstm_conv(const SlstmConvArgs& a, hipStream_t stream);

struct SlstmPointwiseArgs {
  const float* gates;  // [B*T, 4, H] gate-major (i,f,z,o) input pre-activations (Wx)
  const float* ry;     // [B, 4, H]   recurrent contribution for this token
  const float* bias;   // [4, H]
  float* state;        // [4, B, H] in/out
  float* yout;         // [B*T, H]  y written at row b*T + t
  int B, T, t, H;
  int state_B;         // env count of the full state tensor
};
void launch_slstm_pointwise(const SlstmPointwiseArgs& a, hipStream_t stream);

// recurrent projection + pointwise cell of ONE token in one launch (few env rows; xlstm_kernels.hip)
struct SlstmTokenArgs {
  const float* gates;  // [B*T, 4, H] input pre-activations (Wx)
  const float* rt;     // [NH, 4, SDH, SDH] recurrent weights (out, in)
  const float* bias;   // [4, H]
  const float* hprev;  // h_{t-1}: row b at hprev + b * hprev_ld (the state's h plane, or yout rows of token t - 1)
  int64_t hprev_ld;
  float* state;        // [4, state_B, H] in/out (h plane written only when write_h)
  float* yout;         // [B*T, H]
  int B, T, t, H, NH, state_B, write_h;
};
bool slstm_token_supported(int H, int NH);
void launch_slstm_token(const SlstmTokenArgs& a, hipStream_t stream);

// recurrent projection + pointwise cell of ALL T tokens of an env-step in one launch (head dim 128, any slice size; slstm_seq.hip)
struct SlstmSeqArgs {
  const float* gates = nullptr;  // [B*T, 4, H] input pre-activations (Wx)
  const float* rt2 = nullptr;    // [NH, SDH (k), SDH (channel), 4 (gate)] recurrent weights re-packed by launch_slstm_pack_rt
  const uint16_t* rt2h = nullptr;  // != nullptr: the f16x2 form -- two f16 planes of the row-scaled weights (launch_slstm_pack_rt16)
  const float* rinv = nullptr;     // ... and [NH, 4, SDH] inverse scales
  const float* bias = nullptr;   // [4, H]
  float* state = nullptr;        // [4, state_B, H] in/out (h, c, n, m planes)
  float* yout = nullptr;         // [B*T, H]
  int B = 0, T = 0, H = 0, NH = 0, state_B = 0;
};
void launch_slstm_h_range(const float* h, int64_t n, float limit, int* flag, hipStream_t stream);  // flag |= any element outside (-limit, limit), NaN included
bool slstm_seq_supported(int H, int NH, int T);
void launch_slstm_seq(const SlstmSeqArgs& a, hipStream_t stream);
void launch_slstm_pack_rt(const float* rt, float* rt2, int NH, int SDH, hipStream_t stream);  // rt: [NH, 4, out, in]
void launch_slstm_pack_rt16(const float* rt, uint16_t* rt2h, float* rinv, int NH, int SDH, hipStream_t stream);

// a[r, f] = gelu(p[r, f]) * p[r, F + f]      p: [rows, 2F]
void launch_gelu_gate(const float* p, float* a, int rows, int F, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Mamba
// ---------------------------------------------------------------------------------------------
struct MambaConvArgs {
  const float* xz;      // [B*T, 2*d_inner]  in_proj output: x | z
  float* conv_state;    // [B, d_inner, K]
  const float* conv_w;  // [d_inner, K]
  const float* conv_b;
  float* xc;            // [B*T, d_inner] out silu(conv)
  const uint8_t* reset;
  int B, T, d_inner, K;
  float* amax = nullptr;  // optional [B*T][d_inner / 64]: per-wave partial row maxima of xc (f16x2 GEMM's a_amax)
};
void launch_mamba_conv(const MambaConvArgs& a, hipStream_t stream);

struct MambaSsmArgs {
  float* ssm_state;     // [B, d_inner, N]
  const float* xc;      // [B*T, d_inner]
  const float* dtp;     // [B*T, d_inner]  dt_proj(dt) (no bias)
  const float* dt_bias; // [d_inner]
  const float* xdb;     // [B*T, R + 2N]   x_proj output (dt | B | C)
  const float* A_log;   // [d_inner, N]
  const float* Dp;      // [d_inner]
  const float* xz;      // [B*T, 2*d_inner] (z at + d_inner)
  float* y;             // [B*T, d_inner] out
  const uint8_t* reset;
  int B, T, d_inner, N, R;
  float* amax = nullptr;    // optional [B*T][d_inner / 64]: per-64-channel partial row maxima of y (f16x2 GEMM's a_amax)
  const float* dt_wt = nullptr;  // optional dt_proj.weight TRANSPOSED [R, d_inner]: dt_proj evaluated inside the kernel, dtp unused
};
inline bool mamba_ssm_dt_fusable(int N, int R) { return N == 16 && R >= 1 && R <= 128; }
void launch_transpose_f32(const float* src, int rows, int cols, float* dst, hipStream_t stream);
void launch_mamba_ssm(const MambaSsmArgs& a, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// IMPALA-CNN image front end (impala_cnn.hip)
// ---------------------------------------------------------------------------------------------
struct Conv3x3Args {
  const void* in;         // [B, CIN, H, W] float (or uint8 when in_u8: scaled by 1/255 while staging)
  const float* w;         // [COUT, CIN, 3, 3]
  const float* bias;      // [COUT]
  const float* residual;  // optional [B, COUT, H, W], added to the output
  float* out;             // [B, COUT, H, W]
  int B, CIN, COUT, H, W;
  int in_relu, out_relu, in_u8;
};
void launch_conv3x3(const Conv3x3Args& a, hipStream_t stream);
void launch_maxpool3s2(const float* in, float* out, int64_t planes, int H, int W, hipStream_t stream);
void launch_relu(float* x, int64_t n, hipStream_t stream);

// misc
void launch_pad_obs(const float* native, int n_native, const int32_t* inv_index, const float* mean, const float* stdv,
                    float* out, int B, int state_dim, hipStream_t stream);
void launch_stream_copy(float* dst, const float* src, size_t numel, hipStream_t stream);
void launch_stream_read(const float* buf, size_t numel, float* sink, hipStream_t stream);  // read only; sink: 1024 floats
void launch_stream_rmw(float* buf, size_t numel, hipStream_t stream);  // in place, the cell kernel's access pattern
void launch_zero_rows(float* buf, const uint8_t* mask, int B, int64_t row_elems, int64_t outer, int64_t outer_stride,
                      hipStream_t stream);

}  // namespace lram
