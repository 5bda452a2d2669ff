// Engine: owns weights, per-env recurrent state and activation workspace on one MI355X, and issues the
// kernel sequence of one env-step (layer-major: every block consumes all T tokens of the timestep before
// the next block runs, so each block's recurrent state is read and written once per env-step).
// C ABI in include/lram_hip.h.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lram_hip.h"
#include "common.h"

using namespace lram;

namespace {

thread_local std::string g_last_error;

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;
  void alloc(size_t numel) {
    release();
    if (numel == 0) return;
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), numel * sizeof(float)));
    n = numel;
  }
  void zero(hipStream_t s = nullptr) {
    if (p) LRAM_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(float), s));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

struct BlockWeights {  // resolved device pointers (nullptr when absent / optional)
  // common
  const float *norm_g = nullptr, *norm_b = nullptr;
  // mLSTM
  const float *proj_up = nullptr, *conv_w = nullptr, *conv_b = nullptr, *wq = nullptr, *wk = nullptr, *wv = nullptr,
              *wi = nullptr, *bi = nullptr, *wf = nullptr, *bf = nullptr, *on_g = nullptr, *on_b = nullptr,
              *skip = nullptr, *proj_down = nullptr;
  // sLSTM
  const float *gate_w[4] = {nullptr, nullptr, nullptr, nullptr};  // i, f, z, o slots of the cell
  const float *rt = nullptr, *rbias = nullptr, *gn_g = nullptr, *gn_b = nullptr, *ffn_norm_g = nullptr,
              *ffn_norm_b = nullptr, *ffn_up = nullptr, *ffn_down = nullptr;
  // Mamba
  const float *in_proj = nullptr, *in_proj_b = nullptr, *x_proj = nullptr, *dt_proj = nullptr, *dt_bias = nullptr,
              *A_log = nullptr, *Dp = nullptr, *out_proj = nullptr, *out_proj_b = nullptr;
};

struct BlockState {
  DevBuf s0;    // mLSTM C | sLSTM state [4,B,D] | Mamba ssm
  DevBuf n;     // mLSTM n
  DevBuf m;     // mLSTM m
  DevBuf conv;  // conv state
  // lazy matrix memory (mlstm_lazy.hip): window rows and ping-pong bookkeeping, allocated in lazy mode only
  DevBuf wk, wv;    // [B, NH, W, DH] each
  DevBuf coef;      // [2][B, NH, W]
  DevBuf gsc;       // [2][B, NH]
  DevBuf pw;        // [B, NH, 4, kLazyWT] window scores (head dims with several column slices per head only)
};

struct GraphKey {
  const void *obs, *rtg, *rew, *mask, *act, *tok;
  int emb, discrete, B;
  hipStream_t stream;
  bool operator==(const GraphKey& o) const {
    return obs == o.obs && rtg == o.rtg && rew == o.rew && mask == o.mask && act == o.act && tok == o.tok &&
           emb == o.emb && discrete == o.discrete && B == o.B && stream == o.stream;
  }
};

}  // namespace

constexpr int kTokenTapMaxBatch = 1024;  // larger batches skip the per-step copy of the embed_ln tokens (lram_get_taps)

struct lram_engine {
  lram_config cfg{};
  int device = 0;
  std::map<std::string, DevBuf> weights;
  bool finalized = false;
  std::vector<BlockWeights> bw;
  // bf16x3 GEMM: fp32 weight pointer -> its three bf16 planes (built in finalize)
  struct Split {
    uint16_t* p;
    size_t n;
  };
  std::map<const float*, Split> split;
  bool use_bf16x3 = true;  // LRAM_GEMM=f32 selects the exact fp32-MFMA kernel everywhere
  // f16x2 projection kernel (gemm_f16x2.hip): un-batched weights also get two row-scaled f16 planes + inverse scales;
  // LRAM_GEMM=bf16x3 keeps the three-plane bf16 kernel for them too
  bool use_f16x2 = true;
  int f16x2_min_rows = 256;   // LRAM_F16_MIN_ROWS
  struct Split16 {
    uint16_t* planes;  // [2][rows][k] f16
    float* inv;        // [rows] exact inverse of each weight row's power-of-two scale
    size_t rows, k;
  };
  std::map<const float*, Split16> split16;
  bool gemm_presplit = true;   // LRAM_GEMM_PRESPLIT=0: the norms ahead of proj_up / in_proj write fp32 + row maxima (round 3) instead of
                               // the f16x2 GEMM's operand planes (gemm_f16x2p.hip)
  double gemm_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // launches / fp32-equivalent FLOPs per dispatcher family (lram_gemm_counts)
  std::vector<DevBuf> slstm_rt2;  // sLSTM: recurrent weights re-packed per block for slstm_seq.hip: fp32 [head][k][channel][gate], or
                                  // (f16x2 projections, the default) two f16 planes in the same bytes + slstm_rinv, the inverse row scales
  std::vector<DevBuf> slstm_rinv;
  int lazy_cap2_envs = 896;       // LRAM_LAZY_CAP2_ENVS: largest slice whose read pass runs two workgroups per CU (0 = never)
  std::map<const float*, DevBuf> narrow;   // narrow-output weights (Mamba x_proj) packed for gemm_narrow.hip (built in finalize)
  bool gemm_narrow_on = true;     // LRAM_GEMM_NARROW=0: x_proj through the tile GEMMs (split-K + reduce) as before round 6
  int gemm_narrow_min_rows = 256;
  bool upz_beside = true;         // LRAM_UPZ_8P=0: proj_up's z half (issued beside the slice's own state pass) never through the 8-phase kernel
  int mamba_slices_now = 1;       // env slices of the Mamba step under way (run_mamba_stack)
  bool slstm_gates_one = true;    // LRAM_SLSTM_GATES_ONE=0: the four sLSTM gate projections of larger slices as four bf16x3 launches
  bool gemm_narrow_f16 = true;    // LRAM_GEMM_NARROW=2: its exact-fp32 form even where the projections run as f16x2
  bool gn_amax_handover = true;   // LRAM_GN_AMAX=0: proj_down's operand row maxima from their own launch, not from the group norm
  bool gn_planes = true;          // LRAM_GN_AMAX=1: the group norm writes fp32 + partial row maxima (round 5) instead of proj_down's operand planes
  int xlstm_slices_now = 1;       // env slices of the stack pass under way (set by run_xlstm_stack)
  bool slstm_seq_f32 = false;     // LRAM_SLSTM_SEQ=2: its exact-fp32 form even where the projections run as f16x2
  bool slstm_seq = true;          // LRAM_SLSTM_SEQ=0: per-token recurrent GEMM + pointwise launches for slices beyond the token kernel's
  std::vector<DevBuf> gate_coef;  // mLSTM: folded i / f gate coefficients per block (mlstm_front.hip), geometries it covers
  bool front_multi = true;     // LRAM_FRONT_MULTI=0: keep the one-workgroup-per-env front end for large launches too
  int front_min_envs = 256;    // LRAM_FRONT_MIN_ENVS: slices of at least this many env slots take the multi-env front end
  std::vector<DevBuf> dt_wt;   // Mamba: dt_proj.weight transposed to [dt_rank, d_inner] per block (state-update kernel's operand)
  DevBuf ASCALE;  // per-row maxima of a GEMM's A operand computed by launch_row_amax, one region per stream slot (like
  size_t ascale_rows = 0;  // the split-K slabs)
  // row maxima handed over by the kernels that produce the projections' operands, indexed like the rows of X:
  // XN (norm -> proj_up / ffn_up / in_proj), XA (Mamba conv -> x_proj), H (Mamba selective state update -> out_proj)
  DevBuf AMX_XN, AMX_XA, AMX_H;
  // front end / head
  const float *w_state = nullptr, *b_state = nullptr, *w_rtg = nullptr, *b_rtg = nullptr, *w_rew = nullptr,
              *b_rew = nullptr, *eln_g = nullptr, *eln_b = nullptr, *w_head = nullptr, *b_head = nullptr,
              *post_g = nullptr, *post_b = nullptr;
  // IMPALA-CNN image front end (optional: present when the embed_image.* weights were uploaded)
  struct ImgConv {
    const float *w = nullptr, *b = nullptr;
    int cin = 0, cout = 0;
  };
  ImgConv img_conv[3][5];  // [stage][stage conv, res0.conv_0, res0.conv_1, res1.conv_0, res1.conv_1]
  const float *img_lin_w = nullptr, *img_lin_b = nullptr;
  int img_channels = 0, img_flat = 0;  // input channels, flattened feature count of the linear layer
  DevBuf IMG_P, IMG_X0, IMG_X1, IMG_T;
  DevBuf IMG_EMB;                      // [B, D] state-token embeddings of lram_step_images
  // lram_step_images: the frames of the env-step under way (set around step_launches): every env slice runs the IMPALA-CNN on
  // its own frames on its own stream, and the state-pass stream takes fold_bubbles_images folds ahead of the first read pass --
  // the VALU-bound CNN and the HBM-bound folds share the start of the step
  const uint8_t* step_images = nullptr;
  int step_img_c = 0, step_img_h = 0, step_img_w = 0;
  // (206M, 512 slots, same box: two calls 31.03k env-steps/s; one call with 2 / 5 / 8 / 11 / 14 folds ahead 31.36k / 31.68k / 31.81k /
  // 31.65k / 31.31k; the second slice's CNN held back until the first slice's is done: 31.4k -- not kept)
  static constexpr int fold_bubbles_images = 8;
  size_t img_cap = 0;  // batch * input pixels the image buffers were sized for
  // lazy matrix memory: C_base read once per step, rewritten once per `lazy_period` steps (see mlstm_lazy.hip)
  int lazy_mode = 2;        // 0 materialised, 1 lazy, 2 auto (LRAM_STATE / lram_set_state_mode)
  bool lazy = false;        // effective choice for the current batch (decided in state_alloc / set_state_mode)
  bool lazy_ready = false;  // buffers allocated for the current batch
  int lazy_period = 13;
  int gn_fuse = 2;          // LRAM_GN_FUSE: output group norm + skip in the read pass's epilogue, gate in proj_down's
                            // operand staging.  0 off, 1 on, 2 auto = on from 2048 env slots (round 3, same box, two
                            // rounds: 391.1k / 393.5k off vs 395.8k / 397.5k on at 4096 slots; 1024 slots: -0.4 %)
  bool mamba_dt_fuse = true;  // LRAM_MAMBA_DT_FUSE: dt_proj inside the selective-state-update kernel (d_state 16, dt_rank <= 64)
  int slstm_fused_rows = 512;  // LRAM_SLSTM_FUSED_ROWS: slices of slstm_fused_min .. this many envs (at sLSTM head dim <= 128; fewer above:
                               // x 128 / head dim) take the one-launch sLSTM token kernel (0 = never)
  int gemm_skinny_rows = 384;  // LRAM_GEMM_SKINNY_ROWS: GEMMs with 9 .. this many operand rows (half of it for weights above 600k elements) ...
  static constexpr int slstm_gates_rows = 768;  // sLSTM gate projections (head dim <= 128) of up to this many rows on the few-row kernel as well
  int gemm_skinny_min = 5;     // LRAM_GEMM_SKINNY_MIN: fewest operand rows (below: the GEMV path; 16M at 1 env 0.372 vs 0.410 ms, at 2 envs 0.443 vs 0.418)
  static constexpr int gemm_skinny_k = 1024;  // ... and K up to this take the few-row kernel
  static constexpr int fold_bubbles = 2;  // folds before the first read pass; the rest behind the sLSTM blocks, all on the state-pass
                                          // stream (measured on one box: k = 0 -- own stream, one block ahead -- 364k, 1 367k, 2 368k,
                                          // 3 367k, 4 366k env-steps/s)
  int64_t lazy_step = 0;    // steps taken in lazy mode: fold phase and ping-pong parity
  std::vector<int> lazy_bound;  // host-side upper bound of pending tokens per fold class (b % period)
  bool lazy_compact = false;    // this step's fold launches may use the compact grid (no window can overflow)
  bool lazy_dirty = false;      // a lazy step ran since the last materialise: windows may hold pending tokens
  DevBuf LZ_COUNT;          // [2][B] int32 pending tokens per env
  // state + workspace
  int B = 0;
  std::vector<BlockState> st;
  DevBuf X, XN, TOK, HID, U, Q, K, V, XA, H, G, SCAL, RY, LOGITS, RES, DTP;
  DevBuf XN2;   // the norm output as f16x2 operand planes [2][B*T, D] f16 (pre-split projections): its own buffer -- a slice inside an
                // sLSTM block uses XN as fp32 while another slice's mLSTM block holds planes
  DevBuf GATES, AMAT, VEC;           // chunkwise mLSTM prefill work buffers (allocated with the first long chunk)
  DevBuf SEQ_EMB;                    // state embeddings of a stored context [B, L, D] (lram_prefill)
  int tok_cap = 0;                   // tokens per env the activation workspace holds (kMaxTokens until a prefill grows it)
  bool chunk_prefill = true;         // LRAM_PREFILL_CHUNK=0: keep the token-sequential kernels for prefill
  bool chunk_exact_fp32 = false;     // LRAM_PREFILL_CHUNK=2: chunkwise cell on the fp32-input matrix cores (the round 1-5 form)
  // Chunk lanes of lram_prefill: consecutive chunks of a stored context alternate between two activation workspaces and two
  // streams; block i of chunk c + 1 waits for block i of chunk c only (its recurrent state), so two chunks are in flight one
  // block apart -- the matrix-core-bound projections of one beside the HBM-bound state passes of the other, and the
  // token-sequential sLSTM launches of either hidden behind both (one env slice only; LRAM_PREFILL_CHUNK=3: off).
  bool chunk_lanes = true;
  static constexpr int kMaxLanes = 3;
  static constexpr int n_lanes = 3;  // chunks in flight (206M, 64 envs x 512 timesteps, same box: 1 lane 385 ms, 2 lanes 326, 3 lanes 305, 4 lanes 303)
  DevBuf twin[kMaxLanes - 1][21];    // further copies of the per-token activation workspace (see workspace_set())
  std::vector<hipEvent_t> lane_ev[kMaxLanes];             // "block i of the lane's current chunk is done"
  const std::vector<hipEvent_t>* lane_wait = nullptr;     // set by timesteps_launches around run_stack
  const std::vector<hipEvent_t>* lane_rec = nullptr;
  DevBuf SK;                         // split-K partial slabs: one slot per stream that may run a GEMM
  static constexpr size_t kSplitKSlotElems = 6u << 20;  // 6 Mi floats (24 MiB) >= S*M*N for any GEMM the chooser splits
  static constexpr int kSplitKSlots = 9;                 // caller's stream + up to 8 micro-batch streams
  size_t ucols = 0, icols = 0;  // allocated row pitch of U and of Q/K/V/XA/H/G (slice offsets use these)
  // graph replay
  bool graph_mode = false;
  bool graph_valid = false;
  GraphKey graph_key{};
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t capture_stream = nullptr;  // capture needs a non-default stream; replay runs on the caller's
  // micro-batch pipeline: env slices on their own streams, cell kernels serialised on hbm_stream
  int n_micro = 0;  // 0 = auto
  // reference-trajectory modes of the Mamba agent (lram_set_compat_mode)
  int compat_repeat = 1;      // forwards per env-step: action dim i is read from forward min(i, repeat - 1)
  // Repeated forwards share what does not depend on the recurrent state: the (s, rtg, r) token embeddings, and with them
  // layer 0's add + RMSNorm and in_proj (identical inputs in every pass).  Pass 0 keeps them in X0 / U0; later passes skip
  // the front end and layer 0's first stage.  compat_pass / compat_passes: the pass under way, set by step_launches.
  DevBuf X0, U0;
  int compat_pass = 0, compat_passes = 1;
  bool compat_share = true;   // LRAM_COMPAT_SHARE=0: every repeated forward recomputes the front end and layer 0's in_proj
  bool compat_stale = false;  // a reset re-initialises layer 0 only; layers >= 1 keep the previous episode's state
  static constexpr int cell_unroll = 16;  // C rows in flight per thread of the materialised cell kernel
  std::vector<hipStream_t> micro_streams;
  hipStream_t hbm_stream = nullptr;
  std::vector<hipEvent_t> sync_events, edge_events;   // engine-internal edges (device-scope fence) / fork + join with the caller's stream
  size_t sync_used = 0, edge_used = 0;
  bool event_device_scope = true;   // LRAM_EVENT_SCOPE=system: default (system-scope) events for the internal edges too
  // profiling of the dominant recurrent kernel
  bool prof_on = false;
  int prof_every = 1;       // lram_profile_begin_sampled: every n-th lram_step is timed (its launches carry the event pairs)
  int64_t prof_calls = 0;   // lram_step calls since profiling was armed
  bool prof_live = true;    // the call under way is one of the timed ones
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<uint8_t> prof_aux;  // 1: the pair times a fold launch (adds to the total, is not a state-pass launch)
  size_t prof_used = 0;

  ~lram_engine() {
    drop_graph();
    if (capture_stream) (void)hipStreamDestroy(capture_stream);
    if (hbm_stream) (void)hipStreamDestroy(hbm_stream);
    for (hipStream_t ms : micro_streams) (void)hipStreamDestroy(ms);
    for (hipEvent_t ev : sync_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : edge_events) (void)hipEventDestroy(ev);
    for (auto& v : lane_ev)
      for (hipEvent_t ev : v) (void)hipEventDestroy(ev);
    for (auto& e : prof_events) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    for (auto& kv : weights) kv.second.release();
    drop_splits();
    release_state();
  }
  void drop_splits() {
    for (auto& kv : split) (void)hipFree(kv.second.p);
    split.clear();
    for (auto& kv : split16) (void)hipFree(kv.second.planes), (void)hipFree(kv.second.inv);
    split16.clear();
    for (DevBuf& b : dt_wt) b.release();
    dt_wt.clear();
    for (auto& kv : narrow) kv.second.release();
    narrow.clear();
    for (DevBuf& b : gate_coef) b.release();
    gate_coef.clear();
    for (DevBuf& b : slstm_rt2) b.release();
    slstm_rt2.clear();
    for (DevBuf& b : slstm_rinv) b.release();
    slstm_rinv.clear();
  }
  void drop_graph() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    graph_exec = nullptr;
    graph = nullptr;
    graph_valid = false;
  }
  void release_state() {
    for (auto& s : st) {
      s.s0.release();
      s.n.release();
      s.m.release();
      s.conv.release();
      s.wk.release();
      s.wv.release();
      s.coef.release();
      s.gsc.release();
      s.pw.release();
    }
    LZ_COUNT.release();
    lazy_ready = false;
    st.clear();
    for (DevBuf* b : {&X, &XN, &TOK, &HID, &U, &Q, &K, &V, &XA, &H, &G, &SCAL, &RY, &LOGITS, &RES, &DTP, &SK, &GATES,
                      &AMAT, &VEC, &SEQ_EMB, &IMG_EMB, &IMG_P, &IMG_X0, &IMG_X1, &IMG_T, &XN2, &ASCALE, &AMX_XN, &AMX_XA, &AMX_H, &X0, &U0})
      b->release();
    for (auto& t : twin)
      for (DevBuf& b : t) b.release();
    ascale_rows = 0;
    img_cap = 0;
    B = 0;
    tok_cap = 0;
  }
  int dh() const { return cfg.inner / cfg.n_heads; }
  int sdh() const { return cfg.d_model / cfg.n_heads; }
};

namespace {

const float* need(lram_engine* e, const std::string& name, size_t numel) {
  auto it = e->weights.find(name);
  if (it == e->weights.end()) throw Error("lram: missing weight '" + name + "'");
  if (it->second.n != numel)
    throw Error("lram: weight '" + name + "' has " + std::to_string(it->second.n) + " elements, expected " +
                std::to_string(numel));
  return it->second.p;
}
const float* optional(lram_engine* e, const std::string& name, size_t numel) {
  auto it = e->weights.find(name);
  if (it == e->weights.end()) return nullptr;
  if (it->second.n != numel)
    throw Error("lram: weight '" + name + "' has " + std::to_string(it->second.n) + " elements, expected " +
                std::to_string(numel));
  return it->second.p;
}

void validate_config(const lram_config& c) {
  LRAM_REQUIRE(c.abi_version == LRAM_ABI_VERSION, "config.abi_version does not match the library");
  LRAM_REQUIRE(c.backbone == LRAM_BACKBONE_XLSTM || c.backbone == LRAM_BACKBONE_MAMBA, "unknown backbone");
  LRAM_REQUIRE(c.d_model > 0 && c.d_model % 4 == 0 && c.d_model <= 2048, "d_model must be a multiple of 4, <= 2048");
  LRAM_REQUIRE(c.n_blocks > 0 && c.n_blocks <= LRAM_MAX_BLOCKS, "n_blocks out of range");
  LRAM_REQUIRE(c.tokens_per_step >= 1 && c.tokens_per_step <= 4, "tokens_per_step must be in 1..4");
  LRAM_REQUIRE(c.pred_token >= 0 && c.pred_token < c.tokens_per_step, "pred_token out of range");
  LRAM_REQUIRE(c.state_dim > 0 && c.state_dim % 4 == 0, "state_dim must be a positive multiple of 4");
  LRAM_REQUIRE(c.act_dim > 0 && c.n_vocab > 0 && c.n_discrete >= 0 && c.n_discrete <= c.n_vocab &&
                   c.action_channels > 0,
               "bad action head dimensions");
  if (c.backbone == LRAM_BACKBONE_XLSTM) {
    LRAM_REQUIRE(c.n_heads > 0 && c.inner > 0 && c.inner % (c.n_heads * 16) == 0,
                 "xLSTM inner dim must be a multiple of 16 * n_heads");
    LRAM_REQUIRE(c.d_model % c.n_heads == 0, "d_model must be a multiple of n_heads");
    // limits of the step kernels (xlstm_kernels.hip: kMaxGroups, kGnMaxV, the NH template instances), checked here so
    // that lram_create fails up front instead of a launch throwing in the middle of a step.  Every preset of the
    // reference's configs/agent_params/huggingface/xlstm_*.yaml passes (head dims 256 .. 896, inner <= 3584); head dims
    // that are not multiples of 64 (the *_half presets: 352, 544, 720) take 16-column cell slices and no lazy / chunkwise
    // path.  sLSTM blocks additionally need d_model / num_heads to be a multiple of 4 (checked below where they occur).
    LRAM_REQUIRE(c.n_heads == 1 || c.n_heads == 2 || c.n_heads == 4 || c.n_heads == 8, "xLSTM num_heads must be 1, 2, 4 or 8");
    LRAM_REQUIRE(c.inner <= 4096, "xLSTM inner dim (proj_factor * embedding_dim, rounded up to 64) must be <= 4096");
    LRAM_REQUIRE(c.inner / c.n_heads <= 1024, "mLSTM head dim must be <= 1024");
    LRAM_REQUIRE(c.d_model / c.n_heads <= 1024, "sLSTM head dim must be <= 1024");
    LRAM_REQUIRE(c.conv_k == 4, "conv1d_kernel_size must be 4");
    LRAM_REQUIRE(c.qkv_blocksize == 4, "qkv_proj_blocksize must be 4");
    bool any_s = false;
    for (int i = 0; i < c.n_blocks; ++i) any_s |= c.block_is_slstm[i] != 0;
    if (any_s) {
      LRAM_REQUIRE(c.ffn_dim > 0 && c.ffn_dim % 4 == 0, "ffn_dim must be a positive multiple of 4");
      LRAM_REQUIRE(c.d_model % (4 * c.n_heads) == 0, "sLSTM blocks need d_model to be a multiple of 4 * n_heads");
    }
  } else {
    LRAM_REQUIRE(c.d_inner > 0 && c.d_inner % 4 == 0 && c.d_conv == 4 && c.d_state > 0 && c.dt_rank > 0,
                 "bad Mamba dimensions");
  }
}

void make_split(lram_engine* e, const float* w, size_t n);
bool presplit_for(const lram_engine* e, const float* w, int rows, int n, int k);

void finalize(lram_engine* e) {
  const lram_config& c = e->cfg;
  const size_t D = c.d_model;
  e->w_state = need(e, "embed_state.weight", D * c.state_dim);
  e->b_state = need(e, "embed_state.bias", D);
  e->w_rtg = need(e, "embed_return.weight", D);
  e->b_rtg = need(e, "embed_return.bias", D);
  e->w_rew = need(e, "embed_rewards.weight", D);
  e->b_rew = need(e, "embed_rewards.bias", D);
  e->eln_g = need(e, "embed_ln.weight", D);
  e->eln_b = optional(e, "embed_ln.bias", D);
  e->w_head = need(e, "action_net.weight", (size_t)c.act_dim * c.n_vocab * D);
  e->b_head = need(e, "action_net.bias", (size_t)c.act_dim * c.n_vocab);
  e->post_g = need(e, "post_norm.gamma", D);
  e->post_b = optional(e, "post_norm.beta", D);
  e->bw.assign(c.n_blocks, BlockWeights());
  for (int i = 0; i < c.n_blocks; ++i) {
    BlockWeights& w = e->bw[i];
    const std::string p = "b" + std::to_string(i) + ".";
    w.norm_g = need(e, p + "norm.gamma", D);
    w.norm_b = optional(e, p + "norm.beta", D);
    if (c.backbone == LRAM_BACKBONE_MAMBA) {
      const size_t di = c.d_inner, N = c.d_state, R = c.dt_rank;
      w.in_proj = need(e, p + "in_proj", 2 * di * D);
      w.in_proj_b = optional(e, p + "in_proj_b", 2 * di);
      w.conv_w = need(e, p + "conv_w", di * 4);
      w.conv_b = optional(e, p + "conv_b", di);
      w.x_proj = need(e, p + "x_proj", (R + 2 * N) * di);
      w.dt_proj = need(e, p + "dt_proj", di * R);
      w.dt_bias = need(e, p + "dt_bias", di);
      w.A_log = need(e, p + "A_log", di * N);
      w.Dp = need(e, p + "D", di);
      w.out_proj = need(e, p + "out_proj", D * di);
      w.out_proj_b = optional(e, p + "out_proj_b", D);
    } else if (c.block_is_slstm[i]) {
      const size_t NH = c.n_heads, DH = D / NH, F = c.ffn_dim;
      w.conv_w = need(e, p + "conv_w", D * 4);
      w.conv_b = need(e, p + "conv_b", D);
      w.gate_w[0] = need(e, p + "gate_i", NH * DH * DH);
      w.gate_w[1] = need(e, p + "gate_f", NH * DH * DH);
      w.gate_w[2] = need(e, p + "gate_z", NH * DH * DH);
      w.gate_w[3] = need(e, p + "gate_o", NH * DH * DH);
      w.rt = need(e, p + "rt", NH * 4 * DH * DH);
      w.rbias = need(e, p + "rbias", 4 * D);
      w.gn_g = need(e, p + "gn.gamma", D);
      w.gn_b = optional(e, p + "gn.beta", D);
      w.ffn_norm_g = need(e, p + "ffn_norm.gamma", D);
      w.ffn_norm_b = optional(e, p + "ffn_norm.beta", D);
      w.ffn_up = need(e, p + "ffn_up", 2 * F * D);
      w.ffn_down = need(e, p + "ffn_down", D * F);
    } else {
      const size_t inner = c.inner, NH = c.n_heads;
      w.proj_up = need(e, p + "proj_up", 2 * inner * D);
      w.conv_w = need(e, p + "conv_w", inner * 4);
      w.conv_b = need(e, p + "conv_b", inner);
      w.wq = need(e, p + "wq", inner * 4);
      w.wk = need(e, p + "wk", inner * 4);
      w.wv = need(e, p + "wv", inner * 4);
      w.wi = need(e, p + "wi", NH * 3 * inner);
      w.bi = need(e, p + "bi", NH);
      w.wf = need(e, p + "wf", NH * 3 * inner);
      w.bf = need(e, p + "bf", NH);
      w.on_g = need(e, p + "outnorm.gamma", inner);
      w.on_b = optional(e, p + "outnorm.beta", inner);
      w.skip = need(e, p + "skip", inner);
      w.proj_down = need(e, p + "proj_down", D * inner);
    }
  }
  // optional image front end: embed_image.* with the reference's module names (image_encoders.py:39-56)
  e->img_lin_w = nullptr;
  e->img_channels = 0;
  {
    auto it = e->weights.find("embed_image.cnn.0.conv.weight");
    if (it != e->weights.end()) {
      const int chans[3] = {16, 32, 32};
      LRAM_REQUIRE(it->second.n % (16 * 9) == 0, "embed_image.cnn.0.conv.weight has an unexpected size");
      int cin = (int)(it->second.n / (16 * 9));
      e->img_channels = cin;
      for (int sidx = 0; sidx < 3; ++sidx) {
        const int cout = chans[sidx];
        const std::string p = "embed_image.cnn." + std::to_string(sidx) + ".";
        const char* names[5] = {"conv", "residual_0.conv_0", "residual_0.conv_1", "residual_1.conv_0", "residual_1.conv_1"};
        for (int k = 0; k < 5; ++k) {
          lram_engine::ImgConv& cv = e->img_conv[sidx][k];
          cv.cin = k == 0 ? cin : cout;
          cv.cout = cout;
          cv.w = need(e, p + names[k] + ".weight", (size_t)cout * cv.cin * 9);
          cv.b = need(e, p + names[k] + ".bias", (size_t)cout);
        }
        cin = cout;
      }
      auto lw = e->weights.find("embed_image.linear.0.weight");
      LRAM_REQUIRE(lw != e->weights.end() && lw->second.n % D == 0, "embed_image.linear.0.weight missing or mis-sized");
      e->img_flat = (int)(lw->second.n / D);
      e->img_lin_w = lw->second.p;
      e->img_lin_b = need(e, "embed_image.linear.0.bias", D);
    }
  }
  // bf16 split planes of every GEMM weight (LRAM_GEMM=f32 keeps the exact fp32-MFMA kernels instead)
  e->drop_splits();
  if (const char* v = std::getenv("LRAM_GEMM")) {
    e->use_bf16x3 = std::string(v) != "f32";
    e->use_f16x2 = std::string(v) != "f32" && std::string(v) != "bf16x3";
  }
  if (e->use_bf16x3 && e->use_f16x2) {
    // the big un-batched projections: (weight, rows, K); the per-head / per-gate batched GEMMs of the sLSTM block keep
    // bf16x3 (their operand rows would need one scale per head)
    const int D = c.d_model;
    auto rows_of = [&](const float* p, size_t k) -> size_t {
      for (auto& kv : e->weights)
        if (kv.second.p == p) return kv.second.n / k;
      return 0;
    };
    auto add16 = [&](const float* p, size_t k) {
      if (p == nullptr || k == 0 || (k & 7) != 0 || e->split16.count(p)) return;
      const size_t rows = rows_of(p, k);
      if (rows == 0) return;
      lram_engine::Split16 sp{nullptr, nullptr, rows, k};
      LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sp.planes), 2 * split_f16x2_plane_elems(rows, k) * sizeof(uint16_t)));
      LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sp.inv), rows * sizeof(float)));
      launch_split_f16x2(p, (int)rows, (int)k, sp.planes, sp.inv, nullptr);
      e->split16[p] = sp;
    };
    add16(e->w_head, D);
    if (e->img_lin_w != nullptr) add16(e->img_lin_w, (size_t)e->img_flat);
    for (const BlockWeights& w : e->bw) {
      add16(w.proj_up, D), add16(w.proj_down, c.inner), add16(w.ffn_up, D), add16(w.ffn_down, c.ffn_dim);
      add16(w.in_proj, D), add16(w.x_proj, c.d_inner), add16(w.out_proj, c.d_inner);
      // (dt_proj, K = dt_rank = 48: two K tiles, nothing to gain -- 24.8 us vs 20.8 us for bf16x3 at 6144 rows)
    }
  }
  if (e->use_bf16x3) {
    auto numel = [&](const float* p) -> size_t {
      for (auto& kv : e->weights)
        if (kv.second.p == p) return kv.second.n;
      return 0;
    };
    std::vector<const float*> ws = {e->w_head, e->img_lin_w};  // embed_state has K = state_dim (204): rows not 16-byte aligned
    for (const BlockWeights& w : e->bw)
      for (const float* p : {w.proj_up, w.proj_down, w.gate_w[0], w.gate_w[1], w.gate_w[2], w.gate_w[3], w.rt, w.ffn_up,
                             w.ffn_down, w.in_proj, w.x_proj, w.dt_proj, w.out_proj})
        ws.push_back(p);
    for (const float* p : ws)
      if (p != nullptr) make_split(e, p, numel(p));
    LRAM_HIP_CHECK(hipDeviceSynchronize());
  }
  e->gate_coef.assign(e->bw.size(), DevBuf());
  if (c.backbone == LRAM_BACKBONE_XLSTM && mlstm_front_supported(c.inner, c.n_heads, c.conv_k, c.tokens_per_step)) {
    for (size_t i = 0; i < e->bw.size(); ++i) {
      if (c.block_is_slstm[i]) continue;
      const BlockWeights& w = e->bw[i];
      e->gate_coef[i].alloc((size_t)c.inner * 4 * c.n_heads);
      launch_gate_coef(w.wq, w.wk, w.wv, w.wi, w.wf, c.inner, c.n_heads, e->gate_coef[i].p, nullptr);
    }
    LRAM_HIP_CHECK(hipDeviceSynchronize());
  }
  e->slstm_rt2.assign(e->bw.size(), DevBuf());
  e->slstm_rinv.assign(e->bw.size(), DevBuf());
  if (c.backbone == LRAM_BACKBONE_XLSTM && slstm_seq_supported(c.d_model, c.n_heads, c.tokens_per_step)) {
    for (size_t i = 0; i < e->bw.size(); ++i) {
      if (!c.block_is_slstm[i]) continue;
      const size_t sdh = (size_t)c.d_model / c.n_heads;
      e->slstm_rt2[i].alloc((size_t)c.n_heads * 4 * sdh * sdh);
      if (e->use_f16x2 && !e->slstm_seq_f32) {  // (two f16 planes: the bytes of the fp32 copy)
        e->slstm_rinv[i].alloc((size_t)c.n_heads * 4 * sdh);
        launch_slstm_pack_rt16(e->bw[i].rt, reinterpret_cast<uint16_t*>(e->slstm_rt2[i].p), e->slstm_rinv[i].p, c.n_heads, (int)sdh, nullptr);
      } else {
        launch_slstm_pack_rt(e->bw[i].rt, e->slstm_rt2[i].p, c.n_heads, (int)sdh, nullptr);
      }
    }
    LRAM_HIP_CHECK(hipDeviceSynchronize());
  }
  if (c.backbone == LRAM_BACKBONE_MAMBA && e->gemm_narrow_on && gemm_narrow_shape(c.dt_rank + 2 * c.d_state, c.d_inner)) {
    const int nx = c.dt_rank + 2 * c.d_state;
    for (const BlockWeights& w : e->bw) {
      if (w.x_proj == nullptr || e->narrow.count(w.x_proj)) continue;
      DevBuf& pk = e->narrow[w.x_proj];
      pk.alloc(gemm_narrow_pack_elems(nx, c.d_inner));
      launch_gemm_narrow_pack(w.x_proj, nx, c.d_inner, pk.p, nullptr);
    }
    LRAM_HIP_CHECK(hipDeviceSynchronize());
  }
  e->dt_wt.assign(e->bw.size(), DevBuf());
  if (c.backbone == LRAM_BACKBONE_MAMBA && e->mamba_dt_fuse && mamba_ssm_dt_fusable(c.d_state, c.dt_rank)) {
    for (size_t i = 0; i < e->bw.size(); ++i) {
      e->dt_wt[i].alloc((size_t)c.d_inner * c.dt_rank);
      launch_transpose_f32(e->bw[i].dt_proj, c.d_inner, c.dt_rank, e->dt_wt[i].p, nullptr);
    }
    LRAM_HIP_CHECK(hipDeviceSynchronize());
  }
  e->finalized = true;
}

// Activation workspace for `tokens` tokens per env slot (rows b * T + t of every buffer).
size_t workspace_floats_per_token(const lram_config& c) {
  const size_t D = c.d_model;
  if (c.backbone == LRAM_BACKBONE_MAMBA) return 5 * D + 4 * (size_t)c.d_inner + c.dt_rank + 2 * c.d_state + (size_t)c.d_inner;
  const size_t inner = c.inner;
  const size_t ucols = std::max<size_t>(std::max<size_t>(2 * inner, 4 * D), 2 * (size_t)c.ffn_dim);
  const size_t icols = std::max<size_t>(std::max<size_t>(inner, D), (size_t)c.ffn_dim);
  return 4 * D + ucols + 6 * icols + 6 * (size_t)c.n_heads;
}

void alloc_workspace(lram_engine* e, int tokens) {
  const lram_config& c = e->cfg;
  const size_t B = e->B, D = c.d_model, BT = B * (size_t)tokens;
  e->SK.alloc(lram_engine::kSplitKSlotElems * lram_engine::kSplitKSlots);
  e->ascale_rows = BT;
  e->ASCALE.alloc(BT * lram_engine::kSplitKSlots);
  const size_t parts = c.backbone == LRAM_BACKBONE_MAMBA ? std::max<size_t>(1, c.d_inner / 64) : 0;
  e->AMX_XN.alloc(BT);
  if (parts) e->AMX_XA.alloc(BT * parts), e->AMX_H.alloc(BT * parts);
  if (c.backbone == LRAM_BACKBONE_XLSTM && e->use_f16x2) e->AMX_H.alloc(BT * (size_t)c.n_heads);  // group norm -> proj_down: per-head maxima
  e->X.alloc(BT * D);
  e->XN.alloc(BT * D);
  if (e->gemm_presplit && e->use_f16x2) e->XN2.alloc(BT * D);
  e->TOK.alloc(BT * D);
  e->HID.alloc(BT * D);
  e->LOGITS.alloc(B * c.act_dim * c.n_vocab);
  if (c.backbone == LRAM_BACKBONE_MAMBA) {
    const size_t di = c.d_inner;
    e->RES.alloc(BT * D);
    e->U.alloc(BT * 2 * di);                        // xz
    e->XA.alloc(BT * di);                           // xc
    e->Q.alloc(BT * (c.dt_rank + 2 * c.d_state));   // x_proj output
    e->DTP.alloc(BT * di);
    e->H.alloc(BT * di);                            // y
  } else {
    const size_t inner = c.inner;
    const size_t ucols = std::max<size_t>(std::max<size_t>(2 * inner, 4 * D), 2 * (size_t)c.ffn_dim);
    const size_t icols = std::max<size_t>(std::max<size_t>(inner, D), (size_t)c.ffn_dim);
    e->ucols = ucols, e->icols = icols;
    e->U.alloc(BT * ucols);
    e->Q.alloc(BT * icols);
    e->K.alloc(BT * icols);
    e->V.alloc(BT * icols);
    e->XA.alloc(BT * icols);
    e->H.alloc(BT * icols);
    e->G.alloc(BT * icols);
    e->SCAL.alloc(BT * c.n_heads * 4);
    e->RY.alloc(B * 4 * D);
    if (tokens > kMaxTokens) {
      e->GATES.alloc(BT * c.n_heads * 2);
      e->AMAT.alloc(B * c.n_heads * kChunkMaxTokens * kChunkMaxTokens);
      e->VEC.alloc(B * c.n_heads * 3 * kChunkMaxTokens);
    }
  }
  e->tok_cap = tokens;
}

// The per-token activation buffers a chunk of a stored context goes through (xLSTM and Mamba): what a second chunk in flight needs its own copy of.
// (SK / ASCALE are per stream already; LOGITS / TOK belong to the last timestep, which always runs on the primary set.)
std::array<DevBuf*, 21> workspace_set(lram_engine* e) {
  return {&e->X, &e->XN, &e->XN2, &e->HID, &e->U, &e->Q, &e->K, &e->V, &e->XA, &e->H, &e->G, &e->SCAL, &e->RY, &e->GATES,
          &e->AMAT, &e->VEC, &e->AMX_XN, &e->AMX_H, &e->RES, &e->DTP, &e->AMX_XA};
}
void swap_workspace(lram_engine* e, int lane) {   // lane >= 1: primary <-> that lane's copy
  const auto ws = workspace_set(e);
  for (size_t i = 0; i < ws.size(); ++i) std::swap(*ws[i], e->twin[lane - 1][i]);
}
// Second workspace for the chunk lanes, sized like the first; false (no lanes) if the device has no room for it.
bool twin_ready(lram_engine* e) {
  const auto ws = workspace_set(e);
  bool same = true;
  size_t want = 0;
  for (int l = 0; l + 1 < e->n_lanes; ++l)
    for (size_t i = 0; i < ws.size(); ++i) same = same && e->twin[l][i].n == ws[i]->n, want += ws[i]->n * sizeof(float);
  if (same) return true;
  LRAM_HIP_CHECK(hipDeviceSynchronize());
  for (auto& t : e->twin)
    for (DevBuf& b : t) b.release();
  size_t free_b = 0, total_b = 0;
  LRAM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  if (want + ((size_t)2 << 30) > free_b) return false;  // keep 2 GiB of headroom
  for (int l = 0; l + 1 < e->n_lanes; ++l)
    for (size_t i = 0; i < ws.size(); ++i) e->twin[l][i].alloc(ws[i]->n);
  LRAM_HIP_CHECK(hipDeviceSynchronize());
  return true;
}

// Timesteps per state pass for a stored context of L timesteps.  xLSTM geometries the chunkwise kernels cover
// take up to 21 timesteps (63 tokens) per pass, in equal chunks; everything else 4 (the token-sequential kernels).
// Grows the activation workspace on first use when the device has room for it.
int prefill_chunk_steps(lram_engine* e, int L) {
  const lram_config& c = e->cfg;
  const int T = c.tokens_per_step;
  const int seq = kMaxTokens / T;
  if (!e->chunk_prefill || c.backbone != LRAM_BACKBONE_XLSTM || L <= seq || e->graph_mode ||
      !mlstm_chunk_supported(c.inner, c.n_heads, c.conv_k))
    return seq;
  const int max_steps = kChunkMaxTokens / T;
  const int n_chunks = (L + max_steps - 1) / max_steps;
  const int steps = (L + n_chunks - 1) / n_chunks;
  if (steps * T <= kMaxTokens) return seq;
  if (steps * T > e->tok_cap) {
    LRAM_HIP_CHECK(hipDeviceSynchronize());
    size_t free_b = 0, total_b = 0;
    LRAM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const size_t have = workspace_floats_per_token(c) * 4 * (size_t)e->B * e->tok_cap;
    const size_t want = workspace_floats_per_token(c) * 4 * (size_t)e->B * kChunkMaxTokens +
                        (size_t)e->B * c.n_heads * (kChunkMaxTokens + 3) * kChunkMaxTokens * 4;
    if (want > have + free_b - std::min<size_t>(free_b, (size_t)2 << 30)) return seq;  // keep 2 GiB of headroom
    alloc_workspace(e, kChunkMaxTokens);
    LRAM_HIP_CHECK(hipDeviceSynchronize());
  }
  return steps;
}

// ---- lazy matrix memory plumbing ------------------------------------------------------------------------
bool lazy_geometry_ok(const lram_engine* e) {
  return e->cfg.backbone == LRAM_BACKBONE_XLSTM && mlstm_lazy_supported(e->cfg.inner / e->cfg.n_heads, e->cfg.tokens_per_step);
}

void lazy_alloc(lram_engine* e) {
  if (e->lazy_ready || e->B <= 0 || !lazy_geometry_ok(e)) return;
  const lram_config& c = e->cfg;
  const size_t B = e->B, NH = c.n_heads, DH = e->dh();
  for (int i = 0; i < c.n_blocks; ++i) {
    if (c.block_is_slstm[i]) continue;
    BlockState& s = e->st[i];
    s.wk.alloc(B * NH * kLazyWindow * DH);
    s.wv.alloc(B * NH * kLazyWindow * DH);
    s.coef.alloc(2 * B * NH * kLazyWindow);
    s.gsc.alloc(2 * B * NH);
    if (!mlstm_lazy_fused_scores((int)DH)) {
      s.pw.alloc(B * NH * 4 * kLazyWT);
      s.pw.zero();
    }
    s.coef.zero();
  }
  e->LZ_COUNT.alloc(2 * B);
  e->LZ_COUNT.zero();
  for (int i = 0; i < c.n_blocks; ++i) {
    if (c.block_is_slstm[i]) continue;
    BlockState& s = e->st[i];
    for (int p = 0; p < 2; ++p)
      launch_mlstm_lazy_clear(reinterpret_cast<int32_t*>(e->LZ_COUNT.p) + p * B, s.gsc.p + p * B * NH, nullptr, (int)B,
                              (int)NH, nullptr);
  }
  LRAM_HIP_CHECK(hipDeviceSynchronize());
  e->lazy_step = 0;
  e->lazy_ready = true;
}

// auto: lazy where the state pass dominates -- one mLSTM block's matrix memory of at least 128 MiB over the batch
// (16M geometry from 128 env slots, 206M from 21); below that the extra launches per block cost more than the saved
// bytes.  Round 6, one box, lazy vs materialised env-steps/s: 16M at 128 / 256 / 384 envs 151.4k vs 145.4k / 207.2k vs 191.0k /
// 269.8k vs 233.1k; 206M at 8 / 16 / 32 / 64 envs 3.27k vs 3.69k / 6.00k vs 6.13k / 9.78k vs 9.33k / 12.35k vs 11.21k.
// (Rounds 2-5 used 512 MiB, from round 2's kernels: 16M at 256 envs 174k lazy vs 178k materialised then.)
bool lazy_choice(const lram_engine* e) {
  if (e->lazy_mode == 0 || !lazy_geometry_ok(e)) return false;
  if (e->lazy_mode == 1) return true;
  const double dh = e->cfg.inner / e->cfg.n_heads;
  return (double)e->B * e->cfg.n_heads * dh * dh * 4.0 >= 128.0 * 1024 * 1024;
}

bool lazy_active(const lram_engine* e, int T) {
  return e->lazy && e->lazy_ready && !e->graph_mode && T >= 1 && T <= 4;
}

MlstmLazyArgs lazy_args(lram_engine* e, int i, int T, const uint8_t* reset, int b0, int nb) {
  const lram_config& c = e->cfg;
  const size_t NH = c.n_heads, DH = e->dh(), B = e->B;
  const int in = (int)(e->lazy_step & 1), out = 1 - in;
  BlockState& st = e->st[i];
  MlstmLazyArgs a{};
  a.C = st.s0.p + (size_t)b0 * NH * DH * DH;
  a.wk = st.wk.p + (size_t)b0 * NH * kLazyWindow * DH;
  a.wv = st.wv.p + (size_t)b0 * NH * kLazyWindow * DH;
  a.coef_in = st.coef.p + (in * B + b0) * NH * kLazyWindow;
  a.coef_out = st.coef.p + (out * B + b0) * NH * kLazyWindow;
  a.g_in = st.gsc.p + (in * B + b0) * NH;
  a.g_out = st.gsc.p + (out * B + b0) * NH;
  a.count_in = reinterpret_cast<const int32_t*>(e->LZ_COUNT.p) + in * B + b0;
  a.count_out = reinterpret_cast<int32_t*>(e->LZ_COUNT.p) + out * B + b0;
  a.pw = st.pw.p ? st.pw.p + (size_t)b0 * NH * T * kLazyWT : nullptr;
  a.reset = reset ? reset + b0 : nullptr;
  a.B = nb, a.T = T, a.NH = (int)NH, a.DH = (int)DH;
  // the fold phase is taken relative to the env's global index, so slices fold the same envs as the whole batch
  a.phase = (int)((e->lazy_step + b0) % e->lazy_period), a.period = e->lazy_period, a.force = 0;
  return a;
}

// Fold every pending window into C_base and empty the bookkeeping: afterwards the state is the materialised
// reference layout again (export / import, prefill, long encoder calls, leaving lazy mode).
void lazy_materialize(lram_engine* e, hipStream_t s) {
  if (!e->lazy_ready || !e->lazy_dirty) return;
  const lram_config& c = e->cfg;
  const size_t B = e->B, NH = c.n_heads;
  for (int i = 0; i < c.n_blocks; ++i) {
    if (c.block_is_slstm[i]) continue;
    MlstmLazyArgs a = lazy_args(e, i, 1, nullptr, 0, e->B);
    a.force = 1;
    launch_mlstm_lazy_fold(a, s);
  }
  for (int i = 0; i < c.n_blocks; ++i) {
    if (c.block_is_slstm[i]) continue;
    for (int p = 0; p < 2; ++p)
      launch_mlstm_lazy_clear(reinterpret_cast<int32_t*>(e->LZ_COUNT.p) + p * B, e->st[i].gsc.p + p * B * NH, nullptr,
                              (int)B, (int)NH, s);
  }
  e->lazy_bound.assign(e->lazy_period, 0);
  e->lazy_dirty = false;
}

void state_alloc(lram_engine* e, int B) {
  LRAM_REQUIRE(e->finalized, "lram_finalize must be called before lram_state_alloc");
  LRAM_REQUIRE(B > 0 && B <= 65535, "batch must be in 1..65535");
  LRAM_HIP_CHECK(hipSetDevice(e->device));
  e->drop_graph();
  e->release_state();
  const lram_config& c = e->cfg;
  const size_t D = c.d_model;
  e->st.resize(c.n_blocks);
  for (int i = 0; i < c.n_blocks; ++i) {
    BlockState& s = e->st[i];
    if (c.backbone == LRAM_BACKBONE_MAMBA) {
      s.s0.alloc((size_t)B * c.d_inner * c.d_state);
      s.conv.alloc((size_t)B * c.d_inner * c.d_conv);
    } else if (c.block_is_slstm[i]) {
      s.s0.alloc(4 * (size_t)B * D);
      s.conv.alloc((size_t)B * c.conv_k * D);
    } else {
      const size_t DH = e->dh();
      s.s0.alloc((size_t)B * c.n_heads * DH * DH);
      s.n.alloc((size_t)B * c.inner);
      s.m.alloc((size_t)B * c.n_heads);
      s.conv.alloc((size_t)B * c.conv_k * c.inner);
    }
    s.s0.zero();
    s.n.zero();
    s.m.zero();
    s.conv.zero();
  }
  e->B = B;
  alloc_workspace(e, kMaxTokens);
  e->lazy = lazy_choice(e);
  if (e->lazy) lazy_alloc(e);
  LRAM_HIP_CHECK(hipDeviceSynchronize());
}

// ---------------------------------------------------------------------------------------------
// block stack on X [B*T, D] (in place residual stream) -> HID [B*T, D]
// ---------------------------------------------------------------------------------------------
// The few-row kernel's share of the dispatch (see gemm()).
bool takes_skinny(const lram_engine* e, const GemmArgs& g) {
  const bool shape = g.k <= e->gemm_skinny_k && (g.m <= e->gemm_skinny_rows / 2 || (int64_t)g.n * g.k <= 600000);
  return g.m >= e->gemm_skinny_min && g.m <= e->gemm_skinny_rows && shape && gemm_skinny_supported(g);
}
// ... and may the norm ahead of this projection move into its prologue?  (Then the caller skips the norm launch and hands
// the un-normalised rows over with norm_g / norm_b / norm_eps / norm_rms set.)
bool takes_skinny_with_norm(const lram_engine* e, const GemmArgs& g) {
  return takes_skinny(e, g) && gemm_skinny_norm_supported(g);
}

// The narrow-output kernel's share: a whole packed weight (x_proj), enough rows to fill the chip with 16-row workgroups.
bool narrow_takes(const lram_engine* e, const GemmArgs& g) {
  if (!e->gemm_narrow_on || g.m < e->gemm_narrow_min_rows || g.a2 != nullptr || (int)g.ldw != g.k) return false;
  return e->narrow.count(g.w) != 0 && gemm_narrow_supported(g);
}

void count_gemm(lram_engine* e, int family, const GemmArgs& g) {
  e->gemm_counts[family] += 1.0;
  e->gemm_counts[4 + family] += 2.0 * g.m * g.n * g.k * g.nb1 * g.nb2;
}

// ---- which projections take the f16x2 kernels: ONE predicate for the dispatcher and for the producers of the operands ------
// Row threshold: from 256 rows, wider weights earlier (below).  (Rounds 3-5: 1024 / 512, from
// the time the f16x2 kernels needed a row-maximum launch per projection; the producers hand the maxima over since round 5.)
// Round 6, one box, one env slice, env-steps/s with the old / new thresholds: 206M at 64 / 128 / 256 envs 12.4k / 17.9k / 24.8k ->
// 15.0k / 21.1k / 25.1k; Mamba-48M at 128 / 256 envs 84.2k / 149.6k -> 99.2k / 178.7k; 16M at 64 / 128 / 256 envs 95.8k / 150.6k /
// 207.4k -> 95.8k / 151.7k / 219.5k (16M at 64 envs = 192 rows on f16x2: 94.0k, hence 256 for the narrow weights).
// Below 256 rows by weight size: >= 2.5 M elements (206M stack) from 48 rows (206M at 16 envs 6.13k -> 6.65k), >= 1.1 M (Mamba-48M's
// in_proj / out_proj; not the 16M stack's 2048 x 512) from 96 (Mamba-48M at 32 / 64 envs 37.4k / 50.7k -> 40.1k / 52.7k).
bool f16x2_rows(const lram_engine* e, int rows, int n, int k) {
  const int64_t nk = (int64_t)n * k;
  return e->use_f16x2 && (rows >= e->f16x2_min_rows || (rows >= 96 && nk >= 1100000) || (rows >= 48 && nk >= 2500000));
}
// The f16 planes of the weight tensor that contains w (a GEMM may address a row range of a weight: proj_up's halves): fills the
// operand fields of g and returns true when w starts on a whole row of a split weight whose K equals ldw.
bool f16x2_weight(const lram_engine* e, const float* w, int ldw, GemmArgs* g) {
  auto it = e->split16.upper_bound(w);
  if (it == e->split16.begin()) return false;
  --it;
  if (!(w < it->first + it->second.rows * it->second.k) || (int)it->second.k != ldw) return false;
  const size_t row0 = (size_t)(w - it->first) / it->second.k;
  if (row0 * it->second.k != (size_t)(w - it->first)) return false;   // planes are addressed by whole rows
  if (g != nullptr) {
    g->w2 = it->second.planes + row0 * 32;  // K-tile-major planes
    g->w2_plane = (int64_t)split_f16x2_plane_elems(it->second.rows, it->second.k), g->w2_kt = (int64_t)it->second.rows * 32;
    g->w_inv = it->second.inv + row0;
  }
  return true;
}

// Does the projection `rows x k` against weight w take the f16x2 kernel with BOTH operands pre-split (gemm_f16x2p.hip)?  The
// producer of A (a row norm) asks before it chooses its output format, gemm() asks the same question through the a2 operand:
// the two cannot drift apart.  K a multiple of the kernel's 32-deep tile (d_model <= 2048: the norm kernels' limit, checked by
// validate_config).
bool presplit_for(const lram_engine* e, const float* w, int rows, int n, int k) {
  if (!e->gemm_presplit || (k & 31) != 0 || e->XN2.p == nullptr || !f16x2_rows(e, rows, n, k)) return false;
  // (the kernel's LDS-DMA addresses an operand's two planes with 32-bit byte offsets: gemm_f16x2p_supported)
  GemmArgs probe;
  if ((int64_t)e->XN2.n * 4 >= (1ll << 31) || !f16x2_weight(e, w, k, &probe)) return false;
  return 4 * probe.w2_plane < (1ll << 31);
}

int stream_slot(const lram_engine* e, hipStream_t s) {  // split-K slab / row-maximum region of the stream a GEMM runs on
  for (size_t i = 0; i < e->micro_streams.size() && i + 1 < (size_t)lram_engine::kSplitKSlots; ++i)
    if (e->micro_streams[i] == s) return (int)i + 1;
  return 0;
}

// GEMM dispatch: f16x2 (both operands pre-split, or A split while it is staged) for the big un-batched projections, the
// few-row kernel for tens of rows, bf16x3 for the batched per-head GEMMs and whatever is left, exact fp32 MFMA as the fallback.
void gemm(lram_engine* e, GemmArgs& g, hipStream_t s) {
  if (e->SK.p != nullptr) {
    g.splitk_ws = e->SK.p + (size_t)stream_slot(e, s) * lram_engine::kSplitKSlotElems;
    g.splitk_ws_elems = (int64_t)lram_engine::kSplitKSlotElems;
  }
  if (g.a2 != nullptr) {  // A handed over as f16x2 operand planes by its producer (presplit_for() said this GEMM takes them)
    LRAM_REQUIRE(f16x2_weight(e, g.w, (int)g.ldw, &g) && gemm_f16x2p_supported(g),
                 "gemm: pre-split A operand for a projection the pre-split kernel does not serve");
    launch_gemm_f16x2p(g, s);
    count_gemm(e, 0, g);
    return;
  }
  if (narrow_takes(e, g)) {  // narrow outputs (Mamba x_proj): one launch, no split-K slabs / reduce launch
    if (e->use_f16x2 && e->gemm_narrow_f16 && g.a_amax != nullptr && f16x2_weight(e, g.w, (int)g.ldw, &g) && gemm_narrow16_supported(g)) {
      launch_gemm_narrow16(g, s);   // f16x2 split products (the operand's row maxima come from its producer)
      count_gemm(e, 0, g);
      return;
    }
    g.w2 = nullptr, g.w_inv = nullptr, g.w2_kt = 0;
    launch_gemm_narrow(g, e->narrow.find(g.w)->second.p, s);   // exact fp32
    count_gemm(e, 2, g);
    return;
  }
  if (f16x2_rows(e, g.m, g.n, g.k) && g.nb1 * g.nb2 == 1 && e->ASCALE.p != nullptr && (size_t)g.m <= e->ascale_rows) {
    if (f16x2_weight(e, g.w, (int)g.ldw, &g) && gemm_f16x2_supported(g)) {
      if (g.a_amax == nullptr) {  // no producer handed the row maxima over: one small launch ahead of the GEMM
        float* sc = e->ASCALE.p + (size_t)stream_slot(e, s) * e->ascale_rows;
        launch_row_amax(g.a, g.lda, g.gate, g.ldg, g.m, g.k, sc, s);
        g.a_amax = sc, g.amax_parts = 1;
      }
      launch_gemm_f16x2(g, s);
      count_gemm(e, 0, g);
      return;
    }
    g.w2 = nullptr, g.w_inv = nullptr, g.w2_kt = 0;
  }
  // few operand rows (more than the GEMV's 8, at most gemm_skinny_rows): one 32 x 32 fp32 matrix-core tile per workgroup, operands
  // straight into registers, no split-K slab / reduce launch
  // Where it wins (same box each, `profiles/r03_ab_gemm_few_rows.txt`): K <= 1024 -- a lane group walks its K range in rounds
  // of 8 float4, one memory round trip each, so a long K is a long serial chain where the tile kernels' split-K spreads it
  // over workgroups (Mamba x_proj / out_proj, K = 1536: -3 % each at 32 envs; the 206M stack's K = 1280 / 2560: -7 % at 64
  // envs) -- and up to 192 operand rows, 384 for weights of at most 600k elements (every 32-row tile re-reads the weight).
  // 16M at 4 / 12 / 32 / 64 / 128 envs: +17 / +17 / +16 / +12 / +10 %; C1 (2 blocks, D = 128) at 32 envs: 0.130 -> 0.093 ms.
  if (takes_skinny(e, g)) {
    launch_gemm_skinny(g, s);
    count_gemm(e, 3, g);
    return;
  }
  if (e->use_bf16x3 && !gemm_small_m(g)) {
    // planes of the weight tensor that contains g.w (a GEMM may address a row range of a weight: proj_up's halves)
    auto it = e->split.upper_bound(g.w);
    if (it != e->split.begin() && (--it, g.w < it->first + it->second.n)) {
      g.w3 = it->second.p + (g.w - it->first);
      g.w3_plane = (int64_t)it->second.n;
      if (gemm_bf16x3_supported(g)) {
        launch_gemm_bf16x3(g, s);
        count_gemm(e, 1, g);
        return;
      }
    }
  }
  LRAM_REQUIRE(g.gate == nullptr && g.act_silu_from < 0, "gemm: gated operand / output activation need the bf16x3 kernel");
  launch_gemm_f32(g, s);
  count_gemm(e, 2, g);
}

void make_split(lram_engine* e, const float* w, size_t n) {
  if (w == nullptr || e->split.count(w)) return;
  uint16_t* p = nullptr;
  LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), 3 * n * sizeof(uint16_t)));
  launch_split_bf16x3(w, p, n, nullptr);
  e->split[w] = lram_engine::Split{p, n};
}

void prof_record(lram_engine* e, hipStream_t s, bool start, bool aux = false) {
  if (!e->prof_on || !e->prof_live) return;
  if (start) {
    if (e->prof_used == e->prof_events.size()) {
      hipEvent_t a, b;
      // (timing only: no system-scope fence -- the header's own advice for events that measure)
      const unsigned flags = e->event_device_scope ? hipEventDisableSystemFence : hipEventDefault;
      LRAM_HIP_CHECK(hipEventCreateWithFlags(&a, flags));
      LRAM_HIP_CHECK(hipEventCreateWithFlags(&b, flags));
      e->prof_events.emplace_back(a, b);
      e->prof_aux.push_back(0);
    }
    e->prof_aux[e->prof_used] = aux ? 1 : 0;
    LRAM_HIP_CHECK(hipEventRecord(e->prof_events[e->prof_used].first, s));
  } else {
    LRAM_HIP_CHECK(hipEventRecord(e->prof_events[e->prof_used].second, s));
    ++e->prof_used;
  }
}

// A contiguous range of env slots processed on its own stream.  All activation buffers are indexed by
// row b*T + t, so a slice simply works on rows [b0*T, (b0+nb)*T) of the shared buffers.
struct Slice {
  int b0, nb;
  hipStream_t s;
};

// `dst` waits for everything enqueued so far on `src` (event from the engine's pool; also legal under
// stream capture, where it becomes a graph edge).
// boundary = false: both streams are the engine's own (slice streams, state-pass stream).  Those events are created with
// hipEventDisableSystemFence: a default event performs a SYSTEM-scope release / acquire when it is recorded -- cache write-back
// and invalidation for the host's and other devices' benefit -- ~150 times per env-step, between kernels of one device whose
// launches already order their memory at device scope.  boundary = true (fork from / join into the caller's stream): default
// events, the caller may hand the results to a copy engine or the host next.
hipEvent_t ring_event(lram_engine* e) {   // an event of the engine's own ring (device scope), for a record / wait pair placed apart
  constexpr size_t kRing = 512;
  std::vector<hipEvent_t>& pool = e->sync_events;
  if (pool.size() < kRing && e->sync_used >= pool.size()) {
    hipEvent_t nev;
    LRAM_HIP_CHECK(hipEventCreateWithFlags(&nev, hipEventDisableTiming | (e->event_device_scope ? hipEventDisableSystemFence : 0u)));
    pool.push_back(nev);
  }
  return pool[e->sync_used++ % pool.size()];
}

void stream_after(lram_engine* e, hipStream_t dst, hipStream_t src, bool boundary = false) {
  if (dst == src) return;
  // ring of events: a wait captures the record that precedes it at call time, so re-recording an event later
  // (next timestep / next call) cannot disturb waits that are already enqueued
  constexpr size_t kRing = 512;
  std::vector<hipEvent_t>& pool = boundary ? e->edge_events : e->sync_events;
  size_t& used = boundary ? e->edge_used : e->sync_used;
  if (pool.size() < kRing && used >= pool.size()) {
    hipEvent_t nev;
    unsigned flags = hipEventDisableTiming;
    if (!boundary && e->event_device_scope) flags |= hipEventDisableSystemFence;
    LRAM_HIP_CHECK(hipEventCreateWithFlags(&nev, flags));
    pool.push_back(nev);
  }
  hipEvent_t ev = pool[used++ % pool.size()];
  LRAM_HIP_CHECK(hipEventRecord(ev, src));
  LRAM_HIP_CHECK(hipStreamWaitEvent(dst, ev, 0));
}

// Slices for this call.  One slice on the caller's stream unless micro-batching is on: then n_micro slices on
// engine-owned streams plus one stream that serialises the HBM-bound cell kernels (see run_xlstm_stack).
std::vector<Slice> make_slices(lram_engine* e, hipStream_t s, hipStream_t* hbm) {
  int n = e->n_micro;
  if (n == 0) {
    // auto: two slices where the second one has something long to hide behind.  xLSTM: one mLSTM block's matrix memory over the
    // batch of at least 512 MiB (16M from 512 env slots, 206M from 82); Mamba: from 1024 env slots.  Round 6, one box, one vs two
    // slices, env-steps/s: 206M at 64 / 96 / 128 / 256 envs 15.1k vs 14.9k / 18.5k vs 18.5k / 21.1k vs 21.6k / 25.1k vs 29.1k
    // (rounds 2-5 split from 512 envs only); 16M at 256 / 512 / 640 envs 219.8k vs 194.1k / 296.2k vs 297.1k / 304.8k vs 314.1k;
    // Mamba-48M at 512 / 768 / 1024 / 1536 envs 288.6k vs 275.0k / 380.4k vs 367.7k / 410.5k vs 416.9k / 432.8k vs 479.4k.
    if (e->cfg.backbone == LRAM_BACKBONE_MAMBA) {
      n = e->B >= 1024 ? 2 : 1;
    } else {
      const double dh = e->cfg.n_heads > 0 ? (double)e->cfg.inner / e->cfg.n_heads : 0.0;
      n = (double)e->B * e->cfg.n_heads * dh * dh * 4.0 >= 512.0 * 1024 * 1024 ? 2 : 1;
    }
  }
  if (e->graph_mode) n = 1;  // graph replay targets small, launch-bound batches: one slice, one stream
  n = std::max(1, std::min(n, std::min(e->B, 8)));
  *hbm = s;
  if (n == 1) return {Slice{0, e->B, s}};
  while ((int)e->micro_streams.size() < n) {
    hipStream_t ns;
    LRAM_HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
    e->micro_streams.push_back(ns);
  }
  if (!e->hbm_stream) LRAM_HIP_CHECK(hipStreamCreateWithFlags(&e->hbm_stream, hipStreamNonBlocking));
  *hbm = e->hbm_stream;
  std::vector<Slice> out;
  const int base = e->B / n, rem = e->B % n;
  int b0 = 0;
  for (int i = 0; i < n; ++i) {
    const int nb = base + (i < rem ? 1 : 0);
    out.push_back(Slice{b0, nb, e->micro_streams[i]});
    b0 += nb;
  }
  return out;
}

void fork_slices(lram_engine* e, const std::vector<Slice>& sl, hipStream_t hbm, hipStream_t s) {
  for (const Slice& x : sl) stream_after(e, x.s, s, true);
  stream_after(e, hbm, s, true);
}
void join_slices(lram_engine* e, const std::vector<Slice>& sl, hipStream_t hbm, hipStream_t s) {
  for (const Slice& x : sl) stream_after(e, s, x.s, true);
  stream_after(e, s, hbm, true);
}

// ---- mLSTM block, split at the cell kernel -----------------------------------------------------------
// proj_up in two halves pays from 2048 env slots (measured at 16M: 4096 slots 370k -> 374k env-steps/s, 1024 slots 292k
// -> 287k, 32 slots 45.4k -> 40.0k: below that the extra launch costs more than the shorter critical path gives)
// lean front end: the lazy read pass of the fused-score geometries rebuilds q, k, v itself (no q / k / v round trip through HBM)
bool lean_front(const lram_engine* e, int T) {
  return lazy_active(e, T) && mlstm_lazy_fused_scores(e->cfg.inner / e->cfg.n_heads);
}

bool split_up_now(const lram_engine* e) { return e->B >= 2048 && !e->graph_mode; }

// Output group norm + learnable skip inside the lazy read pass's epilogue (its workgroup holds a head's whole output
// row), silu(z) written by proj_up's epilogue and multiplied onto proj_down's operand while that GEMM stages it: no
// group-norm launch on the slice's chain, no [rows, inner] round trip for h.
// (Measured and removed, profiles/EXPERIMENTS.md: the output gate and proj_down's row maxima in that epilogue too -- the
// row-maximum launches and 0.7 GB of chain reads went, the step got 1.8 % SLOWER because the read pass, the critical queue,
// got 14 us longer; proj_down's row scales from a Cauchy-Schwarz bound instead of a row-maximum launch: -1.5 %; the gated
// operand pre-split by a row kernel: -0.5 %.)
bool gn_fused(const lram_engine* e, int T) {
  const int dh = e->cfg.inner / e->cfg.n_heads;
  return (e->gn_fuse == 1 || (e->gn_fuse == 2 && e->B >= 2048)) && lean_front(e, T) && e->use_bf16x3 && (dh == 256 || dh == 128) && T <= 4 &&
         e->cfg.inner % 8 == 0 && e->cfg.d_model % 8 == 0 && e->B >= 64;  // (fewer rows take the GEMV path)
}

void mlstm_front(lram_engine* e, int i, int T, const uint8_t* reset, const Slice& sl) {
  const lram_config& c = e->cfg;
  const int D = c.d_model, inner = c.inner, NH = c.n_heads, rows = sl.nb * T;
  const size_t r0 = (size_t)sl.b0 * T, b0 = sl.b0;
  const BlockWeights& w = e->bw[i];
  BlockState& st = e->st[i];
  float* amx = e->use_f16x2 ? e->AMX_XN.p + r0 : nullptr;  // the norm hands proj_up's operand row maxima over
  // proj_up in two halves: the x_m half feeds the conv / q / k / v front end and is on the block's critical path; the
  // z half is only needed by the output gate after the state pass and is issued beside it (mlstm_up_z)
  GemmArgs up;
  up.a = e->XN.p + r0 * D, up.lda = D, up.w = w.proj_up, up.ldw = D, up.c = e->U.p + r0 * e->ucols, up.ldc = 2 * inner;
  up.m = rows, up.n = split_up_now(e) ? inner : 2 * inner, up.k = D;
  if (gn_fused(e, T) && !split_up_now(e)) up.act_silu_from = inner;  // the z half is stored as silu(z)
  if (!split_up_now(e) && takes_skinny_with_norm(e, up)) {
    // few rows: the norm runs in the projection's prologue (each workgroup normalises its 32 rows in registers)
    up.a = e->X.p + r0 * D, up.norm_g = w.norm_g, up.norm_b = w.norm_b, up.norm_eps = c.ln_eps, up.norm_rms = c.norm_is_rms;
    launch_gemm_skinny(up, sl.s);
    count_gemm(e, 3, up);
  } else {
    // f16x2 with both operands pre-split: the norm writes the two operand planes (into XN's memory: 2 x 2 bytes per
    // element) and the rows' inverse scales (into AMX_XN) instead of fp32 + row maxima; both halves of proj_up read them
    const bool ps = presplit_for(e, w.proj_up, rows, inner, D);
    uint16_t* xn2 = reinterpret_cast<uint16_t*>(e->XN2.p) + r0 * 32;  // K-tile-major planes: [D / 32][B * T][32]
    const int64_t xn2_kt = ps ? (int64_t)(e->XN2.n / D) * 32 : 0;
    launch_row_norm(e->X.p + r0 * D, D, ps ? nullptr : e->XN.p + r0 * D, D, w.norm_g, w.norm_b, rows, D, c.ln_eps,
                    c.norm_is_rms, sl.s, nullptr, ps ? nullptr : amx, nullptr, ps ? xn2 : nullptr, (int64_t)e->XN2.n,
                    ps ? amx : nullptr, xn2_kt);
    up.a_amax = amx;
    if (ps) up.a = nullptr, up.a_amax = nullptr, up.a2 = xn2, up.a2_plane = (int64_t)e->XN2.n, up.a2_kt = xn2_kt, up.a2_inv = amx;
    if (e->lane_rec != nullptr) up.beside_memory_bound = 2;   // a chunk lane of lram_prefill
    gemm(e, up, sl.s);
  }
  if (e->front_multi && lean_front(e, T) && sl.nb >= e->front_min_envs && e->gate_coef[i].p != nullptr &&
      mlstm_front_supported(inner, NH, c.conv_k, T)) {
    // large launches of the lean path: several env slots per workgroup, weights in registers (mlstm_front.hip)
    MlstmFrontArgs fa;
    fa.u = e->U.p + r0 * e->ucols, fa.ldu = 2 * inner, fa.conv_state = st.conv.p + b0 * c.conv_k * inner;
    fa.n_state = st.n.p + b0 * inner, fa.m_state = st.m.p + b0 * NH;
    fa.conv_w = w.conv_w, fa.conv_b = w.conv_b, fa.wq = w.wq, fa.wk = w.wk, fa.gc = e->gate_coef[i].p, fa.bi = w.bi, fa.bf = w.bf;
    fa.xa = e->XA.p + r0 * e->icols, fa.scal = e->SCAL.p + r0 * NH * 4, fa.reset = reset ? reset + b0 : nullptr;
    fa.B = sl.nb, fa.T = T, fa.inner = inner, fa.NH = NH, fa.K = c.conv_k;
    launch_mlstm_front(fa, sl.s);
    return;
  }
  MlstmPreArgs pa;
  pa.u = e->U.p + r0 * e->ucols, pa.conv_state = st.conv.p + b0 * c.conv_k * inner, pa.n_state = st.n.p + b0 * inner;
  pa.m_state = st.m.p + b0 * NH;
  pa.conv_w = w.conv_w, pa.conv_b = w.conv_b, pa.wq = w.wq, pa.wk = w.wk, pa.wv = w.wv;
  pa.wi = w.wi, pa.bi = w.bi, pa.wf = w.wf, pa.bf = w.bf;
  pa.q = e->Q.p + r0 * e->icols, pa.k = e->K.p + r0 * e->icols, pa.v = e->V.p + r0 * e->icols;
  pa.xa = e->XA.p + r0 * e->icols;
  pa.scal = e->SCAL.p + r0 * NH * 4, pa.reset = reset ? reset + b0 : nullptr;
  pa.B = sl.nb, pa.T = T, pa.inner = inner, pa.NH = NH, pa.K = c.conv_k;
  pa.lean = lean_front(e, T) ? 1 : 0;
  if (T > kMaxTokens) {
    LRAM_REQUIRE(T <= e->tok_cap && e->AMAT.p != nullptr, "chunkwise prefill workspace not allocated");
    pa.gates = e->GATES.p + r0 * NH * 2;
    pa.amat = e->AMAT.p + b0 * NH * kChunkMaxTokens * kChunkMaxTokens;
    pa.vec = e->VEC.p + b0 * NH * 3 * kChunkMaxTokens;
  }
  launch_mlstm_pre(pa, sl.s);
}

void mlstm_up_z(lram_engine* e, int i, int T, const Slice& sl) {
  if (!split_up_now(e)) return;
  const lram_config& c = e->cfg;
  const int D = c.d_model, inner = c.inner, rows = sl.nb * T;
  const size_t r0 = (size_t)sl.b0 * T;
  GemmArgs up;
  up.a = e->XN.p + r0 * D, up.lda = D, up.w = e->bw[i].proj_up + (size_t)inner * D, up.ldw = D;
  up.c = e->U.p + r0 * e->ucols + inner, up.ldc = 2 * inner, up.m = rows, up.n = inner, up.k = D;
  if (e->use_f16x2) up.a_amax = e->AMX_XN.p + r0;  // written by this block's norm launch (mlstm_front)
  if (presplit_for(e, e->bw[i].proj_up, rows, inner, D)) {  // (same decision as mlstm_front: XN2 holds operand planes)
    up.a = nullptr, up.a_amax = nullptr;
    up.a2 = reinterpret_cast<uint16_t*>(e->XN2.p) + r0 * 32, up.a2_plane = (int64_t)e->XN2.n, up.a2_kt = (int64_t)(e->XN2.n / D) * 32;
    up.a2_inv = e->AMX_XN.p + r0;
  }
  if (gn_fused(e, T)) up.act_silu_from = 0;
  up.beside_memory_bound = e->upz_beside ? 1 : 0;   // (issued beside this slice's own state pass)
  gemm(e, up, sl.s);
}

void mlstm_cell(lram_engine* e, int i, int T, const uint8_t* reset, const Slice& sl, hipStream_t s) {
  const lram_config& c = e->cfg;
  const int inner = c.inner, NH = c.n_heads, DH = e->dh();
  const size_t r0 = (size_t)sl.b0 * T, b0 = sl.b0;
  MlstmCellArgs ca;
  ca.C = e->st[i].s0.p + b0 * NH * DH * DH, ca.q = e->Q.p + r0 * e->icols, ca.k = e->K.p + r0 * e->icols;
  ca.v = e->V.p + r0 * e->icols, ca.scal = e->SCAL.p + r0 * NH * 4, ca.h = e->H.p + r0 * e->icols;
  ca.reset = reset ? reset + b0 : nullptr, ca.B = sl.nb, ca.T = T, ca.NH = NH, ca.DH = DH;
  // Large launches: one cell workgroup per CU (84 KB of LDS each; a second one does not fit, two 37 KB GEMM
  // workgroups of the other slice do).  Measured on MI355X at B=4096/16M: 1.61 ms -> 1.47 ms per launch
  // (5.5 -> 6.0 TB/s) standalone; see DESIGN.md section 6.
  const long wgs = (long)sl.nb * NH * ((DH % 256 == 0) ? DH / 256 : (DH % 128 == 0) ? DH / 128 : DH / 64);
  ca.min_lds_bytes = wgs >= 1024 ? 84 * 1024 : 0;
  ca.unroll = e->cell_unroll;
  if (T > kMaxTokens) {
    ca.amat = e->AMAT.p + b0 * NH * kChunkMaxTokens * kChunkMaxTokens;
    ca.vec = e->VEC.p + b0 * NH * 3 * kChunkMaxTokens;
    ca.chunk_exact_fp32 = e->chunk_exact_fp32 ? 1 : 0;
  }
  prof_record(e, s, true);
  launch_mlstm_cell(ca, s);
  prof_record(e, s, false);
}

void mlstm_back(lram_engine* e, int i, int T, const Slice& sl) {
  const lram_config& c = e->cfg;
  const int D = c.d_model, inner = c.inner, NH = c.n_heads, DH = e->dh(), rows = sl.nb * T;
  const size_t r0 = (size_t)sl.b0 * T;
  const BlockWeights& w = e->bw[i];
  float* X = e->X.p + r0 * D;
  if (gn_fused(e, T)) {  // H holds GN(h) + skip * xa, U's z half silu(z)
    GemmArgs dn;
    dn.a = e->H.p + r0 * e->icols, dn.lda = inner, dn.w = w.proj_down, dn.ldw = inner, dn.c = X, dn.ldc = D, dn.residual = X;
    dn.m = rows, dn.n = D, dn.k = inner;
    dn.gate = e->U.p + r0 * e->ucols + inner, dn.ldg = 2 * inner;
    gemm(e, dn, sl.s);
    return;
  }
  GroupNormArgs ga;
  ga.h = e->H.p + r0 * e->icols, ga.gamma = w.on_g, ga.beta = w.on_b, ga.skip = w.skip, ga.xa = e->XA.p + r0 * e->icols;
  ga.u = e->U.p + r0 * e->ucols, ga.out = e->G.p + r0 * e->icols, ga.rows = rows, ga.NH = NH, ga.DH = DH, ga.mode = 0;
  ga.eps = c.ln_eps;
  // the norm's waves (one per row and head) hand proj_down's operand row maxima over as NH partial maxima per row: the
  // row_amax launch between the two (8-11 us on every block of a chain-bound slice's chain) goes
  // (LRAM_GN_AMAX=0, test switch: the standalone row-maximum launch instead; bit-identical by construction -- a maximum of
  // partial maxima is exact -- and tests/test_gpu_realbatch.py holds it to that)
  const bool hand_over = e->gn_amax_handover && e->AMX_H.p != nullptr && f16x2_rows(e, rows, D, inner);
  GemmArgs dn;
  dn.a = e->G.p + r0 * e->icols, dn.lda = inner, dn.w = w.proj_down, dn.ldw = inner, dn.c = X, dn.ldc = D, dn.residual = X;
  dn.m = rows, dn.n = D, dn.k = inner;
  // ... or (round 6; one env slice: stored contexts, small and mid-size batches) the norm writes proj_down's operand itself: the two
  // f16 planes of the row scaled by its maximum over all heads -- the same 4 bytes per element as the fp32 row, into G's memory --
  // and the projection runs on the pre-split kernel (LDS-DMA staging, no conversion in its loop: 15-28 % faster on every
  // down-projection shape alone, profiles/r06_gemm_durations.txt).  Bit-identical to the hand-over path.  Same box, hand-over vs
  // planes: C5's prefill 297.9 -> 292.0 ms, 206M at 64 envs 14.98k -> 15.18k env-steps/s; NOT inside the two-slice pipelines, where
  // the pre-split kernel's 48 KB workgroups wait for the other slice's read pass to leave a CU: 16M at 1024 slots 378.6k -> 368.8k,
  // 206M at 512 slots +-0.
  const int64_t bt = (int64_t)(e->G.n / e->icols);   // rows the workspace holds
  GemmArgs probe;
  const bool planes = hand_over && e->gn_planes && e->xlstm_slices_now == 1 && e->gemm_presplit && NH <= 8 && (inner & 31) == 0 && bt * inner * 4 < (1ll << 31) &&
                      f16x2_weight(e, w.proj_down, inner, &probe) && 4 * probe.w2_plane < (1ll << 31);
  if (planes) {
    ga.out = nullptr;
    ga.h2 = reinterpret_cast<uint16_t*>(e->G.p) + r0 * 32, ga.h2_plane = bt * inner, ga.h2_kt = bt * 32, ga.h2_inv = e->AMX_H.p + r0;
    dn.a = nullptr, dn.a2 = ga.h2, dn.a2_plane = ga.h2_plane, dn.a2_kt = ga.h2_kt, dn.a2_inv = ga.h2_inv;
  } else {
    ga.amax = hand_over ? e->AMX_H.p + r0 * NH : nullptr;
    if (hand_over) dn.a_amax = ga.amax, dn.amax_parts = NH;
  }
  launch_group_norm(ga, sl.s);
  gemm(e, dn, sl.s);
}

// Fills g4's operand tables for the four sLSTM gate projections as one bf16x3 launch; false where that kernel cannot serve it.
bool slstm_gates_one_bf16x3(const lram_engine* e, GemmArgs& g4, const BlockWeights& w, const float* XC, const float* XN, float* gates,
                            int Hs) {
  if (!e->use_bf16x3 || !e->slstm_gates_one) return false;
  int64_t plane = -1;
  for (int g = 0; g < 4; ++g) {
    auto it = e->split.find(w.gate_w[g]);
    if (it == e->split.end() || (plane >= 0 && (int64_t)it->second.n != plane)) return false;
    plane = (int64_t)it->second.n;
    g4.a_tab[g] = (g < 2) ? XC : XN, g4.w_tab[g] = w.gate_w[g], g4.c_tab[g] = gates + (int64_t)g * Hs;
    g4.w3_tab[g] = it->second.p;
  }
  g4.w3 = g4.w3_tab[0], g4.w3_plane = plane;
  return gemm_bf16x3_supported(g4) && !gemm_small_m(g4);
}

void slstm_block(lram_engine* e, int i, int T, const uint8_t* reset, const Slice& sl) {
  const lram_config& c = e->cfg;
  const int D = c.d_model, NH = c.n_heads, SDH = e->sdh(), F = c.ffn_dim, Hs = D, rows = sl.nb * T;
  const size_t r0 = (size_t)sl.b0 * T, b0 = sl.b0;
  const BlockWeights& w = e->bw[i];
  BlockState& st = e->st[i];
  hipStream_t s = sl.s;
  float* X = e->X.p + r0 * D;
  float* XN = e->XN.p + r0 * D;
  float* XC = e->Q.p + r0 * e->icols;          // silu(conv(xn))
  float* gates = e->U.p + r0 * e->ucols;  // [rows, 4, H]
  float* RY = e->RY.p + b0 * 4 * Hs;     // [nb, 4, H]
  float* Y = e->H.p + r0 * e->icols;          // [rows, H]
  float* Ubuf = e->U.p + r0 * e->ucols;
  float* Gbuf = e->G.p + r0 * e->icols;
  float* state = st.s0.p + b0 * Hs;      // [4, B, H] viewed from env b0 (leading-axis stride e->B * H)
  launch_row_norm(X, D, XN, D, w.norm_g, w.norm_b, rows, D, c.ln_eps, c.norm_is_rms, s);
  SlstmConvArgs sa;
  sa.xn = XN, sa.conv_state = st.conv.p + b0 * c.conv_k * D, sa.slstm_state = state, sa.conv_w = w.conv_w;
  sa.conv_b = w.conv_b, sa.xc = XC, sa.reset = reset ? reset + b0 : nullptr, sa.B = sl.nb, sa.T = T, sa.D = D;
  sa.K = c.conv_k, sa.state_B = e->B;
  launch_slstm_conv(sa, s);
  GemmArgs g4;  // few rows: the four gate projections (per-head blocks, i / f on the conv branch, z / o on the norm) as ONE launch
  g4.a = XC, g4.lda = D, g4.sA1 = SDH, g4.w = w.gate_w[0], g4.ldw = SDH, g4.sW1 = (int64_t)SDH * SDH;
  g4.c = gates, g4.ldc = 4 * Hs, g4.sC1 = SDH, g4.m = rows, g4.n = SDH, g4.k = SDH, g4.nb1 = NH, g4.nb2 = 4;
  // (head dim <= 128: up to 768 rows as well -- 16M at 256 envs +2.6 %; at 6144 rows -1.5 %, 206M's 320-wide heads at 768 rows -1 %)
  const bool gates_big = rows <= e->slstm_gates_rows && gemm_skinny_supported(g4) && g4.k <= 128;
  // (every table entry must meet the few-row kernel's 16-byte alignment, not only entry 0 that g4.a / g4.w stand for: a
  // misaligned later entry falls back to the four separate launches instead of failing inside launch_gemm_skinny)
  bool tab_aligned = true;
  for (int g = 0; g < 4; ++g)
    tab_aligned = tab_aligned && ((reinterpret_cast<uintptr_t>((g < 2) ? XC : XN) | reinterpret_cast<uintptr_t>(w.gate_w[g])) & 15) == 0;
  if (tab_aligned && (takes_skinny(e, g4) || gates_big)) {
    for (int g = 0; g < 4; ++g)
      g4.a_tab[g] = (g < 2) ? XC : XN, g4.w_tab[g] = w.gate_w[g], g4.c_tab[g] = gates + (int64_t)g * Hs;
    launch_gemm_skinny(g4, s);
    count_gemm(e, 3, g4);
  } else if (slstm_gates_one_bf16x3(e, g4, w, XC, XN, gates, Hs)) {
    // larger slices: the same ONE launch on the bf16x3 kernel (operand tables; every gate's tiles in one grid instead of four
    // short launches of 48-144 workgroups each on the slice's chain) -- bit-identical to the four launches
    launch_gemm_bf16x3(g4, s);
    count_gemm(e, 1, g4);
  } else {
    for (int g = 0; g < 4; ++g) {
      GemmArgs ga;
      ga.a = (g < 2) ? XC : XN, ga.lda = D, ga.sA1 = SDH;
      ga.w = w.gate_w[g], ga.ldw = SDH, ga.sW1 = (int64_t)SDH * SDH;
      ga.c = gates + (int64_t)g * Hs, ga.ldc = 4 * Hs, ga.sC1 = SDH;
      ga.m = rows, ga.n = SDH, ga.k = SDH, ga.nb1 = NH;
      gemm(e, ga, s);
    }
  }
  // few env rows (up to slstm_fused_rows): recurrent projection + pointwise cell as ONE lean launch per token instead of a
  // batched matrix-core GEMM (fixed latency of a 128-row tile) and the pointwise kernel
  // (measured, same box each: 16M 1 env +1.9 %, 8 +4.2 %, 12 +6.6 %, 32 +6.0 %, 128 +3.3 %, 512-env slices +1.6 %, 1024-env
  // slices +-0; 206M 16 envs +6.1 %, 64 +2.9 %, 256-env slices -1.4 %: the row limit scales with 128 / head dim)
  const bool tok_fused = e->slstm_fused_rows > 0 &&
                         (int64_t)sl.nb * std::max(SDH, 128) <= (int64_t)e->slstm_fused_rows * 128 && slstm_token_supported(Hs, NH);
  for (int t = 0; tok_fused && t < T; ++t) {
    SlstmTokenArgs ta;
    ta.gates = gates, ta.rt = w.rt, ta.bias = w.rbias, ta.state = state, ta.yout = Y;
    ta.hprev = t == 0 ? state : Y + (int64_t)(t - 1) * Hs, ta.hprev_ld = t == 0 ? Hs : (int64_t)T * Hs;
    ta.B = sl.nb, ta.T = T, ta.t = t, ta.H = Hs, ta.NH = NH, ta.state_B = e->B, ta.write_h = (t == T - 1 && t > 0) ? 1 : 0;
    launch_slstm_token(ta, s);
  }
  if (tok_fused && T == 1)  // the single launch read the state's h plane: it is refreshed from the output rows afterwards
    LRAM_HIP_CHECK(hipMemcpyAsync(state, Y, (size_t)sl.nb * Hs * sizeof(float), hipMemcpyDeviceToDevice, s));
  // slices beyond the token kernel's: the whole step's recurrence as ONE launch (head dim 128; slstm_seq.hip)
  const bool seq = !tok_fused && e->slstm_seq && e->slstm_rt2[i].p != nullptr && slstm_seq_supported(Hs, NH, T);
  if (seq) {
    SlstmSeqArgs qa;
    qa.gates = gates, qa.bias = w.rbias, qa.state = state, qa.yout = Y;
    if (e->slstm_rinv[i].p != nullptr)
      qa.rt2h = reinterpret_cast<const uint16_t*>(e->slstm_rt2[i].p), qa.rinv = e->slstm_rinv[i].p;
    else
      qa.rt2 = e->slstm_rt2[i].p;
    qa.B = sl.nb, qa.T = T, qa.H = Hs, qa.NH = NH, qa.state_B = e->B;
    launch_slstm_seq(qa, s);
  }
  for (int t = 0; !tok_fused && !seq && t < T; ++t) {
    GemmArgs ra;
    ra.a = state, ra.lda = Hs, ra.sA1 = SDH, ra.sA2 = 0;
    ra.w = w.rt, ra.ldw = SDH, ra.sW1 = 4 * (int64_t)SDH * SDH, ra.sW2 = (int64_t)SDH * SDH;
    ra.c = RY, ra.ldc = 4 * Hs, ra.sC1 = SDH, ra.sC2 = Hs;
    ra.m = sl.nb, ra.n = SDH, ra.k = SDH, ra.nb1 = NH, ra.nb2 = 4;
    gemm(e, ra, s);
    SlstmPointwiseArgs pw;
    pw.gates = gates, pw.ry = RY, pw.bias = w.rbias, pw.state = state, pw.yout = Y;
    pw.B = sl.nb, pw.T = T, pw.t = t, pw.H = Hs, pw.state_B = e->B;
    launch_slstm_pointwise(pw, s);
  }
  GroupNormArgs gn;
  gn.h = Y, gn.gamma = w.gn_g, gn.beta = w.gn_b, gn.out = X, gn.rows = rows, gn.NH = NH, gn.DH = SDH;
  gn.mode = 1, gn.eps = c.ln_eps, gn.skip = nullptr, gn.xa = nullptr, gn.u = nullptr;
  launch_group_norm(gn, s);
  float* amx = e->use_f16x2 ? e->AMX_XN.p + r0 : nullptr;
  GemmArgs up;
  up.a = XN, up.lda = D, up.w = w.ffn_up, up.ldw = D, up.c = Ubuf, up.ldc = 2 * F;
  up.m = rows, up.n = 2 * F, up.k = D;
  if (takes_skinny_with_norm(e, up)) {  // few rows: the FFN's norm inside the projection's prologue
    up.a = X, up.norm_g = w.ffn_norm_g, up.norm_b = w.ffn_norm_b, up.norm_eps = c.ln_eps, up.norm_rms = c.norm_is_rms;
    launch_gemm_skinny(up, s);
    count_gemm(e, 3, up);
  } else {
    launch_row_norm(X, D, XN, D, w.ffn_norm_g, w.ffn_norm_b, rows, D, c.ln_eps, c.norm_is_rms, s, nullptr, amx);
    up.a_amax = amx;
    gemm(e, up, s);
  }
  launch_gelu_gate(Ubuf, Gbuf, rows, F, s);
  GemmArgs dn;
  dn.a = Gbuf, dn.lda = F, dn.w = w.ffn_down, dn.ldw = F, dn.c = X, dn.ldc = D, dn.residual = X;
  dn.m = rows, dn.n = D, dn.k = F;
  gemm(e, dn, s);
}

// Block stack on X [B*T, D] (in-place residual stream) -> HID.  With more than one slice the HBM-bound cell
// kernels of all slices are serialised on `hbm` while each slice's projections / norms run on its own stream:
// while slice A's matrix memory streams through HBM, slice B's fp32-MFMA GEMMs use the otherwise idle matrix
// cores (and vice versa one half-layer later).
void run_xlstm_stack(lram_engine* e, int T, const uint8_t* reset, const std::vector<Slice>& sl, hipStream_t hbm) {
  const lram_config& c = e->cfg;
  const int D = c.d_model;
  const bool lazy = lazy_active(e, T);
  e->xlstm_slices_now = (int)sl.size();
  if (lazy) {
    // Upper bound of pending tokens per fold class (env index mod period), tracked on the host: while no class can
    // overflow its window before its turn, the fold launch only covers the envs whose turn it is.
    const int P = e->lazy_period;
    e->lazy_compact = true;
    if ((int)e->lazy_bound.size() != P) {
      e->lazy_bound.assign(P, kLazyWindow);
      e->lazy_compact = false;
    }
    const int c_due = (P - (int)(e->lazy_step % P)) % P;
    for (int cls = 0; cls < P; ++cls) {
      if (cls == c_due)
        e->lazy_bound[cls] = 0;
      else if (e->lazy_bound[cls] + T > kLazyWindow)
        e->lazy_compact = false;
      e->lazy_bound[cls] = std::min(e->lazy_bound[cls] + T, 4 * kLazyWindow);
    }
  }
  // This step's folds depend on nothing this step computes (window rows, coefficients and counts are last step's).
  // Two slices: they go onto the state-pass stream itself, into the two stretches of a step where that stream has nothing to
  // run -- fold_bubbles of them before the first read pass (the step's front end and block 0's projections are still under
  // way), the rest while both slices are inside an sLSTM block -- instead of beside the read passes, which they slow down.
  // One slice (everything on the caller's stream): fold(i) right ahead of block i.
  // (Measured and removed, profiles/EXPERIMENTS.md: folds on their own stream one block ahead of the cells, every fold queued
  // at the step start, folds fused with the readout of the envs they rewrite, gaps / staggered front ends.)
  const bool bubbles = lazy && sl.size() > 1;
  // One slice (everything else on the caller's stream): ALL of the step's folds go to a side stream at the step's start -- they
  // depend on nothing this step computes -- and the read pass of block i waits for fold i alone, instead of every fold sitting
  // on the one stream ahead of its block (206M at 64 envs: 17 folds of ~21 us each = 8 % of the step).
  // From 256 MiB of matrix memory per block (16M: 256 envs, 206M: 41); below, the extra stream's events cost more than the folds.
  // Same box, folds on the one stream vs on the side stream, env-steps/s: 206M at 32 / 64 envs 10.56k vs 10.56k / 15.18k vs 15.67k;
  // 16M at 128 / 256 / 448 envs 155.6k vs 149.5k / 224.2k vs 226.7k / 287.4k vs 297.2k.
  const double dh_ = c.n_heads > 0 ? (double)c.inner / c.n_heads : 0.0;
  // (Only where the ONE slice is the automatic choice: a forced single slice -- lram_set_micro_batches(1), bench.py's "chip to
  // itself" measurement of the state pass -- keeps every kernel of the pass alone on the chip.)
  const bool side_folds = lazy && sl.size() == 1 && e->n_micro == 0 && (double)e->B * c.n_heads * dh_ * dh_ * 4.0 >= 256.0 * 1024 * 1024;
  hipStream_t fold_stream = hbm;
  std::vector<hipEvent_t> fold_done(side_folds ? c.n_blocks : 0, nullptr);
  if (side_folds) {
    if (!e->hbm_stream) LRAM_HIP_CHECK(hipStreamCreateWithFlags(&e->hbm_stream, hipStreamNonBlocking));
    fold_stream = e->hbm_stream;
    stream_after(e, fold_stream, sl[0].s);
  }
  std::vector<char> folded(c.n_blocks, 0);
  auto launch_folds = [&](int i) {  // one launch per block over all env slots: folds do not care about the slices
    MlstmLazyArgs la = lazy_args(e, i, T, reset, 0, e->B);
    la.compact = e->lazy_compact ? 1 : 0;
    prof_record(e, fold_stream, true, true);
    launch_mlstm_lazy_fold(la, fold_stream);
    prof_record(e, fold_stream, false, true);
    folded[i] = 1;
  };
  auto next_mlstm = [&](int i) {
    for (int k = i + 1; k < c.n_blocks; ++k)
      if (!c.block_is_slstm[k]) return k;
    return -1;
  };
  if (bubbles) {
    int k = 0;
    const int ahead = e->step_images != nullptr ? e->fold_bubbles_images : lram_engine::fold_bubbles;
    for (int i = next_mlstm(-1); i >= 0 && k < ahead; i = next_mlstm(i), ++k) launch_folds(i);
  }
  if (side_folds)
    for (int i = next_mlstm(-1); i >= 0; i = next_mlstm(i)) {
      launch_folds(i);
      fold_done[i] = ring_event(e);
      LRAM_HIP_CHECK(hipEventRecord(fold_done[i], fold_stream));
    }
  for (int i = 0; i < c.n_blocks; ++i) {
    if (i > 0 && e->lane_rec) LRAM_HIP_CHECK(hipEventRecord((*e->lane_rec)[i - 1], sl[0].s));   // (chunk lanes: one slice, one stream)
    if (e->lane_wait) LRAM_HIP_CHECK(hipStreamWaitEvent(sl[0].s, (*e->lane_wait)[i], 0));
    if (c.block_is_slstm[i]) {
      // (enqueued BEFORE the sLSTM block's ~50 launches: with short kernels the host is only just ahead of the device
      // there, and folds queued behind them reached the state-pass stream 0.26 ms after it had gone idle -- 206M, 512 slots)
      if (bubbles) {
        // the folds still outstanding run behind the previous block's read passes, shared out over this and the later sLSTM
        // blocks of the stack (206M: three stretches, five folds each, instead of fifteen in the first and none in the
        // other two); at least the blocks whose read passes come before the next sLSTM block
        int left = 0, stretches = 0, must = 0;
        for (int k = next_mlstm(i); k >= 0; k = next_mlstm(k)) left += folded[k] ? 0 : 1;
        for (int k = i; k < c.n_blocks; ++k) stretches += c.block_is_slstm[k] ? 1 : 0;
        for (int k = i + 1; k < c.n_blocks && !c.block_is_slstm[k]; ++k) must += folded[k] ? 0 : 1;
        int take = left;
        if (stretches > 1) take = std::max((take + stretches - 1) / stretches, std::min(must, take));
        for (int k = next_mlstm(i); k >= 0 && take > 0; k = next_mlstm(k))
          if (!folded[k]) launch_folds(k), --take;
      }
      for (const Slice& x : sl) slstm_block(e, i, T, reset, x);
      continue;
    }
    if (lazy && !folded[i]) launch_folds(i);  // (one slice, or a stack without an sLSTM block: the fold ahead of its read passes)
    for (const Slice& x : sl) {
      mlstm_front(e, i, T, reset, x);
      if (lazy) {
        // lazy matrix memory: on the HBM stream the read-only pass with the window scores, the window attention and the
        // step's bookkeeping
        MlstmLazyArgs la = lazy_args(e, i, T, reset, x.b0, x.nb);
        const size_t r0 = (size_t)x.b0 * T;
        la.q = e->Q.p + r0 * e->icols, la.k = e->K.p + r0 * e->icols, la.v = e->V.p + r0 * e->icols;
        la.scal = e->SCAL.p + r0 * c.n_heads * 4, la.h = e->H.p + r0 * e->icols;
        // the read-only pass's occupancy cap (LDS per workgroup; 0 = the launcher's default of three workgroups per CU, 41 KB,
        // at 256-wide heads).  Slices below ~900 envs are CHAIN-bound -- the slice's projections / front end take longer than the
        // other slice's read pass -- and two read-pass workgroups per CU (54 KB) leave room for the two-stage projection
        // workgroups (155 VGPRs, 48 KB) to start beside them: 16M at 640 / 768 / 896 / 1024 / 1152 / 1280 / 1408 slots +1.5 / +2.6 /
        // +4.0 / +2.5 / +4.6 / +3.8 / +1.1 %, 1536-1792 +0.3-1 %, 2048 -1.3 %, 4096 -1.3 % (profiles/r05_ab_read_pass_lds_cap.txt)
        la.min_lds_bytes = (sl.size() >= 2 && x.nb <= e->lazy_cap2_envs && la.DH == 256) ? 54 * 1024 : 0;
        if (!mlstm_lazy_fused_scores(la.DH)) launch_mlstm_lazy_book(la, x.s);  // scores beside the front end
        if (lean_front(e, T)) {
          const BlockWeights& w = e->bw[i];
          la.lean_xa = e->XA.p + r0 * e->icols, la.lean_u = e->U.p + r0 * e->ucols;
          la.lean_wq = w.wq, la.lean_wk = w.wk, la.lean_wv = w.wv;
          if (gn_fused(e, T)) la.gn_g = w.on_g, la.gn_b = w.on_b, la.gn_skip = w.skip, la.gn_eps = c.ln_eps;
        }
        stream_after(e, hbm, x.s);
        if (side_folds) LRAM_HIP_CHECK(hipStreamWaitEvent(hbm, fold_done[i], 0));
        prof_record(e, hbm, true);
        launch_mlstm_lazy_cell(la, hbm);
        prof_record(e, hbm, false);
        mlstm_up_z(e, i, T, x);  // on the slice's stream, beside its own state pass
        stream_after(e, x.s, hbm);
        continue;
      }
      stream_after(e, hbm, x.s);
      mlstm_cell(e, i, T, reset, x, hbm);
      mlstm_up_z(e, i, T, x);
      stream_after(e, x.s, hbm);
    }
    for (const Slice& x : sl) mlstm_back(e, i, T, x);
  }
  if (e->lane_rec) LRAM_HIP_CHECK(hipEventRecord((*e->lane_rec)[c.n_blocks - 1], sl[0].s));
  if (lazy) {
    ++e->lazy_step;
    e->lazy_dirty = true;
  }
  for (const Slice& x : sl) {
    const size_t r0 = (size_t)x.b0 * T;
    launch_row_norm(e->X.p + r0 * D, D, e->HID.p + r0 * D, D, e->post_g, e->post_b, x.nb * T, D, c.ln_eps,
                    c.norm_is_rms, x.s);
  }
}

// ---- Mamba block, cut at its projections ---------------------------------------------------------------
// stage 0: add + RMSNorm | in_proj      stage 1: conv | x_proj, dt_proj      stage 2: selective state update | out_proj
void mamba_stage(lram_engine* e, int i, int stage, int T, const uint8_t* reset, const Slice& sl) {
  const lram_config& c = e->cfg;
  const int D = c.d_model, di = c.d_inner, N = c.d_state, R = c.dt_rank, rows = sl.nb * T, ldx = R + 2 * N;
  const size_t r0 = (size_t)sl.b0 * T, b0 = sl.b0;
  const BlockWeights& w = e->bw[i];
  BlockState& st = e->st[i];
  float* X = e->X.p + r0 * D;
  float* RES = e->RES.p + r0 * D;
  float* XN = e->XN.p + r0 * D;
  // shared repeated forwards (step_launches): layer 0's residual input and in_proj output are the same in every pass
  const bool share = e->compat_passes > 1;
  if (share && i == 0 && stage == 0 && e->compat_pass > 0) return;
  float* U = (share && i == 0) ? e->U0.p + r0 * 2 * di : e->U.p + r0 * 2 * di;
  float* RES_out = (share && i == 0) ? e->X0.p + r0 * D : RES;   // layer 0: RES = the embedded tokens, kept in X0
  const float* RES_in = i == 0 ? nullptr : ((share && i == 1) ? e->X0.p + r0 * D : RES);
  float* XA = e->XA.p + r0 * di;
  float* Q = e->Q.p + r0 * ldx;
  float* DTP = e->DTP.p + r0 * di;
  float* H = e->H.p + r0 * di;
  // compat_stale (reference InferenceParams.reset(), decision_mamba.py:20-25 + models/decision_mamba.py:130-149):
  // only layer 0 starts the episode from an empty state
  const uint8_t* rs = (reset && !(e->compat_stale && i > 0)) ? reset + b0 : nullptr;
  hipStream_t gs = sl.s;
  // f16x2 projections: the kernels that produce their operands hand the row maxima over -- the norm writes XN's (one
  // wave per row), the conv and the state-update kernels one partial maximum per wave (d_inner / 64 per row, plain
  // stores; the GEMM's prologue takes their maximum).  Atomic maxima were measured first: +20 us on the conv launch,
  // +13 us on the state update (147k single-lane atomics per launch), as much as the row-maximum launches they replaced.
  // dt_proj (K = dt_rank) inside the state-update kernel instead of a GEMM launch + its [rows, d_inner] round trip
  const bool dt_fused = e->mamba_dt_fuse && mamba_ssm_dt_fusable(N, R) && e->dt_wt[i].p != nullptr;
  const bool amx = e->use_f16x2 && di % 64 == 0 && N == 16 && T <= 4;
  const int parts = di / 64;
  float* amx_xn = amx ? e->AMX_XN.p + r0 : nullptr;
  float* amx_xa = amx ? e->AMX_XA.p + r0 * parts : nullptr;
  float* amx_h = amx ? e->AMX_H.p + r0 * parts : nullptr;
  // in_proj with both operands pre-split: the norm writes XN as two f16 planes + inverse row scales (see mlstm_front)
  const bool ps_in = amx && presplit_for(e, w.in_proj, rows, 2 * di, D);
  uint16_t* xn2 = reinterpret_cast<uint16_t*>(e->XN2.p) + r0 * 32;  // K-tile-major planes: [D / 32][B * T][32]
  const int64_t xn2_kt = ps_in ? (int64_t)(e->XN2.n / D) * 32 : 0;
  if (stage == 0) {
    launch_add_rms_norm(X, RES_in, RES_out, ps_in ? nullptr : XN, w.norm_g, rows, D, c.norm_eps, sl.s, ps_in ? nullptr : amx_xn,
                        ps_in ? xn2 : nullptr, (int64_t)e->XN2.n, ps_in ? amx_xn : nullptr, xn2_kt);
  } else if (stage == 1) {
    MambaConvArgs ca;
    ca.xz = U, ca.conv_state = st.conv.p + b0 * di * c.d_conv, ca.conv_w = w.conv_w, ca.conv_b = w.conv_b, ca.xc = XA;
    // (x_proj's operand row maxima are not needed where it runs in the exact-fp32 form of the narrow-output kernel)
    const bool xp_narrow32 = e->gemm_narrow_on && rows >= e->gemm_narrow_min_rows && e->narrow.count(w.x_proj) != 0 &&
                             !(e->use_f16x2 && e->gemm_narrow_f16);
    ca.reset = rs, ca.B = sl.nb, ca.T = T, ca.d_inner = di, ca.K = c.d_conv, ca.amax = xp_narrow32 ? nullptr : amx_xa;
    launch_mamba_conv(ca, sl.s);
  } else {
    MambaSsmArgs sa;
    sa.ssm_state = st.s0.p + b0 * di * N, sa.xc = XA, sa.dtp = DTP, sa.dt_bias = w.dt_bias, sa.xdb = Q;
    sa.A_log = w.A_log, sa.Dp = w.Dp, sa.xz = U, sa.y = H, sa.reset = rs;
    sa.B = sl.nb, sa.T = T, sa.d_inner = di, sa.N = N, sa.R = R, sa.amax = amx_h;
    if (dt_fused) sa.dt_wt = e->dt_wt[i].p, sa.dtp = nullptr;
    prof_record(e, sl.s, true);
    launch_mamba_ssm(sa, sl.s);
    prof_record(e, sl.s, false);
  }
  if (stage == 0) {
    GemmArgs in;
    in.a = XN, in.lda = D, in.w = w.in_proj, in.ldw = D, in.c = U, in.ldc = 2 * di, in.bias = w.in_proj_b;
    in.m = rows, in.n = 2 * di, in.k = D, in.a_amax = amx_xn;
    if (ps_in) in.a = nullptr, in.a_amax = nullptr, in.a2 = xn2, in.a2_plane = (int64_t)e->XN2.n, in.a2_kt = xn2_kt, in.a2_inv = amx_xn;
    in.beside_memory_bound = (e->mamba_slices_now > 1 || e->lane_rec != nullptr) ? 1 : 0;   // (the other slice's conv / state update / norm run beside it)
    gemm(e, in, gs);
  } else if (stage == 1) {
    GemmArgs xp;
    xp.a = XA, xp.lda = di, xp.w = w.x_proj, xp.ldw = di, xp.c = Q, xp.ldc = ldx;
    xp.m = rows, xp.n = ldx, xp.k = di, xp.a_amax = amx_xa, xp.amax_parts = amx ? parts : 1;
    gemm(e, xp, gs);
    if (!dt_fused) {
      GemmArgs dp;
      dp.a = Q, dp.lda = ldx, dp.w = w.dt_proj, dp.ldw = R, dp.c = DTP, dp.ldc = di;
      dp.m = rows, dp.n = di, dp.k = R;
      gemm(e, dp, gs);
    }
  } else {
    GemmArgs op;
    op.a = H, op.lda = di, op.w = w.out_proj, op.ldw = di, op.c = X, op.ldc = D, op.bias = w.out_proj_b;
    op.m = rows, op.n = D, op.k = di, op.a_amax = amx_h, op.amax_parts = amx ? parts : 1;
    gemm(e, op, gs);
  }
}

// Mamba is projection-bound.  With two env slices on their own streams the memory-bound kernels
// of one slice (norm, conv, the selective state update) overlap the projections of the other; slice 1 is enqueued
// one stage behind slice 0 so the two do not start in lockstep.  No cross-stream events between fork and join:
// serialising the projections on a third stream costs more in event hand-offs than it gains (measured: 250k vs
// 298k single-stream vs 320k free-running env-steps/s at B = 2048, Mamba-48M).
void run_mamba_stack(lram_engine* e, int T, const uint8_t* reset, const std::vector<Slice>& sl, hipStream_t) {
  const lram_config& c = e->cfg;
  const int D = c.d_model;
  const int n_stages = 3 * c.n_blocks;
  const int ns = (int)sl.size();
  e->mamba_slices_now = ns;
  for (int k = 0; k < n_stages + ns - 1; ++k)   // slice j is enqueued j stages behind slice 0
    for (int j = 0; j < ns; ++j)
      if (k - j >= 0 && k - j < n_stages) {
        const int layer = (k - j) / 3, stage = (k - j) % 3;
        // chunk lanes of lram_prefill (one slice): layer i of this chunk after layer i of the chunk before it (conv + SSM state)
        if (stage == 0 && e->lane_wait) LRAM_HIP_CHECK(hipStreamWaitEvent(sl[j].s, (*e->lane_wait)[layer], 0));
        mamba_stage(e, layer, stage, T, reset, sl[j]);
        if (stage == 2 && e->lane_rec) LRAM_HIP_CHECK(hipEventRecord((*e->lane_rec)[layer], sl[j].s));
      }
  for (const Slice& x : sl) {
    const size_t r0 = (size_t)x.b0 * T;
    launch_add_rms_norm(e->X.p + r0 * D, e->RES.p + r0 * D, nullptr, e->HID.p + r0 * D, e->post_g, x.nb * T, D,
                        c.norm_eps, x.s);
  }
}

void run_stack(lram_engine* e, int T, const uint8_t* reset, const std::vector<Slice>& sl, hipStream_t hbm) {
  if (e->cfg.backbone == LRAM_BACKBONE_MAMBA)
    run_mamba_stack(e, T, reset, sl, hbm);
  else
    run_xlstm_stack(e, T, reset, sl, hbm);
}

// uint8 frames [B, C, H, W] -> state-token embeddings [B, d_model] (reference: embed_image(x / 255),
// online_decision_transformer_model.py:523-526 + image_encoders.py:58-66)
// Image work buffers for B frames of H x W (synchronises when it has to grow them: never called between a fork and a join)
void image_buffers(lram_engine* e, int H, int W) {
  const size_t B = e->B, px = B * H * W;
  if (px <= e->img_cap) return;
  LRAM_HIP_CHECK(hipDeviceSynchronize());
  const size_t hp = (H - 1) / 2 + 1, wp = (W - 1) / 2 + 1;
  e->IMG_P.alloc(B * 16 * H * W);       // stage-1 conv output before its pool (the largest tensor)
  e->IMG_X0.alloc(B * 32 * hp * wp);    // pooled maps never exceed 32 channels at half resolution
  e->IMG_X1.alloc(B * 32 * hp * wp);
  e->IMG_T.alloc(B * 32 * hp * wp);
  e->img_cap = px;
}

// envs b0 .. b0 + nb - 1 (`images` / `out` point at env b0's frame / row; every env slice keeps to its own fixed region of the
// work buffers, so slices at different stages of the CNN never touch each other's maps)
void embed_images(lram_engine* e, const uint8_t* images, int C, int H, int W, float* out, hipStream_t s, int b0 = 0, int nb = -1) {
  LRAM_REQUIRE(e->img_lin_w != nullptr, "lram_embed_images: no embed_image.* weights were uploaded");
  LRAM_REQUIRE(C == e->img_channels, "lram_embed_images: channel count does not match embed_image.cnn.0.conv.weight");
  const int B = nb < 0 ? e->B : nb, D = e->cfg.d_model;
  int h = H, w = W;
  for (int k = 0; k < 3; ++k) h = (h - 1) / 2 + 1, w = (w - 1) / 2 + 1;
  LRAM_REQUIRE(32 * h * w == e->img_flat, "lram_embed_images: image size does not match embed_image.linear.0.weight");
  LRAM_REQUIRE((size_t)e->B * H * W <= e->img_cap, "image work buffers not allocated");
  const size_t hp0 = (H - 1) / 2 + 1, wp0 = (W - 1) / 2 + 1;
  float* const P = e->IMG_P.p + (size_t)b0 * 16 * H * W;
  float* const X0 = e->IMG_X0.p + (size_t)b0 * 32 * hp0 * wp0;
  float* const X1 = e->IMG_X1.p + (size_t)b0 * 32 * hp0 * wp0;
  float* const Tb = e->IMG_T.p + (size_t)b0 * 32 * hp0 * wp0;
  const void* in = images;
  int in_u8 = 1;
  h = H, w = W;
  for (int sidx = 0; sidx < 3; ++sidx) {
    const lram_engine::ImgConv* cv = e->img_conv[sidx];
    auto conv = [&](const lram_engine::ImgConv& c, const void* src, int u8, int relu_in, const float* res, float* dst,
                    int relu_out) {
      Conv3x3Args a;
      a.in = src, a.w = c.w, a.bias = c.b, a.residual = res, a.out = dst;
      a.B = B, a.CIN = c.cin, a.COUT = c.cout, a.H = h, a.W = w, a.in_relu = relu_in, a.out_relu = relu_out, a.in_u8 = u8;
      launch_conv3x3(a, s);
    };
    conv(cv[0], in, in_u8, 0, nullptr, P, 0);
    launch_maxpool3s2(P, X0, (int64_t)B * cv[0].cout, h, w, s);
    h = (h - 1) / 2 + 1, w = (w - 1) / 2 + 1;
    conv(cv[1], X0, 0, 1, nullptr, Tb, 0);
    conv(cv[2], Tb, 0, 1, X0, X1, 0);
    conv(cv[3], X1, 0, 1, nullptr, Tb, 0);
    conv(cv[4], Tb, 0, 1, X1, X0, sidx == 2 ? 1 : 0);  // act_flatten's ReLU on the last map
    in = X0;
    in_u8 = 0;
  }
  // stage s > 0 reads X0 and writes P, then pools back into X0: no aliasing within a launch
  GemmArgs g;
  g.a = X0, g.lda = e->img_flat, g.w = e->img_lin_w, g.ldw = e->img_flat, g.c = out, g.ldc = D;
  g.bias = e->img_lin_b, g.m = B, g.n = D, g.k = e->img_flat;
  gemm(e, g, s);
  launch_relu(out, (int64_t)B * D, s);
}

// L consecutive timesteps for every env slot (L = 1: one env-step).  Inputs are [B, L, .] / [B, L] row-major; the
// reset mask applies before the first timestep; the action head runs on the last timestep only (and only if an
// output buffer is given).  One fork / join of the slice streams brackets the whole call.
void timesteps_launches(lram_engine* e, const float* obs, int emb, const float* rtg, const float* rew, int L,
                        const uint8_t* reset, int discrete, float* actions, int32_t* tokens, hipStream_t s,
                        int col_begin = 0, int shared_passes = 0, int fork_join = 3) {
  const lram_config& c = e->cfg;
  const int D = c.d_model, T = c.tokens_per_step;
  const int64_t obs_w = emb ? D : c.state_dim;
  e->sync_used = 0, e->edge_used = 0;
  // Stored context is consumed in chunks: every block then reads and writes its recurrent state once per chunk
  // instead of once per timestep.  Up to 4 timesteps (12 tokens) per chunk through the token-sequential kernels,
  // up to 21 (63 tokens) through the chunkwise matrix-core kernels (mlstm_chunk.hip).
  const int kChunk = L > 1 ? prefill_chunk_steps(e, L) : 1;
  if (L > 1 || !lazy_active(e, T)) lazy_materialize(e, s);  // stored contexts go through the materialised kernels
  // Stored contexts: the state embeddings of ALL timesteps as one GEMM ahead of the chunks (rows b * L + l, as the input lies),
  // instead of one few-row GEMM per timestep (206M, 64 envs x 512 timesteps: 1024 launches of 12-26 us -> 1 + one per chunk)
  const float* seq_emb = nullptr;
  if (L > 1 && T == 3 && D % 4 == 0 && shared_passes <= 1) {
    if (emb) {
      seq_emb = obs;
    } else if ((size_t)e->B * L * D <= ((size_t)1 << 29)) {   // <= 2 GiB
      if (e->SEQ_EMB.n < (size_t)e->B * L * D) {
        LRAM_HIP_CHECK(hipDeviceSynchronize());
        e->SEQ_EMB.alloc((size_t)e->B * L * D);
      }
      GemmArgs ge;
      ge.a = obs, ge.lda = c.state_dim, ge.w = e->w_state, ge.ldw = c.state_dim, ge.c = e->SEQ_EMB.p, ge.ldc = D;
      ge.bias = e->b_state, ge.m = e->B * L, ge.n = D, ge.k = c.state_dim;
      gemm(e, ge, s);
      seq_emb = e->SEQ_EMB.p;
    }
  }
  // chunk lanes (see lram_engine::chunk_lanes): the last chunk -- the one the action head reads -- is on lane 0 = the primary
  // workspace and the caller's stream.  Where they apply they replace the automatic env slices of large batches as well: whole-batch
  // launches, three chunks in flight (16M, 1024 envs x 252 timesteps: 224.4 -> 215.5 ms; 206M, 512 envs x 63: 295.3 -> 274.0 ms).
  const int n_chunks = (L + kChunk - 1) / kChunk;
  // (Mamba's stored contexts and the xLSTM geometries without a chunkwise form go through the token-sequential kernels in chunks
  // of 4 timesteps: the lanes apply to them as they are)
  const bool lanes = e->n_micro <= 1 && n_chunks >= 2 && e->chunk_lanes && !e->graph_mode && shared_passes <= 1 && twin_ready(e);
  hipStream_t hbm = s;
  const std::vector<Slice> sl = lanes ? std::vector<Slice>{Slice{0, e->B, s}} : make_slices(e, s, &hbm);
  const bool multi = sl.size() > 1;
  if (multi && (fork_join & 1)) fork_slices(e, sl, hbm, s);   // (repeated forwards: one fork ahead of the first, one join behind the last)
  const int NL = lanes ? e->n_lanes : 1;
  hipStream_t lane_s[lram_engine::kMaxLanes] = {s, s, s};
  if (lanes) {
    while ((int)e->micro_streams.size() < NL - 1) {
      hipStream_t ns;
      LRAM_HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
      e->micro_streams.push_back(ns);
    }
    for (int k = 1; k < NL; ++k) lane_s[k] = e->micro_streams[k - 1];
    for (auto& v : e->lane_ev)
      while ((int)v.size() < c.n_blocks) {
        hipEvent_t nev;
        LRAM_HIP_CHECK(hipEventCreateWithFlags(&nev, hipEventDisableTiming | (e->event_device_scope ? hipEventDisableSystemFence : 0u)));
        v.push_back(nev);
      }
    for (int k = 1; k < NL; ++k) stream_after(e, lane_s[k], s, true);
  }
  int Tc = T, last_steps = 1;
  for (int l = 0, ci = 0; l < L; l += kChunk, ++ci) {
    const int Lc = std::min(kChunk, L - l);
    Tc = T * Lc;
    last_steps = Lc;
    const int lane = (n_chunks - 1 - ci) % NL;
    const std::vector<Slice> lane_sl = {Slice{0, e->B, lane_s[lane]}};
    const std::vector<Slice>& use = lanes ? lane_sl : sl;
    // (scope guard: an exception out of a launch below must not leave the engine on a lane's workspace or with lane events set)
    struct LaneScope {
      lram_engine* e;
      int lane;
      ~LaneScope() {
        e->lane_wait = e->lane_rec = nullptr;
        if (lane) swap_workspace(e, lane);
      }
    } lane_scope{e, lane};
    if (lane) swap_workspace(e, lane);
    for (const Slice& x : use) {
      if (shared_passes > 1 && col_begin > 0) break;  // the tokens of this env-step were embedded by pass 0 (X0 / U0)
      const size_t r0 = (size_t)x.b0 * Tc, b0 = x.b0;
      float* X = e->X.p + r0 * D;
      if (seq_emb != nullptr) {  // stored context: the chunk's token rows in one launch
        launch_embed_chunk(X, seq_emb + (b0 * L + l) * D, (int64_t)L * D, rtg + b0 * L + l, rew + b0 * L + l, L, e->w_rtg, e->b_rtg,
                           e->w_rew, e->b_rew, x.nb, Lc, Tc, D, x.s);
        launch_row_norm(X, D, X, D, e->eln_g, e->eln_b, x.nb * Tc, D, 1e-5f, 0, x.s);
        continue;
      }
      if (e->step_images != nullptr)   // lram_step_images: this slice's frames -> its rows of `obs` (= IMG_EMB), on its own stream
        embed_images(e, e->step_images + b0 * e->step_img_c * e->step_img_h * e->step_img_w, e->step_img_c, e->step_img_h,
                     e->step_img_w, e->IMG_EMB.p + b0 * D, x.s, (int)b0, x.nb);
      for (int j = 0; j < Lc; ++j) {
        const float* o = obs + (b0 * L + l + j) * obs_w;
        float* Xj = X + (size_t)(T * j) * D;  // token slots 3j .. 3j+2 of every env row group
        if (emb) {
          launch_scatter_token0(Xj, o, (int64_t)L * D, x.nb, Tc, D, x.s);
        } else {
          GemmArgs ge;
          ge.a = o, ge.lda = (int64_t)L * c.state_dim, ge.w = e->w_state, ge.ldw = c.state_dim, ge.c = Xj;
          ge.ldc = (int64_t)Tc * D, ge.bias = e->b_state, ge.m = x.nb, ge.n = D, ge.k = c.state_dim;
          gemm(e, ge, x.s);
        }
        // (a single timestep per call: the scalar tokens are built by the embed_ln launch below)
        if (Lc > 1 || T != 3)
          launch_embed_scalars(Xj, rtg + b0 * L + l + j, rew + b0 * L + l + j, L, e->w_rtg, e->b_rtg, e->w_rew, e->b_rew,
                               x.nb, Tc, D, x.s);
      }
      // embed_ln in place; single env-steps of small batches also keep a copy for lram_get_taps (written by the same launch)
      ScalarTokens stok;
      const bool stok_on = Lc == 1 && T == 3;
      if (stok_on) {
        stok.rtg = rtg + b0 * L + l, stok.rew = rew + b0 * L + l, stok.in_stride = L, stok.T = T;
        stok.w_rtg = e->w_rtg, stok.b_rtg = e->b_rtg, stok.w_rew = e->w_rew, stok.b_rew = e->b_rew;
      }
      launch_row_norm(X, D, X, D, e->eln_g, e->eln_b, x.nb * Tc, D, 1e-5f, 0, x.s,
                      (L == 1 && e->B <= kTokenTapMaxBatch) ? e->TOK.p + r0 * D : nullptr, nullptr, stok_on ? &stok : nullptr);
    }
    if (lanes) e->lane_wait = ci > 0 ? &e->lane_ev[(lane + 1) % NL] : nullptr, e->lane_rec = &e->lane_ev[lane];
    run_stack(e, Tc, l == 0 ? reset : nullptr, use, lanes ? lane_s[lane] : hbm);
  }
  for (int k = 1; k < NL; ++k) stream_after(e, s, lane_s[k], true);
  if (actions != nullptr) {
    const int64_t nlog = (int64_t)c.act_dim * c.n_vocab;
    const int pred = T * (last_steps - 1) + c.pred_token;  // rtg token of the last timestep in the last chunk
    // shared repeated forwards: pass p only has to produce action dim p (the last pass every dim from its own on), so
    // the head evaluates that column block of action_net alone
    const int col_end = (shared_passes > 1 && col_begin + 1 < shared_passes) ? col_begin + 1 : c.act_dim;
    const int col0 = shared_passes > 1 ? col_begin : 0;
    for (const Slice& x : sl) {
      const size_t r0 = (size_t)x.b0 * Tc, b0 = x.b0;
      GemmArgs gh;
      gh.a = e->HID.p + (r0 + pred) * D, gh.lda = (int64_t)Tc * D, gh.w = e->w_head + (size_t)col0 * c.n_vocab * D, gh.ldw = D;
      gh.c = e->LOGITS.p + b0 * nlog + (size_t)col0 * c.n_vocab, gh.ldc = nlog, gh.bias = e->b_head + (size_t)col0 * c.n_vocab;
      gh.m = x.nb, gh.n = (col_end - col0) * c.n_vocab, gh.k = D;
      gemm(e, gh, x.s);
      launch_action_argmax(e->LOGITS.p + b0 * nlog, actions + b0 * c.act_dim,
                           tokens ? tokens + b0 * c.act_dim : nullptr, x.nb, c.act_dim, c.n_vocab, c.n_discrete,
                           c.action_channels, c.tok_min, c.tok_max, discrete, col_begin, x.s, col_end);
    }
  }
  if (multi && (fork_join & 2)) join_slices(e, sl, hbm, s);
}

// Do the repeated forwards of the Mamba reference-trajectory mode share the token front end and layer 0's in_proj?
bool compat_shares(const lram_engine* e, int discrete) {
  const int passes = discrete ? 1 : std::max(1, std::min(e->compat_repeat, e->cfg.act_dim));
  return passes > 1 && e->cfg.backbone == LRAM_BACKBONE_MAMBA && e->compat_share && e->cfg.n_blocks >= 2;
}
// ... then pass 0 keeps them in X0 / U0.  Called by lram_step BEFORE any stream capture begins: hipMalloc on a thread with
// an active capture fails with hipErrorStreamCaptureUnsupported and invalidates the capture (graph mode + repeated forwards).
void compat_prepare(lram_engine* e, int discrete) {
  if (e->B <= 0 || !compat_shares(e, discrete)) return;
  const size_t bt = (size_t)e->B * e->cfg.tokens_per_step;
  if (e->X0.n < bt * e->cfg.d_model) e->X0.alloc(bt * e->cfg.d_model);
  if (e->U0.n < bt * 2 * e->cfg.d_inner) e->U0.alloc(bt * 2 * e->cfg.d_inner);
}

void step_launches(lram_engine* e, const float* obs, int emb, const float* rtg, const float* rew,
                   const uint8_t* reset, int discrete, float* actions, int32_t* tokens, hipStream_t s) {
  // compat_repeat (reference DiscreteDecisionMamba.get_action_pred, src/algos/decision_mamba.py:107-122): the same
  // (state, rtg, reward) tokens go through the stack once per action dim with the cache on, and action dim i is the
  // prediction of forward i.  Forward p writes action columns >= p, so column i keeps forward min(i, repeat - 1).
  const int passes = discrete ? 1 : std::max(1, std::min(e->compat_repeat, e->cfg.act_dim));
  const bool share = compat_shares(e, discrete);
  if (share) {  // (allocated by compat_prepare ahead of this call: never inside a stream capture)
    const size_t bt = (size_t)e->B * e->cfg.tokens_per_step;
    LRAM_REQUIRE(e->X0.n >= bt * e->cfg.d_model && e->U0.n >= bt * 2 * e->cfg.d_inner,
                 "shared repeated forwards: workspace not prepared");
  }
  e->compat_passes = share ? passes : 1;
  for (int p = 0; p < passes; ++p) {
    e->compat_pass = share ? p : 0;
    // every forward runs on the same slice streams: a slice's forward p + 1 follows its forward p in stream order (state, X0 / U0,
    // logits are per slice), so the slices are forked once and joined once instead of draining the two-slice pipeline per forward
    // (Mamba-48M at 2048 slots, 4 forwards per env-step, same box: 140.35k -> 141.0k env-steps/s)
    const int fj = passes > 1 ? ((p == 0 ? 1 : 0) | (p == passes - 1 ? 2 : 0)) : 3;
    timesteps_launches(e, obs, emb, rtg, rew, 1, p == 0 ? reset : nullptr, discrete, actions, tokens, s, p,
                       share ? passes : 0, fj);
  }
  e->compat_pass = 0, e->compat_passes = 1;
}

struct StateView {
  float* p;
  size_t n;
};
StateView state_view(const lram_engine* e, int block, int which) {
  if (block < 0 || block >= (int)e->st.size()) return {nullptr, 0};
  const BlockState& s = e->st[block];
  const bool mlstm = e->cfg.backbone == LRAM_BACKBONE_XLSTM && !e->cfg.block_is_slstm[block];
  switch (which) {
    case 0: return {s.s0.p, s.s0.n};
    case 1: return mlstm ? StateView{s.n.p, s.n.n} : StateView{nullptr, 0};
    case 2: return mlstm ? StateView{s.m.p, s.m.n} : StateView{nullptr, 0};
    case 3: return {s.conv.p, s.conv.n};
    default: return {nullptr, 0};
  }
}

template <typename Fn>
int32_t guarded(Fn&& fn) {
  try {
    fn();
    g_last_error.clear();
    return 0;
  } catch (const std::exception& ex) {
    g_last_error = ex.what();
    return 1;
  } catch (...) {
    g_last_error = "lram: unknown error";
    return 1;
  }
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* lram_last_error(void) { return g_last_error.c_str(); }

int32_t lram_abi_version(void) { return LRAM_ABI_VERSION; }

int32_t lram_create(const lram_config* cfg, int32_t device, lram_engine** out) {
  return guarded([&] {
    LRAM_REQUIRE(cfg != nullptr && out != nullptr, "lram_create: null argument");
    validate_config(*cfg);
    int ndev = 0;
    LRAM_HIP_CHECK(hipGetDeviceCount(&ndev));
    LRAM_REQUIRE(device >= 0 && device < ndev, "lram_create: no such HIP device");
    LRAM_HIP_CHECK(hipSetDevice(device));
    auto e = std::make_unique<lram_engine>();
    e->cfg = *cfg;
    e->device = device;
    // Environment knobs (measurement / test switches; the table is in DESIGN.md section 5)
    gemm_knobs_reload();   // the projection launchers' process-wide knobs: read here, never on the step path
    if (const char* v = std::getenv("LRAM_PREFILL_CHUNK")) e->chunk_prefill = std::atoi(v) != 0, e->chunk_exact_fp32 = std::atoi(v) == 2, e->chunk_lanes = std::atoi(v) != 3;
    if (const char* v = std::getenv("LRAM_STATE")) {
      const std::string m(v);
      e->lazy_mode = m == "lazy" ? 1 : (m == "eager" || m == "materialised" || m == "materialized") ? 0 : 2;
    }
    if (const char* v = std::getenv("LRAM_LAZY_PERIOD")) e->lazy_period = std::max(1, std::min(14, std::atoi(v)));
    if (const char* v = std::getenv("LRAM_F16_MIN_ROWS")) e->f16x2_min_rows = std::max(9, std::atoi(v));
    if (const char* v = std::getenv("LRAM_GEMM_PRESPLIT")) e->gemm_presplit = std::atoi(v) != 0;
    if (const char* v = std::getenv("LRAM_COMPAT_SHARE")) e->compat_share = std::atoi(v) != 0;
    if (const char* v = std::getenv("LRAM_MAMBA_DT_FUSE")) e->mamba_dt_fuse = std::atoi(v) != 0;
    if (const char* v = std::getenv("LRAM_GN_FUSE")) e->gn_fuse = std::max(0, std::min(2, std::atoi(v)));
    if (const char* v = std::getenv("LRAM_GN_AMAX")) e->gn_amax_handover = std::atoi(v) != 0, e->gn_planes = std::atoi(v) >= 2;
    if (const char* v = std::getenv("LRAM_SLSTM_GATES_ONE")) e->slstm_gates_one = std::atoi(v) != 0;
    if (const char* v = std::getenv("LRAM_UPZ_8P")) e->upz_beside = std::atoi(v) != 0;
    if (const char* v = std::getenv("LRAM_GEMM_NARROW")) e->gemm_narrow_on = std::atoi(v) != 0, e->gemm_narrow_f16 = std::atoi(v) != 2;
    if (const char* v = std::getenv("LRAM_SLSTM_FUSED_ROWS")) e->slstm_fused_rows = std::max(0, std::atoi(v));
    if (const char* v = std::getenv("LRAM_SLSTM_SEQ")) e->slstm_seq = std::atoi(v) != 0, e->slstm_seq_f32 = std::atoi(v) == 2;
    if (const char* v = std::getenv("LRAM_LAZY_CAP2_ENVS")) e->lazy_cap2_envs = std::max(0, std::atoi(v));
    if (const char* v = std::getenv("LRAM_GEMM_SKINNY_ROWS")) e->gemm_skinny_rows = std::max(0, std::atoi(v));
    if (const char* v = std::getenv("LRAM_GEMM_SKINNY_MIN")) e->gemm_skinny_min = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("LRAM_FRONT_MULTI")) e->front_multi = std::atoi(v) != 0;
    if (const char* v = std::getenv("LRAM_FRONT_MIN_ENVS")) e->front_min_envs = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("LRAM_EVENT_SCOPE")) e->event_device_scope = std::string(v) != "system";
    *out = e.release();
  });
}

int32_t lram_destroy(lram_engine* e) {
  return guarded([&] {
    if (e) {
      (void)hipSetDevice(e->device);
      delete e;
    }
  });
}

int32_t lram_set_weight(lram_engine* e, const char* name, const float* host_data, size_t numel) {
  return guarded([&] {
    LRAM_REQUIRE(e && name && host_data && numel > 0, "lram_set_weight: bad argument");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    DevBuf& b = e->weights[name];
    b.alloc(numel);
    LRAM_HIP_CHECK(hipMemcpy(b.p, host_data, numel * sizeof(float), hipMemcpyHostToDevice));
    e->finalized = false;
    e->drop_graph();
  });
}

int32_t lram_finalize(lram_engine* e) {
  return guarded([&] {
    LRAM_REQUIRE(e != nullptr, "lram_finalize: null engine");
    finalize(e);
  });
}

int32_t lram_state_alloc(lram_engine* e, int32_t batch) {
  return guarded([&] {
    LRAM_REQUIRE(e != nullptr, "lram_state_alloc: null engine");
    state_alloc(e, batch);
  });
}

int64_t lram_state_bytes_per_env(const lram_engine* e) {
  if (!e) return 0;
  const lram_config& c = e->cfg;
  int64_t elems = 0;
  for (int i = 0; i < c.n_blocks; ++i) {
    if (c.backbone == LRAM_BACKBONE_MAMBA) {
      elems += (int64_t)c.d_inner * (c.d_state + c.d_conv);
    } else if (c.block_is_slstm[i]) {
      elems += (int64_t)c.d_model * (4 + c.conv_k);
    } else {
      const int64_t DH = c.inner / c.n_heads;
      elems += (int64_t)c.n_heads * DH * DH + c.inner + c.n_heads + (int64_t)c.conv_k * c.inner;
    }
  }
  return elems * 4;
}

int32_t lram_reset(lram_engine* e, const uint8_t* dev_env_mask, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0, "lram_reset: state not allocated");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = e->B;
    for (int i = 0; i < e->cfg.n_blocks; ++i) {
      if (e->compat_stale && i > 0) break;  // reference Mamba reset: layers >= 1 keep their cached state (Q1)
      BlockState& st = e->st[i];
      const bool slstm = e->cfg.backbone == LRAM_BACKBONE_XLSTM && e->cfg.block_is_slstm[i];
      if (slstm)
        launch_zero_rows(st.s0.p, dev_env_mask, B, e->cfg.d_model, 4, (int64_t)B * e->cfg.d_model, s);
      else
        launch_zero_rows(st.s0.p, dev_env_mask, B, (int64_t)(st.s0.n / B), 1, 0, s);
      if (st.n.p) launch_zero_rows(st.n.p, dev_env_mask, B, (int64_t)(st.n.n / B), 1, 0, s);
      if (st.m.p) launch_zero_rows(st.m.p, dev_env_mask, B, (int64_t)(st.m.n / B), 1, 0, s);
      launch_zero_rows(st.conv.p, dev_env_mask, B, (int64_t)(st.conv.n / B), 1, 0, s);
      if (e->lazy_ready && st.gsc.p != nullptr)  // pending window of a reset env is dropped with its C_base
        for (int p = 0; p < 2; ++p)
          launch_mlstm_lazy_clear(reinterpret_cast<int32_t*>(e->LZ_COUNT.p) + (size_t)p * B,
                                  st.gsc.p + (size_t)p * B * e->cfg.n_heads, dev_env_mask, B, e->cfg.n_heads, s);
    }
  });
}

// Every public entry that launches the stack counts as one call of a sampled profile (lram_profile_begin_sampled): whether ITS
// launches are timed is decided here, not inherited from whatever call came before.
static void prof_tick(lram_engine* e) { e->prof_live = !e->prof_on || (e->prof_calls++ % e->prof_every) == 0; }

int32_t lram_step(lram_engine* e, const float* dev_obs, int32_t obs_is_embedding, const float* dev_rtg,
                  const float* dev_reward, const uint8_t* dev_reset_mask, int32_t discrete, float* dev_actions,
                  int32_t* dev_tokens, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0, "lram_step: state not allocated (call lram_state_alloc)");
    LRAM_REQUIRE(dev_obs && dev_rtg && dev_reward && dev_actions, "lram_step: null device pointer");
    LRAM_REQUIRE(e->cfg.tokens_per_step == 3, "lram_step: the (state, rtg, reward) front end needs tokens_per_step == 3");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    compat_prepare(e, discrete);  // (workspace of the shared repeated forwards: outside any capture)
    prof_tick(e);
    if (e->graph_mode && !(e->prof_on && e->prof_live)) {  // (a sampled run's un-timed steps keep the graph path)
      GraphKey key{};
      key.obs = dev_obs, key.rtg = dev_rtg, key.rew = dev_reward, key.mask = dev_reset_mask, key.act = dev_actions;
      key.tok = dev_tokens, key.emb = obs_is_embedding, key.discrete = discrete, key.B = e->B, key.stream = s;
      if (!(e->graph_valid && key == e->graph_key)) {
        e->drop_graph();
        if (!e->capture_stream) LRAM_HIP_CHECK(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
        hipStream_t cs = e->capture_stream;
        LRAM_HIP_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        try {
          step_launches(e, dev_obs, obs_is_embedding, dev_rtg, dev_reward, dev_reset_mask, discrete, dev_actions,
                        dev_tokens, cs);
        } catch (...) {
          hipGraph_t g = nullptr;
          (void)hipStreamEndCapture(cs, &g);
          if (g) (void)hipGraphDestroy(g);
          throw;
        }
        LRAM_HIP_CHECK(hipStreamEndCapture(cs, &e->graph));
        LRAM_HIP_CHECK(hipGraphInstantiate(&e->graph_exec, e->graph, nullptr, nullptr, 0));
        e->graph_key = key;
        e->graph_valid = true;
      }
      LRAM_HIP_CHECK(hipGraphLaunch(e->graph_exec, s));
    } else {
      step_launches(e, dev_obs, obs_is_embedding, dev_rtg, dev_reward, dev_reset_mask, discrete, dev_actions,
                    dev_tokens, s);
    }
  });
}

int32_t lram_step_images(lram_engine* e, const uint8_t* dev_images, int32_t channels, int32_t height, int32_t width,
                         const float* dev_rtg, const float* dev_reward, const uint8_t* dev_reset_mask, int32_t discrete,
                         float* dev_actions, int32_t* dev_tokens, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0, "lram_step_images: state not allocated (call lram_state_alloc)");
    LRAM_REQUIRE(dev_images && dev_rtg && dev_reward && dev_actions && channels > 0 && height > 0 && width > 0,
                 "lram_step_images: bad argument");
    LRAM_REQUIRE(e->cfg.tokens_per_step == 3, "lram_step_images: the (state, rtg, reward) front end needs tokens_per_step == 3");
    LRAM_REQUIRE(e->img_lin_w != nullptr, "lram_step_images: no embed_image.* weights were uploaded");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    image_buffers(e, height, width);
    if (e->IMG_EMB.n < (size_t)e->B * e->cfg.d_model) {
      LRAM_HIP_CHECK(hipDeviceSynchronize());
      e->IMG_EMB.alloc((size_t)e->B * e->cfg.d_model);
    }
    compat_prepare(e, discrete);
    prof_tick(e);
    struct Scope {   // (the frames belong to this call only)
      lram_engine* e;
      ~Scope() { e->step_images = nullptr; }
    } scope{e};
    e->step_images = dev_images, e->step_img_c = channels, e->step_img_h = height, e->step_img_w = width;
    // (launch-per-kernel path also in graph mode: a captured step would pin one frame buffer)
    step_launches(e, e->IMG_EMB.p, 1, dev_rtg, dev_reward, dev_reset_mask, discrete, dev_actions, dev_tokens, s);
  });
}

int32_t lram_prefill(lram_engine* e, const float* dev_obs_seq, int32_t obs_is_embedding, const float* dev_rtg_seq,
                     const float* dev_reward_seq, int32_t timesteps, const uint8_t* dev_reset_mask, int32_t discrete,
                     float* dev_actions, int32_t* dev_tokens, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0, "lram_prefill: state not allocated (call lram_state_alloc)");
    LRAM_REQUIRE(dev_obs_seq && dev_rtg_seq && dev_reward_seq, "lram_prefill: null device pointer");
    LRAM_REQUIRE(timesteps >= 1, "lram_prefill: timesteps must be >= 1");
    LRAM_REQUIRE(e->cfg.tokens_per_step == 3, "lram_prefill: the (state, rtg, reward) front end needs tokens_per_step == 3");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    prof_tick(e);
    timesteps_launches(e, dev_obs_seq, obs_is_embedding, dev_rtg_seq, dev_reward_seq, timesteps, dev_reset_mask, discrete,
                       dev_actions, dev_tokens, static_cast<hipStream_t>(stream));
  });
}

int32_t lram_encoder_step(lram_engine* e, const float* dev_inputs_embeds, int32_t tokens,
                          const uint8_t* dev_reset_mask, float* dev_hidden_out, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0, "lram_encoder_step: state not allocated");
    LRAM_REQUIRE(dev_inputs_embeds && dev_hidden_out, "lram_encoder_step: null device pointer");
    const bool chunk_ok = e->cfg.backbone == LRAM_BACKBONE_XLSTM && e->chunk_prefill && !e->graph_mode &&
                          mlstm_chunk_supported(e->cfg.inner, e->cfg.n_heads, e->cfg.conv_k);
    LRAM_REQUIRE((tokens >= 1 && tokens <= 4) || tokens == 6 || tokens == 9 || tokens == 12 ||
                     (chunk_ok && tokens > kMaxTokens && tokens <= kChunkMaxTokens),
                 "lram_encoder_step: tokens must be 1..4, 6, 9 or 12 (13..64 too on xLSTM geometries with a head dim "
                 "that is a multiple of 128)");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    if (tokens > e->tok_cap) {
      LRAM_HIP_CHECK(hipDeviceSynchronize());
      alloc_workspace(e, kChunkMaxTokens);
      LRAM_HIP_CHECK(hipDeviceSynchronize());
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    prof_tick(e);
    const size_t bytes = sizeof(float) * (size_t)e->B * tokens * e->cfg.d_model;
    if (!lazy_active(e, tokens)) lazy_materialize(e, s);
    LRAM_HIP_CHECK(hipMemcpyAsync(e->X.p, dev_inputs_embeds, bytes, hipMemcpyDeviceToDevice, s));
    e->sync_used = 0, e->edge_used = 0;
    hipStream_t hbm;
    const std::vector<Slice> sl = make_slices(e, s, &hbm);
    if (sl.size() > 1) fork_slices(e, sl, hbm, s);
    run_stack(e, tokens, dev_reset_mask, sl, hbm);
    if (sl.size() > 1) join_slices(e, sl, hbm, s);
    LRAM_HIP_CHECK(hipMemcpyAsync(dev_hidden_out, e->HID.p, bytes, hipMemcpyDeviceToDevice, s));
  });
}

int32_t lram_get_taps(lram_engine* e, float* dev_tokens_embed, float* dev_hidden, float* dev_logits, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0, "lram_get_taps: state not allocated");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t btd = sizeof(float) * (size_t)e->B * e->cfg.tokens_per_step * e->cfg.d_model;
    if (dev_tokens_embed) {
      LRAM_REQUIRE(e->B <= kTokenTapMaxBatch,
                   "lram_get_taps: the embed_ln token tap is kept for batches of up to 1024 env slots only (it costs a "
                   "copy of the token buffer per step); pass NULL for it");
      LRAM_HIP_CHECK(hipMemcpyAsync(dev_tokens_embed, e->TOK.p, btd, hipMemcpyDeviceToDevice, s));
    }
    if (dev_hidden) LRAM_HIP_CHECK(hipMemcpyAsync(dev_hidden, e->HID.p, btd, hipMemcpyDeviceToDevice, s));
    if (dev_logits)
      LRAM_HIP_CHECK(hipMemcpyAsync(dev_logits, e->LOGITS.p, sizeof(float) * e->LOGITS.n, hipMemcpyDeviceToDevice, s));
  });
}

int64_t lram_state_numel(const lram_engine* e, int32_t block, int32_t which) {
  if (!e || e->B <= 0) return 0;
  return (int64_t)state_view(e, block, which).n;
}

int32_t lram_state_export(lram_engine* e, int32_t block, int32_t which, float* dev_dst, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0 && dev_dst, "lram_state_export: bad argument");
    StateView v = state_view(e, block, which);
    LRAM_REQUIRE(v.p != nullptr, "lram_state_export: no such state tensor");
    lazy_materialize(e, static_cast<hipStream_t>(stream));
    LRAM_HIP_CHECK(hipMemcpyAsync(dev_dst, v.p, v.n * sizeof(float), hipMemcpyDeviceToDevice,
                                  static_cast<hipStream_t>(stream)));
  });
}

int32_t lram_state_import(lram_engine* e, int32_t block, int32_t which, const float* dev_src, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0 && dev_src, "lram_state_import: bad argument");
    StateView v = state_view(e, block, which);
    LRAM_REQUIRE(v.p != nullptr, "lram_state_import: no such state tensor");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool slstm = e->cfg.backbone == LRAM_BACKBONE_XLSTM && e->cfg.block_is_slstm[block];
    if (slstm && which == 0 && e->slstm_rinv[block].p != nullptr) {
      // The f16x2 form of the sLSTM step (slstm_seq16_kernel) keeps h_t in LDS as two binary16 planes of 2^12 h: every state the
      // recurrence itself produces has |h| < 1, a foreign one need not (|h| >= 16 overflows binary16 to inf and the next step
      // spreads NaN).  A rare call: one small reduction over the h plane [B, D] and a host synchronisation are affordable.
      LRAM_HIP_CHECK(hipSetDevice(e->device));
      const int64_t n = (int64_t)e->B * e->cfg.d_model;   // plane 0 of [4, B, D]
      int* dflag = nullptr;
      int hflag = 0;
      LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&dflag), sizeof(int)));
      try {
        LRAM_HIP_CHECK(hipMemsetAsync(dflag, 0, sizeof(int), s));
        launch_slstm_h_range(dev_src, n, 15.9f, dflag, s);
        LRAM_HIP_CHECK(hipMemcpyAsync(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost, s));
        LRAM_HIP_CHECK(hipStreamSynchronize(s));
      } catch (...) {
        (void)hipFree(dflag);
        throw;
      }
      (void)hipFree(dflag);
      LRAM_REQUIRE(hflag == 0,
                   "lram_state_import: sLSTM hidden plane holds |h| >= 16 (or NaN): outside what the recurrence produces (|h| < 1) "
                   "and outside the binary16 planes of the f16x2 step kernel; import a state the model produced, or run the "
                   "engine with LRAM_SLSTM_SEQ=2 / LRAM_GEMM=f32 (exact fp32 recurrence, no range limit)");
    }
    lazy_materialize(e, s);
    LRAM_HIP_CHECK(hipMemcpyAsync(v.p, dev_src, v.n * sizeof(float), hipMemcpyDeviceToDevice, s));
  });
}

int32_t lram_set_graph_mode(lram_engine* e, int32_t enable) {
  return guarded([&] {
    LRAM_REQUIRE(e != nullptr, "lram_set_graph_mode: null engine");
    if (enable != 0 && e->lazy_ready) {  // graph replay bakes kernel arguments: it runs on the materialised state
      LRAM_HIP_CHECK(hipSetDevice(e->device));
      // steps may still be in flight on a non-blocking caller stream, which the null stream does not order against:
      // drain the device before the folds touch the windows and C
      LRAM_HIP_CHECK(hipDeviceSynchronize());
      lazy_materialize(e, nullptr);
      LRAM_HIP_CHECK(hipDeviceSynchronize());
    }
    e->graph_mode = enable != 0;
    if (!e->graph_mode) e->drop_graph();
  });
}

int32_t lram_set_state_mode(lram_engine* e, int32_t mode, int32_t fold_period) {
  return guarded([&] {
    LRAM_REQUIRE(e != nullptr, "lram_set_state_mode: null engine");
    LRAM_REQUIRE(mode >= 0 && mode <= 2, "lram_set_state_mode: mode must be 0 (materialised), 1 (lazy) or 2 (auto)");
    LRAM_REQUIRE(fold_period == 0 || (fold_period >= 1 && fold_period * e->cfg.tokens_per_step + 4 <= kLazyWindow),
                 "lram_set_state_mode: fold_period out of range (the window holds 48 tokens)");
    LRAM_REQUIRE(mode != 1 || lazy_geometry_ok(e),
                 "lram_set_state_mode: lazy matrix memory needs an xLSTM head dim that is a multiple of 128");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    if (e->lazy_ready) {  // leave the current mode with a materialised state
      LRAM_HIP_CHECK(hipDeviceSynchronize());  // pending steps on non-blocking streams first (see lram_set_graph_mode)
      lazy_materialize(e, nullptr);
      LRAM_HIP_CHECK(hipDeviceSynchronize());
    }
    e->lazy_mode = mode;
    if (fold_period > 0) e->lazy_period = fold_period;
    e->lazy_bound.clear();
    e->lazy = e->B > 0 && lazy_choice(e);
    if (e->lazy) lazy_alloc(e);
  });
}

int32_t lram_get_state_mode(const lram_engine* e) { return (e != nullptr && e->lazy && e->lazy_ready) ? 1 : 0; }

int32_t lram_lazy_peek(lram_engine* e, int32_t block, int32_t which, float* dev_dst, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0 && dev_dst, "lram_lazy_peek: bad argument");
    LRAM_REQUIRE(e->lazy && e->lazy_ready, "lram_lazy_peek: the lazy representation is not in effect");
    LRAM_REQUIRE(block >= 0 && block < e->cfg.n_blocks && !e->cfg.block_is_slstm[block] && which >= 0 && which <= 2,
                 "lram_lazy_peek: no such tensor");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t B = e->B, NH = e->cfg.n_heads;
    const int side = (int)(e->lazy_step & 1);  // what the next step reads = what the last one wrote
    if (which == 0) {
      LRAM_HIP_CHECK(hipMemcpyAsync(dev_dst, e->st[block].gsc.p + side * B * NH, B * NH * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else if (which == 1) {
      LRAM_HIP_CHECK(hipMemcpyAsync(dev_dst, e->st[block].m.p, B * NH * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
      launch_lazy_counts_as_float(reinterpret_cast<const int32_t*>(e->LZ_COUNT.p) + side * B, dev_dst, (int)B, s);
    }
  });
}

int32_t lram_set_micro_batches(lram_engine* e, int32_t n) {
  return guarded([&] {
    LRAM_REQUIRE(e != nullptr && n >= 0 && n <= 8, "lram_set_micro_batches: n must be in 0..8 (0 = auto)");
    e->n_micro = n;
    e->drop_graph();
  });
}

int32_t lram_set_compat_mode(lram_engine* e, int32_t mamba_repeat, int32_t stale_state) {
  return guarded([&] {
    LRAM_REQUIRE(e != nullptr, "lram_set_compat_mode: null engine");
    LRAM_REQUIRE(mamba_repeat >= 1 && mamba_repeat <= 64, "lram_set_compat_mode: mamba_repeat must be in 1..64");
    LRAM_REQUIRE(e->cfg.backbone == LRAM_BACKBONE_MAMBA || (mamba_repeat == 1 && stale_state == 0),
                 "lram_set_compat_mode: the repeated-forward / stale-state quirks belong to the reference's Mamba agent "
                 "(src/algos/decision_mamba.py); the xLSTM agent does a single forward and drops the whole cache");
    if (e->compat_repeat != mamba_repeat || e->compat_stale != (stale_state != 0)) e->drop_graph();
    e->compat_repeat = mamba_repeat;
    e->compat_stale = stale_state != 0;
  });
}

int32_t lram_get_compat_mode(const lram_engine* e, int32_t* mamba_repeat, int32_t* stale_state) {
  if (e == nullptr) return 1;
  if (mamba_repeat) *mamba_repeat = e->compat_repeat;
  if (stale_state) *stale_state = e->compat_stale ? 1 : 0;
  return 0;
}

int32_t lram_profile_begin(lram_engine* e) { return lram_profile_begin_sampled(e, 1); }

int32_t lram_profile_begin_sampled(lram_engine* e, int32_t every_n_steps) {
  return guarded([&] {
    LRAM_REQUIRE(e != nullptr && every_n_steps >= 1, "lram_profile_begin_sampled: null engine / every_n_steps < 1");
    e->prof_on = true;
    e->prof_used = 0;
    e->prof_every = every_n_steps;
    e->prof_calls = 0;
    e->prof_live = true;
  });
}

int32_t lram_profile_end(lram_engine* e, double* total_ms, int64_t* n_launches) {
  return guarded([&] {
    LRAM_REQUIRE(e && total_ms && n_launches, "lram_profile_end: bad argument");
    double tot = 0.0;
    size_t n_aux = 0;
    for (size_t i = 0; i < e->prof_used; ++i) {
      LRAM_HIP_CHECK(hipEventSynchronize(e->prof_events[i].second));
      float ms = 0.f;
      LRAM_HIP_CHECK(hipEventElapsedTime(&ms, e->prof_events[i].first, e->prof_events[i].second));
      tot += ms;
      if (e->prof_aux[i]) ++n_aux;
    }
    *total_ms = tot;
    *n_launches = (int64_t)(e->prof_used - n_aux);
    e->prof_on = false;
    e->prof_live = true;
    e->prof_used = 0;
  });
}

int32_t lram_profile_end_split(lram_engine* e, double* main_ms, int64_t* n_main, double* aux_ms, int64_t* n_aux) {
  return guarded([&] {
    LRAM_REQUIRE(e && main_ms && n_main && aux_ms && n_aux, "lram_profile_end_split: bad argument");
    double tm = 0.0, ta = 0.0;
    int64_t nm = 0, na = 0;
    for (size_t i = 0; i < e->prof_used; ++i) {
      LRAM_HIP_CHECK(hipEventSynchronize(e->prof_events[i].second));
      float ms = 0.f;
      LRAM_HIP_CHECK(hipEventElapsedTime(&ms, e->prof_events[i].first, e->prof_events[i].second));
      if (e->prof_aux[i]) {
        ta += ms;
        ++na;
      } else {
        tm += ms;
        ++nm;
      }
    }
    *main_ms = tm, *n_main = nm, *aux_ms = ta, *n_aux = na;
    e->prof_on = false;
    e->prof_live = true;
    e->prof_used = 0;
  });
}

int32_t lram_gemm_counts(lram_engine* e, double* out8, int32_t reset) {
  if (e == nullptr || out8 == nullptr) return 1;
  for (int i = 0; i < 8; ++i) out8[i] = e->gemm_counts[i];
  if (reset)
    for (int i = 0; i < 8; ++i) e->gemm_counts[i] = 0.0;
  return 0;
}

int32_t lram_gemm_f32(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c, int64_t ldc,
                      const float* dev_bias, int32_t accumulate, int32_t m, int32_t n, int32_t k, void* stream) {
  return guarded([&] {
    gemm_knobs_reload();   // standalone test / micro-benchmark entry: the launch knobs as the environment has them NOW
    GemmArgs g;
    g.a = dev_a, g.lda = lda, g.w = dev_w, g.ldw = ldw, g.c = dev_c, g.ldc = ldc, g.bias = dev_bias;
    g.residual = accumulate ? dev_c : nullptr;
    g.m = m, g.n = n, g.k = k;
    launch_gemm_f32(g, static_cast<hipStream_t>(stream));
  });
}

int32_t lram_gemm_skinny(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c, int64_t ldc,
                         const float* dev_bias, int32_t accumulate, int32_t m, int32_t n, int32_t k, void* stream) {
  return guarded([&] {
    gemm_knobs_reload();   // standalone test / micro-benchmark entry: the launch knobs as the environment has them NOW
    GemmArgs g;
    g.a = dev_a, g.lda = lda, g.w = dev_w, g.ldw = ldw, g.c = dev_c, g.ldc = ldc, g.bias = dev_bias;
    g.residual = accumulate ? dev_c : nullptr;
    g.m = m, g.n = n, g.k = k;
    launch_gemm_skinny(g, static_cast<hipStream_t>(stream));
  });
}

int32_t lram_gemm_narrow(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c, int64_t ldc,
                         const float* dev_bias, int32_t accumulate, int32_t m, int32_t n, int32_t k, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(ldw == k && accumulate == 0, "lram_gemm_narrow: W must be contiguous [n, k]; no accumulation");
    LRAM_REQUIRE(gemm_narrow_shape(n, k), "lram_gemm_narrow: n <= 96, k a multiple of 64, >= 256");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* packed = nullptr;
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&packed), gemm_narrow_pack_elems(n, k) * sizeof(float)));
    try {
      launch_gemm_narrow_pack(dev_w, n, k, packed, s);
      GemmArgs g;
      g.a = dev_a, g.lda = lda, g.w = dev_w, g.ldw = ldw, g.c = dev_c, g.ldc = ldc, g.bias = dev_bias;
      g.m = m, g.n = n, g.k = k;
      launch_gemm_narrow(g, packed, s);
      LRAM_HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      (void)hipFree(packed);
      throw;
    }
    (void)hipFree(packed);
  });
}

int32_t lram_gemm_narrow_f16x2(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c, int64_t ldc,
                               const float* dev_bias, int32_t accumulate, int32_t m, int32_t n, int32_t k, void* stream) {
  return guarded([&] {
    gemm_knobs_reload();
    LRAM_REQUIRE(ldw == k && accumulate == 0, "lram_gemm_narrow_f16x2: W must be contiguous [n, k]; no accumulation");
    LRAM_REQUIRE(gemm_narrow_shape(n, k), "lram_gemm_narrow_f16x2: n <= 96, k a multiple of 64, >= 256");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t numel = split_f16x2_plane_elems((size_t)n, (size_t)k);
    uint16_t* planes = nullptr;
    float* scales = nullptr;  // [n] inverse weight scales, then [m] row maxima of A
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&planes), 2 * numel * sizeof(uint16_t)));
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scales), ((size_t)n + m) * sizeof(float)));
    try {
      launch_split_f16x2(dev_w, n, k, planes, scales, s);
      launch_row_amax(dev_a, lda, nullptr, 0, m, k, scales + n, s);
      GemmArgs g;
      g.a = dev_a, g.lda = lda, g.w = dev_w, g.ldw = ldw, g.c = dev_c, g.ldc = ldc, g.bias = dev_bias;
      g.m = m, g.n = n, g.k = k, g.w2 = planes, g.w2_plane = (int64_t)numel, g.w2_kt = 32 * (int64_t)n, g.w_inv = scales;
      g.a_amax = scales + n, g.amax_parts = 1;
      launch_gemm_narrow16(g, s);
      LRAM_HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      (void)hipFree(planes);
      (void)hipFree(scales);
      throw;
    }
    (void)hipFree(planes);
    (void)hipFree(scales);
  });
}

int32_t lram_gemm_bf16x3(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c, int64_t ldc,
                         const float* dev_bias, int32_t accumulate, int32_t m, int32_t n, int32_t k, void* stream) {
  return guarded([&] {
    gemm_knobs_reload();   // standalone test / micro-benchmark entry: the launch knobs as the environment has them NOW
    LRAM_REQUIRE(ldw == k, "lram_gemm_bf16x3: W must be contiguous [n, k]");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t numel = (size_t)n * k;
    uint16_t* planes = nullptr;
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&planes), 3 * numel * sizeof(uint16_t)));
    try {
      launch_split_bf16x3(dev_w, planes, numel, s);
      GemmArgs g;
      g.a = dev_a, g.lda = lda, g.w = dev_w, g.ldw = ldw, g.c = dev_c, g.ldc = ldc, g.bias = dev_bias;
      g.residual = accumulate ? dev_c : nullptr;
      g.m = m, g.n = n, g.k = k, g.w3 = planes, g.w3_plane = (int64_t)numel;
      launch_gemm_bf16x3(g, s);
      LRAM_HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      (void)hipFree(planes);
      throw;
    }
    (void)hipFree(planes);
  });
}

int32_t lram_gemm_f16x2(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c, int64_t ldc,
                        const float* dev_bias, int32_t accumulate, int32_t m, int32_t n, int32_t k, void* stream) {
  return guarded([&] {
    gemm_knobs_reload();   // standalone test / micro-benchmark entry: the launch knobs as the environment has them NOW
    LRAM_REQUIRE(ldw == k, "lram_gemm_f16x2: W must be contiguous [n, k]");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t numel = split_f16x2_plane_elems((size_t)n, (size_t)k);  // (K-tile-major planes)
    uint16_t* planes = nullptr;
    float* scales = nullptr;  // [n] inverse weight scales, then [m] activation scales
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&planes), 2 * numel * sizeof(uint16_t)));
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scales), ((size_t)n + m) * sizeof(float)));
    try {
      launch_split_f16x2(dev_w, n, k, planes, scales, s);
      launch_row_amax(dev_a, lda, nullptr, 0, m, k, scales + n, s);
      GemmArgs g;
      g.a = dev_a, g.lda = lda, g.w = dev_w, g.ldw = ldw, g.c = dev_c, g.ldc = ldc, g.bias = dev_bias;
      g.residual = accumulate ? dev_c : nullptr;
      g.m = m, g.n = n, g.k = k, g.w2 = planes, g.w2_plane = (int64_t)numel, g.w2_kt = 32 * (int64_t)n, g.w_inv = scales, g.a_amax = scales + n;
      launch_gemm_f16x2(g, s);
      LRAM_HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      (void)hipFree(planes);
      (void)hipFree(scales);
      throw;
    }
    (void)hipFree(planes);
    (void)hipFree(scales);
  });
}

int32_t lram_gemm_f16x2_presplit(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c, int64_t ldc,
                                 const float* dev_bias, int32_t accumulate, int32_t m, int32_t n, int32_t k, void* stream) {
  return guarded([&] {
    gemm_knobs_reload();   // standalone test / micro-benchmark entry: the launch knobs as the environment has them NOW
    LRAM_REQUIRE(ldw == k, "lram_gemm_f16x2_presplit: W must be contiguous [n, k]");
    LRAM_REQUIRE(k % 32 == 0 && k <= 3072, "lram_gemm_f16x2_presplit: k must be a multiple of 32, <= 3072");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t wn = (size_t)n * k, an = (size_t)m * k;
    uint16_t *wp = nullptr, *ap = nullptr;
    float* scales = nullptr;  // [n] inverse weight scales, then [m] inverse activation scales
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&wp), 2 * wn * sizeof(uint16_t)));
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ap), 2 * an * sizeof(uint16_t)));
    LRAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scales), ((size_t)n + m) * sizeof(float)));
    try {
      launch_split_f16x2(dev_w, n, k, wp, scales, s);
      launch_row_split_f16x2(dev_a, lda, nullptr, 0, m, k, ap, 32 * (int64_t)m, (int64_t)an, scales + n, s);
      GemmArgs g;
      g.lda = k, g.w = dev_w, g.ldw = ldw, g.c = dev_c, g.ldc = ldc, g.bias = dev_bias;
      g.residual = accumulate ? dev_c : nullptr;
      g.m = m, g.n = n, g.k = k, g.w2 = wp, g.w2_plane = (int64_t)wn, g.w2_kt = 32 * (int64_t)n, g.w_inv = scales;
      g.a2 = ap, g.a2_plane = (int64_t)an, g.a2_kt = 32 * (int64_t)m, g.a2_inv = scales + n;
      launch_gemm_f16x2p(g, s);
      LRAM_HIP_CHECK(hipStreamSynchronize(s));
    } catch (...) {
      (void)hipFree(wp), (void)hipFree(ap), (void)hipFree(scales);
      throw;
    }
    (void)hipFree(wp), (void)hipFree(ap), (void)hipFree(scales);
  });
}

int32_t lram_pad_obs(const float* dev_native, int32_t n_native, const int32_t* dev_inv_index, const float* dev_mean,
                     const float* dev_std, float* dev_out, int32_t batch, int32_t state_dim, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(dev_native && dev_out && batch > 0 && state_dim > 0 && n_native > 0, "lram_pad_obs: bad argument");
    LRAM_REQUIRE(dev_inv_index != nullptr || n_native <= state_dim, "lram_pad_obs: observation wider than state_dim");
    LRAM_REQUIRE((dev_mean == nullptr) == (dev_std == nullptr), "lram_pad_obs: mean and std go together");
    launch_pad_obs(dev_native, n_native, dev_inv_index, dev_mean, dev_std, dev_out, batch, state_dim,
                   static_cast<hipStream_t>(stream));
  });
}

int32_t lram_embed_images(lram_engine* e, const uint8_t* dev_images, int32_t channels, int32_t height, int32_t width,
                          float* dev_embeddings, void* stream) {
  return guarded([&] {
    LRAM_REQUIRE(e && e->B > 0, "lram_embed_images: state not allocated (call lram_state_alloc)");
    LRAM_REQUIRE(dev_images && dev_embeddings && channels > 0 && height > 0 && width > 0, "lram_embed_images: bad argument");
    LRAM_HIP_CHECK(hipSetDevice(e->device));
    image_buffers(e, height, width);
    embed_images(e, dev_images, channels, height, width, dev_embeddings, static_cast<hipStream_t>(stream));
  });
}

int32_t lram_stream_copy(float* dev_dst, const float* dev_src, size_t numel, void* stream) {
  return guarded([&] { launch_stream_copy(dev_dst, dev_src, numel, static_cast<hipStream_t>(stream)); });
}

int32_t lram_stream_read(const float* dev_buf, size_t numel, float* dev_sink, void* stream) {
  return guarded([&] { launch_stream_read(dev_buf, numel, dev_sink, static_cast<hipStream_t>(stream)); });
}

int32_t lram_stream_rmw(float* dev_buf, size_t numel, void* stream) {
  return guarded([&] { launch_stream_rmw(dev_buf, numel, static_cast<hipStream_t>(stream)); });
}

}  // extern "C"
