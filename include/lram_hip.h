/*
 * lram_hip.h -- C ABI of the MI355X-native recurrent action-inference engine.
 *
 * The reference (ml-jku/LRAM) has no C interface: its drop-in boundary for the rollout hot path is the
 * Python `self.encoder(inputs_embeds, past_key_values, use_cache)` operator of the policy
 * (src/algos/models/decision_xlstm.py:138-169, src/algos/models/decision_mamba.py:109-166) together
 * with the embed / head code around it (src/algos/models/online_decision_transformer_model.py:392-461).
 * Each entry point below names the reference interface it stands in for.  Host code (the lram_amd package)
 * binds this library with ctypes; INTEGRATION.md shows the stub a maintainer adds on the LRAM side.
 *
 * Conventions
 *   - plain C, no torch / C++ types; every pointer argument says host or device.
 *   - every function returns 0 on success, non-zero on error; lram_last_error() gives the text.
 *   - one engine = one GPU, one stream per call (caller passes a hipStream_t as void*, NULL = default
 *     stream); no internal threads; calls on one engine must be externally serialised.
 *   - all floating point data is fp32, row-major, contiguous unless a stride is given.
 */
#ifndef LRAM_HIP_H
#define LRAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRAM_ABI_VERSION 1

#define LRAM_BACKBONE_XLSTM 0
#define LRAM_BACKBONE_MAMBA 1

#define LRAM_MAX_BLOCKS 64

/* Model description.  Mirrors the fields the reference reads from `agent_params.huggingface`
 * (configs/agent_params/huggingface/xlstm_*.yaml, mamba_*.yaml -> xLSTMConfig / MambaConfig,
 * src/algos/models/decision_xlstm.py:104-116, src/algos/models/decision_mamba.py:15-49) and from
 * `agent_params.model_kwargs` (configs/agent_params/model_kwargs/multi_domain.yaml). */
typedef struct lram_config {
  int32_t abi_version;      /* must be LRAM_ABI_VERSION */
  int32_t backbone;         /* LRAM_BACKBONE_* */
  int32_t d_model;          /* hidden_size / embedding_dim / d_model */
  int32_t n_blocks;         /* n_layer / num_blocks */
  int32_t tokens_per_step;  /* 3: (state, return-to-go, reward), discrete_decision_transformer_model.py:265-275 */
  int32_t pred_token;       /* 1: action is read at the rtg token, tok_to_pred_pos["a"] */
  /* xLSTM */
  int32_t n_heads;          /* mlstm.num_heads == slstm.num_heads (4) */
  int32_t conv_k;           /* conv1d_kernel_size (4) */
  int32_t qkv_blocksize;    /* qkv_proj_blocksize (4) */
  int32_t inner;            /* mLSTM inner dim = ceil64(2 * d_model) */
  int32_t ffn_dim;          /* sLSTM-block gated FFN dim = ceil64(1.3 * d_model) */
  int32_t block_is_slstm[LRAM_MAX_BLOCKS]; /* 1 where the block index is in `slstm_at` */
  int32_t norm_is_rms;      /* 1 when HF config has rms_norm (decision_xlstm.py:190-191) */
  float   ln_eps;           /* 1e-5 */
  /* Mamba */
  int32_t d_inner;          /* expand * d_model */
  int32_t d_state;          /* 16 */
  int32_t d_conv;           /* 4 */
  int32_t dt_rank;          /* ceil(d_model / 16) */
  float   norm_eps;         /* 1e-5 */
  /* token front end / head (multi_domain_discrete_dt_model.py:12-81) */
  int32_t state_dim;        /* 204 (max_state_dim) */
  int32_t act_dim;          /* 8 (max_act_dim) */
  int32_t n_vocab;          /* 274 = discrete_actions + action_channels */
  int32_t n_discrete;       /* 18, also the tokenizer shift */
  int32_t action_channels;  /* 256 */
  float   tok_min;          /* -1 */
  float   tok_max;          /* +1 */
} lram_config;

typedef struct lram_engine lram_engine; /* opaque */

/* Text of the last error raised on this thread ("" if none). */
const char* lram_last_error(void);

/* ABI version the library was built with. */
int32_t lram_abi_version(void);
/* 64 hex digits: sha256 over the library's sources, headers and compiler flags (lram_amd/build.py::source_hash).  The
 * Python side rebuilds when it differs from the checked-out tree; tests/conftest.py and bench.py assert / report it. */
const char* lram_build_id(void);

/* Create an engine on HIP device `device`.  Replaces the construction of the policy's encoder,
 * xLSTMEncoder.__init__ / MambaEncoder.__init__ (decision_xlstm.py:119-136, decision_mamba.py:52-107). */
int32_t lram_create(const lram_config* cfg, int32_t device, lram_engine** out);

/* Destroy the engine and free all device memory it owns. */
int32_t lram_destroy(lram_engine* e);

/* Upload one weight tensor (host fp32, `numel` elements) under an engine-side name.  Names and shapes
 * are listed by lram_amd/weights.py::engine_layout (derived from the reference checkpoint keys,
 * `policy.state_dict()`, src/algos/decision_transformer_sb3.py:1246-1280).  Replaces load_state_dict. */
int32_t lram_set_weight(lram_engine* e, const char* name, const float* host_data, size_t numel);

/* Check that every weight the configured model needs is present with the right size; resolves the
 * kernel-side pointer tables.  Must be called once after the last lram_set_weight. */
int32_t lram_finalize(lram_engine* e);

/* Allocate (or re-allocate) zeroed recurrent state and activation workspace for `batch` env slots.
 * Replaces `past_key_values = None` / InferenceParams(max_batch_size) (decision_mamba.py:33-38). */
int32_t lram_state_alloc(lram_engine* e, int32_t batch);

/* Bytes of recurrent state held per env slot (the S_env of SURVEY.md 8d). */
int64_t lram_state_bytes_per_env(const lram_engine* e);

/* Zero the recurrent state of the env slots whose mask byte is non-zero (device uint8[batch]); NULL
 * resets every slot.  Replaces `model.past_key_values = None; model.inference_params.reset()`
 * (src/callbacks/evaluation.py:116-119,247-250). */
int32_t lram_reset(lram_engine* e, const uint8_t* dev_env_mask, void* stream);

/* One env-step for all `batch` slots: embed (state, rtg, reward) -> embed_ln -> tokens_per_step
 * recurrent token steps through the block stack -> post norm -> action head -> argmax -> inv_tokenize.
 * Replaces policy.forward(..., use_inference_cache=True, past_key_values=...) as called from
 * get_action_pred (src/algos/discrete_decision_transformer_sb3.py:60-68) for every env at once.
 *   dev_obs        device float[batch, state_dim]  (zero-padded obs, decision_xlstm.py:16-19), or, when
 *                  obs_is_embedding != 0, device float[batch, d_model] = embed_image(obs/255) computed by
 *                  the caller (ImpalaCNN stays in PyTorch/MIOpen)
 *   dev_rtg        device float[batch]             returns-to-go (already divided by reward_scale)
 *   dev_reward     device float[batch]             reward token (0 in the reference loop)
 *   dev_reset_mask device uint8[batch] or NULL     slots to reset before this step
 *   discrete       0: continuous head (argmax over n_vocab per action dim, de-tokenised to fp32)
 *                  1: discrete head (argmax over the first n_discrete logits of action dim 0)
 *   dev_actions    device float[batch, act_dim]    out; discrete: column 0 holds the action index
 *   dev_tokens     device int32[batch, act_dim] or NULL   out; raw argmax token ids */
int32_t lram_step(lram_engine* e, const float* dev_obs, int32_t obs_is_embedding, const float* dev_rtg,
                  const float* dev_reward, const uint8_t* dev_reset_mask, int32_t discrete,
                  float* dev_actions, int32_t* dev_tokens, void* stream);

/* Context (re-)prime: `timesteps` consecutive env-steps of stored trajectory data in ONE call, without the action
 * head on the intermediate steps -- the recurrent counterpart of feeding `eval_context_len` timesteps through
 * policy.forward without a cache (src/algos/decision_transformer_sb3.py:628-640,663-666: after
 * `reset_inf_cache_freq` fires the reference re-embeds the context but keeps only the last 3 tokens, SURVEY 3.5 Q5;
 * `chunkwise_step`, decision_xlstm.py:158-159).  Equal to `timesteps` sequential lram_step calls up to
 * fp32 rounding, but processed in chunks: each block's recurrent state is read and written once per chunk.
 * xLSTM with a head dim that is a multiple of 128 (the 16M and 206M geometries): up to 21 timesteps (63 tokens) per
 * chunk through the chunkwise matrix-core kernels (csrc/mlstm_chunk.hip: intra-chunk attention form + one rank-T
 * update of C); otherwise, or with LRAM_PREFILL_CHUNK=0 in the environment at lram_create, 4 timesteps (12 tokens)
 * per chunk through the token-sequential kernels.  The first long prefill grows the activation workspace to 64
 * tokens per env slot (device-synchronising, once).  From two chunks on, three chunks are in flight on engine-owned
 * streams (block i of chunk c + 1 waits for block i of chunk c only): the engine then holds two further copies of that
 * workspace and, for raw observations, the [batch, timesteps, d_model] state embeddings of the context (allocated on the
 * first such call; one chunk at a time if the device has less than 2 GiB to spare, or with LRAM_PREFILL_CHUNK=3).  Results
 * are bit-identical to one chunk at a time and are on `stream` when the call returns.
 *   dev_obs_seq    device float[batch, timesteps, state_dim] (or [batch, timesteps, d_model] embeddings)
 *   dev_rtg_seq, dev_reward_seq   device float[batch, timesteps]
 *   dev_reset_mask applied before the first timestep;  dev_actions (nullable): action at the LAST timestep. */
int32_t lram_prefill(lram_engine* e, const float* dev_obs_seq, int32_t obs_is_embedding, const float* dev_rtg_seq,
                     const float* dev_reward_seq, int32_t timesteps, const uint8_t* dev_reset_mask,
                     int32_t discrete, float* dev_actions, int32_t* dev_tokens, void* stream);

/* Encoder-only operator: inputs_embeds[batch, tokens, d_model] -> last_hidden_state of the same shape
 * (after post_blocks_norm / norm_f), state advanced by `tokens` (1..4, 6, 9 or 12; any count in 13..64 as well on
 * xLSTM geometries the chunkwise kernels cover, see lram_prefill).  This is the exact plug point of
 * `self.encoder(**encoder_inputs)` (online_decision_transformer_model.py:448). */
int32_t lram_encoder_step(lram_engine* e, const float* dev_inputs_embeds, int32_t tokens,
                          const uint8_t* dev_reset_mask, float* dev_hidden_out, void* stream);

/* Debug/parity taps of the last lram_step: copies into caller device buffers when non-NULL.
 *   dev_tokens_embed float[batch, tokens_per_step, d_model]  embed_ln output (batches of up to 1024 env slots: larger
 *                                                             ones skip the per-step copy this tap costs; pass NULL)
 *   dev_hidden       float[batch, tokens_per_step, d_model]  encoder output (after the final norm)
 *   dev_logits       float[batch, act_dim * n_vocab]         action_net output */
int32_t lram_get_taps(lram_engine* e, float* dev_tokens_embed, float* dev_hidden, float* dev_logits,
                      void* stream);

/* Recurrent-state tensors in the reference's `past_key_values` layout.  `which`:
 *   xLSTM mLSTM block: 0 = C [B,NH,DH,DH]  1 = n [B,NH,DH,1]  2 = m [B,NH,1,1]  3 = conv [B,K,inner]
 *   xLSTM sLSTM block: 0 = slstm_state [4,B,D] (y,c,n,m)      3 = conv [B,K,D]
 *   Mamba layer      : 0 = ssm_state [B,d_inner,d_state]      3 = conv_state [B,d_inner,d_conv]
 * lram_state_numel returns the element count (0 if the tensor does not exist for that block).
 * Invariant checked at import: an sLSTM hidden plane (slstm_state[0] = y) must satisfy |y| < 16 (no NaN) wherever the step runs
 * its recurrent products on binary16 planes of 2^12 y (the default with f16x2 projections; the recurrence itself only produces
 * |y| < 1).  lram_state_import refuses anything else (one reduction + host synchronisation on that tensor); LRAM_SLSTM_SEQ=2 or
 * LRAM_GEMM=f32 select the exact-fp32 recurrence, which has no such limit. */
int64_t lram_state_numel(const lram_engine* e, int32_t block, int32_t which);
int32_t lram_state_export(lram_engine* e, int32_t block, int32_t which, float* dev_dst, void* stream);
int32_t lram_state_import(lram_engine* e, int32_t block, int32_t which, const float* dev_src, void* stream);

/* Capture the kernel sequence of lram_step for the current batch / pointer set into a hipGraph and
 * replay it on later identical calls (launch-latency removal for small batches). enable = 0 disables. */
int32_t lram_set_graph_mode(lram_engine* e, int32_t enable);

/* State representation of the mLSTM matrix memory for lram_step / short lram_encoder_step calls.
 *   mode 0: C_t is materialised -- read and rewritten once per env-step (the reference's representation).
 *   mode 1: lazy -- C_t = g * C_base + sum_j c_j khat_j v_j^T.  A step reads C_base once and appends its tokens to a
 *           window of up to 48 tokens; C_base is rewritten ("folded", fp32 matrix cores) once every `fold_period` steps
 *           per env (0 = keep the current period, default 13; the folds of different envs are staggered).  Same
 *           mathematics as recurrent_step_stabilized_simple ([3P]), ~0.6x the HBM bytes of the
 *           materialised update.  Needs an xLSTM head dim that is a multiple of 128.
 *   mode 2 (default): lazy where one block's matrix memory over the batch is at least 128 MiB (16M: 128 env slots, 206M: 21), else materialised.
 * lram_state_export / import, lram_prefill, hipGraph mode and calls with more than 4 tokens fold every pending window
 * first, so they always see the reference state layout.  LRAM_STATE=eager|lazy|auto (and LRAM_LAZY_PERIOD) in the
 * environment at lram_create set the initial mode.  lram_profile_end in lazy mode: total_ms includes the fold launches,
 * n_launches counts the read passes. */
int32_t lram_set_state_mode(lram_engine* e, int32_t mode, int32_t fold_period);
/* 1 when the lazy representation is in effect for the allocated batch, else 0. */
int32_t lram_get_state_mode(const lram_engine* e);
/* Lazy representation, looked at WITHOUT folding (an export would change the fold schedule of the run it observes): copies
 * for mLSTM block `block`
 *   which 0: the scale g of C_base accumulated since the env's last fold   float[B, NH]
 *   which 1: the stabiliser state m                                         float[B, NH]
 *   which 2: pending window tokens per env (as floats)                       float[B]
 * into dev_dst, as of the last completed step.  Evidence hook of the long-horizon parity tests (the range g and m cover over
 * a 1000-step episode: evaluation.py:130-177 never clears the cache inside an episode); fails in materialised mode. */
int32_t lram_lazy_peek(lram_engine* e, int32_t block, int32_t which, float* dev_dst, void* stream);

/* Micro-batch pipeline (xLSTM): the env slots are processed as `n` slices on engine-owned HIP streams; the
 * HBM-bound matrix-memory kernels of all slices run back to back on one stream while the other slices'
 * fp32-MFMA projections overlap them.  n = 1 disables it, 0 = automatic (2 slices where one mLSTM block's matrix memory over the batch reaches 512 MiB -- 16M from 512 env slots,
 * 206M from 82 -- and for Mamba from 1024 env slots), max 8.
 * Results do not depend on n (envs are independent). */
int32_t lram_set_micro_batches(lram_engine* e, int32_t n);

/* Reference-trajectory modes of the Mamba agent (Mamba engines only; defaults 1 / 0 = one state advance per
 * env-step, a reset empties every layer).
 *   mamba_repeat R > 1: DiscreteDecisionMamba.get_action_pred (src/algos/decision_mamba.py:107-122) calls the policy
 *     once per action dim with the inference cache on, so the conv / ssm state advances R = env_act_dim times per
 *     env-step on the same (state, rtg, reward) tokens and action dim i is the prediction of forward i.  lram_step
 *     then runs R forwards (reset mask applied before the first); columns >= R hold forward R - 1.  Discrete heads
 *     (act_dim 1 in the reference) always take one forward.
 *   stale_state != 0: InferenceParams.reset() (src/algos/decision_mamba.py:20-25) only zeroes seqlen_offset, and
 *     MambaEncoder.forward bumps it inside the layer loop (src/algos/models/decision_mamba.py:130-149): layer 0 runs
 *     its full scan from an empty state, layers >= 1 step on from the previous episode's cache.  The reset mask of
 *     lram_step / lram_prefill and lram_reset then re-initialise layer 0 only. */
int32_t lram_set_compat_mode(lram_engine* e, int32_t mamba_repeat, int32_t stale_state);
int32_t lram_get_compat_mode(const lram_engine* e, int32_t* mamba_repeat, int32_t* stale_state);

/* Per-kernel timing of the recurrent step, measured with HIP events on the stream the kernels are
 * launched on.  lram_profile_begin arms it; every later lram_step records one (start, stop) event pair
 * around the mLSTM cell-update launches (xLSTM) or the selective-state-update launches (Mamba).
 * lram_profile_end synchronises and returns total milliseconds and number of launches timed. */
int32_t lram_profile_begin(lram_engine* e);
/* The same, timing only every n-th lram_step (the first one after this call included): each timed launch is bracketed by two
 * event packets on the state-pass queue, and at 21 launches per step that bookkeeping costs the headline 1.3-1.8 % when every
 * step carries it (profiles/r05_ab_kernel_timing.txt).  lram_profile_end* report the sampled launches only. */
int32_t lram_profile_begin_sampled(lram_engine* e, int32_t every_n_steps);
int32_t lram_profile_end(lram_engine* e, double* total_ms, int64_t* n_launches);
/* The same, with the lazy mode's fold launches (timed on their own stream) reported apart from the state-pass
 * launches, so that each figure can be held against the per-kernel averages of a rocprofv3 --kernel-trace run. */
int32_t lram_profile_end_split(lram_engine* e, double* main_ms, int64_t* n_main, double* aux_ms, int64_t* n_aux);

/* Measurement aid: how many projection launches each kernel family of the dispatcher (engine.hip::gemm) has served since
 * lram_create or the last call with `reset` != 0 -- out[0] f16x2, out[1] bf16x3, out[2] exact fp32 MFMA (tile / GEMV),
 * out[3] few-row fp32 kernel, and out[4..7] the fp32-equivalent FLOPs (2 M N K, summed) of the same four.  bench.py labels
 * its workload line from these instead of from the LRAM_GEMM environment variable.  (The reference has no counterpart:
 * nn.Linear calls at src/algos/models/decision_mamba.py:78-93 / [3P] xlstm proj_up / proj_down go to the vendor BLAS.) */
int32_t lram_gemm_counts(lram_engine* e, double* out8, int32_t reset);

/* Standalone kernel entry points used by tests and micro-benchmarks. */
/* C[M,N] = A[M,K] * W[N,K]^T (+ bias[N]) (+ residual C_in)   fp32, MFMA 32x32x2 f32 (exact k-ordered fma chain) */
int32_t lram_gemm_f32(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c,
                      int64_t ldc, const float* dev_bias, int32_t accumulate, int32_t m, int32_t n,
                      int32_t k, void* stream);
/* Same contract through the bf16x3 kernel the engine uses for its projections (each fp32 operand split into
 * three bf16 pieces, six bf16 MFMA products accumulated in fp32: fp32-level accuracy, not bit-identical to
 * lram_gemm_f32).  W must be contiguous [n, k], k a multiple of 8.  Splits W into a temporary and synchronises
 * the stream: test / micro-benchmark entry, not a hot-path call. */
int32_t lram_gemm_bf16x3(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c,
                         int64_t ldc, const float* dev_bias, int32_t accumulate, int32_t m, int32_t n,
                         int32_t k, void* stream);
/* Same contract through the few-row kernel (gemm_f32.hip::gemm_skinny_kernel: one 32 x 32 tile of the exact fp32 matrix
 * instruction per workgroup, K split over the waves, operands straight into registers; the engine takes it for GEMMs of
 * 9 .. 384 operand rows and K <= 1024).  k >= 32, k / lda / ldw multiples of 4, 16-byte aligned operands.  Test entry. */
int32_t lram_gemm_skinny(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c,
                         int64_t ldc, const float* dev_bias, int32_t accumulate, int32_t m, int32_t n,
                         int32_t k, void* stream);
/* Same contract through the narrow-output kernel (gemm_narrow.hip: 16 rows x all n <= 96 columns per workgroup, exact fp32 matrix
 * instruction, A staged coalesced through LDS, W packed in fragment order; the engine takes it for Mamba's x_proj --
 * mamba_ssm.Mamba.x_proj, reached from src/algos/models/decision_mamba.py:130-147 -- from 256 operand rows).  W contiguous
 * [n, k], k a multiple of 64 and >= 256, lda a multiple of 4, accumulate must be 0.  Packs W into a temporary and synchronises
 * the stream: test / micro-benchmark entry. */
int32_t lram_gemm_narrow(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c,
                         int64_t ldc, const float* dev_bias, int32_t accumulate, int32_t m, int32_t n,
                         int32_t k, void* stream);
/* The same kernel family with the products formed as in lram_gemm_f16x2 (rows scaled by a power of two, two binary16 pieces per
 * operand, hi*hi + hi*lo + lo*hi on the f16 matrix instruction, exact un-scaling): the form the engine runs for x_proj wherever
 * its projections run as f16x2 and the conv kernel hands the operand's row maxima over.  Same argument rules as lram_gemm_narrow;
 * splits W and takes A's row maxima in temporaries, synchronises the stream: test / micro-benchmark entry. */
int32_t lram_gemm_narrow_f16x2(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c,
                               int64_t ldc, const float* dev_bias, int32_t accumulate, int32_t m, int32_t n,
                               int32_t k, void* stream);
/* Same contract through the f16x2 kernel (each fp32 operand row scaled by a power of two and split into two binary16
 * pieces, three f16 MFMA products accumulated in fp32, exact un-scaling: fp32-level accuracy at half the matrix-core
 * work of bf16x3; gemm_f16x2.hip).  Splits W and computes A's row scales into temporaries, synchronises the stream:
 * test / micro-benchmark entry. */
int32_t lram_gemm_f16x2(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c,
                        int64_t ldc, const float* dev_bias, int32_t accumulate, int32_t m, int32_t n,
                        int32_t k, void* stream);
/* The f16x2 arithmetic with the activation operand pre-split too (gemm_f16x2p.hip): A goes through the row-split kernel
 * (two f16 planes of the row-scaled rows + inverse scales -- what the engine's norm kernels write directly), both operands
 * are then staged global -> LDS by DMA and the inner loop is MFMAs only.  k a multiple of 32, <= 3072.  Same accuracy
 * contract as lram_gemm_f16x2 (the pieces and the products are the same; the result is bit-identical to it).
 * Test / micro-benchmark entry. */
int32_t lram_gemm_f16x2_presplit(const float* dev_a, int64_t lda, const float* dev_w, int64_t ldw, float* dev_c,
                                 int64_t ldc, const float* dev_bias, int32_t accumulate, int32_t m, int32_t n,
                                 int32_t k, void* stream);
/* Image observations: uint8 frames [batch, channels, height, width] -> state-token embeddings [batch, d_model]
 * through the IMPALA CNN (3 x [conv3x3 -> maxpool(3,2,1) -> 2 residual blocks], 16/32/32 channels, ReLU, flatten,
 * Linear, ReLU).  Replaces `self.embed_image(state.float() / 255)` (online_decision_transformer_model.py:523-526;
 * module src/algos/models/image_encoders.py:10-131, built at multi_domain_discrete_dt_model.py:43-46).  Needs the
 * `embed_image.*` tensors (reference names) uploaded before lram_finalize; the result is passed to lram_step /
 * lram_prefill with obs_is_embedding = 1. */
int32_t lram_embed_images(lram_engine* e, const uint8_t* dev_images, int32_t channels, int32_t height, int32_t width,
                          float* dev_embeddings, void* stream);
/* One env-step from image observations: lram_embed_images + lram_step(obs_is_embedding = 1) as ONE call -- what the reference's
 * forward does with image states (`compute_inputs`: `state_embeddings = self.embed_image(states / 255)` then the token stack,
 * online_decision_transformer_model.py:463-530).  Same results as the two calls; the engine runs the CNN per env slice on the
 * slice's own stream and starts the step's state-pass work that does not depend on the observation (the lazy matrix memory's
 * folds) beside it.  Frames uint8 [batch, channels, height, width]; other arguments as lram_step. */
int32_t lram_step_images(lram_engine* e, const uint8_t* dev_images, int32_t channels, int32_t height, int32_t width,
                         const float* dev_rtg, const float* dev_reward, const uint8_t* dev_reset_mask, int32_t discrete,
                         float* dev_actions_out, int32_t* dev_tokens_out, void* stream);

/* Observation front end on the device: native obs [batch, n_native] -> model input [batch, state_dim].
 * dev_inv_index == NULL: zero-pad (DecisionXLSTM.pad_inputs, src/algos/decision_xlstm.py:16-19); otherwise
 * int32[state_dim] giving, per output dim, the native column it is filled from or -1 (DMControl full-space
 * mapping, src/envs/dmcontrol_utils.py:35-59; Mimicgen, mimicgen_utils.py:190-197).  Optional fused
 * normalisation (x - mean) / std over the padded vector (src/algos/decision_transformer_sb3.py:650-651). */
int32_t lram_pad_obs(const float* dev_native, int32_t n_native, const int32_t* dev_inv_index, const float* dev_mean,
                     const float* dev_std, float* dev_out, int32_t batch, int32_t state_dim, void* stream);

/* Diagnostic: runs the mLSTM front-end kernel beside a bf16x3 GEMM on a second stream `iters` times and counts
 * output elements that differ from a solo run (must be 0; see lram_amd/csrc/selftest.hip for the gfx950
 * packed-fp32 / bf16-MFMA co-execution hazard this guards against). */
int32_t lram_selftest_concurrent(int32_t iters, int64_t* n_diff);
/* STREAM-like device copy (float4), used by bench.py to measure the achievable HBM rate on the box. */
int32_t lram_stream_copy(float* dev_dst, const float* dev_src, size_t numel, void* stream);

/* Measurement aid: read-only stream over `numel` floats (a multiple of 1 Mi) with the access shape of the lazy mLSTM read
 * pass (mlstm_lazy.hip: whole 1 KiB rows per wave, several rows in flight per lane, non-temporal loads) and none of its
 * arithmetic; per-workgroup sums are accumulated into dev_sink (1024 floats).  The practical ceiling for "read the state
 * once" that bench.py reports beside the read pass's rate. */
int32_t lram_stream_read(const float* dev_buf, size_t numel, float* dev_sink, void* stream);

/* Measurement aid: in-place read-modify-write stream (x *= 1) over `numel` floats (a multiple of 65536) with the
 * access pattern of the mLSTM cell kernel and none of its arithmetic -- the practical ceiling for "read the state
 * once, write it once" that bench.py reports beside the cell kernel's rate. */
int32_t lram_stream_rmw(float* dev_buf, size_t numel, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRAM_HIP_H */
