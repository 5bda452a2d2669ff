"""ctypes binding of liblram_hip.so (C ABI: include/lram_hip.h) and the `Engine` host object.

PyTorch is used for device memory, streams and (elsewhere) torch.distributed only: every tensor handed
to the library is passed as a raw device pointer.  There is no CPU / eager fallback: if the HIP library
is missing or fails to load, construction raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from .config import ModelSpec
from .weights import engine_layout

LRAM_ABI_VERSION = 1
LRAM_MAX_BLOCKS = 64
_LIB_NAME = "liblram_hip.so"


class LramConfig(ctypes.Structure):
    """Field-for-field mirror of `lram_config` in include/lram_hip.h."""
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("backbone", ctypes.c_int32), ("d_model", ctypes.c_int32),
        ("n_blocks", ctypes.c_int32), ("tokens_per_step", ctypes.c_int32), ("pred_token", ctypes.c_int32),
        ("n_heads", ctypes.c_int32), ("conv_k", ctypes.c_int32), ("qkv_blocksize", ctypes.c_int32),
        ("inner", ctypes.c_int32), ("ffn_dim", ctypes.c_int32), ("block_is_slstm", ctypes.c_int32 * LRAM_MAX_BLOCKS),
        ("norm_is_rms", ctypes.c_int32), ("ln_eps", ctypes.c_float),
        ("d_inner", ctypes.c_int32), ("d_state", ctypes.c_int32), ("d_conv", ctypes.c_int32),
        ("dt_rank", ctypes.c_int32), ("norm_eps", ctypes.c_float),
        ("state_dim", ctypes.c_int32), ("act_dim", ctypes.c_int32), ("n_vocab", ctypes.c_int32),
        ("n_discrete", ctypes.c_int32), ("action_channels", ctypes.c_int32),
        ("tok_min", ctypes.c_float), ("tok_max", ctypes.c_float),
    ]


# name -> (restype, argtypes); every symbol include/lram_hip.h declares
_VP = ctypes.c_void_p
_SYMBOLS = {
    "lram_last_error": (ctypes.c_char_p, []),
    "lram_abi_version": (ctypes.c_int32, []),
    "lram_build_id": (ctypes.c_char_p, []),
    "lram_create": (ctypes.c_int32, [ctypes.POINTER(LramConfig), ctypes.c_int32, ctypes.POINTER(_VP)]),
    "lram_destroy": (ctypes.c_int32, [_VP]),
    "lram_set_weight": (ctypes.c_int32, [_VP, ctypes.c_char_p, _VP, ctypes.c_size_t]),
    "lram_finalize": (ctypes.c_int32, [_VP]),
    "lram_state_alloc": (ctypes.c_int32, [_VP, ctypes.c_int32]),
    "lram_state_bytes_per_env": (ctypes.c_int64, [_VP]),
    "lram_reset": (ctypes.c_int32, [_VP, _VP, _VP]),
    "lram_step": (ctypes.c_int32, [_VP, _VP, ctypes.c_int32, _VP, _VP, _VP, ctypes.c_int32, _VP, _VP, _VP]),
    "lram_prefill": (ctypes.c_int32, [_VP, _VP, ctypes.c_int32, _VP, _VP, ctypes.c_int32, _VP, ctypes.c_int32, _VP, _VP,
                                      _VP]),
    "lram_encoder_step": (ctypes.c_int32, [_VP, _VP, ctypes.c_int32, _VP, _VP, _VP]),
    "lram_get_taps": (ctypes.c_int32, [_VP, _VP, _VP, _VP, _VP]),
    "lram_state_numel": (ctypes.c_int64, [_VP, ctypes.c_int32, ctypes.c_int32]),
    "lram_state_export": (ctypes.c_int32, [_VP, ctypes.c_int32, ctypes.c_int32, _VP, _VP]),
    "lram_state_import": (ctypes.c_int32, [_VP, ctypes.c_int32, ctypes.c_int32, _VP, _VP]),
    "lram_set_graph_mode": (ctypes.c_int32, [_VP, ctypes.c_int32]),
    "lram_set_micro_batches": (ctypes.c_int32, [_VP, ctypes.c_int32]),
    "lram_set_compat_mode": (ctypes.c_int32, [_VP, ctypes.c_int32, ctypes.c_int32]),
    "lram_get_compat_mode": (ctypes.c_int32, [_VP, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "lram_profile_begin": (ctypes.c_int32, [_VP]),
    "lram_profile_begin_sampled": (ctypes.c_int32, [_VP, ctypes.c_int32]),
    "lram_profile_end": (ctypes.c_int32, [_VP, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "lram_profile_end_split": (ctypes.c_int32, [_VP, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "lram_gemm_f32": (ctypes.c_int32, [_VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_gemm_skinny": (ctypes.c_int32, [_VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_gemm_narrow": (ctypes.c_int32, [_VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_gemm_narrow_f16x2": (ctypes.c_int32, [_VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_gemm_bf16x3": (ctypes.c_int32, [_VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_gemm_f16x2": (ctypes.c_int32, [_VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_gemm_f16x2_presplit": (ctypes.c_int32, [_VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP, ctypes.c_int64, _VP,
                                                  ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_embed_images": (ctypes.c_int32, [_VP, _VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP, _VP]),
    "lram_step_images": (ctypes.c_int32, [_VP, _VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _VP, _VP, _VP, ctypes.c_int32,
                         _VP, _VP, _VP]),
    "lram_set_state_mode": (ctypes.c_int32, [_VP, ctypes.c_int32, ctypes.c_int32]),
    "lram_get_state_mode": (ctypes.c_int32, [_VP]),
    "lram_lazy_peek": (ctypes.c_int32, [_VP, ctypes.c_int32, ctypes.c_int32, _VP, _VP]),
    "lram_stream_rmw": (ctypes.c_int32, [_VP, ctypes.c_size_t, _VP]),
    "lram_stream_read": (ctypes.c_int32, [_VP, ctypes.c_size_t, _VP, _VP]),
    "lram_gemm_counts": (ctypes.c_int32, [_VP, ctypes.POINTER(ctypes.c_double), ctypes.c_int32]),
    "lram_pad_obs": (ctypes.c_int32, [_VP, ctypes.c_int32, _VP, _VP, _VP, _VP, ctypes.c_int32, ctypes.c_int32, _VP]),
    "lram_selftest_concurrent": (ctypes.c_int32, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]),
    "lram_stream_copy": (ctypes.c_int32, [_VP, _VP, ctypes.c_size_t, _VP]),
}

_lib = None


def library_path() -> str:
    """The in-tree library; LRAM_LIB_VARIANT=<name> picks csrc/_variants/<name>.so instead -- A/B measurements of two
    builds inside one GPU call (scripts/gpu_ab.sh), never set in tests or by the driver."""
    here = os.path.dirname(os.path.abspath(__file__))
    variant = os.environ.get("LRAM_LIB_VARIANT")
    if variant:
        return os.path.join(here, "csrc", "_variants", variant + ".so")
    return os.path.join(here, "csrc", _LIB_NAME)


def load_library():
    """dlopen the in-tree HIP library and bind every declared symbol.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the engine.")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.lram_abi_version() != LRAM_ABI_VERSION:
        raise RuntimeError("liblram_hip.so ABI version mismatch; rebuild the library")
    _lib = lib
    return lib


class LramError(RuntimeError):
    pass


def _check(lib, rc):
    if rc != 0:
        raise LramError(lib.lram_last_error().decode("utf-8", "replace"))


def make_config(spec: ModelSpec) -> LramConfig:
    c = LramConfig()
    c.abi_version = LRAM_ABI_VERSION
    c.backbone = 1 if spec.backbone == "mamba" else 0
    c.d_model, c.n_blocks = spec.d_model, spec.n_blocks
    c.tokens_per_step, c.pred_token = spec.tokens_per_step, spec.pred_token
    c.n_heads, c.conv_k, c.qkv_blocksize = spec.n_heads, spec.conv_k, spec.qkv_blocksize
    c.inner, c.ffn_dim = spec.inner, spec.ffn_dim
    if spec.n_blocks > LRAM_MAX_BLOCKS:
        raise ValueError(f"n_blocks {spec.n_blocks} > {LRAM_MAX_BLOCKS}")
    for i in spec.slstm_at:
        c.block_is_slstm[i] = 1
    c.norm_is_rms = int(spec.rms_norm)
    c.ln_eps = spec.ln_eps
    c.d_inner, c.d_state, c.d_conv = spec.d_inner, spec.d_state, spec.d_conv
    c.dt_rank = int(spec.dt_rank) if spec.backbone == "mamba" else 0
    c.norm_eps = spec.norm_eps
    c.state_dim, c.act_dim, c.n_vocab = spec.state_dim, spec.act_dim, spec.n_vocab
    c.n_discrete, c.action_channels = spec.n_discrete, spec.action_channels
    c.tok_min, c.tok_max = -1.0, 1.0
    return c


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _chk_dev(t: torch.Tensor, dtype, shape, device, name):
    if t.device != device or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous {dtype} tensor of shape {tuple(shape)} on {device}, got "
                         f"{t.dtype} {tuple(t.shape)} on {t.device} (contiguous={t.is_contiguous()})")


class Engine:
    """One engine per GPU: weights + per-env recurrent state resident in HBM, one batched env-step per call.

    Mirrors what the reference keeps on the agent (`policy`, `past_key_values` / `inference_params`,
    src/algos/decision_transformer_sb3.py:86-104, src/algos/decision_mamba.py:29-38) for B envs at once."""

    def __init__(self, spec: ModelSpec, state_dict: Dict[str, torch.Tensor], batch: int, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("lram_amd.Engine needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = load_library()
        self.spec = spec
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.batch = 0
        self._h = ctypes.c_void_p()
        cfg = make_config(spec)
        _check(self.lib, self.lib.lram_create(ctypes.byref(cfg), self.device.index, ctypes.byref(self._h)))
        self.load_weights(state_dict)
        self.alloc(batch)

    # -- lifetime --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.lram_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights / state -------------------------------------------------------------------------
    def load_weights(self, state_dict: Dict[str, torch.Tensor]):
        packed = engine_layout(self.spec, state_dict)
        for name, t in packed.items():
            t = t.contiguous()
            _check(self.lib, self.lib.lram_set_weight(self._h, name.encode(), ctypes.c_void_p(t.data_ptr()), t.numel()))
        _check(self.lib, self.lib.lram_finalize(self._h))

    def alloc(self, batch: int):
        _check(self.lib, self.lib.lram_state_alloc(self._h, int(batch)))
        self.batch = int(batch)
        B, A = self.batch, self.spec.act_dim
        self._actions = torch.empty(B, A, dtype=torch.float32, device=self.device)
        self._tokens = torch.empty(B, A, dtype=torch.int32, device=self.device)

    def state_bytes_per_env(self) -> int:
        return int(self.lib.lram_state_bytes_per_env(self._h))

    def reset(self, env_mask: Optional[torch.Tensor] = None):
        """Zero recurrent state of masked env slots (all when None): `past_key_values = None`."""
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            _chk_dev(env_mask, torch.uint8, (self.batch,), self.device, "env_mask")
        _check(self.lib, self.lib.lram_reset(self._h, _ptr(env_mask), _stream_ptr(self.device)))

    # -- the hot path ----------------------------------------------------------------------------
    def step(self, obs: torch.Tensor, rtg: torch.Tensor, reward: torch.Tensor,
             reset_mask: Optional[torch.Tensor] = None, discrete: bool = False, obs_is_embedding: bool = False,
             out_actions: Optional[torch.Tensor] = None, out_tokens: Optional[torch.Tensor] = None):
        """One env-step for all slots.  Inputs are device tensors (already resident in HBM).
        Returns (actions float32 [B, act_dim], tokens int32 [B, act_dim]); valid once the current
        stream has executed.  discrete=True: column 0 holds the action index."""
        B, spec = self.batch, self.spec
        _chk_dev(obs, torch.float32, (B, spec.d_model if obs_is_embedding else spec.state_dim), self.device, "obs")
        _chk_dev(rtg, torch.float32, (B,), self.device, "rtg")
        _chk_dev(reward, torch.float32, (B,), self.device, "reward")
        if reset_mask is not None:
            _chk_dev(reset_mask, torch.uint8, (B,), self.device, "reset_mask")
        actions = self._actions if out_actions is None else out_actions
        tokens = self._tokens if out_tokens is None else out_tokens
        _chk_dev(actions, torch.float32, (B, spec.act_dim), self.device, "out_actions")
        _chk_dev(tokens, torch.int32, (B, spec.act_dim), self.device, "out_tokens")
        _check(self.lib, self.lib.lram_step(self._h, _ptr(obs), int(obs_is_embedding), _ptr(rtg), _ptr(reward),
                                            _ptr(reset_mask), int(discrete), _ptr(actions), _ptr(tokens),
                                            _stream_ptr(self.device)))
        return actions, tokens

    def prefill(self, obs_seq: torch.Tensor, rtg_seq: torch.Tensor, reward_seq: torch.Tensor,
                reset_mask: Optional[torch.Tensor] = None, discrete: bool = False, obs_is_embedding: bool = False,
                want_action: bool = True):
        """L stored timesteps in one call ([B, L, state_dim], [B, L], [B, L]); == L sequential step() calls.
        Returns (actions, tokens) of the last timestep (None when want_action is False)."""
        B, spec = self.batch, self.spec
        L = obs_seq.shape[1]
        _chk_dev(obs_seq, torch.float32, (B, L, spec.d_model if obs_is_embedding else spec.state_dim), self.device,
                 "obs_seq")
        _chk_dev(rtg_seq, torch.float32, (B, L), self.device, "rtg_seq")
        _chk_dev(reward_seq, torch.float32, (B, L), self.device, "reward_seq")
        if reset_mask is not None:
            _chk_dev(reset_mask, torch.uint8, (B,), self.device, "reset_mask")
        act = self._actions if want_action else None
        tok = self._tokens if want_action else None
        _check(self.lib, self.lib.lram_prefill(self._h, _ptr(obs_seq), int(obs_is_embedding), _ptr(rtg_seq),
                                               _ptr(reward_seq), int(L), _ptr(reset_mask), int(discrete), _ptr(act),
                                               _ptr(tok), _stream_ptr(self.device)))
        return (act, tok) if want_action else (None, None)

    def embed_images(self, images: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 frames [B, C, H, W] -> state-token embeddings [B, d_model] through the IMPALA CNN kernels
        (`self.embed_image(state / 255)`, online_decision_transformer_model.py:523-526); feed the result to
        step(..., obs_is_embedding=True).  Needs the embed_image.* weights in the state dict."""
        B, D = self.batch, self.spec.d_model
        if images.dim() != 4:
            raise ValueError("images must be [B, C, H, W]")
        _chk_dev(images, torch.uint8, (B, *images.shape[1:]), self.device, "images")
        if out is None:
            out = torch.empty(B, D, dtype=torch.float32, device=self.device)
        _chk_dev(out, torch.float32, (B, D), self.device, "out")
        _check(self.lib, self.lib.lram_embed_images(self._h, _ptr(images), int(images.shape[1]), int(images.shape[2]),
                                                    int(images.shape[3]), _ptr(out), _stream_ptr(self.device)))
        return out

    def step_images(self, images: torch.Tensor, rtg: torch.Tensor, reward: torch.Tensor,
                    reset_mask: Optional[torch.Tensor] = None, discrete: bool = False,
                    out_actions: Optional[torch.Tensor] = None, out_tokens: Optional[torch.Tensor] = None):
        """One env-step from uint8 frames [B, C, H, W]: embed_images + step(obs_is_embedding=True) as one call (the reference's
        forward embeds image states inside `compute_inputs`, online_decision_transformer_model.py:463-530).  Same results as the two
        calls; the CNN runs per env slice beside the step's observation-independent state-pass work."""
        B, spec = self.batch, self.spec
        if images.dim() != 4:
            raise ValueError("images must be [B, C, H, W]")
        _chk_dev(images, torch.uint8, (B, *images.shape[1:]), self.device, "images")
        _chk_dev(rtg, torch.float32, (B,), self.device, "rtg")
        _chk_dev(reward, torch.float32, (B,), self.device, "reward")
        if reset_mask is not None:
            _chk_dev(reset_mask, torch.uint8, (B,), self.device, "reset_mask")
        actions = self._actions if out_actions is None else out_actions
        tokens = self._tokens if out_tokens is None else out_tokens
        _chk_dev(actions, torch.float32, (B, spec.act_dim), self.device, "out_actions")
        _chk_dev(tokens, torch.int32, (B, spec.act_dim), self.device, "out_tokens")
        _check(self.lib, self.lib.lram_step_images(self._h, _ptr(images), int(images.shape[1]), int(images.shape[2]),
                                                   int(images.shape[3]), _ptr(rtg), _ptr(reward), _ptr(reset_mask),
                                                   int(discrete), _ptr(actions), _ptr(tokens), _stream_ptr(self.device)))
        return actions, tokens

    def encoder_step(self, inputs_embeds: torch.Tensor, reset_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`self.encoder(inputs_embeds=[B,T,D], use_cache=True)` plug point (decision_xlstm.py:138-169)."""
        B, D = self.batch, self.spec.d_model
        T = inputs_embeds.shape[1]
        _chk_dev(inputs_embeds, torch.float32, (B, T, D), self.device, "inputs_embeds")
        if reset_mask is not None:
            _chk_dev(reset_mask, torch.uint8, (B,), self.device, "reset_mask")
        out = torch.empty_like(inputs_embeds)
        _check(self.lib, self.lib.lram_encoder_step(self._h, _ptr(inputs_embeds), int(T), _ptr(reset_mask), _ptr(out),
                                                    _stream_ptr(self.device)))
        return out

    def taps(self):
        """(embed_ln tokens [B,T,D], encoder hidden [B,T,D], logits [B, act_dim*n_vocab]) of the last step."""
        B, s = self.batch, self.spec
        hid = torch.empty(B, s.tokens_per_step, s.d_model, dtype=torch.float32, device=self.device)
        tok = torch.empty_like(hid) if B <= 1024 else None   # the token tap is kept for small batches only
        logits = torch.empty(B, s.act_dim * s.n_vocab, dtype=torch.float32, device=self.device)
        _check(self.lib, self.lib.lram_get_taps(self._h, _ptr(tok), _ptr(hid), _ptr(logits), _stream_ptr(self.device)))
        return tok, hid, logits

    # -- past_key_values-compatible state access -------------------------------------------------
    def _state_shape(self, block: int, which: int):
        s, B = self.spec, self.batch
        if s.backbone == "mamba":
            return {0: (B, s.d_inner, s.d_state), 3: (B, s.d_inner, s.d_conv)}.get(which)
        if block in s.slstm_at:
            return {0: (4, B, s.d_model), 3: (B, s.conv_k, s.d_model)}.get(which)
        dh = s.head_dim
        return {0: (B, s.n_heads, dh, dh), 1: (B, s.n_heads, dh, 1), 2: (B, s.n_heads, 1, 1),
                3: (B, s.conv_k, s.inner)}.get(which)

    def export_state_tensor(self, block: int, which: int) -> torch.Tensor:
        shape = self._state_shape(block, which)
        n = int(self.lib.lram_state_numel(self._h, block, which))
        if shape is None or n == 0:
            raise KeyError(f"no state tensor (block={block}, which={which})")
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        assert out.numel() == n
        _check(self.lib, self.lib.lram_state_export(self._h, block, which, _ptr(out), _stream_ptr(self.device)))
        return out

    def import_state_tensor(self, block: int, which: int, t: torch.Tensor):
        shape = self._state_shape(block, which)
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is None or tuple(t.shape) != tuple(shape):
            raise ValueError(f"state tensor (block={block}, which={which}) must have shape {shape}, got {tuple(t.shape)}")
        _check(self.lib, self.lib.lram_state_import(self._h, block, which, _ptr(t), _stream_ptr(self.device)))

    def export_past_key_values(self):
        """State in the reference's layout: xLSTM nested dict of tuples, Mamba
        {layer_idx: (conv_state, ssm_state)} (src/algos/decision_mamba.py:9-25)."""
        s = self.spec
        if s.backbone == "mamba":
            return {i: (self.export_state_tensor(i, 3), self.export_state_tensor(i, 0)) for i in range(s.n_blocks)}
        out = {}
        for i in range(s.n_blocks):
            if i in s.slstm_at:
                out[f"block_{i}"] = {"slstm_state": self.export_state_tensor(i, 0),
                                     "conv_state": (self.export_state_tensor(i, 3),)}
            else:
                out[f"block_{i}"] = {"mlstm_state": tuple(self.export_state_tensor(i, w) for w in (0, 1, 2)),
                                     "conv_state": (self.export_state_tensor(i, 3),)}
        return out

    def import_past_key_values(self, pkv):
        s = self.spec
        if s.backbone == "mamba":
            for i, (conv, ssm) in pkv.items():
                self.import_state_tensor(int(i), 3, conv)
                self.import_state_tensor(int(i), 0, ssm)
            return
        for i in range(s.n_blocks):
            blk = pkv[f"block_{i}"]
            if i in s.slstm_at:
                self.import_state_tensor(i, 0, blk["slstm_state"])
            else:
                for w, t in enumerate(blk["mlstm_state"]):
                    self.import_state_tensor(i, w, t)
            self.import_state_tensor(i, 3, blk["conv_state"][0])

    # -- launch-latency removal / measurement ----------------------------------------------------
    def set_graph_mode(self, enable: bool):
        _check(self.lib, self.lib.lram_set_graph_mode(self._h, int(enable)))

    def set_state_mode(self, mode, fold_period: int = 0):
        """mode: False / 0 / "eager" = materialised C (the reference's representation); True / 1 / "lazy" = read-once
        matrix memory with a pending-token window folded every `fold_period` steps; 2 / "auto" (default) = lazy where
        the state pass dominates (lram_set_state_mode)."""
        names = {"eager": 0, "materialised": 0, "lazy": 1, "auto": 2}
        m = names[mode] if isinstance(mode, str) else int(mode)
        _check(self.lib, self.lib.lram_set_state_mode(self._h, m, int(fold_period)))

    @property
    def state_mode(self) -> str:
        return "lazy" if self.lib.lram_get_state_mode(self._h) else "materialised"

    def lazy_peek(self, block: int, which: str) -> torch.Tensor:
        """Lazy representation looked at without folding: 'g' [B, NH] (scale of C_base since the last fold), 'm' [B, NH]
        (stabiliser state), 'pending' [B] (window tokens) -- lram_lazy_peek."""
        w = {"g": 0, "m": 1, "pending": 2}[which]
        shape = (self.batch,) if w == 2 else (self.batch, self.spec.n_heads)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        _check(self.lib, self.lib.lram_lazy_peek(self._h, block, w, _ptr(out), _stream_ptr(self.device)))
        return out

    def set_micro_batches(self, n: int):
        """Env slices pipelined on separate HIP streams (0 = auto, 1 = off); results are independent of n."""
        _check(self.lib, self.lib.lram_set_micro_batches(self._h, int(n)))

    def set_compat_mode(self, mamba_repeat: int = 1, stale_state: bool = False):
        """Reference-trajectory modes of the Mamba agent (lram_set_compat_mode):
        `mamba_repeat` forwards per env-step with action dim i read from forward i
        (src/algos/decision_mamba.py:107-122), `stale_state`: a reset re-initialises layer 0 only (:20-25)."""
        _check(self.lib, self.lib.lram_set_compat_mode(self._h, int(mamba_repeat), int(bool(stale_state))))

    @property
    def compat_mode(self):
        r, st = ctypes.c_int32(1), ctypes.c_int32(0)
        self.lib.lram_get_compat_mode(self._h, ctypes.byref(r), ctypes.byref(st))
        return {"mamba_repeat": int(r.value), "stale_state": bool(st.value)}

    def profile_begin_sampled(self, every_n_steps: int):
        """Time every n-th step only (lram_profile_begin_sampled): 1/n of the event bookkeeping on the state-pass queue."""
        _check(self.lib, self.lib.lram_profile_begin_sampled(self._h, int(every_n_steps)))

    def profile_begin(self):
        _check(self.lib, self.lib.lram_profile_begin(self._h))

    def profile_end(self):
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        _check(self.lib, self.lib.lram_profile_end(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


    def gemm_counts(self, reset: bool = False) -> dict:
        """Projection launches and fp32-equivalent FLOPs per kernel family of the dispatcher since the last reset
        (lram_gemm_counts): what the engine actually ran, whatever LRAM_GEMM says."""
        buf = (ctypes.c_double * 8)()
        _check(self.lib, self.lib.lram_gemm_counts(self._h, buf, 1 if reset else 0))
        names = ("f16x2", "bf16x3", "f32", "few_row_f32")
        return {n: {"launches": int(buf[i]), "flop": float(buf[4 + i])} for i, n in enumerate(names)}

    def profile_end_split(self):
        """(state-pass ms, state-pass launches, fold ms, fold launches) -- lram_profile_end_split."""
        m, a = ctypes.c_double(0.0), ctypes.c_double(0.0)
        nm, na = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self.lib, self.lib.lram_profile_end_split(self._h, ctypes.byref(m), ctypes.byref(nm), ctypes.byref(a),
                                                         ctypes.byref(na)))
        return m.value, nm.value, a.value, na.value


def gemm_f32(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, accumulate: bool = False, kernel: str = "f32") -> torch.Tensor:
    """out[M,N] = a[M,K] @ w[N,K]^T (+ bias) (+ out) through the library's fp32-MFMA kernel (kernel="f32") or the
    bf16x3 kernel (kernel="bf16x3") the engine uses for its projections."""
    lib = load_library()
    if kernel == "f16x2p8":   # the pre-split entry with the 8-phase 256 x 256 kernel forced (the entry re-reads the launch knobs)
        import os
        old = os.environ.get("LRAM_GEMM_TILE")
        os.environ["LRAM_GEMM_TILE"] = "256"
        try:
            return gemm_f32(a, w, bias, out, accumulate, "f16x2p")
        finally:
            if old is None:
                del os.environ["LRAM_GEMM_TILE"]
            else:
                os.environ["LRAM_GEMM_TILE"] = old
    fn = {"f32": lib.lram_gemm_f32, "bf16x3": lib.lram_gemm_bf16x3, 
          "f16x2": lib.lram_gemm_f16x2, "f16x2p": lib.lram_gemm_f16x2_presplit, "skinny": lib.lram_gemm_skinny,
          "narrow": lib.lram_gemm_narrow, "narrow16": lib.lram_gemm_narrow_f16x2}[kernel]
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _check(lib, fn(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out), out.stride(0), _ptr(bias),
                                  int(accumulate), M, N, K, _stream_ptr(a.device)))
    return out


def pad_obs(native: torch.Tensor, state_dim: int, inv_index: Optional[torch.Tensor] = None,
            mean: Optional[torch.Tensor] = None, std: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Device-side observation front end (lram_pad_obs): scatter / zero-pad to `state_dim` (+ normalise)."""
    lib = load_library()
    B, n = native.shape
    if out is None:
        out = torch.empty(B, state_dim, dtype=torch.float32, device=native.device)
    _check(lib, lib.lram_pad_obs(_ptr(native), n, _ptr(inv_index), _ptr(mean), _ptr(std), _ptr(out), B, state_dim,
                                 _stream_ptr(native.device)))
    return out


def stream_copy(dst: torch.Tensor, src: torch.Tensor):
    lib = load_library()
    _check(lib, lib.lram_stream_copy(_ptr(dst), _ptr(src), src.numel(), _stream_ptr(src.device)))


def stream_read(buf: torch.Tensor, sink: torch.Tensor):
    """Read-only stream with the lazy read pass's access shape (lram_stream_read): the practical HBM ceiling for reading
    the recurrent state once.  `sink`: 1024 floats on the same device."""
    lib = load_library()
    _check(lib, lib.lram_stream_read(_ptr(buf), buf.numel(), _ptr(sink), _stream_ptr(buf.device)))


def stream_rmw(buf: torch.Tensor):
    """In-place x *= 1 stream with the cell kernel's access pattern (lram_stream_rmw): the practical HBM ceiling for
    reading the recurrent state once and writing it once."""
    lib = load_library()
    _check(lib, lib.lram_stream_rmw(_ptr(buf), buf.numel(), _stream_ptr(buf.device)))
