"""Weights: reference checkpoint key layout, seeded initialisation, engine packing, SB3-zip loading.

Key names are those of the reference's `policy.state_dict()` (written by `save()`,
src/algos/decision_transformer_sb3.py:1246-1280; listed in SURVEY.md Appendix A).  Backbone sub-keys follow the
attribute names of the third-party `xlstm` / `mamba_ssm` modules the reference instantiates
(src/algos/models/decision_xlstm.py:130-133, src/algos/models/decision_mamba.py:78-94).
"""
from __future__ import annotations

import io
import math
import zipfile
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .config import ModelSpec

IMPALA_CHANNELS = (16, 32, 32)  # src/algos/models/image_encoders.py:39-43 (model_size 1)
# scheme "trained_like": mLSTM gate weight scales in units of 1/sqrt(fan_in) (measured on the 16M stack: input-gate
# pre-activations of standard deviation ~5, i.e. about +-16 over a trajectory; forget-gate ones ~1.3 around the 3..6 bias).
# Stacks deeper than 8 blocks take the smaller input-gate scale (pre-activations +-10, m still beyond 8): measured on the
# 20-block 206M stack (fp32 CPU oracle vs its float64 evaluation, 10 env-steps, worst hidden row relative to the largest
# entry): scale 3: 2.5e-4, 4: 1.6e-4, 5: 3.0e-3, 6: 6.5e-2 -- at 6 NO fp32 evaluation of the reference means anything (two fp32
# CPU runs of the reference would disagree in the second digit), so there is nothing to hold an engine to.
TRAINED_LIKE_IGATE, TRAINED_LIKE_IGATE_DEEP, TRAINED_LIKE_FGATE = 6.0, 4.0, 1.0


# ----------------------------------------------------------------------------------------------
# key layout
# ----------------------------------------------------------------------------------------------
def reference_layout(spec: ModelSpec, with_image_encoder: bool = False) -> Dict[str, Tuple[int, ...]]:
    """name -> shape of every reference state-dict entry the inference path reads."""
    D = spec.d_model
    lay: Dict[str, Tuple[int, ...]] = {
        "embed_state.weight": (D, spec.state_dim), "embed_state.bias": (D,),
        "embed_return.weight": (D, 1), "embed_return.bias": (D,),
        "embed_rewards.weight": (D, 1), "embed_rewards.bias": (D,),
        "embed_ln.weight": (D,), "embed_ln.bias": (D,),
        "action_net.0.weight": (spec.act_dim * spec.n_vocab, D), "action_net.0.bias": (spec.act_dim * spec.n_vocab,),
    }
    if with_image_encoder:
        cin = spec.image_shape[0]
        hw = spec.image_shape[1]
        for b, cout in enumerate(IMPALA_CHANNELS):
            p = f"embed_image.cnn.{b}."
            lay[p + "conv.weight"] = (cout, cin, 3, 3)
            lay[p + "conv.bias"] = (cout,)
            for r in range(2):
                for cv in range(2):
                    lay[f"{p}residual_{r}.conv_{cv}.weight"] = (cout, cout, 3, 3)
                    lay[f"{p}residual_{r}.conv_{cv}.bias"] = (cout,)
            cin = cout
            hw = (hw + 2 - 3) // 2 + 1  # MaxPool2d(3, 2, padding=1)
        lay["embed_image.linear.0.weight"] = (D, cin * hw * hw)
        lay["embed_image.linear.0.bias"] = (D,)
    if spec.backbone == "xlstm":
        inner, NH = spec.inner, spec.n_heads
        sdh = D // NH
        F = spec.ffn_dim
        norms = []
        for i in range(spec.n_blocks):
            p = f"encoder.layers.blocks.{i}."
            lay[p + "xlstm_norm.weight"] = (D,)
            norms.append((p + "xlstm_norm", D))
            if i in spec.slstm_at:
                x = p + "xlstm."
                lay[x + "conv1d.conv.weight"] = (D, 1, spec.conv_k)
                lay[x + "conv1d.conv.bias"] = (D,)
                for g in ("igate", "fgate", "zgate", "ogate"):
                    lay[x + g + ".weight"] = (NH, sdh, sdh)
                lay[x + "slstm_cell._recurrent_kernel_"] = (NH, sdh, 4, sdh)
                lay[x + "slstm_cell._bias_"] = (NH, 4, sdh)
                lay[x + "group_norm.weight"] = (D,)
                norms.append((x + "group_norm", D))
                lay[p + "ffn_norm.weight"] = (D,)
                norms.append((p + "ffn_norm", D))
                lay[p + "ffn.proj_up.weight"] = (2 * F, D)
                lay[p + "ffn.proj_down.weight"] = (D, F)
            else:
                x = p + "xlstm."
                lay[x + "proj_up.weight"] = (2 * inner, D)
                for n in ("q_proj", "k_proj", "v_proj"):
                    lay[x + n + ".weight"] = (inner // spec.qkv_blocksize, spec.qkv_blocksize, spec.qkv_blocksize)
                lay[x + "conv1d.conv.weight"] = (inner, 1, spec.conv_k)
                lay[x + "conv1d.conv.bias"] = (inner,)
                lay[x + "mlstm_cell.igate.weight"] = (NH, 3 * inner)
                lay[x + "mlstm_cell.igate.bias"] = (NH,)
                lay[x + "mlstm_cell.fgate.weight"] = (NH, 3 * inner)
                lay[x + "mlstm_cell.fgate.bias"] = (NH,)
                lay[x + "mlstm_cell.outnorm.weight"] = (inner,)
                norms.append((x + "mlstm_cell.outnorm", inner))
                lay[x + "learnable_skip"] = (inner,)
                lay[x + "proj_down.weight"] = (D, inner)
        lay["encoder.layers.post_blocks_norm.weight"] = (D,)
        norms.append(("encoder.layers.post_blocks_norm", D))
        if spec.ln_bias:
            for key, n in norms:
                lay[key + ".bias"] = (n,)
    else:
        di, N, R = spec.d_inner, spec.d_state, spec.dt_rank
        for i in range(spec.n_blocks):
            p = f"encoder.layers.{i}."
            lay[p + "norm.weight"] = (D,)
            m = p + "mixer."
            lay[m + "in_proj.weight"] = (2 * di, D)
            lay[m + "conv1d.weight"] = (di, 1, spec.d_conv)
            lay[m + "conv1d.bias"] = (di,)
            lay[m + "x_proj.weight"] = (R + 2 * N, di)
            lay[m + "dt_proj.weight"] = (di, R)
            lay[m + "dt_proj.bias"] = (di,)
            lay[m + "A_log"] = (di, N)
            lay[m + "D"] = (di,)
            lay[m + "out_proj.weight"] = (D, di)
        lay["encoder.norm_f.weight"] = (D,)
    return lay


# ----------------------------------------------------------------------------------------------
# seeded initialisation
# ----------------------------------------------------------------------------------------------
def init_state_dict(spec: ModelSpec, seed: int = 0, scheme: str = "exercise", with_image_encoder: bool = False
                    ) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights in the reference key layout.

    scheme "exercise": every tensor random with O(1) activations (recurrent kernel R non-zero, gate weights
        non-zero, norm weights != default) so that parity tests exercise every term of the recurrences.
    scheme "reference": follows the reference / package initialisers where they matter for the dynamics
        (HF normal(0, 0.02) for Linear, sLSTM recurrent kernel zeros and power-law forget bias, mLSTM gate
        weights zero with forget bias linspace(3, 6), Mamba A_log = log(1..N), D = 1, dt bias = softplus^-1
        of log-uniform [1e-3, 1e-1]) -- the distribution a freshly constructed reference model has
        (src/algos/models/decision_xlstm.py:170-171,210-213: post_init -> reset_parameters).
    scheme "trained_like": the long-memory corner a trained checkpoint sits in
        (src/algos/decision_transformer_sb3.py:1120-1184 loads such weights): dense weights at 1/sqrt(fan_in)
        (O(1) activations), mLSTM forget bias linspace(3, 6) KEPT (f ~ 0.95-0.998: hundreds of steps of memory),
        input-gate weights large enough that the pre-activation spans about +-15 (the stabiliser m leaves
        [-8, 8]), forget-gate weights moderate, sLSTM recurrent kernel non-zero with the unscaled power-law
        forget bias, Mamba dt_proj.bias at BOTH ends of [1e-3, 1e-1] and A_log up to log(16) + 2.
    """
    assert scheme in ("exercise", "reference", "trained_like"), scheme
    ex = scheme != "reference"          # tensors that are random wherever the model is not freshly initialised
    tl = scheme == "trained_like"
    # Portable generator: PCG64 uniform floats are exact integer->float conversions, so the same seed gives
    # bit-identical weights on every host (torch.randn / exp / log go through SIMD-dependent libm paths).
    rng = np.random.Generator(np.random.PCG64(seed))
    lay = reference_layout(spec, with_image_encoder)
    sd: Dict[str, torch.Tensor] = {}

    def randn(shape, std):
        """zero-mean uniform with standard deviation `std`"""
        u = rng.random(tuple(shape), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
        return torch.from_numpy(u * np.float32(std * math.sqrt(3.0)))

    def f64(arr):
        return torch.from_numpy(np.asarray(arr, dtype=np.float64).astype(np.float32))

    for name, shape in lay.items():
        fan_in = shape[-1] if len(shape) >= 2 else 1
        if name.endswith("A_log"):
            t = f64(np.log(np.arange(1, shape[1] + 1, dtype=np.float64))).repeat(shape[0], 1)
            if tl:     # decay rates up to e^2 x the initial ones: A = -exp(A_log) down to -16 e^2
                t = t + torch.from_numpy(rng.random(tuple(shape), dtype=np.float32) * np.float32(2.0))
            elif ex:
                t = t + randn(shape, 0.1)
        elif name.endswith("mixer.D"):
            t = torch.ones(shape) + (randn(shape, 0.1) if ex else 0)
        elif name.endswith("dt_proj.bias"):
            u = rng.random(tuple(shape), dtype=np.float32).astype(np.float64)
            if tl:     # both ends of the initialiser's range, nothing in between
                u = (u >= 0.5).astype(np.float64)
            dt = np.maximum(np.exp(u * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3)), 1e-4)
            t = f64(dt + np.log(-np.expm1(-dt)))
        elif name.endswith("learnable_skip"):
            t = torch.ones(shape) + (randn(shape, 0.1) if ex else 0)
        elif ("norm" in name and name.endswith(".weight") and "embed_ln" not in name and spec.backbone == "xlstm"
              and not spec.rms_norm) or name.endswith("outnorm.weight") or name.endswith("group_norm.weight"):
            # xlstm LayerNorm stores gamma - 1 (residual weight), default 0
            t = randn(shape, 0.1) if ex else torch.zeros(shape)
        elif "norm" in name and name.endswith(".weight") or name == "embed_ln.weight":
            t = torch.ones(shape) + (randn(shape, 0.1) if ex else 0)
        elif name.endswith("_recurrent_kernel_"):
            t = randn(shape, 1.0 / math.sqrt(shape[1])) if ex else torch.zeros(shape)
        elif name.endswith("slstm_cell._bias_"):
            NH, _, dh = shape
            t = torch.zeros(shape)
            blk = int(name.split("blocks.")[1].split(".")[0])
            ratio = blk / (spec.n_blocks - 1) if spec.n_blocks > 1 else 0.0
            # [3P] powerlaw_blockdependent forget-gate bias (slot 1 = "f")
            t[:, 1, :] = f64(-(-5.0 + 12.0 * (np.arange(dh, dtype=np.float64) / max(dh - 1, 1)) ** (0.3 + 1.3 * ratio)))
            if tl:
                t = t + randn(shape, 0.2)
            elif ex:
                t = t * 0.25 + randn(shape, 0.2)
        elif name.endswith("mlstm_cell.fgate.bias"):
            t = torch.linspace(3.0, 6.0, shape[0])
            if scheme == "exercise":
                t = torch.linspace(0.5, 3.0, shape[0])
        elif name.endswith("mlstm_cell.igate.bias"):
            t = randn(shape, 0.1)
        elif name.endswith("mlstm_cell.igate.weight") or name.endswith("mlstm_cell.fgate.weight"):
            k = ((TRAINED_LIKE_IGATE if spec.n_blocks <= 8 else TRAINED_LIKE_IGATE_DEEP) if "igate" in name
                 else TRAINED_LIKE_FGATE) if tl else 1.0
            t = randn(shape, k / math.sqrt(shape[-1])) if ex else torch.zeros(shape)
        elif name.endswith(".bias"):
            t = randn(shape, 0.02 if scheme == "reference" else 0.1)
        elif name in ("embed_return.weight", "embed_rewards.weight"):
            t = randn(shape, 0.02 if scheme == "reference" else 0.5)
        elif len(shape) == 4:  # conv2d
            t = randn(shape, 1.0 / math.sqrt(shape[1] * 9))
        elif "conv1d" in name and name.endswith("weight"):
            t = randn(shape, 1.0 / math.sqrt(shape[-1]))
        elif scheme == "reference" and len(shape) == 2:
            t = randn(shape, 0.02)
        else:
            t = randn(shape, 1.0 / math.sqrt(fan_in))
        sd[name] = t.contiguous()
    return sd


# ----------------------------------------------------------------------------------------------
# engine packing
# ----------------------------------------------------------------------------------------------
def _gamma(spec: ModelSpec, w: torch.Tensor, residual: bool) -> torch.Tensor:
    return (1.0 + w) if residual else w


def engine_layout(spec: ModelSpec, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Reference state dict -> engine-side named fp32 tensors (what lram_set_weight receives).

    Transforms: xlstm LayerNorm residual weights -> gamma = 1 + w (plain weight when rms_norm);
    sLSTM `_recurrent_kernel_` (head, in, gate, out) -> (head, gate, out, in) so the recurrent GEMM reads
    K-contiguous rows; `_bias_` (head, gate, dh) -> (gate, head*dh); the package's gate-name swap
    (cell input-gate slot <- module `fgate`, forget-gate slot <- module `igate`, see oracle/xlstm_ref.py)."""
    def f32(t):
        return t.detach().to(torch.float32).contiguous().cpu()

    out: Dict[str, torch.Tensor] = {
        "embed_state.weight": f32(sd["embed_state.weight"]), "embed_state.bias": f32(sd["embed_state.bias"]),
        "embed_return.weight": f32(sd["embed_return.weight"]).reshape(-1),
        "embed_return.bias": f32(sd["embed_return.bias"]),
        "embed_rewards.weight": f32(sd["embed_rewards.weight"]).reshape(-1),
        "embed_rewards.bias": f32(sd["embed_rewards.bias"]),
        "embed_ln.weight": f32(sd["embed_ln.weight"]),
        "action_net.weight": f32(sd["action_net.0.weight"]), "action_net.bias": f32(sd["action_net.0.bias"]),
    }
    if "embed_ln.bias" in sd and sd["embed_ln.bias"] is not None:
        out["embed_ln.bias"] = f32(sd["embed_ln.bias"])
    # IMPALA-CNN image front end: uploaded under the reference's own module names (csrc/impala_cnn.hip)
    for k, v in sd.items():
        if k.startswith("embed_image."):
            out[k] = f32(v).reshape(-1)
    if spec.backbone == "xlstm":
        res = not spec.rms_norm
        for i in range(spec.n_blocks):
            p = f"encoder.layers.blocks.{i}."
            x = p + "xlstm."
            e = f"b{i}."
            out[e + "norm.gamma"] = f32(_gamma(spec, sd[p + "xlstm_norm.weight"], res))
            if spec.ln_bias and (p + "xlstm_norm.bias") in sd and not spec.rms_norm:
                out[e + "norm.beta"] = f32(sd[p + "xlstm_norm.bias"])
            if i in spec.slstm_at:
                out[e + "conv_w"] = f32(sd[x + "conv1d.conv.weight"]).reshape(spec.d_model, spec.conv_k)
                out[e + "conv_b"] = f32(sd[x + "conv1d.conv.bias"])
                out[e + "gate_i"] = f32(sd[x + "fgate.weight"])   # sic: package wiring
                out[e + "gate_f"] = f32(sd[x + "igate.weight"])   # sic
                out[e + "gate_z"] = f32(sd[x + "zgate.weight"])
                out[e + "gate_o"] = f32(sd[x + "ogate.weight"])
                out[e + "rt"] = f32(sd[x + "slstm_cell._recurrent_kernel_"].permute(0, 2, 3, 1))
                out[e + "rbias"] = f32(sd[x + "slstm_cell._bias_"].permute(1, 0, 2)).reshape(-1)
                out[e + "gn.gamma"] = f32(1.0 + sd[x + "group_norm.weight"])
                if spec.ln_bias and (x + "group_norm.bias") in sd:
                    out[e + "gn.beta"] = f32(sd[x + "group_norm.bias"])
                out[e + "ffn_norm.gamma"] = f32(_gamma(spec, sd[p + "ffn_norm.weight"], res))
                if spec.ln_bias and (p + "ffn_norm.bias") in sd and not spec.rms_norm:
                    out[e + "ffn_norm.beta"] = f32(sd[p + "ffn_norm.bias"])
                out[e + "ffn_up"] = f32(sd[p + "ffn.proj_up.weight"])
                out[e + "ffn_down"] = f32(sd[p + "ffn.proj_down.weight"])
            else:
                out[e + "proj_up"] = f32(sd[x + "proj_up.weight"])
                out[e + "conv_w"] = f32(sd[x + "conv1d.conv.weight"]).reshape(spec.inner, spec.conv_k)
                out[e + "conv_b"] = f32(sd[x + "conv1d.conv.bias"])
                out[e + "wq"] = f32(sd[x + "q_proj.weight"]).reshape(-1)
                out[e + "wk"] = f32(sd[x + "k_proj.weight"]).reshape(-1)
                out[e + "wv"] = f32(sd[x + "v_proj.weight"]).reshape(-1)
                out[e + "wi"] = f32(sd[x + "mlstm_cell.igate.weight"])
                out[e + "bi"] = f32(sd[x + "mlstm_cell.igate.bias"])
                out[e + "wf"] = f32(sd[x + "mlstm_cell.fgate.weight"])
                out[e + "bf"] = f32(sd[x + "mlstm_cell.fgate.bias"])
                out[e + "outnorm.gamma"] = f32(1.0 + sd[x + "mlstm_cell.outnorm.weight"])
                if spec.ln_bias and (x + "mlstm_cell.outnorm.bias") in sd:
                    out[e + "outnorm.beta"] = f32(sd[x + "mlstm_cell.outnorm.bias"])
                out[e + "skip"] = f32(sd[x + "learnable_skip"])
                out[e + "proj_down"] = f32(sd[x + "proj_down.weight"])
        out["post_norm.gamma"] = f32(_gamma(spec, sd["encoder.layers.post_blocks_norm.weight"], res))
        if spec.ln_bias and "encoder.layers.post_blocks_norm.bias" in sd and not spec.rms_norm:
            out["post_norm.beta"] = f32(sd["encoder.layers.post_blocks_norm.bias"])
    else:
        for i in range(spec.n_blocks):
            p = f"encoder.layers.{i}."
            m = p + "mixer."
            e = f"b{i}."
            out[e + "norm.gamma"] = f32(sd[p + "norm.weight"])
            out[e + "in_proj"] = f32(sd[m + "in_proj.weight"])
            if (m + "in_proj.bias") in sd:
                out[e + "in_proj_b"] = f32(sd[m + "in_proj.bias"])
            out[e + "conv_w"] = f32(sd[m + "conv1d.weight"]).reshape(spec.d_inner, spec.d_conv)
            if (m + "conv1d.bias") in sd:
                out[e + "conv_b"] = f32(sd[m + "conv1d.bias"])
            out[e + "x_proj"] = f32(sd[m + "x_proj.weight"])
            out[e + "dt_proj"] = f32(sd[m + "dt_proj.weight"])
            out[e + "dt_bias"] = f32(sd[m + "dt_proj.bias"])
            out[e + "A_log"] = f32(sd[m + "A_log"])
            out[e + "D"] = f32(sd[m + "D"])
            out[e + "out_proj"] = f32(sd[m + "out_proj.weight"])
            if (m + "out_proj.bias") in sd:
                out[e + "out_proj_b"] = f32(sd[m + "out_proj.bias"])
        out["post_norm.gamma"] = f32(sd["encoder.norm_f.weight"])
    return out


# ----------------------------------------------------------------------------------------------
# SB3 zip checkpoints
# ----------------------------------------------------------------------------------------------
def strip_prefixes(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Drop the `module.` (DDP) and `_orig_mod.` (torch.compile) prefixes the reference strips in
    load_model_weights (src/algos/decision_transformer_sb3.py:1138-1158)."""
    return filter_policy_dict(sd, {"load_state_head": True})


def filter_policy_dict(policy_dict: Dict[str, torch.Tensor], load_kwargs: Optional[dict] = None,
                       compile: bool = False) -> Dict[str, torch.Tensor]:
    """The key handling of the reference's `load_model_weights`
    (src/algos/decision_transformer_sb3.py:1120-1168), in its order:
      * the first "module." of every key is removed (DDP), keys then listed in `exclude_params` are dropped:
        the action heads unless `load_action_head` (default True), `predict_state.*` unless `load_state_head`
        (default False);
      * `img_encoder_only` keeps keys containing "embed_image"; `exclude_heads` drops keys containing action_pred /
        predict_state / predict_reward / predict_return;
      * the first "_orig_mod." is removed when the receiving model is not compiled (it is added when the model is
        compiled and the checkpoint was not);
      * legacy `mu.* / log_std.*` heads are renamed to `mu.0.* / log_std.0.*`.
    The result is what the reference hands to `policy.load_state_dict(..., strict=False)`; pinned against the executed
    reference method by tests/golden/reference_vectors.json `load_model_weights_trace`."""
    kw = load_kwargs if load_kwargs is not None else {}
    exclude = []
    if not kw.get("load_action_head", True):
        for name in ("action_net", "action_pred", "mu", "log_std"):
            exclude += [f"{name}.weight", f"{name}.bias", f"{name}.0.weight", f"{name}.0.bias", f"{name}.1.weight",
                        f"{name}.1.bias"]
    if not kw.get("load_state_head", False):
        exclude += ["predict_state.weight", "predict_state.bias"]
    out = {k.replace("module.", "", 1): v for k, v in policy_dict.items() if k.replace("module.", "", 1) not in exclude}
    if kw.get("img_encoder_only", False):
        out = {k: v for k, v in out.items() if "embed_image" in k}
    if kw.get("exclude_heads", False):
        for frag in ("action_pred", "predict_state", "predict_reward", "predict_return"):
            out = {k: v for k, v in out.items() if frag not in k}
    from_compiled = bool(out) and next(iter(out)).startswith("_orig_mod.")
    if not compile:
        out = {k.replace("_orig_mod.", "", 1): v for k, v in out.items()}
    elif not from_compiled:
        out = {f"_orig_mod.{k}": v for k, v in out.items()}
    if "mu.weight" in out:
        out["mu.0.weight"], out["mu.0.bias"] = out.pop("mu.weight"), out.pop("mu.bias")
        out["log_std.0.weight"], out["log_std.0.bias"] = out.pop("log_std.weight"), out.pop("log_std.bias")
    return out


def load_report(spec: ModelSpec, sd: Dict[str, torch.Tensor], with_image_encoder: bool = False):
    """(missing, unexpected) against the keys the inference path reads -- what `load_state_dict(strict=False)` prints
    in the reference (decision_transformer_sb3.py:1169-1173); heads the rollout never evaluates are not expected."""
    lay = reference_layout(spec, with_image_encoder)
    missing = [k for k in lay if k not in sd and k != "embed_ln.bias"]
    unexpected = [k for k in sd if k not in lay]
    return missing, unexpected


def load_sb3_zip(path: str, load_kwargs: Optional[dict] = None
                 ) -> Tuple[Dict[str, torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Read `policy.pth` (+ `state_mean` / `state_std` from `pytorch_variables.pth`) out of an SB3 zip written by the
    reference's save() (decision_transformer_sb3.py:1246-1280, agent_utils.py:165-202: `data` json, one `<name>.pth` per
    state dict, `pytorch_variables.pth`, `system_info.txt`) and apply load_model_weights' key handling
    (`filter_policy_dict`).  Returns (state dict, state_mean, state_std)."""
    with zipfile.ZipFile(path) as zf:
        names = set(zf.namelist())
        if "policy.pth" not in names:
            raise KeyError(f"{path}: no policy.pth in archive ({sorted(names)})")
        sd = torch.load(io.BytesIO(zf.read("policy.pth")), map_location="cpu", weights_only=True)
        mean = std = None
        if "pytorch_variables.pth" in names:
            pv = torch.load(io.BytesIO(zf.read("pytorch_variables.pth")), map_location="cpu", weights_only=False)
            if isinstance(pv, dict) and "state_mean" in pv and "state_std" in pv:   # :1182-1184
                mean, std = pv["state_mean"], pv["state_std"]
    return filter_policy_dict(sd, load_kwargs), mean, std


def save_sb3_zip(path: str, sd: Dict[str, torch.Tensor], state_mean=None, state_std=None, prefix: str = "",
                 optimizer_state: Optional[dict] = None) -> None:
    """Write an SB3-style zip with the members the reference's save() produces (agent_utils.py:165-202): `data`,
    `pytorch_variables.pth`, `policy.pth` (+ `optimizer.pth`), `system_info.txt`.  Test helper for the loader."""
    with zipfile.ZipFile(path, "w") as zf:
        zf.writestr("data", "{}")
        buf = io.BytesIO()
        torch.save({"state_mean": state_mean, "state_std": state_std}, buf)
        zf.writestr("pytorch_variables.pth", buf.getvalue())
        buf = io.BytesIO()
        torch.save({prefix + k: v for k, v in sd.items()}, buf)
        zf.writestr("policy.pth", buf.getvalue())
        if optimizer_state is not None:
            buf = io.BytesIO()
            torch.save(optimizer_state, buf)
            zf.writestr("optimizer.pth", buf.getvalue())
        zf.writestr("system_info.txt", "lram_amd test checkpoint")


def check_state_dict(spec: ModelSpec, sd: Dict[str, torch.Tensor], with_image_encoder: bool = False) -> None:
    """Raise with the full list of missing / mis-shaped keys (load_state_dict(strict) behaviour for the
    keys the inference path needs; extra keys such as predict_state.* are ignored)."""
    lay = reference_layout(spec, with_image_encoder)
    missing = [k for k in lay if k not in sd and not k.endswith("embed_ln.bias")]
    bad = [f"{k}: {tuple(sd[k].shape)} != {lay[k]}" for k in lay if k in sd and tuple(sd[k].shape) != lay[k]]
    if missing or bad:
        raise KeyError(f"state dict does not match the model spec; missing={missing[:8]}{'...' if len(missing) > 8 else ''} "
                       f"shape mismatches={bad[:8]}")


def count_params(sd: Dict[str, torch.Tensor]) -> int:
    return int(sum(int(np.prod(v.shape)) for v in sd.values()))
