"""Model description + loader for the reference's Hydra `agent_params` config group.

The reference builds its policy from `config.agent_params` (src/algos/builder.py:12-104):
`kind` selects the model / agent classes (src/algos/__init__.py:34-80), `huggingface` becomes the kwargs
of xLSTMConfig / MambaConfig (src/algos/models/decision_xlstm.py:104-116, decision_mamba.py:15-49),
`model_kwargs` the kwargs of the model class, `replay_buffer_kwargs.{max_state_dim,max_act_dim}` the
padded state / action dims (builder.py:31-37).  Hydra/OmegaConf are not required here: `load_agent_params`
composes the same YAML tree (defaults list, group overrides `a/b=c`, value overrides `a.b=c`, `+a.b=c`,
`${...}` interpolations incl. the `multiply` resolver of src/utils/misc.py:11-22).

`ModelSpec` is the engine-facing, fully resolved description; unknown keys raise (the reference uses
dacite strict mode for `xlstm_config`, decision_xlstm.py:130-132).
"""
from __future__ import annotations

import copy
import dataclasses
import math
import os
import re
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import yaml

XLSTM_KINDS = ("DecisionXLSTM", "DiscreteDecisionXLSTM", "MDDXLSTM")
MAMBA_KINDS = ("DecisionMamba", "DiscreteDecisionMamba", "MDDMamba")


def _ceil_multiple(x: float, m: int) -> int:
    """[3P] xlstm UpProjConfigMixin._set_proj_up_dim: round proj_factor * dim up to a multiple of 64."""
    return int(math.ceil(x / m) * m)


@dataclass
class ModelSpec:
    """Everything the engine (and the oracle) needs to know about one policy."""
    backbone: str = "xlstm"             # "xlstm" | "mamba"
    kind: str = "MDDXLSTM"              # reference agent/model kind
    d_model: int = 512
    n_blocks: int = 8
    tokens_per_step: int = 3            # (s, rtg, r)  discrete_decision_transformer_model.py:265-275
    pred_token: int = 1                 # tok_to_pred_pos["a"] = s_dim = 1
    # xLSTM
    n_heads: int = 4
    conv_k: int = 4
    qkv_blocksize: int = 4
    mlstm_proj_factor: float = 2.0
    ffn_proj_factor: float = 1.3
    slstm_at: List[int] = field(default_factory=list)
    ln_eps: float = 1e-5
    ln_bias: bool = False               # +agent_params.huggingface.ln_bias (decision_xlstm.py:19-26,134-136)
    rms_norm: bool = False              # +agent_params.huggingface.rms_norm (decision_xlstm.py:190-191)
    chunkwise_step: bool = False        # accepted, prefill == sequential steps
    context_length: int = 150
    # Mamba
    d_state: int = 16
    d_conv: int = 4
    expand: int = 2
    dt_rank: int = 0                    # 0 -> ceil(d_model / 16)
    norm_eps: float = 1e-5
    # front end / head  (multi_domain.yaml, multi_domain_discrete_dt_model.py:12-81)
    state_dim: int = 204
    act_dim: int = 8
    action_channels: int = 256
    n_discrete: int = 18
    image_shape: Optional[Sequence[int]] = (3, 64, 64)
    # agent-level switches read by the rollout surface
    max_length: int = 50
    use_inference_cache: bool = True
    reset_inf_cache_freq: Optional[int] = None

    def __post_init__(self):
        if self.backbone not in ("xlstm", "mamba"):
            raise ValueError(f"unknown backbone {self.backbone!r}")
        if self.backbone == "mamba" and self.dt_rank in (0, "auto", None):
            self.dt_rank = math.ceil(self.d_model / 16)
        self.slstm_at = sorted(int(i) for i in self.slstm_at)
        if any(i < 0 or i >= self.n_blocks for i in self.slstm_at):
            raise ValueError("slstm_at index out of range")

    # derived sizes -------------------------------------------------------------------------
    @property
    def inner(self) -> int:
        return _ceil_multiple(self.mlstm_proj_factor * self.d_model, 64)

    @property
    def head_dim(self) -> int:
        return self.inner // self.n_heads

    @property
    def ffn_dim(self) -> int:
        return _ceil_multiple(self.ffn_proj_factor * self.d_model, 64)

    @property
    def d_inner(self) -> int:
        return int(self.expand * self.d_model)

    @property
    def n_vocab(self) -> int:
        return self.n_discrete + self.action_channels

    def state_bytes_per_env(self) -> int:
        """fp32 recurrent state per env slot ('Per-env recurrent state')."""
        if self.backbone == "mamba":
            return 4 * self.n_blocks * self.d_inner * (self.d_state + self.d_conv)
        total = 0
        for i in range(self.n_blocks):
            if i in self.slstm_at:
                total += self.d_model * (4 + self.conv_k)
            else:
                dh = self.head_dim
                total += self.n_heads * dh * dh + self.inner + self.n_heads + self.conv_k * self.inner
        return 4 * total

    def mlstm_matrix_bytes_per_env(self) -> int:
        """Bytes of matrix memory C per env over all mLSTM blocks."""
        if self.backbone != "xlstm":
            return 0
        n_m = self.n_blocks - len(self.slstm_at)
        return 4 * n_m * self.n_heads * self.head_dim * self.head_dim


# ----------------------------------------------------------------------------------------------
# YAML composition (Hydra-compatible subset)
# ----------------------------------------------------------------------------------------------
_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _get_path(root: Dict[str, Any], path: str):
    cur: Any = root
    for part in path.split("."):
        if not isinstance(cur, dict) or part not in cur:
            raise KeyError(f"interpolation key not found: {path}")
        cur = cur[part]
    return cur


def _resolve_value(val, root):
    if not isinstance(val, str):
        return val
    for _ in range(16):
        m = _INTERP.search(val)
        if m is None:
            return val
        expr = m.group(1).strip()
        if ":" in expr:
            name, args = expr.split(":", 1)
            parts = [_resolve_value(a.strip(), root) for a in args.split(",")]
            if name == "multiply":      # src/utils/misc.py `multiply` resolver
                rep = 1
                for p in parts:
                    rep = rep * (yaml.safe_load(p) if isinstance(p, str) else p)
            elif name == "maybe_split":  # src/utils/misc.py `maybe_split` resolver
                rep = parts[0].split(",") if isinstance(parts[0], str) else parts[0]
            else:
                raise KeyError(f"unknown resolver {name!r}")
        else:
            rep = _resolve_value(_get_path(root, expr), root)
        if m.span() == (0, len(val)):
            return rep
        val = val[: m.start()] + str(rep) + val[m.end():]
    raise ValueError(f"interpolation too deep: {val}")


def _resolve_tree(node, root):
    if isinstance(node, dict):
        return {k: _resolve_tree(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve_tree(v, root) for v in node]
    return _resolve_value(node, root)


def _load_yaml(path: str) -> Dict[str, Any]:
    with open(path, "r") as fh:
        return yaml.safe_load(fh) or {}


def _set_path(root: Dict[str, Any], path: str, value, create: bool):
    parts = path.split(".")
    cur = root
    for p in parts[:-1]:
        if p not in cur or not isinstance(cur[p], dict):
            if not create:
                raise KeyError(f"override key not found: {path} (use +{path}=... to add)")
            cur[p] = {}
        cur = cur[p]
    if parts[-1] not in cur and not create:
        raise KeyError(f"override key not found: {path} (use +{path}=... to add)")
    cur[parts[-1]] = value


def load_agent_params(config_dir: str, name: str = "multi_domain", overrides: Sequence[str] = ()) -> Dict[str, Any]:
    """Compose `agent_params/<name>.yaml` from a reference-style config tree and apply overrides.

    Returns the resolved `agent_params` dict (what `OmegaConf.to_container(config.agent_params,
    resolve=True)` yields upstream, src/algos/builder.py:15)."""
    base = os.path.join(config_dir, "agent_params")
    cfg = _load_yaml(os.path.join(base, f"{name}.yaml"))
    defaults = cfg.pop("defaults", []) or []
    group_choice: Dict[str, Optional[str]] = {}
    for item in defaults:
        if isinstance(item, dict):
            for g, choice in item.items():
                group_choice[g] = choice
    value_overrides = []
    for ov in overrides:
        key, _, val = ov.partition("=")
        if key.startswith("agent_params/"):
            group_choice[key[len("agent_params/"):]] = val
        else:
            value_overrides.append((key, val))
    own = copy.deepcopy(cfg)
    composed: Dict[str, Any] = {}
    for g, choice in group_choice.items():
        if choice in (None, "null", "None"):
            continue
        composed[g] = _load_yaml(os.path.join(base, g, f"{choice}.yaml"))
    # the primary file's own keys are merged over the group files (Hydra default: _self_ last)
    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = copy.deepcopy(v)
    merge(composed, own)
    root = {"agent_params": composed, "run_params": {"total_timesteps": 0}}
    # top-level scalars of the primary config (configs/config.yaml: SSD_DATA_DIR, seed, device, ...) are interpolation
    # targets of the group files (data_paths: ${SSD_DATA_DIR}/...)
    top = os.path.join(config_dir, "config.yaml")
    if os.path.exists(top):
        for k, v in (_load_yaml(top) or {}).items():
            if k not in ("defaults", "hydra", "agent_params") and not isinstance(v, (dict, list)):
                root.setdefault(k, v)
    for key, val in value_overrides:
        create = key.startswith("+")
        key = key.lstrip("+")
        if not key.startswith("agent_params."):
            continue  # env_params / run_params overrides are not this package's business
        _set_path(root, key, yaml.safe_load(val), create)
    return _resolve_tree(root["agent_params"], root)


_XLSTM_HF_KEYS = {"max_ep_len", "max_length", "n_layer", "hidden_size", "n_head", "xlstm_config", "ln_bias",
                  "rms_norm", "chunkwise_step", "activation_function", "use_fast_attn", "n_positions",
                  "output_attentions", "n_embd"}
_MAMBA_HF_KEYS = {"max_ep_len", "max_length", "n_layer", "n_head", "n_embd", "d_model", "d_intermediate", "d_state",
                  "d_conv", "expand", "norm_epsilon", "conv_bias", "bias", "rms_norm", "fused_add_norm",
                  "residual_in_fp32", "dt_rank", "output_attentions", "activation_function", "use_fast_attn",
                  "n_positions", "hidden_size", "dtype"}
_XLSTM_CFG_KEYS = {"mlstm_block", "slstm_block", "context_length", "num_blocks", "embedding_dim", "slstm_at",
                   "add_post_blocks_norm", "bias", "dropout"}
_MODEL_KWARGS_KEYS = {"reward_condition", "tokenize_a", "tokenize_rtg", "action_channels", "discrete_actions",
                      "state_dim", "image_shape", "relative_pos_embds", "use_time_embds", "action_condition",
                      "shared_a_head", "inf_dummy_batch_size", "rtg_condition", "max_act_dim", "encoder_kwargs",
                      "embed_bias_init"}


def _strict(d: Dict[str, Any], allowed, what: str):
    unknown = set(d) - set(allowed)
    if unknown:
        raise KeyError(f"unknown key(s) in {what}: {sorted(unknown)}")


def spec_from_agent_params(ap: Dict[str, Any]) -> ModelSpec:
    """agent_params dict -> ModelSpec.  Only the recurrent kinds are accepted (this package is the
    recurrent action-inference path; the GPT-2 DT baseline is out of scope)."""
    kind = ap.get("kind")
    hf = dict(ap.get("huggingface") or {})
    mk = dict(ap.get("model_kwargs") or {})
    rb = dict(ap.get("replay_buffer_kwargs") or {})
    _strict(mk, _MODEL_KWARGS_KEYS, "agent_params.model_kwargs")
    # the engine implements the (s, rtg, r) token layout of multi_domain.yaml
    if not mk.get("reward_condition", False) or mk.get("action_condition", True) or not mk.get("rtg_condition", True):
        raise ValueError("engine supports reward_condition=True, rtg_condition=True, action_condition=False "
                         "(configs/agent_params/model_kwargs/multi_domain.yaml)")
    if not mk.get("shared_a_head", False) or not mk.get("tokenize_a", True) or mk.get("use_time_embds", True):
        raise ValueError("engine supports shared_a_head=True, tokenize_a=True, use_time_embds=False")
    common = dict(
        kind=kind,
        state_dim=int(rb.get("max_state_dim") or mk.get("state_dim", 204)),
        act_dim=int(rb.get("max_act_dim") or mk.get("max_act_dim") or 8),
        action_channels=int(mk.get("action_channels", 256)),
        n_discrete=int(mk.get("discrete_actions", 18)),
        image_shape=tuple(mk["image_shape"]) if mk.get("image_shape") else None,
        max_length=int(hf.get("max_length", 50)),
        use_inference_cache=bool(ap.get("use_inference_cache", True)),
        reset_inf_cache_freq=ap.get("reset_inf_cache_freq"),
    )
    if kind in XLSTM_KINDS:
        _strict(hf, _XLSTM_HF_KEYS, "agent_params.huggingface")
        xc = dict(hf.get("xlstm_config") or {})
        _strict(xc, _XLSTM_CFG_KEYS, "agent_params.huggingface.xlstm_config")
        ml = dict((xc.get("mlstm_block") or {}).get("mlstm") or {})
        sb = dict(xc.get("slstm_block") or {})
        sl = dict(sb.get("slstm") or {})
        ff = dict(sb.get("feedforward") or {})
        if ff.get("act_fn", "gelu") != "gelu":
            raise ValueError("only the gelu gated feed-forward is implemented")
        if sl and int(sl.get("num_heads", 4)) != int(ml.get("num_heads", 4)):
            raise ValueError("mLSTM and sLSTM head counts must match")
        slstm_at = xc.get("slstm_at") or []
        n_blocks = int(xc.get("num_blocks", hf.get("n_layer", 1)))
        if slstm_at == "all":
            slstm_at = list(range(n_blocks))
        return ModelSpec(
            backbone="xlstm", d_model=int(xc.get("embedding_dim", hf.get("hidden_size"))), n_blocks=n_blocks,
            n_heads=int(ml.get("num_heads", hf.get("n_head", 4))), conv_k=int(ml.get("conv1d_kernel_size", 4)),
            qkv_blocksize=int(ml.get("qkv_proj_blocksize", 4)), mlstm_proj_factor=float(ml.get("proj_factor", 2.0)),
            ffn_proj_factor=float(ff.get("proj_factor", 1.3)), slstm_at=list(slstm_at),
            ln_bias=bool(hf.get("ln_bias", False)), rms_norm=bool(hf.get("rms_norm", False)),
            chunkwise_step=bool(hf.get("chunkwise_step", False)), context_length=int(xc.get("context_length", 150)),
            **common)
    if kind in MAMBA_KINDS:
        _strict(hf, _MAMBA_HF_KEYS, "agent_params.huggingface")
        if int(hf.get("d_intermediate", 0)) != 0:
            raise ValueError("Mamba blocks with an MLP (d_intermediate > 0) are not implemented")
        if not hf.get("rms_norm", True):
            raise ValueError("Mamba with LayerNorm (rms_norm=False) is not implemented")
        return ModelSpec(
            backbone="mamba", d_model=int(hf.get("d_model", 2560)), n_blocks=int(hf.get("n_layer", 64)),
            d_state=int(hf.get("d_state", 16)), d_conv=int(hf.get("d_conv", 4)), expand=int(hf.get("expand", 2)),
            dt_rank=hf.get("dt_rank", "auto") if hf.get("dt_rank", "auto") != "auto" else 0,
            norm_eps=float(hf.get("norm_epsilon", 1e-5)), **common)
    raise ValueError(f"agent kind {kind!r} is not a recurrent LRAM kind; expected one of {XLSTM_KINDS + MAMBA_KINDS}")


def engine_limits(spec: ModelSpec) -> List[str]:
    """Reasons why `lram_create` would refuse this geometry (the mirror of `validate_config` in csrc/engine.hip); empty when
    the engine runs it.  Every preset of the reference's configs/agent_params/huggingface passes as an mLSTM-only stack;
    sLSTM blocks additionally need d_model / num_heads to be a multiple of 4, which `xlstm_mediumplus_half` (266) and
    `xlstm_large_half` (358) do not meet."""
    why: List[str] = []
    if spec.d_model % 4 or spec.d_model > 2048:
        why.append(f"d_model {spec.d_model} must be a multiple of 4 and <= 2048")
    if spec.state_dim % 4:
        why.append(f"state_dim {spec.state_dim} must be a multiple of 4")
    if spec.backbone == "xlstm":
        if spec.n_heads not in (1, 2, 4, 8):
            why.append(f"num_heads {spec.n_heads} must be 1, 2, 4 or 8")
        elif spec.inner % (16 * spec.n_heads):
            why.append(f"mLSTM head dim {spec.inner / spec.n_heads:g} must be a multiple of 16")
        if spec.inner > 4096:
            why.append(f"mLSTM inner dim {spec.inner} must be <= 4096")
        if spec.n_heads and spec.inner // spec.n_heads > 1024:
            why.append(f"mLSTM head dim {spec.inner // spec.n_heads} must be <= 1024")
        if spec.d_model % spec.n_heads:
            why.append(f"d_model {spec.d_model} must be a multiple of num_heads")
        if spec.slstm_at and spec.d_model % (4 * spec.n_heads):
            why.append(f"sLSTM blocks need d_model {spec.d_model} to be a multiple of 4 * num_heads")
        if spec.conv_k != 4 or spec.qkv_blocksize != 4:
            why.append("conv1d_kernel_size and qkv_proj_blocksize must be 4")
    else:
        if spec.d_inner % 4 or spec.d_conv != 4:
            why.append("Mamba d_inner must be a multiple of 4 and d_conv 4")
    return why


# ----------------------------------------------------------------------------------------------
# named presets = BASELINE.json configs (shapes from the reference YAMLs)
# ----------------------------------------------------------------------------------------------
def preset(name: str) -> ModelSpec:
    presets = {
        # C1: xLSTM[1:0] 2-layer d_model=128
        "xlstm_c1": dict(backbone="xlstm", kind="MDDXLSTM", d_model=128, n_blocks=2, slstm_at=[]),
        # C2 / headline: xLSTM[7:1] 16M = xlstm_medium.yaml + slstm_at=[1] (reference README.md:189)
        "xlstm_16m": dict(backbone="xlstm", kind="MDDXLSTM", d_model=512, n_blocks=8, slstm_at=[1]),
        # C3: Mamba 48M = mamba_mediumplus.yaml
        "mamba_48m": dict(backbone="mamba", kind="MDDMamba", d_model=768, n_blocks=12),
        # C4/C5: xLSTM[7:1] 206M = xlstm_huge.yaml + slstm_at=[1,3,5] (reference README.md:234)
        "xlstm_206m": dict(backbone="xlstm", kind="MDDXLSTM", d_model=1280, n_blocks=20, slstm_at=[1, 3, 5]),
        # small shapes for tests
        "xlstm_tiny": dict(backbone="xlstm", kind="MDDXLSTM", d_model=128, n_blocks=3, slstm_at=[1], state_dim=20,
                           act_dim=4),
        "mamba_tiny": dict(backbone="mamba", kind="MDDMamba", d_model=64, n_blocks=2, state_dim=20, act_dim=4),
    }
    if name not in presets:
        raise KeyError(f"unknown preset {name!r}; have {sorted(presets)}")
    return ModelSpec(**presets[name])


def spec_to_dict(spec: ModelSpec) -> Dict[str, Any]:
    return dataclasses.asdict(spec)
