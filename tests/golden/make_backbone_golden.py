#!/usr/bin/env python3
"""Backbone pinning kit: golden vectors of the recurrent blocks FROM THE THIRD-PARTY PACKAGES THEMSELVES.

The arithmetic of SURVEY.md rows a6-a9 lives in `xlstm` (unpinned, /root/reference/README.md:94-97) and
`mamba_ssm==2.1.0` (+ `causal-conv1d==1.3.0.post1`, README.md:99-103), which the reference calls at
  src/algos/models/decision_xlstm.py:130-133,155-166   xLSTMBlockStack(cfg) built with dacite from the Hydra dict; .step
  src/algos/models/decision_mamba.py:78-94,130-147     create_block(...) per layer; Block(hidden, residual, inference_params)
Neither package is in the build container, so the oracle's restatement of them is unpinned.
This script closes that gap on ANY machine that has the packages (`pip install xlstm`, `pip install mamba-ssm==2.1.0
causal-conv1d==1.3.0.post1` -- the latter needs a GPU):

    python tests/golden/make_backbone_golden.py            # writes tests/golden/backbone_xlstm.npz / backbone_mamba.npz

It builds the blocks exactly as the reference does (same config dict / same factory), overwrites every parameter with
seeded ASYMMETRIC random values (recurrent kernel, biases and all four gate projections non-zero and different, so a
transposed axis or a swapped gate cannot cancel), runs 6 single-token `step` calls from the empty state for 2 envs and
stores inputs, per-step outputs, the final recurrent state and the package's own `state_dict()` -- data only.
`tests/test_backbone_golden.py` then feeds that state dict through the oracle (CPU) and through
`lram_amd.weights.engine_layout` + the HIP engine (`-m gpu`); both are skipped while the fixtures are absent.

`--from-oracle DIR` writes fixtures of the same format computed by the oracle itself (labelled source=oracle) -- only
for testing the test plumbing; such a file proves nothing about parity and is never committed under tests/golden/.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

XLSTM_CFG = {   # the reference's configs/agent_params/huggingface/xlstm_*.yaml shape, small sizes, one sLSTM block
    "mlstm_block": {"mlstm": {"conv1d_kernel_size": 4, "qkv_proj_blocksize": 4, "num_heads": 2}},
    "slstm_block": {"slstm": {"backend": "vanilla", "num_heads": 2, "conv1d_kernel_size": 4,
                              "bias_init": "powerlaw_blockdependent"},
                    "feedforward": {"proj_factor": 1.3, "act_fn": "gelu"}},
    "context_length": 48, "num_blocks": 3, "embedding_dim": 64, "slstm_at": [1],
}
MAMBA_CFG = {"d_model": 64, "n_layer": 2, "d_state": 16, "d_conv": 4, "expand": 2, "dt_rank": 4, "bias": False,
             "conv_bias": True, "norm_epsilon": 1e-5, "rms_norm": True, "residual_in_fp32": True, "d_intermediate": 0}
B, STEPS = 2, 6


def _asymmetric_(module, seed):
    """Every parameter <- seeded N(0, s) with a per-tensor scale, then a ramp along the LAST axis so that no two slices
    along any axis are statistically alike.  Norm weights stay small (the xLSTM norms store gamma - 1)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            scale = 0.1 if ("norm" in name and p.dim() == 1) else (0.5 if p.dim() == 1 else 1.5 / max(p.shape[-1], 1) ** 0.5)
            v = torch.randn(p.shape, generator=g) * scale
            ramp = torch.linspace(0.7, 1.3, p.shape[-1]) if p.dim() >= 1 and p.shape[-1] > 1 else torch.ones(1)
            v = v * ramp
            if name.endswith("A_log"):           # Mamba: A = -exp(A_log) must stay a decay
                v = torch.log(torch.rand(p.shape, generator=g) * 4.0 + 0.5)
            if name.endswith("dt_proj.bias"):
                v = torch.rand(p.shape, generator=g) * 0.5 - 2.0
            p.copy_(v.to(p.dtype).to(p.device))


def _flatten_state(prefix, obj, out):
    if torch.is_tensor(obj):
        out[prefix] = obj.detach().float().cpu().numpy()
    elif isinstance(obj, dict):
        for k, v in obj.items():
            _flatten_state(f"{prefix}/{k}", v, out)
    elif isinstance(obj, (tuple, list)):
        for i, v in enumerate(obj):
            _flatten_state(f"{prefix}/{i}", v, out)
    elif obj is not None:
        raise TypeError(f"unexpected state entry at {prefix}: {type(obj)}")


def _save(path, meta, arrays):
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print("wrote", path, "(%d arrays, source=%s)" % (len(arrays), meta["source"]))


# ---- xLSTM ----------------------------------------------------------------------------------------------------
def xlstm_from_package(out_dir):
    import importlib.metadata as md
    from dacite import Config as DaciteConfig, from_dict
    from xlstm import xLSTMBlockStack, xLSTMBlockStackConfig
    cfg = from_dict(data_class=xLSTMBlockStackConfig, data=json.loads(json.dumps(XLSTM_CFG)), config=DaciteConfig(strict=True))
    stack = xLSTMBlockStack(cfg).float().eval()      # decision_xlstm.py:132-133
    _asymmetric_(stack, seed=11)
    x = torch.randn(B, STEPS, XLSTM_CFG["embedding_dim"], generator=torch.Generator().manual_seed(12))
    ys, state = [], None
    with torch.no_grad():
        for t in range(STEPS):                          # decision_xlstm.py:162-164
            y, state = stack.step(x[:, t].unsqueeze(1), state)
            ys.append(y.clone())
    arrays = {"x": x.numpy(), "y": torch.cat(ys, dim=1).numpy()}
    _flatten_state("state", state, arrays)
    for k, v in stack.state_dict().items():
        arrays["sd/" + k] = v.detach().float().cpu().numpy()
    meta = {"source": "package", "package": "xlstm", "version": md.version("xlstm"), "config": XLSTM_CFG, "B": B, "steps": STEPS,
            "torch": torch.__version__, "call": "xLSTMBlockStack(cfg).step(x[:, t:t+1], state), state=None first"}
    _save(os.path.join(out_dir, "backbone_xlstm.npz"), meta, arrays)


def xlstm_from_oracle(out_dir):
    from lram_amd import init_state_dict
    from oracle import xlstm_ref
    spec = spec_from_xlstm_cfg(XLSTM_CFG)
    sd = {k: v for k, v in init_state_dict(spec, seed=3).items() if k.startswith("encoder.layers.")}
    x = torch.randn(B, STEPS, spec.d_model, generator=torch.Generator().manual_seed(12))
    ys, state = [], None
    for t in range(STEPS):
        y, state = xlstm_ref.stack_step(spec, sd, x[:, t].unsqueeze(1), state)
        ys.append(y)
    arrays = {"x": x.numpy(), "y": torch.cat(ys, dim=1).numpy()}
    _flatten_state("state", state, arrays)
    for k, v in sd.items():
        arrays["sd/" + k[len("encoder.layers."):]] = v.numpy()
    _save(os.path.join(out_dir, "backbone_xlstm.npz"),
          {"source": "oracle", "package": "xlstm", "version": None, "config": XLSTM_CFG, "B": B, "steps": STEPS}, arrays)


def spec_from_xlstm_cfg(cfg):
    import dataclasses
    from lram_amd import preset
    m, s = cfg["mlstm_block"]["mlstm"], cfg["slstm_block"]
    return dataclasses.replace(preset("xlstm_tiny"), d_model=cfg["embedding_dim"], n_blocks=cfg["num_blocks"],
                               n_heads=m["num_heads"], conv_k=m["conv1d_kernel_size"], qkv_blocksize=m["qkv_proj_blocksize"],
                               ffn_proj_factor=s["feedforward"]["proj_factor"], slstm_at=list(cfg["slstm_at"]),
                               context_length=cfg["context_length"])


# ---- Mamba ----------------------------------------------------------------------------------------------------
def mamba_from_package(out_dir):
    import importlib.metadata as md
    from mamba_ssm.models.mixer_seq_simple import create_block
    from mamba_ssm.utils.generation import InferenceParams
    c = MAMBA_CFG
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    ssm_cfg = {"d_state": c["d_state"], "d_conv": c["d_conv"], "expand": c["expand"], "dt_rank": c["dt_rank"],
               "bias": c["bias"], "conv_bias": c["conv_bias"]}          # models/decision_mamba.py:60-62
    layers = torch.nn.ModuleList([
        create_block(c["d_model"], d_intermediate=c["d_intermediate"], ssm_cfg=ssm_cfg, norm_epsilon=c["norm_epsilon"],
                     rms_norm=c["rms_norm"], residual_in_fp32=c["residual_in_fp32"], fused_add_norm=False, layer_idx=i)
        for i in range(c["n_layer"])]).float().to(dev).eval()         # models/decision_mamba.py:78-94 (un-fused add+norm)
    _asymmetric_(layers, seed=21)
    x = torch.randn(B, STEPS, c["d_model"], generator=torch.Generator().manual_seed(22)).to(dev)
    ip = InferenceParams(max_seqlen=64, max_batch_size=B)
    for i, layer in enumerate(layers):
        ip.key_value_memory_dict[i] = layer.allocate_inference_cache(B, 64, dtype=torch.float32)
    ip.seqlen_offset = 1          # every token through Mamba.step on the cached states (the empty state = zeros)
    hs, rs = [], []
    with torch.no_grad():
        for t in range(STEPS):
            hidden, residual = x[:, t].unsqueeze(1), None
            for layer in layers:                        # models/decision_mamba.py:130-147
                hidden, residual = layer(hidden, residual, inference_params=ip)
            hs.append(hidden.clone()), rs.append(residual.clone())
    arrays = {"x": x.cpu().numpy(), "hidden": torch.cat(hs, dim=1).cpu().numpy(), "residual": torch.cat(rs, dim=1).cpu().numpy()}
    for i in range(c["n_layer"]):
        conv, ssm = ip.key_value_memory_dict[i]
        arrays[f"state/{i}/conv"], arrays[f"state/{i}/ssm"] = conv.float().cpu().numpy(), ssm.float().cpu().numpy()
    for k, v in layers.state_dict().items():
        arrays["sd/" + k] = v.detach().float().cpu().numpy()
    meta = {"source": "package", "package": "mamba_ssm", "version": md.version("mamba_ssm"), "config": c, "B": B, "steps": STEPS,
            "device": dev, "call": "create_block(...)(hidden, residual, inference_params) per layer, seqlen_offset=1, zero caches"}
    _save(os.path.join(out_dir, "backbone_mamba.npz"), meta, arrays)


def spec_from_mamba_cfg(c):
    import dataclasses
    from lram_amd import preset
    return dataclasses.replace(preset("mamba_tiny"), d_model=c["d_model"], n_blocks=c["n_layer"], d_state=c["d_state"],
                               d_conv=c["d_conv"], expand=c["expand"], dt_rank=c["dt_rank"], norm_eps=c["norm_epsilon"])


def oracle_mamba_layers(spec, sd, x_t, state, prefix="encoder.layers."):
    """One token through the oracle's layers, returning what the package's Block chain returns BEFORE norm_f:
    (hidden, residual) of the last layer (models/decision_mamba.py:130-147) and the new per-layer states."""
    from oracle import mamba_ref
    hidden, residual, new = x_t, None, {}
    for i in range(spec.n_blocks):
        p = f"{prefix}{i}."
        residual = hidden if residual is None else hidden + residual
        normed = mamba_ref.rms_norm(residual, sd[p + "norm.weight"], spec.norm_eps)
        hidden, c, s = mamba_ref.mamba_step(sd, p + "mixer.", normed, state[i][0], state[i][1], spec.dt_rank, spec.d_state)
        new[i] = (c, s)
    return hidden, residual, new


def mamba_from_oracle(out_dir):
    from lram_amd import init_state_dict
    from oracle import mamba_ref
    c = MAMBA_CFG
    spec = spec_from_mamba_cfg(c)
    sd = {k: v for k, v in init_state_dict(spec, seed=4).items() if k.startswith("encoder.layers.")}
    x = torch.randn(B, STEPS, spec.d_model, generator=torch.Generator().manual_seed(22))
    state = mamba_ref.zero_state(spec, B)
    hs, rs = [], []
    for t in range(STEPS):
        hidden, residual, state = oracle_mamba_layers(spec, sd, x[:, t], state)
        hs.append(hidden.unsqueeze(1)), rs.append(residual.unsqueeze(1))
    arrays = {"x": x.numpy(), "hidden": torch.cat(hs, dim=1).numpy(), "residual": torch.cat(rs, dim=1).numpy()}
    for i in range(spec.n_blocks):
        arrays[f"state/{i}/conv"], arrays[f"state/{i}/ssm"] = state[i][0].numpy(), state[i][1].numpy()
    for k, v in sd.items():
        arrays["sd/" + k[len("encoder.layers."):]] = v.numpy()
    _save(os.path.join(out_dir, "backbone_mamba.npz"),
          {"source": "oracle", "package": "mamba_ssm", "version": None, "config": c, "B": B, "steps": STEPS}, arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--from-oracle", metavar="DIR", default="", help="plumbing self-test fixtures (NOT parity evidence)")
    a = ap.parse_args()
    if a.from_oracle:
        os.makedirs(a.from_oracle, exist_ok=True)
        xlstm_from_oracle(a.from_oracle), mamba_from_oracle(a.from_oracle)
        return 0
    done = 0
    for name, fn in (("xlstm", xlstm_from_package), ("mamba_ssm", mamba_from_package)):
        try:
            fn(a.out)
            done += 1
        except ImportError as e:
            print(f"[skip] {name}: {e} -- install the package on this machine and re-run")
    return 0 if done else 1


if __name__ == "__main__":
    sys.exit(main())
