"""Host logic: Hydra-style agent_params composition, strictness, model sizes, weight layout, checkpoints."""
import os

import dataclasses

import pytest
import torch
import yaml

from lram_amd import init_state_dict, load_agent_params, preset, spec_from_agent_params
from lram_amd.config import ModelSpec
from lram_amd.weights import (check_state_dict, count_params, engine_layout, load_sb3_zip, reference_layout,
                              save_sb3_zip)


def _write_tree(root):
    """A config tree with the reference's schema (configs/agent_params/...), own contents."""
    ap = os.path.join(root, "agent_params")
    for sub in ("huggingface", "model_kwargs", "replay_buffer_kwargs"):
        os.makedirs(os.path.join(ap, sub))
    yaml.safe_dump({"kind": "MDDT", "stochastic_policy": False, "use_amp": True,
                    "offline_steps": "${run_params.total_timesteps}",
                    "defaults": [{"huggingface": "dt_small"}, {"model_kwargs": "md"}, {"replay_buffer_kwargs": "rb"}],
                    "huggingface": {"max_length": 50},
                    "eval_context_len": "${agent_params.huggingface.max_length}"},
                   open(os.path.join(ap, "multi_domain.yaml"), "w"))
    yaml.safe_dump({"n_layer": 4, "hidden_size": 128}, open(os.path.join(ap, "huggingface", "dt_small.yaml"), "w"))
    yaml.safe_dump({"max_ep_len": 1000, "max_length": 50, "n_layer": 8, "hidden_size": 512, "n_head": 4,
                    "xlstm_config": {
                        "mlstm_block": {"mlstm": {"conv1d_kernel_size": 4, "qkv_proj_blocksize": 4,
                                                  "num_heads": "${agent_params.huggingface.n_head}"}},
                        "slstm_block": {"slstm": {"backend": "cuda", "num_heads": "${agent_params.huggingface.n_head}",
                                                  "conv1d_kernel_size": 4, "bias_init": "powerlaw_blockdependent"},
                                        "feedforward": {"proj_factor": 1.3, "act_fn": "gelu"}},
                        "context_length": "${multiply:${agent_params.huggingface.max_length},3}",
                        "num_blocks": "${agent_params.huggingface.n_layer}",
                        "embedding_dim": "${agent_params.huggingface.hidden_size}"}},
                   open(os.path.join(ap, "huggingface", "xl_med.yaml"), "w"))
    yaml.safe_dump({"max_length": 50, "n_embd": 512, "n_layer": 12, "n_head": 1, "max_ep_len": 1000, "d_model": 768,
                    "d_intermediate": 0, "output_attentions": True},
                   open(os.path.join(ap, "huggingface", "mb.yaml"), "w"))
    yaml.safe_dump({"reward_condition": True, "tokenize_a": True, "tokenize_rtg": False, "action_channels": 256,
                    "discrete_actions": 18, "state_dim": 204, "image_shape": [3, 64, 64], "relative_pos_embds": False,
                    "use_time_embds": False, "action_condition": False, "shared_a_head": True},
                   open(os.path.join(ap, "model_kwargs", "md.yaml"), "w"))
    yaml.safe_dump({"kind": "domain", "max_act_dim": 8, "max_state_dim": 204},
                   open(os.path.join(ap, "replay_buffer_kwargs", "rb.yaml"), "w"))


def test_compose_xlstm_from_yaml_tree(tmp_path):
    _write_tree(str(tmp_path))
    ap = load_agent_params(str(tmp_path), "multi_domain", [
        "agent_params/huggingface=xl_med", "agent_params.kind=MDDXLSTM",
        "+agent_params.huggingface.xlstm_config.slstm_at=[1]", "+agent_params.use_inference_cache=True",
        "+agent_params.reset_inf_cache_freq=100", "env_params=ignored"])
    assert ap["kind"] == "MDDXLSTM" and ap["eval_context_len"] == 50
    assert ap["huggingface"]["xlstm_config"]["context_length"] == 150
    assert ap["huggingface"]["xlstm_config"]["mlstm_block"]["mlstm"]["num_heads"] == 4
    spec = spec_from_agent_params(ap)
    assert (spec.backbone, spec.d_model, spec.n_blocks, spec.slstm_at) == ("xlstm", 512, 8, [1])
    assert (spec.inner, spec.head_dim, spec.ffn_dim, spec.n_vocab) == (1024, 256, 704, 274)
    assert spec.reset_inf_cache_freq == 100 and spec.state_dim == 204 and spec.act_dim == 8


def test_compose_mamba_and_errors(tmp_path):
    _write_tree(str(tmp_path))
    ap = load_agent_params(str(tmp_path), "multi_domain", ["agent_params/huggingface=mb", "agent_params.kind=MDDMamba"])
    spec = spec_from_agent_params(ap)
    assert (spec.backbone, spec.d_model, spec.n_blocks, spec.d_inner, spec.dt_rank) == ("mamba", 768, 12, 1536, 48)
    with pytest.raises(KeyError):  # value override of a missing key needs '+'
        load_agent_params(str(tmp_path), "multi_domain", ["agent_params.nope=1"])
    with pytest.raises(ValueError):  # transformer baseline is not this engine's business
        spec_from_agent_params(load_agent_params(str(tmp_path), "multi_domain", []))
    bad = load_agent_params(str(tmp_path), "multi_domain", ["agent_params/huggingface=xl_med", "agent_params.kind=MDDXLSTM",
                                                            "+agent_params.huggingface.xlstm_config.typo=1"])
    with pytest.raises(KeyError):  # strict like dacite (decision_xlstm.py:132)
        spec_from_agent_params(bad)


def test_presets_match_survey_sizes():
    # per-env recurrent state, and parameter counts of README.md:183-241
    assert preset("xlstm_16m").state_bytes_per_env() == 7 * (1048576 + 4096 + 16 + 16384) + 8192 + 8192
    assert abs(preset("xlstm_206m").state_bytes_per_env() / 1e6 - 112.4) < 0.1
    assert preset("mamba_48m").state_bytes_per_env() == 12 * (98304 + 24576)
    assert preset("xlstm_c1").state_bytes_per_env() == 2 * (4 * 64 * 64 * 4 + 1024 + 16 + 4096)
    assert 13e6 < count_params(init_state_dict(preset("xlstm_16m"), 0)) < 17e6
    assert 45e6 < count_params(init_state_dict(preset("mamba_48m"), 0)) < 50e6


def test_engine_layout_transforms():
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=1)
    lay = engine_layout(spec, sd)
    p = "encoder.layers.blocks.1.xlstm."
    assert torch.equal(lay["b1.gate_i"], sd[p + "fgate.weight"])  # package gate wiring (oracle/xlstm_ref.py)
    assert torch.equal(lay["b1.gate_f"], sd[p + "igate.weight"])
    R = sd[p + "slstm_cell._recurrent_kernel_"]
    assert lay["b1.rt"].shape == (4, 4, 32, 32) and float(lay["b1.rt"][2, 1, 5, 7]) == float(R[2, 7, 1, 5])
    assert torch.equal(lay["b0.norm.gamma"], 1.0 + sd["encoder.layers.blocks.0.xlstm_norm.weight"])
    rms = ModelSpec(backbone="xlstm", d_model=128, n_blocks=1, rms_norm=True, state_dim=20, act_dim=4)
    sdr = init_state_dict(rms, seed=1)
    assert torch.equal(engine_layout(rms, sdr)["b0.norm.gamma"], sdr["encoder.layers.blocks.0.xlstm_norm.weight"])
    for name, t in lay.items():
        assert t.dtype == torch.float32 and t.is_contiguous(), name


def _ref_vectors():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")) as fh:
        return json.load(fh)


def test_checkpoint_key_filter_equals_the_executed_reference_load_model_weights():
    """100 cases (DDP / torch.compile prefixes in both orders, load_kwargs variants, compiled receiver or not) produced
    by executing decision_transformer_sb3.py:1120-1184 (make_golden_from_reference.py::load_model_weights_trace)."""
    from lram_amd.weights import filter_policy_dict
    v = _ref_vectors()["load_model_weights_trace"]
    assert len(v["cases"]) == 100
    for case in v["cases"]:
        canned = {case["prefix"] + k: i for i, k in enumerate(v["checkpoint_keys"])}
        got = filter_policy_dict(canned, case["load_kwargs"], compile=case["compile"])
        assert list(got.keys()) == case["loaded_keys"], case
        assert case["strict"] is False
        assert (case["state_mean"] == "MEAN") == case["with_variables"]


def test_checkpoint_key_names_come_from_reference_code():
    """Every non-third-party path component of the keys `reference_layout` expects is a module attribute the
    reference's own classes assign (AST walk) or a parameter of the HF base class; the action head is a Sequential
    (`make_head` executed) -> `action_net.0.*`."""
    names = _ref_vectors()["checkpoint_key_names"]
    top = set(names["OnlineDecisionTransformerModel"]) | set(names["DiscreteDTModel"]) | \
        set(names["MultiDomainDiscreteDTModel"]) | {k.split(".")[0] for k in names["hf_DecisionTransformerModel_params"]}
    assert names["make_head_params"] == ["0.bias", "0.weight"]
    for spec_name, img in (("xlstm_16m", True), ("mamba_48m", False)):
        spec = preset(spec_name)
        for key in reference_layout(spec, with_image_encoder=img):
            parts = key.split(".")
            assert parts[0] in top, key
            if parts[0] == "action_net":
                assert parts[1] == "0" and parts[2] in ("weight", "bias")
            if parts[0] == "embed_image":
                assert parts[1] in names["ImpalaCNN"], key
                if parts[1] == "cnn":
                    assert parts[3] in names["ImpalaCNNBlock"], key
                    if parts[3].startswith("residual_"):
                        assert parts[4] in names["ImpalaCNNResidual"], key
            if parts[0] == "encoder":
                enc = names["xLSTMEncoder"] if spec.backbone == "xlstm" else names["MambaEncoder"]
                assert parts[1] in enc, key
    for k in ("embed_state.weight", "embed_return.bias", "embed_ln.weight"):
        assert k in names["hf_DecisionTransformerModel_params"]


def test_every_reference_preset_resolves_and_engine_limits_are_stated():
    """`reference_presets` = the resolved agent_params of all 16 recurrent presets of the reference's Hydra tree (composed
    by this package's loader from the reference's own YAML files, values only): each becomes a ModelSpec with the sizes
    the [3P] packages derive (inner = ceil64(2 D), ffn = ceil64(1.3 D), dt_rank = ceil(D / 16)), and the engine's geometry
    limits admit every one of them as an mLSTM-only stack; with sLSTM blocks two of the *_half presets drop out."""
    import math
    from lram_amd.config import engine_limits, spec_from_agent_params
    presets = _ref_vectors()["reference_presets"]
    assert len(presets) == 16
    unsupported = set()
    for name, ap in presets.items():
        spec = spec_from_agent_params(ap)
        hf = ap["huggingface"]
        if name.startswith("xlstm_"):
            D = hf["hidden_size"]
            assert spec.backbone == "xlstm" and spec.d_model == D and spec.n_blocks == hf["n_layer"] and spec.n_heads == 4
            assert spec.inner == math.ceil(2 * D / 64) * 64 and spec.ffn_dim == math.ceil(1.3 * D / 64) * 64
            assert spec.conv_k == 4 and spec.qkv_blocksize == 4 and spec.context_length == 150
        else:
            D = hf["d_model"]
            assert spec.backbone == "mamba" and spec.d_model == D and spec.n_blocks == hf["n_layer"]
            assert spec.d_inner == 2 * D and spec.d_state == 16 and spec.dt_rank == math.ceil(D / 16)
        assert spec.state_dim == 204 and spec.act_dim == 8 and spec.n_vocab == 274 and spec.max_length == hf["max_length"]
        assert engine_limits(spec) == [], (name, engine_limits(spec))
        if spec.backbone == "xlstm" and engine_limits(dataclasses.replace(spec, slstm_at=[1])):
            unsupported.add(name)
    assert unsupported == {"xlstm_mediumplus_half", "xlstm_large_half"}     # sLSTM head dims 266 / 358
    # the configurations BASELINE.json names
    for name in ("xlstm_medium", "xlstm_huge", "mamba_mediumplus"):
        assert engine_limits(spec_from_agent_params(presets[name])) == []


def test_sb3_zip_roundtrip_and_prefix_strip(tmp_path):
    import zipfile
    spec = preset("mamba_tiny")
    sd = init_state_dict(spec, seed=2)
    sd["predict_state.weight"] = torch.zeros(3, 3)  # dropped: load_state_head defaults to False upstream
    sd["predict_return.weight"] = torch.zeros(1, 3)  # extra keys are ignored like strict=False upstream
    path = str(tmp_path / "model.zip")
    save_sb3_zip(path, sd, state_mean=torch.zeros(20), state_std=torch.ones(20), prefix="module._orig_mod.",
                 optimizer_state={"state": {}, "param_groups": []})
    assert set(zipfile.ZipFile(path).namelist()) == {"data", "pytorch_variables.pth", "policy.pth", "optimizer.pth",
                                                     "system_info.txt"}   # agent_utils.py:165-202
    sd2, mean, std = load_sb3_zip(path)
    # as upstream: the exclusion list is matched after removing "module." only, so behind a compile prefix the state
    # head survives (and is then ignored by strict=False); from a DDP-only checkpoint it is dropped
    assert "predict_state.weight" in sd2 and "predict_return.weight" in sd2
    path_ddp = str(tmp_path / "model_ddp.zip")
    save_sb3_zip(path_ddp, sd, prefix="module.")
    sd3, mean3, _ = load_sb3_zip(path_ddp)
    assert "predict_state.weight" not in sd3 and "predict_return.weight" in sd3 and mean3 is None
    from lram_amd.weights import load_report
    assert load_report(spec, sd3) == ([], ["predict_return.weight"])
    check_state_dict(spec, sd2)
    assert all(torch.equal(sd[k], sd2[k]) for k in reference_layout(spec))
    assert mean.shape == (20,) and std.shape == (20,)
    del sd2["encoder.layers.0.mixer.A_log"]
    with pytest.raises(KeyError):
        check_state_dict(spec, sd2)


def test_engine_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lram_amd.engine import Engine
    spec = preset("xlstm_tiny")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(spec, init_state_dict(spec, 0), 2)


def test_dmc_index_tables():
    from lram_amd import obs
    from lram_amd.rollout import CHEETAH_RUN_OBS_INDEX
    assert obs.DMC_FULL_OBS_DIM == 204
    st = obs.dmc_start_index()
    assert st["velocity"] == 14 and st["position"] == 41 and st["height"] == 203   # dmcontrol_utils.py:44-49
    inv = obs.dmc_inverse_index(obs.CHEETAH_RUN_SPEC)
    filled = (inv >= 0).nonzero().flatten().tolist()
    assert sorted(filled) == sorted(CHEETAH_RUN_OBS_INDEX)
    assert inv[41].item() == 0 and inv[14].item() == 8 and inv[22].item() == 16
