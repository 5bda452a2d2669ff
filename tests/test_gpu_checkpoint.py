"""GPU, SURVEY 8 row f1: an SB3-style checkpoint zip -> load_sb3_zip -> engine, against the oracle run on the ORIGINAL
state dict.  The zip has what the reference's save() writes (agent_utils.py:165-202) and what its load has to cope with
(decision_transformer_sb3.py:1120-1184): DDP + torch.compile key prefixes, heads the rollout never evaluates,
optimizer state, `state_mean` / `state_std` in pytorch_variables.pth."""
import pytest
import torch

from lram_amd import init_state_dict, preset
from lram_amd.weights import check_state_dict, load_report, load_sb3_zip, save_sb3_zip
from oracle.dt_ref import OraclePolicy
from tests.helpers import assert_actions_match, make_inputs, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,prefix,discrete", [("xlstm_tiny", "module._orig_mod.", False), ("mamba_tiny", "module.", False),
                                                  ("xlstm_16m", "_orig_mod.", False), ("xlstm_tiny", "", True)])
def test_checkpoint_zip_to_engine_matches_oracle_on_original_weights(hip_lib, tmp_path, name, prefix, discrete):
    from lram_amd.agent import RecurrentAgent
    spec = preset(name)
    sd = init_state_dict(spec, seed=11, with_image_encoder=(name == "xlstm_tiny"))
    g = torch.Generator().manual_seed(4)
    mean = torch.randn(spec.state_dim, generator=g) * 0.2
    std = torch.rand(spec.state_dim, generator=g) + 0.5
    ckpt = dict(sd)
    D = spec.d_model
    ckpt.update({"embed_timestep.weight": torch.randn(1000, D, generator=g), "predict_state.weight": torch.randn(spec.state_dim, D, generator=g),
                 "predict_state.bias": torch.zeros(spec.state_dim), "predict_return.weight": torch.randn(1, D, generator=g),
                 "predict_return.bias": torch.zeros(1), "predict_reward.weight": torch.randn(1, D, generator=g),
                 "predict_reward.bias": torch.zeros(1), "embed_action_disc.weight": torch.randn(275, D, generator=g)})
    path = str(tmp_path / "rl_model.zip")
    save_sb3_zip(path, ckpt, state_mean=mean, state_std=std, prefix=prefix,
                 optimizer_state={"state": {}, "param_groups": [{"lr": 1e-4}]})
    loaded, mean2, std2 = load_sb3_zip(path)
    check_state_dict(spec, loaded, with_image_encoder=(name == "xlstm_tiny"))
    missing, unexpected = load_report(spec, loaded, with_image_encoder=(name == "xlstm_tiny"))
    assert missing == [] and all(k.split(".")[0] in ("embed_timestep", "predict_state", "predict_return", "predict_reward",
                                                      "embed_action_disc") for k in unexpected)
    assert torch.equal(mean2, mean) and torch.equal(std2, std)
    B = 4
    agent = RecurrentAgent(spec, loaded, n_envs=B, device="cuda:0", state_mean=mean2, state_std=std2, discrete=discrete)
    ora = OraclePolicy(spec, sd, state_mean=mean, state_std=std)          # original weights, original statistics
    ties = 0
    for t, (obs, rtg, rew, mask) in enumerate(make_inputs(spec, B, 6, seed=5)):
        a = agent.predict_batch(obs.cuda(), rtg.cuda(), None, mask.cuda())
        ref, dbg = ora.step(obs, rtg, rew, mask, discrete=discrete, return_debug=True)
        torch.cuda.synchronize()
        ties += assert_actions_match(a, ref, dbg["logits"], spec, discrete, what=f"{name} checkpoint step {t}")
        _, hidden, _ = agent.engine.taps()
        assert rel_err(hidden, dbg["hidden"]) < 2e-4
    assert ties == 0
    agent.engine.close()


def test_checkpoint_missing_backbone_key_is_reported(hip_lib, tmp_path):
    from lram_amd.engine import Engine
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=1)
    bad = {k: v for k, v in sd.items() if not k.endswith("learnable_skip")}
    path = str(tmp_path / "bad.zip")
    save_sb3_zip(path, bad)
    loaded, _, _ = load_sb3_zip(path)
    missing, _ = load_report(spec, loaded)
    assert missing and all(k.endswith("learnable_skip") for k in missing)
    with pytest.raises(KeyError):
        check_state_dict(spec, loaded)
    with pytest.raises(Exception):
        Engine(spec, loaded, 2, device="cuda:0")       # the engine refuses incomplete weights (no silent defaults)
