"""GPU: the HIP engine against the committed known-answer trajectories (tests/golden/kat_*.npz), and the
reference-compatible agent surface / batched rollout driver on the device."""
import os

import numpy as np
import pytest
import torch

from lram_amd import init_state_dict, preset
from tests.golden.make_kat import KATS, weights_l1
from tests.helpers import assert_actions_match, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", sorted(KATS))
def test_engine_reproduces_kat(hip_lib, name):
    from lram_amd.engine import Engine
    kw = dict(KATS[name])
    kat = np.load(os.path.join(GOLD, f"kat_{name}.npz"))
    spec = preset(kw.get("preset", name))
    sd = init_state_dict(spec, seed=kw["seed"])
    assert abs(weights_l1(sd) / float(kat["weights_l1"]) - 1.0) < 1e-9
    eng = Engine(spec, sd, kw["B"], device="cuda:0")
    for t in range(kw["steps"]):
        a, _ = eng.step(torch.from_numpy(kat["obs"][t]).cuda(), torch.from_numpy(kat["rtg"][t]).cuda(),
                        torch.from_numpy(kat["rew"][t]).cuda(), torch.from_numpy(kat["mask"][t]).cuda(),
                        discrete=kw["discrete"])
        torch.cuda.synchronize()
        a = a.cpu().numpy()
        if kw["discrete"]:
            assert np.array_equal(a[:, :1].astype(np.int64), kat["actions"][t].astype(np.int64)), (name, t)  # bit-exact
        else:
            assert np.abs(a - kat["actions"][t]).max() <= 1e-4, (name, t)  # identical bins
        _, hidden, _ = eng.taps()
        assert rel_err(hidden, torch.from_numpy(kat["hidden"][t])) < 2e-4
    if spec.backbone == "mamba":
        assert rel_err(eng.export_state_tensor(0, 3), torch.from_numpy(kat["state0_a"])) < 2e-4
        assert rel_err(eng.export_state_tensor(0, 0), torch.from_numpy(kat["state0_b"])) < 2e-4
    else:
        C = eng.export_state_tensor(0, 0)
        assert rel_err(C[:, :, :8, :8], torch.from_numpy(kat["state0_a"])) < 2e-4
        assert rel_err(eng.export_state_tensor(0, 1), torch.from_numpy(kat["state0_b"])) < 2e-4
    eng.close()


def test_agent_predict_surface_single_env(hip_lib):
    """predict()/get_action_pred()/get_action with the reference signatures, driven like
    custom_evaluate_policy drives them (growing context tensors, one env), against the oracle."""
    from lram_amd.agent import RecurrentAgent
    from oracle.dt_ref import OraclePolicy
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=31)
    agent = RecurrentAgent(spec, sd, n_envs=1, device="cuda:0", target_return=450.0, reward_scale=100.0)
    ora = OraclePolicy(spec, sd)
    dev = agent.device
    obs_dim, act_dim = 17, 3
    g = torch.Generator().manual_seed(0)
    states = torch.zeros((1, obs_dim), device=dev)
    states[0] = torch.rand(obs_dim, generator=g).to(dev)
    actions = torch.zeros((0, act_dim), device=dev)
    rewards = torch.zeros(0, device=dev)
    target_return = torch.tensor(agent.compute_target_return_val(), device=dev).reshape(1, 1)
    timesteps = torch.zeros((1, 1), dtype=torch.long, device=dev)
    agent.inference_params.reset()
    agent.past_key_values = None
    for t in range(6):
        actions = torch.cat([actions, torch.zeros((1, act_dim), device=dev)], dim=0)
        rewards = torch.cat([rewards, torch.zeros(1, device=dev)])
        action, _ = agent.predict(agent.policy, states, actions, rewards, target_return, timesteps,
                                  context_len=agent.eval_context_len, is_eval=True, env_act_dim=act_dim)
        obs_pad = torch.cat([states[-1].cpu(), torch.zeros(spec.state_dim - obs_dim)]).view(1, -1)
        ref = ora.step(obs_pad, target_return[0, -1].cpu().view(1), torch.zeros(1))[0, :act_dim]
        assert action.shape == (act_dim,) and float((action.cpu() - ref).abs().max()) <= 1e-4
        actions[-1] = action
        rewards[-1] = 1.0 / agent.get_reward_scale_for_env()
        nxt = torch.rand(obs_dim, generator=g).to(dev).view(1, -1)
        states = torch.cat([states, nxt], dim=0)
        target_return = torch.cat([target_return, (target_return[0, -1] - rewards[-1]).view(1, 1)], dim=1)
        timesteps = torch.cat([timesteps, torch.full((1, 1), t + 1, device=dev)], dim=1)
    pkv = agent.past_key_values
    assert set(pkv) == {f"block_{i}" for i in range(spec.n_blocks)}
    assert pkv["block_0"]["mlstm_state"][0].shape == (1, 4, 64, 64)
    a1, a2 = agent.get_action(agent.policy, states[None, -1:], actions[None], rewards.view(1, -1, 1),
                              target_return.view(1, -1, 1), timesteps, None, True, None, env_act_dim=act_dim)
    assert torch.equal(a1, a2)
    agent.engine.close()


def test_batched_rollout_on_device_with_images(hip_lib):
    """Discrete image domain end to end: uint8 frames -> IMPALA-CNN kernels (lram_embed_images) -> engine -> argmax
    over 18."""
    from lram_amd.agent import RecurrentAgent
    from lram_amd.config import ModelSpec
    from lram_amd.rollout import BatchedRollout, SyntheticVecEnv
    from oracle.dt_ref import OraclePolicy
    spec = ModelSpec(backbone="xlstm", d_model=128, n_blocks=2, slstm_at=[1], state_dim=20, act_dim=4)
    sd = init_state_dict(spec, seed=41, with_image_encoder=True)
    B = 6
    agent = RecurrentAgent(spec, sd, n_envs=B, device="cuda:0", discrete=True)
    env = SyntheticVecEnv(B, act_dim=1, ep_len=5, device="cuda:0", image_shape=(3, 64, 64), seed=3)
    ro = BatchedRollout(agent, env, target_return=90.0, reward_scale=10.0, env_act_dim=1)
    ora = OraclePolicy(spec, sd)
    ties = 0
    for t in range(7):
        obs, rtg, mask = ro.obs.clone(), ro.rtg.clone(), ro.reset_mask.clone()
        a = ro.step().clone()
        ref, dbg = ora.step(obs.cpu(), rtg.cpu(), torch.zeros(B), mask.cpu(), discrete=True, return_debug=True)
        assert a.dtype == torch.int64 and a.shape == (B, 1)
        # discrete head: bit-exact; the same 2e-4 tie rule as everywhere else, and no ties on the committed seed
        ties += assert_actions_match(a, ref, dbg["logits"], spec, discrete=True, what=f"image rollout step {t}")
    assert ties == 0
    agent.engine.close()


@pytest.mark.parametrize("d_model,B", [(128, 5), (1280, 3)])
def test_image_encoder_kernels_match_oracle(hip_lib, d_model, B):
    """lram_embed_images (hand-written conv3x3 / maxpool / linear) == the oracle's ImpalaCNN restatement
    (image_encoders.py:10-131) == the PyTorch/MIOpen module, on uint8 frames; odd batch, both model widths."""
    from lram_amd.config import ModelSpec
    from lram_amd.engine import Engine
    from tests.torch_image_encoder import ImageEncoder
    from oracle.dt_ref import impala_cnn
    spec = ModelSpec(backbone="xlstm", d_model=d_model, n_blocks=2, slstm_at=[1])
    sd = init_state_dict(spec, seed=7, with_image_encoder=True)
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (B, 3, 64, 64), generator=g, dtype=torch.uint8)
    ref = impala_cnn(sd, "embed_image.", img.float() / 255.0)
    eng = Engine(spec, sd, B, device="cuda:0")
    out = eng.embed_images(img.cuda())
    out2 = eng.embed_images(img.cuda())          # second call reuses the buffers
    torch.cuda.synchronize()
    assert out.shape == (B, d_model)
    scale = float(ref.abs().max())
    assert float((out.cpu() - ref).abs().max()) <= 2e-5 * scale
    assert torch.equal(out, out2)
    mi = ImageEncoder.from_state_dict(sd, (3, 64, 64), d_model).cuda()(img.cuda())
    assert float((out - mi).abs().max()) <= 5e-5 * scale
    with pytest.raises(Exception):
        eng.embed_images(torch.zeros(B, 4, 64, 64, dtype=torch.uint8).cuda())     # wrong channel count
    eng.close()
    plain = Engine(spec, init_state_dict(spec, seed=7), B, device="cuda:0")
    with pytest.raises(Exception):
        plain.embed_images(img.cuda())                                            # no embed_image.* weights
    plain.close()
