"""GPU: the reference-trajectory modes of the Mamba agent (lram_set_compat_mode).

`mamba_agent_trace` in tests/golden/reference_vectors.json was produced by EXECUTING the reference's
DiscreteDecisionMamba.get_action_pred / InferenceParams.reset() / MambaEncoder.forward
(tests/golden/make_golden_from_reference.py::mamba_agent_trace); the engine in compat mode must return those actions."""
import json
import os

import pytest
import torch

from lram_amd import init_state_dict, preset
from oracle.dt_ref import OraclePolicy
from tests.helpers import assert_actions_match, make_inputs, rel_err
from tests.test_oracle_golden import _mamba_trace_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _trace():
    with open(os.path.join(GOLD, "reference_vectors.json")) as fh:
        return json.load(fh)["mamba_agent_trace"]


def test_engine_compat_mode_reproduces_the_executed_reference_trace(hip_lib):
    from lram_amd.engine import Engine
    v = _trace()
    spec, sd, obs, rtg, masks, want = _mamba_trace_inputs(v)
    R, B = v["env_act_dim"], obs.shape[1]
    eng = Engine(spec, sd, B, device="cuda:0")
    eng.set_compat_mode(R, True)
    assert eng.compat_mode == {"mamba_repeat": R, "stale_state": True}
    zero = torch.zeros(B, device="cuda")
    for t in range(obs.shape[0]):
        a, _ = eng.step(obs[t].cuda(), rtg[t].cuda(), zero, masks[t].cuda())
        torch.cuda.synchronize()
        assert float((a.cpu()[:, :R] - want[t]).abs().max()) <= 1e-4, t
    eng.close()


def test_agent_compat_flags_single_env_surface(hip_lib):
    """RecurrentAgent(compat_mamba_repeat, compat_stale_state) through the reference's single-env surface:
    get_action_pred per step, inference_params.reset() at the episode ends (evaluation.py:248-251)."""
    from lram_amd.agent import RecurrentAgent
    v = _trace()
    spec, sd, obs, rtg, masks, want = _mamba_trace_inputs(v)
    R = v["env_act_dim"]
    for e in range(obs.shape[1]):
        agent = RecurrentAgent(spec, sd, n_envs=1, device="cuda:0", compat_mamba_repeat=True, compat_stale_state=True)
        assert agent.trajectory_mode == {"compat_mamba_repeat": True, "compat_stale_state": True}
        for t in range(obs.shape[0]):
            a, _ = agent.get_action_pred(agent.policy, obs[t, e].view(1, 1, -1), torch.zeros(1, 1, spec.act_dim),
                                         torch.zeros(1, 1, 1), rtg[t, e].view(1, 1, 1), torch.tensor([[t]]), None, True,
                                         None, is_eval=True, env_act_dim=R)
            assert a.shape == (R,)
            assert float((a.cpu() - want[t, e]).abs().max()) <= 1e-4, (e, t)
            if t in v["episode_end_after_step"][e]:
                agent.inference_params.reset()
        agent.engine.close()


@pytest.mark.parametrize("repeat,stale", [(4, False), (1, True), (3, True)])
def test_compat_modes_match_oracle_with_random_resets(hip_lib, repeat, stale):
    """Batched, random staggered resets: engine (mode) == oracle (same mode), actions and final conv / ssm state; and the
    modes really differ from the default trajectory."""
    from lram_amd.engine import Engine
    spec = preset("mamba_tiny")
    sd = init_state_dict(spec, seed=3)
    B, steps = 9, 10
    seq = make_inputs(spec, B, steps, seed=77, reset_prob=0.2)
    eng = Engine(spec, sd, B, device="cuda:0")
    eng.set_compat_mode(repeat, stale)
    ora = OraclePolicy(spec, sd, mamba_repeat=repeat, stale_state=stale)
    plain = OraclePolicy(spec, sd)
    ties = diff_from_plain = 0
    for obs, rtg, rew, mask in seq:
        a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        ties += assert_actions_match(a, ref, dbg["logits"], spec, what=f"repeat={repeat} stale={stale}")
        diff_from_plain += int((plain.step(obs, rtg, rew, mask) != ref).sum())
    assert ties == 0
    assert diff_from_plain > 0
    pkv = eng.export_past_key_values()
    for i in range(spec.n_blocks):
        assert rel_err(pkv[i][0], ora.state[i][0]) < 2e-4
        assert rel_err(pkv[i][1], ora.state[i][1]) < 2e-4
    # lram_reset in stale mode empties layer 0 only
    eng.reset()
    pkv2 = eng.export_past_key_values()
    assert float(pkv2[0][1].abs().max()) == 0.0
    if stale:
        assert torch.equal(pkv2[1][1], pkv[1][1])
    else:
        assert float(pkv2[1][1].abs().max()) == 0.0
    eng.close()


@pytest.mark.parametrize("scheme,steps", [("exercise", 4), ("reference", 30), ("trained_like", 30)])
def test_compat_mode_at_mamba_48m_shapes(hip_lib, scheme, steps):
    """The reference Mamba agent's trajectory (4 forwards per env-step, layer-0-only resets) at Mamba-48M shapes, also on the weight
    distributions a fresh / a trained model has (dt bias from the initialiser's range resp. at both of its ends, A_log up to
    log 16 + 2: lram_amd/weights.py), 120 state advances per env."""
    from lram_amd.engine import Engine
    spec = preset("mamba_48m")
    sd = init_state_dict(spec, seed=0, scheme=scheme)
    B = 3
    seq = make_inputs(spec, B, steps, seed=5, reset_prob=0.3 if scheme == "exercise" else 0.05)
    if scheme == "trained_like":
        for x in seq:
            x[0][:, 3] *= 30.0
    eng = Engine(spec, sd, B, device="cuda:0")
    eng.set_compat_mode(4, True)   # Meta-World: env_act_dim 4
    ora = OraclePolicy(spec, sd, mamba_repeat=4, stale_state=True)
    ties = 0
    for obs, rtg, rew, mask in seq:
        a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        ties += assert_actions_match(a, ref, dbg["logits"], spec, what="mamba_48m compat")
    assert ties == 0
    eng.close()


def test_compat_mode_is_rejected_on_xlstm(hip_lib):
    from lram_amd.agent import RecurrentAgent
    from lram_amd.engine import Engine, LramError
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=0)
    eng = Engine(spec, sd, 2, device="cuda:0")
    with pytest.raises(LramError):
        eng.set_compat_mode(4, False)
    with pytest.raises(LramError):
        eng.set_compat_mode(1, True)
    eng.set_compat_mode(1, False)
    eng.close()
    with pytest.raises(ValueError):
        RecurrentAgent(spec, sd, n_envs=1, device="cuda:0", compat_mamba_repeat=True)


def test_shared_repeated_forwards_equal_the_unshared_ones(hip_lib, monkeypatch):
    """LRAM_COMPAT_SHARE (default on): the repeated forwards of the reference-trajectory mode share the token front end and
    layer 0's norm + in_proj and evaluate only the head columns a pass needs.  Same actions and the same final state as
    recomputing everything in every pass (Mamba-48M shapes, random resets, 4 forwards per env-step)."""
    from lram_amd.engine import Engine
    spec = preset("mamba_48m")
    sd = init_state_dict(spec, seed=3)
    B, R = 9, 4
    seq = make_inputs(spec, B, 6, seed=77, reset_prob=0.2)
    outs = {}
    for share in ("1", "0"):
        monkeypatch.setenv("LRAM_COMPAT_SHARE", share)
        eng = Engine(spec, sd, B, device="cuda:0")
        eng.set_compat_mode(R, True)
        acts = []
        for obs, rtg, rew, mask in seq:
            a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
            acts.append(a.clone())
        torch.cuda.synchronize()
        outs[share] = (torch.stack(acts), eng.export_state_tensor(spec.n_blocks - 1, 0).clone(), eng.export_state_tensor(0, 0).clone())
        eng.close()
    monkeypatch.delenv("LRAM_COMPAT_SHARE")
    assert float((outs["1"][0][..., :R] - outs["0"][0][..., :R]).abs().max()) <= 1e-4
    assert rel_err(outs["1"][1], outs["0"][1]) < 1e-5 and rel_err(outs["1"][2], outs["0"][2]) < 1e-5


def test_graph_replay_with_shared_repeated_forwards(hip_lib):
    """hipGraph decode + the reference-trajectory mode (repeated forwards sharing the front end): the X0 / U0 workspace is
    allocated BEFORE the capture begins (a hipMalloc inside a capture would invalidate it and every later call would retry
    and fail -- round-3 advisor finding).  Same actions and state as the un-captured engine, over a second and third replay."""
    from lram_amd.engine import Engine
    spec = preset("mamba_tiny")
    sd = init_state_dict(spec, seed=5)
    B, R = 3, 3
    seq = make_inputs(spec, B, 5, seed=78, reset_prob=0.0)
    eager = Engine(spec, sd, B, device="cuda:0")
    eager.set_compat_mode(R, True)
    graph = Engine(spec, sd, B, device="cuda:0")
    graph.set_compat_mode(R, True)
    graph.set_graph_mode(True)
    d_obs, d_rtg = torch.empty(B, spec.state_dim, device="cuda"), torch.empty(B, device="cuda")
    d_rew = torch.zeros(B, device="cuda")
    for obs, rtg, rew, _ in seq:
        a_e, _ = eager.step(obs.cuda(), rtg.cuda(), rew.cuda(), None)
        d_obs.copy_(obs), d_rtg.copy_(rtg)
        a_g, _ = graph.step(d_obs, d_rtg, d_rew, None)      # fixed device buffers: one capture, then replays
        torch.cuda.synchronize()
        assert float((a_e[:, :R] - a_g[:, :R]).abs().max()) <= 1e-4
    assert rel_err(graph.export_state_tensor(0, 0), eager.export_state_tensor(0, 0)) < 1e-5
    eager.close(), graph.close()
