"""Pin the oracle against vectors produced by executing the reference's own importable files
(tests/golden/make_golden_from_reference.py) and against the committed oracle KATs."""
import json
import os

import numpy as np
import pytest
import torch

from lram_amd import init_state_dict, preset
from oracle import dt_ref, xlstm_ref
from tests.golden.make_kat import KATS, weights_l1
from tests.helpers import rel_err

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ref_vectors():
    with open(os.path.join(GOLD, "reference_vectors.json")) as fh:
        return json.load(fh)


def test_minmax_tokenizer_matches_reference(ref_vectors):
    for key, shift in (("minmax_shift18", 18), ("minmax_shift0", 0)):
        v = ref_vectors[key]
        x = torch.tensor(v["x"])
        tok = dt_ref.minmax_tokenize(x, 256, shift)
        assert tok.tolist() == v["tokens"]
        n = len(v["inv_table"])
        inv = dt_ref.minmax_inv_tokenize(torch.arange(n), 256, shift)
        assert inv.tolist() == v["inv_table"]  # exact in fp32
    # spot values quoted in SURVEY.md 8c
    t = dt_ref.minmax_tokenize(torch.tensor([-1, -0.999, 0, 0.5, 0.9999, 1.0]), 256, 18)
    assert t.tolist() == [18, 18, 146, 210, 273, 273]


def test_rms_norm_matches_reference(ref_vectors):
    for case in ref_vectors["llama_rms_norm"]:
        x, w, y = torch.tensor(case["x"]), torch.tensor(case["weight"]), torch.tensor(case["y"])
        out = xlstm_ref.rms_norm(x, w, case["eps"])
        assert torch.equal(out, y) or float((out - y).abs().max()) < 1e-6


@pytest.mark.parametrize("name", sorted(KATS))
def test_oracle_reproduces_kat(name):
    kw = dict(KATS[name])
    kat = np.load(os.path.join(GOLD, f"kat_{name}.npz"))
    spec = preset(kw.get("preset", name))
    sd = init_state_dict(spec, seed=kw["seed"])
    assert abs(weights_l1(sd) / float(kat["weights_l1"]) - 1.0) < 1e-9, \
        "seeded weight generation drifted from the committed KAT; regenerate with tests/golden/make_kat.py"
    ora = dt_ref.OraclePolicy(spec, sd)
    for t in range(kw["steps"]):
        a, dbg = ora.step(torch.from_numpy(kat["obs"][t]), torch.from_numpy(kat["rtg"][t]),
                          torch.from_numpy(kat["rew"][t]), torch.from_numpy(kat["mask"][t]),
                          discrete=kw["discrete"], return_debug=True)
        assert np.array_equal(a.numpy().astype(np.float32), kat["actions"][t]), (name, t)
        assert rel_err(dbg["hidden"], torch.from_numpy(kat["hidden"][t])) < 1e-5


def test_obs_full_space_tables_match_the_reference_mapping():
    """lram_amd/obs.py's DMControl / Mimicgen slot tables and index construction reproduce the reference's own
    `map_flattened_obs_to_full_space` (executed by make_golden_from_reference.py) on its inputs."""
    import json
    import os
    from lram_amd import obs
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["obs_full_space"]
    tables = {"dmc": obs.DMC_OBSTYPE_TO_DIM, "mimicgen": obs.MIMICGEN_OBSTYPE_TO_DIM}
    for name, case in vec.items():
        table = tables[name.split("_")[0]]
        assert table == case["obstype_to_dim"] and list(table) == list(case["obstype_to_dim"]), name  # values and order
        assert sum(table.values()) == case["full_dim"]
        inv = obs.inverse_index([tuple(x) for x in case["spec"]], table, 204)
        x = torch.tensor(case["x"])
        full = torch.tensor(case["full"])
        got = obs.apply_inverse_index(x, inv)
        assert got.shape == (x.shape[0], 204)
        assert torch.equal(got[:, : case["full_dim"]], full) and not bool(got[:, case["full_dim"]:].any())


def test_action_head_postprocessing_matches_the_reference_methods():
    """oracle.dt_ref.actions_from_logits == the reference's prepare_action_logits + get_action_from_logits
    (executed by make_golden_from_reference.py with the real min-max tokenizer), continuous and discrete."""
    import json
    import os
    from lram_amd import preset
    from oracle import dt_ref
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["action_from_logits"]
    spec = preset("xlstm_16m")
    logits = torch.tensor(vec["logits"])
    cont, _ = dt_ref.actions_from_logits(spec, logits, discrete=False)
    disc, _ = dt_ref.actions_from_logits(spec, logits, discrete=True)
    assert torch.equal(cont, torch.tensor(vec["continuous"]))
    assert disc.view(-1).tolist() == vec["discrete"]


def test_token_front_end_matches_the_reference_methods():
    """oracle.dt_ref.embed_tokens == the reference's compute_inputs(use_inference_cache=True) chain (executed by
    make_golden_from_reference.py): last timestep only, token order (state, rtg, reward), embed_ln; and the action is
    read at the rtg token (tok_to_pred_pos["a"] == 1 == ModelSpec.pred_token)."""
    import json
    import os
    from lram_amd import preset
    from oracle import dt_ref
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["token_front_end"]
    sd = {k: torch.tensor(v) for k, v in vec["state_dict"].items()}
    states, rtg, rew = torch.tensor(vec["states"]), torch.tensor(vec["returns_to_go"]), torch.tensor(vec["rewards"])
    spec = preset("xlstm_16m")
    got = dt_ref.embed_tokens(spec, sd, states[:, -1], rtg[:, -1, 0], rew[:, -1, 0])
    ref = torch.tensor(vec["stacked_inputs"])
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-6
    assert vec["tok_to_pos"] == {"s": 0, "rtg": 1, "r": 2}
    assert vec["tok_to_pred_pos"]["a"] == spec.pred_token == 1 and spec.tokens_per_step == 3


def test_impala_cnn_matches_the_reference_modules():
    """oracle.dt_ref.impala_cnn == the reference's ImpalaCNNBlock / ImpalaCNNResidual modules chained as ImpalaCNN
    chains them (executed by make_golden_from_reference.py), from uint8 frames to the state-token embedding."""
    import os
    import numpy as np
    from oracle import dt_ref
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "impala_cnn_reference.npz"))
    sd = {k: torch.from_numpy(d[k]) for k in d.files if k.startswith("embed_image.")}
    img = torch.from_numpy(d["images"])
    out = dt_ref.impala_cnn(sd, "embed_image.", img.float() / 255.0)
    ref = torch.from_numpy(d["out"])
    assert out.shape == ref.shape and float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def _mamba_trace_inputs(v):
    spec = preset(v["preset"])
    sd = init_state_dict(spec, seed=v["weight_seed"])
    obs, rtg = torch.tensor(v["obs"]), torch.tensor(v["rtg"])
    steps, B = obs.shape[0], obs.shape[1]
    masks = torch.zeros(steps, B, dtype=torch.uint8)
    masks[0] = 1
    for e, ends in enumerate(v["episode_end_after_step"]):
        for t in ends:
            if t + 1 < steps:
                masks[t + 1, e] = 1  # inference_params.reset() after step t == reset mask before step t + 1
    return spec, sd, obs, rtg, masks, torch.tensor(v["returned_actions"])


def test_oracle_mamba_compat_modes_follow_the_executed_reference_control_flow(ref_vectors):
    """`mamba_agent_trace` = DiscreteDecisionMamba.get_action_pred + InferenceParams.reset() + MambaEncoder.forward
    executed from the reference (stand-in layers around the oracle's mixer math): the oracle in
    (mamba_repeat = env_act_dim, stale_state) mode must reproduce the returned actions; the default mode (one advance
    per env-step, full reset) must NOT -- the two trajectories are different things."""
    v = ref_vectors["mamba_agent_trace"]
    spec, sd, obs, rtg, masks, want = _mamba_trace_inputs(v)
    R = v["env_act_dim"]
    ora = dt_ref.OraclePolicy(spec, sd, mamba_repeat=R, stale_state=True)
    plain = dt_ref.OraclePolicy(spec, sd)
    differs = 0
    for t in range(obs.shape[0]):
        a = ora.step(obs[t], rtg[t], torch.zeros(obs.shape[1]), masks[t])
        assert torch.equal(a[:, :R], want[t]), f"step {t}: {a[:, :R]} vs {want[t]}"
        b = plain.step(obs[t], rtg[t], torch.zeros(obs.shape[1]), masks[t])
        differs += int((b[:, :R] != want[t]).sum())
    assert differs > 0
    # each quirk alone is not the reference trajectory either
    for kw in ({"mamba_repeat": R}, {"stale_state": True}):
        o = dt_ref.OraclePolicy(spec, sd, **kw)
        d = 0
        for t in range(obs.shape[0]):
            d += int((o.step(obs[t], rtg[t], torch.zeros(obs.shape[1]), masks[t])[:, :R] != want[t]).sum())
        assert d > 0, kw


# ---- the reference's xLSTM inference wrapper, executed (make_golden_from_reference.py::xlstm_model_trace) ------------
def _xlstm_trace_env(v, e):
    import torch.nn.functional as Fn
    spec = preset(v["preset"])
    sd = init_state_dict(spec, seed=v["weight_seed"])
    env = v["envs"][e]
    obs = torch.tensor(env["obs"])
    obs = Fn.pad(obs, (0, spec.state_dim - obs.shape[1]))     # pad_inputs: zero-pad to max_state_dim
    return spec, sd, env, obs


def test_oracle_policy_equals_the_executed_xlstm_wrapper(ref_vectors):
    """`xlstm_model_trace`: MultiDomainDiscreteDecisionXLSTMModel.forward -> compute_hidden_states ->
    handle_inference_cache -> xLSTMEncoder.forward -> get_predictions, under the executed agent chain predict ->
    pad_inputs -> get_action_pred, with `encoder.layers.step` = the oracle's block stack.  OraclePolicy (the checker
    every GPU parity test uses) must reproduce it exactly: the hidden state of all three tokens, the logits, the actions,
    with the cache emptied exactly where the reference's past_key_values is None."""
    v = ref_vectors["xlstm_model_trace"]
    assert v["tok_to_pred_pos_a"] == 1 and v["tok_to_pos"] == {"s": [0], "rtg": 1, "r": 2}
    for e in range(len(v["envs"])):
        spec, sd, env, obs = _xlstm_trace_env(v, e)
        assert spec.pred_token == v["tok_to_pred_pos_a"] and spec.tokens_per_step == 3
        ora = dt_ref.OraclePolicy(spec, sd)
        A = env["env_act_dim"]
        n_drops = 0
        for t, want in enumerate(env["returned"]):
            # what reaches the block stack: one [1, 1, D] step call per token, a fresh state exactly on a dropped cache
            calls = env["stack_calls"][t]
            assert [c[0] for c in calls] == [[1, 1, spec.d_model]] * 3
            assert [c[1] for c in calls] == [env["cache_is_none"][t], False, False]
            mask = torch.tensor([1 if env["cache_is_none"][t] else 0], dtype=torch.uint8)
            n_drops += int(mask[0])
            a, dbg = ora.step(obs[t:t + 1], torch.tensor([env["rtg_in"][t]]), torch.zeros(1), mask, return_debug=True)
            # bit-exact, except on the step after a mid-episode cache drop: there the reference embeds the whole
            # context_len-step context in one Linear call before cutting it down to the last 3 tokens, and a 5-row
            # matrix product may round differently from a 1-row one (last bit); actions are exact everywhere
            multi_row = env["cache_is_none"][t] and t > 0 and (t - 1) not in env["episode_end_after_step"]
            if env["cache_is_none"][t]:
                drifted = multi_row        # a fresh state forgets earlier last-bit differences
            for got, ref in ((dbg["hidden"][0], torch.tensor(env["hidden"][t])), (dbg["logits"][0], torch.tensor(env["logits"][t]))):
                if drifted:
                    assert torch.allclose(got, ref, rtol=0, atol=1e-5 * float(ref.abs().max())), (e, t)
                else:
                    assert torch.equal(got, ref), (e, t)
            assert torch.equal(a[0, :A], torch.tensor(want)), (e, t)
        assert n_drops >= 3                                   # start + reset_inf_cache_freq drops (+ the episode end)
        if env["final_state_is_none"]:     # the last step fell on a reset_inf_cache_freq boundary: cache dropped after it
            assert e == 1
            continue
        for got, ref in ((ora.state["block_0"]["mlstm_state"][1], env["final_mlstm_n_block0"]),
                         (ora.state["block_1"]["slstm_state"], env["final_slstm_block1"])):
            ref = torch.tensor(ref)
            assert torch.allclose(got.reshape(-1), ref, rtol=0, atol=1e-5 * float(ref.abs().max()))


class _OracleBackedEngine:
    """Engine stand-in for CPU tests of the agent surface: lram_step / lram_reset semantics on top of OraclePolicy."""

    def __init__(self, spec, sd):
        self.ora, self.device, self._pending = dt_ref.OraclePolicy(spec, sd), torch.device("cpu"), True

    def step(self, obs, rtg, rew, reset_mask, discrete=False, obs_is_embedding=False):
        mask = torch.tensor([1 if self._pending else 0], dtype=torch.uint8)
        self._pending = False
        return self.ora.step(obs, rtg, rew, mask, discrete=discrete), None

    def reset(self, mask=None):
        self._pending = True


def run_agent_over_xlstm_trace(make_agent, v, e, atol=0.0):
    """Drive an agent's reference surface (predict / past_key_values = None) through the rollout bookkeeping of
    custom_evaluate_policy exactly as the trace generator did; returns the largest action difference."""
    spec, sd, env, _ = _xlstm_trace_env(v, e)
    agent = make_agent(spec, sd)
    A, obs_all, env_r = env["env_act_dim"], torch.tensor(env["obs"]), torch.tensor(env["env_rewards"])
    scale, rtg0, ends = v["reward_scale"], v["target_return0"], set(env["episode_end_after_step"])
    states, actions, rewards = obs_all[:1].clone(), torch.zeros((0, A)), torch.zeros(0)
    rtg, ts, t_ep = torch.tensor(rtg0).reshape(1, 1), torch.tensor(0).reshape(1, 1), 0
    worst = 0.0
    for t, want in enumerate(env["returned"]):
        actions = torch.cat([actions, torch.zeros((1, A))])
        rewards = torch.cat([rewards, torch.zeros(1)])
        a, _ = agent.predict(agent.policy, states, actions, rewards, rtg, ts, deterministic=True,
                             context_len=v["context_len"], is_eval=True, env_act_dim=A)
        a = a.detach().cpu()
        worst = max(worst, float((a - torch.tensor(want)).abs().max()))
        assert worst <= atol, (e, t, a, want)
        actions[-1] = a
        rewards[-1] = env_r[t] / scale
        t_ep += 1
        if t in ends:
            states, actions, rewards = obs_all[t + 1: t + 2].clone(), torch.zeros((0, A)), torch.zeros(0)
            rtg, ts, t_ep = torch.tensor(rtg0).reshape(1, 1), torch.tensor(0).reshape(1, 1), 0
            agent.past_key_values = None
            continue
        states = torch.cat([states, obs_all[t + 1: t + 2]])
        rtg = torch.cat([rtg, (rtg[0, -1] - env_r[t] / scale).reshape(1, 1)], dim=1)
        ts = torch.cat([ts, torch.full((1, 1), t_ep)], dim=1)
    return worst


def test_agent_surface_reproduces_the_executed_xlstm_wrapper(ref_vectors):
    """RecurrentAgent.predict / get_action_pred / `past_key_values = None` on an oracle-backed engine stand-in: same
    actions as the executed reference model + agent, bit for bit, over cache drops and an episode end."""
    import dataclasses
    from lram_amd.agent import RecurrentAgent, _InferenceParams
    v = ref_vectors["xlstm_model_trace"]

    def make(spec, sd):
        agent = object.__new__(RecurrentAgent)
        agent.spec = dataclasses.replace(spec, reset_inf_cache_freq=v["reset_inf_cache_freq"])
        agent.engine, agent.device, agent.n_envs, agent.is_discrete = _OracleBackedEngine(spec, sd), torch.device("cpu"), 1, False
        agent.policy, agent.image_encoder, agent.has_image_encoder = agent, None, False
        agent.state_mean = agent.state_std = None
        agent.eval_context_len, agent.reset_inf_cache_freq = v["context_len"], v["reset_inf_cache_freq"]
        agent.reprime_context, agent._zero_reward = False, torch.zeros(1)
        agent.inference_params = _InferenceParams(agent)
        return agent

    for e in range(len(v["envs"])):
        assert run_agent_over_xlstm_trace(make, v, e) == 0.0
