"""GPU edge cases of the hot path: single env, ragged batches, every tokens-per-call count for both backbones,
Mamba agent surface, cache-reset frequency, loud failures on misuse."""
import pytest
import torch

from lram_amd import init_state_dict, preset
from oracle import dt_ref, mamba_ref, xlstm_ref
from tests.helpers import assert_actions_match, make_inputs, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,B", [("xlstm_tiny", 1), ("xlstm_tiny", 3), ("mamba_tiny", 1), ("mamba_tiny", 7),
                                    ("xlstm_tiny", 130)])
def test_odd_batch_sizes(hip_lib, name, B):
    from lram_amd.engine import Engine
    spec = preset(name)
    sd = init_state_dict(spec, seed=B)
    eng = Engine(spec, sd, B, device="cuda:0")
    ora = dt_ref.OraclePolicy(spec, sd)
    for t, (obs, rtg, rew, mask) in enumerate(make_inputs(spec, B, 5, seed=B + 50)):
        a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        assert_actions_match(a, ref, dbg["logits"], spec, what=f"{name} B={B} step {t}")
        assert rel_err(eng.taps()[1], dbg["hidden"]) < 2e-4
    eng.close()


def test_mamba_encoder_step_all_token_counts(hip_lib):
    from lram_amd.engine import Engine
    spec = preset("mamba_tiny")
    sd = init_state_dict(spec, seed=3)
    B = 4
    eng = Engine(spec, sd, B, device="cuda:0")
    state = None
    g = torch.Generator().manual_seed(4)
    for T in (1, 4, 2, 3, 12, 6, 9):
        x = torch.randn(B, T, spec.d_model, generator=g)
        ref, state = mamba_ref.encoder_forward_cached(spec, sd, x, state)
        out = eng.encoder_step(x.cuda())
        torch.cuda.synchronize()
        assert rel_err(out, ref) < 2e-4, T
    pkv = eng.export_past_key_values()
    assert rel_err(pkv[0][0], state[0][0]) < 2e-4 and rel_err(pkv[1][1], state[1][1]) < 2e-4
    eng.close()


def test_prefill_equals_sequential_steps(hip_lib):
    """Context re-prime: feeding a stored trajectory through encoder_step in chunks of up to 4
    tokens leaves the same state and last hidden as token-by-token stepping."""
    from lram_amd.engine import Engine
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=8)
    B, L = 3, 24
    x = torch.randn(B, L, spec.d_model, generator=torch.Generator().manual_seed(1))
    e1 = Engine(spec, sd, B, device="cuda:0")
    e2 = Engine(spec, sd, B, device="cuda:0")
    last1 = None
    for i in range(0, L, 4):
        last1 = e1.encoder_step(x[:, i:i + 4].contiguous().cuda())[:, -1]
    for i in range(L):
        last2 = e2.encoder_step(x[:, i:i + 1].contiguous().cuda())[:, -1]
    torch.cuda.synchronize()
    assert rel_err(last1, last2) < 1e-5
    assert rel_err(e1.export_state_tensor(0, 0), e2.export_state_tensor(0, 0)) < 1e-5
    ref, _ = xlstm_ref.encoder_forward_cached(spec, sd, x, None)
    assert rel_err(last1, ref[:, -1]) < 2e-4
    e1.close(), e2.close()


def test_mamba_agent_surface_and_cache_reset_freq(hip_lib):
    from lram_amd.agent import RecurrentAgent
    import dataclasses
    spec = dataclasses.replace(preset("mamba_tiny"), reset_inf_cache_freq=3)
    sd = init_state_dict(spec, seed=12)
    agent = RecurrentAgent(spec, sd, n_envs=1, device="cuda:0", target_return=90.0, reward_scale=10.0)
    ora = dt_ref.OraclePolicy(spec, sd)
    dev = agent.device
    g = torch.Generator().manual_seed(2)
    obs_hist = torch.rand(1, 9, generator=g).to(dev)
    acts = torch.zeros((0, 2), device=dev)
    rews = torch.zeros(0, device=dev)
    rtg = torch.full((1, 1), agent.compute_target_return_val(), device=dev)
    ts = torch.zeros((1, 1), dtype=torch.long, device=dev)
    for t in range(8):
        acts = torch.cat([acts, torch.zeros((1, 2), device=dev)])
        rews = torch.cat([rews, torch.zeros(1, device=dev)])
        a, _ = agent.predict(agent.policy, obs_hist, acts, rews, rtg, ts, env_act_dim=2, is_eval=True)
        pad = torch.cat([obs_hist[-1].cpu(), torch.zeros(spec.state_dim - 9)]).view(1, -1)
        ref = ora.step(pad, rtg[0, -1].cpu().view(1), torch.zeros(1))[0, :2]
        assert float((a.cpu() - ref).abs().max()) <= 1e-4, t
        if t > 0 and t % 3 == 0:      # decision_transformer_sb3.py:663-666: cache dropped after this step
            ora.reset(1)
        obs_hist = torch.cat([obs_hist, torch.rand(1, 9, generator=g).to(dev)])
        rtg = torch.cat([rtg, rtg[:, -1:] - 0.1], dim=1)
        ts = torch.cat([ts, torch.full((1, 1), t + 1, device=dev)], dim=1)
    kv = agent.past_key_values
    assert set(kv) == {0, 1} and kv[0][0].shape == (1, spec.d_inner, spec.d_conv)
    agent.inference_params.reset()
    assert float(agent.past_key_values[0][1].abs().max()) == 0.0
    agent.engine.close()


def test_misuse_is_loud(hip_lib):
    from lram_amd.engine import Engine, LramError
    spec = preset("xlstm_tiny")
    eng = Engine(spec, init_state_dict(spec, 0), 2, device="cuda:0")
    with pytest.raises(ValueError):
        eng.encoder_step(torch.zeros(2, 5, spec.d_model).cuda()[:, ::1][:1])      # wrong batch
    with pytest.raises(LramError):
        eng.encoder_step(torch.zeros(2, 5, spec.d_model).cuda())                   # more than 4 tokens per call
    with pytest.raises(ValueError):
        eng.step(torch.zeros(2, spec.state_dim, dtype=torch.float64).cuda(), torch.zeros(2).cuda(), torch.zeros(2).cuda())
    with pytest.raises(KeyError):
        eng.export_state_tensor(1, 1)                                              # sLSTM block has no 'n' tensor
    eng.close()
    with pytest.raises(LramError):
        eng.step(torch.zeros(2, spec.state_dim).cuda(), torch.zeros(2).cuda(), torch.zeros(2).cuda())  # closed


@pytest.mark.parametrize("name", ["xlstm_tiny", "mamba_tiny"])
def test_prefill_api_equals_sequential_steps(hip_lib, name):
    """lram_prefill(L timesteps, consumed in chunks of 4 = 12 tokens per state pass) == L lram_step calls: same last
    action, same final state to fp32 rounding; L = 9 exercises chunks of 4, 4 and 1 timesteps."""
    from lram_amd.engine import Engine
    spec = preset(name)
    sd = init_state_dict(spec, seed=17)
    B, L = 5, 9
    seq = make_inputs(spec, B, L, seed=3, reset_prob=0.0)
    obs_seq = torch.stack([x[0] for x in seq], 1).contiguous().cuda()
    rtg_seq = torch.stack([x[1] for x in seq], 1).contiguous().cuda()
    rew_seq = torch.stack([x[2] for x in seq], 1).contiguous().cuda()
    e1 = Engine(spec, sd, B, device="cuda:0")
    e2 = Engine(spec, sd, B, device="cuda:0")
    for obs, rtg, rew, _ in seq:
        a_seq, _ = e1.step(obs.cuda(), rtg.cuda(), rew.cuda(), None)
    a_pre, _ = e2.prefill(obs_seq, rtg_seq, rew_seq, reset_mask=torch.ones(B, dtype=torch.uint8).cuda())
    torch.cuda.synchronize()
    assert torch.equal(a_seq, a_pre)
    for blk in range(spec.n_blocks):
        for which in (0, 3):
            assert rel_err(e1.export_state_tensor(blk, which), e2.export_state_tensor(blk, which)) < 1e-4, (blk, which)
    ora = dt_ref.OraclePolicy(spec, sd)
    for obs, rtg, rew, _ in seq:
        ref = ora.step(obs, rtg, rew)
    assert float((a_pre.cpu() - ref).abs().max()) <= 1e-4
    e1.close(), e2.close()


def test_device_obs_front_end(hip_lib):
    from lram_amd import obs as lobs
    from lram_amd.engine import pad_obs
    g = torch.Generator().manual_seed(0)
    native = torch.rand(37, 17, generator=g) * 2 - 1
    inv = lobs.dmc_inverse_index(lobs.CHEETAH_RUN_SPEC)
    mean, std = torch.randn(204, generator=g), torch.rand(204, generator=g) + 0.5
    ref = torch.zeros(37, 204)
    ref[:, 41:49] = native[:, :8]
    ref[:, 14:23] = native[:, 8:]
    out = pad_obs(native.cuda(), 204, inv.cuda(), mean.cuda(), std.cuda())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), (ref - mean) / std)
    mw = torch.rand(5, 39, generator=g)
    out2 = pad_obs(mw.cuda(), 204)
    torch.cuda.synchronize()
    assert torch.equal(out2.cpu(), torch.cat([mw, torch.zeros(5, 165)], 1))


def test_encoder_step_long_chunks_xlstm(hip_lib):
    """6 / 9 / 12 tokens per state pass (prefill chunks) through the large-T kernels, incl. the 206M head geometry."""
    from lram_amd.config import ModelSpec
    from lram_amd.engine import Engine
    for spec in (preset("xlstm_tiny"), ModelSpec(backbone="xlstm", d_model=1280, n_blocks=2, slstm_at=[1])):
        sd = init_state_dict(spec, seed=23)
        B = 3
        eng = Engine(spec, sd, B, device="cuda:0")
        state = None
        g = torch.Generator().manual_seed(6)
        for T in (12, 3, 9, 6):
            x = torch.randn(B, T, spec.d_model, generator=g)
            ref, state = xlstm_ref.encoder_forward_cached(spec, sd, x, state)
            out = eng.encoder_step(x.cuda())
            torch.cuda.synchronize()
            assert rel_err(out, ref) < 2e-4, (spec.d_model, T)
        pkv = eng.export_past_key_values()
        assert rel_err(pkv["block_0"]["mlstm_state"][0], state["block_0"]["mlstm_state"][0]) < 2e-4
        assert rel_err(pkv["block_1"]["slstm_state"], state["block_1"]["slstm_state"]) < 2e-4
        eng.close()


def test_mamba_two_slice_pipeline_equals_single_stream(hip_lib):
    """Mamba with two env slices (memory-bound kernels of one slice under the projections of the other, slice 1 one
    stage behind slice 0) == the single-stream schedule: same actions, same states, step and prefill, over repeats."""
    from lram_amd.engine import Engine
    spec = preset("mamba_tiny")
    sd = init_state_dict(spec, seed=21)
    B, steps = 37, 6
    seq = make_inputs(spec, B, steps, seed=11)
    dseq = [[t.cuda() for t in x] for x in seq]
    outs = []
    for micro in (1, 2, 2):
        eng = Engine(spec, sd, B, device="cuda:0")
        eng.set_micro_batches(micro)
        acts = []
        for x in dseq:
            a, _ = eng.step(*x)
            acts.append(a.clone())
        obs_seq = torch.stack([x[0] for x in dseq], 1).contiguous()
        rtg_seq = torch.stack([x[1] for x in dseq], 1).contiguous()
        rew_seq = torch.stack([x[2] for x in dseq], 1).contiguous()
        a_pre, _ = eng.prefill(obs_seq, rtg_seq, rew_seq)
        torch.cuda.synchronize()
        outs.append((torch.stack(acts), a_pre.clone(), eng.export_state_tensor(0, 0), eng.export_state_tensor(1, 3)))
        eng.close()
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])
        assert rel_err(o[2], outs[0][2]) < 1e-5 and rel_err(o[3], outs[0][3]) < 1e-5
    ora = dt_ref.OraclePolicy(spec, sd)
    for t, (obs, rtg, rew, mask) in enumerate(seq):
        ref = ora.step(obs, rtg, rew, mask)
        assert float((outs[1][0][t].cpu() - ref).abs().max()) <= 1e-4, t


def test_agent_crosses_a_process_boundary(hip_lib):
    """make_pickleable / pickle / reinit_cuda_kernels (custom_eval_callback.py:22-33, decision_xlstm.py:243-267): the
    rebuilt agent starts from an empty cache and reproduces the original agent's first action."""
    import pickle
    from lram_amd.agent import RecurrentAgent
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=5)
    agent = RecurrentAgent(spec, sd, n_envs=3, device="cuda:0", target_return=90.0, reward_scale=10.0)
    obs = torch.rand(3, 17, generator=torch.Generator().manual_seed(1)).cuda()
    rtg = torch.full((3,), 9.0).cuda()
    first = agent.predict_batch(obs, rtg).clone()
    agent.predict_batch(obs, rtg)
    agent.make_pickleable()
    assert agent.engine is None
    clone = pickle.loads(pickle.dumps(agent))
    clone.policy.reinit_cuda_kernels()
    assert clone.policy is clone and clone.engine is not None
    again = clone.predict_batch(obs, rtg)
    torch.cuda.synchronize()
    assert torch.equal(first, again)
    clone.inference_params.reset()
    clone.engine.close()
