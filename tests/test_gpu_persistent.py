"""GPU: the small-batch paths (lram_set_persistent_mode; SURVEY 8 row g1, the reference's own operating point of one env,
src/callbacks/evaluation.py:80) -- fused phase kernels and the whole-step cooperative kernel, both opt-in because both
measure slower than the generic launch-per-kernel path -- against the CPU oracle and against that path, B in {1, 2, 8}."""
import dataclasses

import pytest
import torch

from lram_amd import init_state_dict, preset
from oracle.dt_ref import OraclePolicy
from tests.helpers import assert_actions_match, make_inputs, rel_err

pytestmark = pytest.mark.gpu


MODES = ("fused", "whole_step")


def _engine(spec, sd, B, mode):
    from lram_amd.engine import Engine
    eng = Engine(spec, sd, B, device="cuda:0")
    eng.set_persistent_mode(mode)
    assert eng.persistent_mode == mode
    return eng


def _check_state(eng, ora, spec, tol=2e-4):
    pkv = eng.export_past_key_values()
    for i in range(spec.n_blocks):
        blk, ref = pkv[f"block_{i}"], ora.state[f"block_{i}"]
        assert rel_err(blk["conv_state"][0], ref["conv_state"][0]) < tol, i
        if "mlstm_state" in blk:
            for j in range(3):
                assert rel_err(blk["mlstm_state"][j], ref["mlstm_state"][j]) < tol, (i, j)
        else:
            assert rel_err(blk["slstm_state"], ref["slstm_state"]) < tol, i


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name,B,steps", [("xlstm_16m", 1, 8), ("xlstm_16m", 2, 6), ("xlstm_16m", 8, 6), ("xlstm_tiny", 5, 10),
                                          ("xlstm_c1", 3, 8)])
def test_persistent_step_matches_oracle(hip_lib, name, B, steps, mode):
    spec = preset(name)
    sd = init_state_dict(spec, seed=5)
    eng = _engine(spec, sd, B, mode)
    ora = OraclePolicy(spec, sd)
    ties = 0
    for t, (obs, rtg, rew, mask) in enumerate(make_inputs(spec, B, steps, seed=321, reset_prob=0.25)):
        a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        tok, hid, logits = eng.taps()
        assert rel_err(tok, dbg["tokens"]) < 1e-5, t
        assert rel_err(hid, dbg["hidden"]) < 2e-4, (t, rel_err(hid, dbg["hidden"]))
        ties += assert_actions_match(a, ref, dbg["logits"], spec, what=f"{name} B={B} step {t}")
    assert ties == 0
    _check_state(eng, ora, spec)
    eng.close()


def test_persistent_step_discrete_head_and_embedding_input(hip_lib):
    """Image domain: frames -> lram_embed_images -> whole-step kernel with obs_is_embedding, argmax over 18."""
    from lram_amd.agent import RecurrentAgent
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=8, with_image_encoder=True)
    B = 2
    agent = RecurrentAgent(spec, sd, n_envs=B, device="cuda:0", discrete=True)
    agent.engine.set_persistent_mode("fused")
    assert agent.engine.persistent_mode == "fused"
    ora = OraclePolicy(spec, sd)
    ties = 0
    for t, (obs, rtg, rew, mask) in enumerate(make_inputs(spec, B, 6, seed=12, image=True)):
        a = agent.predict_batch(obs.cuda(), rtg.cuda(), None, mask.cuda(), env_act_dim=1)
        ref, dbg = ora.step(obs, rtg, rew, mask, discrete=True, return_debug=True)
        ties += assert_actions_match(a, ref, dbg["logits"], spec, discrete=True, what=f"step {t}")
    assert ties == 0
    agent.engine.close()


def test_persistent_equals_launch_path_and_modes_share_state(hip_lib):
    """Same inputs through both paths: tokens identical (up to numerical ties), states close; and a trajectory that
    switches the mode in the middle stays on the oracle's trajectory (the state buffers are the same ones)."""
    spec = preset("xlstm_16m")
    sd = init_state_dict(spec, seed=2)
    B = 2
    seq = make_inputs(spec, B, 8, seed=77, reset_prob=0.2)
    engs = {m: _engine(spec, sd, B, m) for m in ("fused", "whole_step", "generic")}
    mixed = _engine(spec, sd, B, "fused")
    ora = OraclePolicy(spec, sd)
    for t, (obs, rtg, rew, mask) in enumerate(seq):
        d = [x.cuda() for x in (obs, rtg, rew, mask)]
        out = {m: e.step(*d)[1].clone() for m, e in engs.items()}
        if t == 3:
            mixed.set_persistent_mode("whole_step")
        if t == 5:
            mixed.set_persistent_mode("generic")
            assert mixed.persistent_mode == "generic"
        a, _ = mixed.step(*d)
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        assert int((out["fused"] != out["generic"]).sum()) <= 1
        assert torch.equal(out["fused"], out["whole_step"])       # the same phase code, barriers instead of launches
        assert assert_actions_match(a, ref, dbg["logits"], spec, what=f"mixed step {t}") == 0
    for i in range(spec.n_blocks):
        for which in ((0, 3) if i in spec.slstm_at else (0, 1, 2, 3)):
            assert rel_err(engs["fused"].export_state_tensor(i, which), engs["generic"].export_state_tensor(i, which)) < 5e-5
    _check_state(mixed, ora, spec)
    for e in list(engs.values()) + [mixed]:
        e.close()


def test_persistent_206m_two_envs(hip_lib):
    spec = dataclasses.replace(preset("xlstm_206m"), n_blocks=6, slstm_at=[1, 3])   # full width, 6 of the 20 blocks
    sd = init_state_dict(spec, seed=4)
    eng = _engine(spec, sd, 2, "fused")
    ora = OraclePolicy(spec, sd)
    for t, (obs, rtg, rew, mask) in enumerate(make_inputs(spec, 2, 4, seed=9)):
        a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        torch.cuda.synchronize()
        assert rel_err(eng.taps()[1], dbg["hidden"]) < 2e-4
        assert assert_actions_match(a, ref, dbg["logits"], spec, what=f"206m step {t}") == 0
    eng.close()


def test_persistent_mode_rules_and_long_run(hip_lib):
    """auto = the generic path; the small-batch modes apply to <= 8 env slots of an xLSTM stack outside graph mode; 300
    consecutive whole-step launches keep the barrier counter consistent (no timeout, deterministic)."""
    from lram_amd.engine import Engine, LramError
    spec = preset("xlstm_16m")
    sd = init_state_dict(spec, seed=0)
    eng = Engine(spec, sd, 1, device="cuda:0")
    assert eng.persistent_mode == "generic"
    eng.set_persistent_mode("fused")
    assert eng.persistent_mode == "fused"
    eng.set_graph_mode(True)
    assert eng.persistent_mode == "generic"
    eng.set_graph_mode(False)
    eng.set_persistent_mode("whole_step")
    g = torch.Generator(device="cuda:0").manual_seed(3)
    obs = torch.rand(300, 1, spec.state_dim, generator=g, device="cuda:0") * 2 - 1
    rtg, rew = torch.full((1,), 4.0, device="cuda:0"), torch.zeros(1, device="cuda:0")
    outs = []
    for rep in range(2):
        eng.reset()
        toks = []
        for t in range(300):
            toks.append(eng.step(obs[t], rtg, rew, None)[1].clone())
        torch.cuda.synchronize()
        outs.append(torch.stack(toks))
    assert torch.equal(outs[0], outs[1])
    assert eng.persistent_mode == "whole_step"
    eng.close()
    big = Engine(spec, sd, 9, device="cuda:0")
    assert big.persistent_mode == "generic"
    for m in ("fused", "whole_step"):
        with pytest.raises(LramError):
            big.set_persistent_mode(m)
    big.close()
    ms = preset("mamba_tiny")
    m = Engine(ms, init_state_dict(ms, seed=0), 2, device="cuda:0")
    assert m.persistent_mode == "generic"
    m.close()
