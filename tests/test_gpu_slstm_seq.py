"""GPU: the one-launch sLSTM step kernel (csrc/slstm_seq.hip: the T tokens of an env-step back to back inside a workgroup of
16 envs x one head, cell state in registers, R_g h as f16x2 split products on the f16 matrix instruction -- form "1", the
default -- or on the exact fp32 matrix instruction with 32 envs per workgroup -- form "2", what LRAM_GEMM=f32 runs) against
the oracle and against the per-token GEMM + pointwise path it replaces for large slices ([3P] sLSTMLayer.step / slstm_pointwise, reference call site
src/algos/models/decision_xlstm.py:155-166)."""
import os

import pytest
import torch

from lram_amd import init_state_dict, preset
from oracle import dt_ref
from tests.helpers import assert_actions_match, make_inputs, rel_err

pytestmark = pytest.mark.gpu


def _engine(spec, sd, B, seq, micro=1):
    """seq: False / "0" = the per-token path, True / "1" = the step kernel's default (f16x2) form, "2" = its fp32 form."""
    from lram_amd.engine import Engine
    keys = ("LRAM_SLSTM_SEQ", "LRAM_SLSTM_FUSED_ROWS")
    old = {k: os.environ.get(k) for k in keys}
    os.environ["LRAM_SLSTM_SEQ"] = seq if isinstance(seq, str) else ("1" if seq else "0")
    os.environ["LRAM_SLSTM_FUSED_ROWS"] = "0"        # the <= 512-env token kernel off: the path under test serves every size
    try:
        eng = Engine(spec, sd, B, device="cuda:0")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    eng.set_micro_batches(micro)
    return eng


def _run(eng, seq):
    acts = []
    for obs, rtg, rew, mask in seq:
        a, _ = eng.step(obs.cuda(), rtg.cuda(), rew.cuda(), mask.cuda())
        acts.append(a.clone())
    torch.cuda.synchronize()
    return torch.stack(acts).cpu()


@pytest.mark.parametrize("B,micro,form", [(5, 1, "1"), (37, 1, "1"), (41, 2, "1"), (70, 2, "1"), (37, 1, "2"), (70, 2, "2")])
def test_slstm_step_kernel_matches_oracle_and_the_gemm_path(hip_lib, B, micro, form):
    """Ragged env counts (workgroups of 16 / 32 envs), random restarts (the per-element n == 0 first-step rule after a reset),
    12 steps: actions follow the oracle, the sLSTM state planes (h, c, n, m) equal the GEMM path's and the oracle's."""
    spec = preset("xlstm_16m")
    sd = init_state_dict(spec, seed=81)
    seq = make_inputs(spec, B, 12, seed=41, reset_prob=0.15)
    new, old = _engine(spec, sd, B, form, micro), _engine(spec, sd, B, False, micro)
    a_new, a_old = _run(new, seq), _run(old, seq)
    ora = dt_ref.OraclePolicy(spec, sd)
    ties = 0
    for t, (obs, rtg, rew, mask) in enumerate(seq):
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        ties += assert_actions_match(a_new[t], ref, dbg["logits"], spec, what=f"sLSTM step kernel (form {form}), step {t}")
    assert ties == 0
    assert float((a_new - a_old).abs().max()) <= 1e-4
    blk = spec.slstm_at[0]
    s_new, s_old = new.export_state_tensor(blk, 0), old.export_state_tensor(blk, 0)
    assert rel_err(s_new, s_old) < 2e-5, rel_err(s_new, s_old)
    assert rel_err(s_new, ora.state[f"block_{blk}"]["slstm_state"]) < 2e-4
    assert rel_err(new.export_state_tensor(blk, 3), old.export_state_tensor(blk, 3)) < 1e-6     # conv state: untouched by it
    new.close(), old.close()


@pytest.mark.parametrize("d_model,B,form", [(768, 37, "1"), (1024, 33, "1"), (1280, 40, "1"), (1536, 34, "1"),
                                            (1024, 33, "2"), (1280, 40, "2")])
def test_slstm_step_kernel_other_head_dims(hip_lib, d_model, B, form):
    """Round 5: the kernel is templated on the sLSTM head dim -- 192 (xlstm_mediumplus), 256 (xlstm_large), 320 (the 206M stack's
    sLSTM blocks, ten waves per workgroup), 384 (xlstm_hugeplus) beside the 16M model's 128: ragged env counts, restarts, 6 steps
    against the oracle and the per-token GEMM path, state planes included."""
    from lram_amd.config import ModelSpec
    spec = ModelSpec(backbone="xlstm", d_model=d_model, n_blocks=2, slstm_at=[1])
    sd = init_state_dict(spec, seed=83)
    seq = make_inputs(spec, B, 6, seed=43, reset_prob=0.15)
    new, old = _engine(spec, sd, B, form), _engine(spec, sd, B, False)
    a_new, a_old = _run(new, seq), _run(old, seq)
    ora = dt_ref.OraclePolicy(spec, sd)
    ties = 0
    for t, (obs, rtg, rew, mask) in enumerate(seq):
        ref, dbg = ora.step(obs, rtg, rew, mask, return_debug=True)
        ties += assert_actions_match(a_new[t], ref, dbg["logits"], spec, what=f"sLSTM step kernel (form {form}), head dim {d_model // 4}, step {t}")
    assert ties == 0
    assert float((a_new - a_old).abs().max()) <= 1e-4
    s_new, s_old = new.export_state_tensor(1, 0), old.export_state_tensor(1, 0)
    assert rel_err(s_new, s_old) < 2e-5, rel_err(s_new, s_old)
    assert rel_err(s_new, ora.state["block_1"]["slstm_state"]) < 2e-4
    new.close(), old.close()


def test_slstm_step_kernel_single_token_calls(hip_lib):
    """T = 1 encoder calls (lram_encoder_step, one token at a time) through the same kernel equal three-token calls."""
    spec = preset("xlstm_16m")
    sd = init_state_dict(spec, seed=82)
    B = 9
    x = torch.randn(B, 3, spec.d_model, generator=torch.Generator().manual_seed(5)).cuda()
    a, b = _engine(spec, sd, B, True), _engine(spec, sd, B, True)
    y3 = a.encoder_step(x.contiguous()).clone()
    y1 = torch.cat([b.encoder_step(x[:, t:t + 1].contiguous()).clone() for t in range(3)], dim=1)
    torch.cuda.synchronize()
    assert rel_err(y1, y3) < 2e-5
    blk = spec.slstm_at[0]
    assert rel_err(a.export_state_tensor(blk, 0), b.export_state_tensor(blk, 0)) < 2e-5
    a.close(), b.close()
