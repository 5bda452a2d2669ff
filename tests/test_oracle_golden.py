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
