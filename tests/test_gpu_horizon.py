"""The step path at the reference's real episode length (tests/golden/make_horizon_fixture.py).

The reference loop runs an episode to `done` -- 1000 steps on DMControl, 200 on Meta-World -- without clearing the cache
(src/callbacks/evaluation.py:130-177, src/algos/decision_transformer_sb3.py:663-666): the recurrent state integrates 3000
(600) tokens, the mLSTM stabiliser m drifts, the lazy matrix memory folds ~77 times per env.  The engine is driven by
lram_step over the fixture's inputs; actions / logits / hidden at the marked steps and the recurrent state at the end of the
episode and of the run are compared with the CPU oracle's (fp32 fixture; float64 fixture for the conditioning rule of
tests/helpers.py::assert_close_or_as_close_as_fp32_oracle, escape hatch capped at 5 % of the compared rows).

Three engine configurations on the xLSTM trajectory: lazy at 8 slots, materialised at 8 slots, and the headline
configuration -- 4096 slots, two slices, the lean front end / fused group norm / pre-split projections, fold period 13 --
with the fixture's 8 envs planted among 4088 slots of random traffic (every fold phase of the period is hit)."""
import json
import os

import numpy as np
import pytest
import torch

from lram_amd import init_state_dict, preset
from tests.golden.make_horizon_fixture import (CASES, SCHEMES, SSM_ENVS, WEIGHT_SEED, case_envs, fixture_name, horizon_inputs,
                                               probe, weight_checksum)
from tests.helpers import (assert_actions_match, assert_close_or_as_close_as_fp32_oracle, rel_err, relaxed_rows_fraction,
                           relaxed_rows_reset)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REPORT = {}


def _fixtures(case, scheme="exercise"):
    c = CASES[case]
    fx = np.load(os.path.join(GOLD, fixture_name(case, scheme)))
    fx64 = np.load(os.path.join(GOLD, fixture_name(case, scheme, fp64=True)))
    assert abs(float(fx64["weight_checksum"]) - float(fx["weight_checksum"])) <= 1e-9 * float(fx["weight_checksum"])
    return c, fx, fx64


def _closer(fx, fx64):
    def close(got, key, what, tol=2e-4):
        want32, want64 = torch.from_numpy(fx[key]), torch.from_numpy(fx64[key])
        got = got.detach().cpu().reshape(want32.shape)
        if got.dim() == 1:
            got, want32, want64 = got.unsqueeze(0), want32.unsqueeze(0), want64.unsqueeze(0)
        assert_close_or_as_close_as_fp32_oracle(got, want32, want64, tol=tol, what=what)
    return close


def _write_report():
    out = os.path.join(os.path.dirname(GOLD), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "horizon_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    if os.environ.get("LRAM_TEST_REPORT"):
        print("[report] horizon:", json.dumps(REPORT, sort_keys=True))


def _run_xlstm(mode, slots, where, case="xlstm", scheme="exercise"):
    """Drive lram_step over the fixture trajectory of `case` on the weight distribution `scheme`.  `where`: slot index of each
    fixture env."""
    from lram_amd.engine import Engine
    c, fx, fx64 = _fixtures(case, scheme)
    FB = case_envs(case)
    close = _closer(fx, fx64)
    spec = preset(c["preset"])
    sd = init_state_dict(spec, seed=WEIGHT_SEED, scheme=scheme)
    assert abs(weight_checksum(sd) - float(fx["weight_checksum"])) <= 1e-9 * float(fx["weight_checksum"]), \
        "seeded weights differ from the ones the fixture was computed with"
    obs, rtg, mask = horizon_inputs(spec, case, scheme)
    n_steps = obs.shape[0]
    eng = Engine(spec, sd, slots, device="cuda:0")
    if mode is not None:
        eng.set_state_mode(mode)
    lazy = eng.state_mode == "lazy"
    assert lazy == (mode != "eager")
    where = torch.as_tensor(where, device="cuda")
    d_obs_all, d_rtg_all, d_mask_all = obs.cuda(), rtg.cuda(), mask.cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    d_obs = torch.zeros(slots, spec.state_dim, device="cuda")
    d_rtg = torch.full((slots,), c["rtg0"], device="cuda")
    d_rew = torch.zeros(slots, device="cuda")
    d_mask = torch.zeros(slots, dtype=torch.uint8, device="cuda")
    relaxed_rows_reset()
    ties = 0
    g_lo, m_lo, m_hi, pend_hi = float("inf"), float("inf"), float("-inf"), 0
    name = f"{c['preset']}_{mode or 'auto'}_{slots}" + ("" if scheme == "exercise" else "_" + scheme)
    n_hi = 0.0
    for t in range(n_steps):
        if slots > FB:   # background traffic: random observations, restarts about every 300 steps, own rtg schedule
            d_obs[:, :c["native"]] = torch.rand(slots, c["native"], generator=g, device="cuda") * 2 - 1
            if scheme == "trained_like":   # the background traffic carries the outlier channel too
                d_obs[:, 3] *= 30.0
            d_mask.copy_((torch.rand(slots, generator=g, device="cuda") < (1.0 if t == 0 else 0.0033)).to(torch.uint8))
            d_rtg.copy_(torch.where(d_mask.bool(), torch.full_like(d_rtg, c["rtg0"]), d_rtg - c["drtg"]))
        d_obs[where] = d_obs_all[t]
        d_rtg[where] = d_rtg_all[t]
        d_mask[where] = d_mask_all[t]
        a, _ = eng.step(d_obs, d_rtg, d_rew, d_mask)
        if lazy and (t % 7 == 0 or t + 1 in c["marks"]):   # looked at without folding (lram_lazy_peek)
            for blk in c["blocks"]:
                gg, mm = eng.lazy_peek(blk, "g")[where], eng.lazy_peek(blk, "m")[where]
                g_lo, m_lo, m_hi = min(g_lo, float(gg.min())), min(m_lo, float(mm.min())), max(m_hi, float(mm.max()))
            pend_hi = max(pend_hi, int(eng.lazy_peek(c["blocks"][0], "pending")[where].max()))
        if t + 1 in c["marks"]:
            torch.cuda.synchronize()
            _, hidden, logits = eng.taps()
            want_logits = torch.from_numpy(fx[f"logits_{t + 1}"])
            try:
                close(hidden[where], f"hidden_{t + 1}", f"{name} step {t + 1} hidden")
                close(logits[where], f"logits_{t + 1}", f"{name} step {t + 1} logits")
                ties += assert_actions_match(a[where], torch.from_numpy(fx[f"actions_{t + 1}"]), want_logits, spec,
                                             what=f"{name} step {t + 1}")
            except AssertionError as ex:
                raise AssertionError(f"first diverging mark: step {t + 1}: {ex}") from None
        if t + 1 == c["episode"] or t + 1 == n_steps:
            # (an export folds every pending window: done only where the fixture holds the state, i.e. twice per run)
            tag = "ep" if t + 1 == c["episode"] else "end"
            for i in c["blocks"]:
                cm = eng.export_state_tensor(i, 0)[where]
                r = probe(cm.shape[-1]).cuda()
                close(cm @ r, f"{tag}_b{i}_Cr", f"{name} {tag} block {i} C r")
                close(r @ cm, f"{tag}_b{i}_rC", f"{name} {tag} block {i} r C")
                close(cm.abs().amax(dim=(-1, -2)), f"{tag}_b{i}_Cabsmax", f"{name} {tag} block {i} max |C|")
                nn = eng.export_state_tensor(i, 1)[where].squeeze(-1)
                n_hi = max(n_hi, float(nn.abs().max()))
                close(nn, f"{tag}_b{i}_n", f"{name} {tag} block {i} n")
                # (the stabiliser: 1e-4 on the well-conditioned distributions; the long-memory one -- gate pre-activations of +-10 ...
                # +-16 on an observation with a x 30 outlier channel -- gets the bar every other tensor has, 2e-4 or the float64 rule)
                if scheme == "trained_like":
                    close(eng.export_state_tensor(i, 2)[where].reshape(fx[f"{tag}_b{i}_m"].shape), f"{tag}_b{i}_m", f"{name} {tag} block {i} m")
                else:
                    assert rel_err(eng.export_state_tensor(i, 2)[where], fx[f"{tag}_b{i}_m"]) < 1e-4, (tag, i)
                close(eng.export_state_tensor(i, 3)[where], f"{tag}_b{i}_conv", f"{name} {tag} block {i} conv")
            close(eng.export_state_tensor(c["slstm"], 0)[:, where], f"{tag}_b{c['slstm']}_slstm", f"{name} {tag} sLSTM state")
    frac = relaxed_rows_fraction()
    REPORT[name] = {"steps": n_steps, "action_ties_below_2e-4": ties, "rows_on_the_float64_rule": round(frac, 4),
                    "oracle_m_range": {f"b{i}": [float(x) for x in fx[f"m_range_b{i}"]] for i in c["blocks"]}}
    if f"n_absmax_b{c['blocks'][0]}" in fx.files:
        REPORT[name]["oracle_n_absmax"] = {f"b{i}": float(fx[f"n_absmax_b{i}"]) for i in c["blocks"]}
    if lazy:
        REPORT[name].update({"engine_m_min": m_lo, "engine_m_max": m_hi, "g_min_between_folds": g_lo,
                             "max_pending_window_tokens": pend_hi, "engine_n_absmax_at_the_state_marks": n_hi})
        assert g_lo > 0.0, "the scale of C_base underflowed between folds"
    _write_report()
    assert frac <= 0.05, frac
    eng.close()
    torch.cuda.empty_cache()


# The weight distributions (lram_amd/weights.py::init_state_dict): "exercise" = every term busy, forgets within tens of steps;
# "reference" = a freshly built reference model (src/algos/models/decision_xlstm.py:170-171,210-213); "trained_like" = the
# long-memory corner of a loaded checkpoint (src/algos/decision_transformer_sb3.py:1120-1184): f ~ 0.95-0.998, input-gate
# pre-activations of +-15 (m far outside [-8, 8]), one observation channel 30 x the others.
@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("mode", ["lazy", "eager"])
def test_xlstm_16m_1000_step_episode_vs_oracle_fixture(hip_lib, mode, scheme):
    _run_xlstm(mode, case_envs("xlstm"), list(range(case_envs("xlstm"))), scheme=scheme)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_xlstm_16m_1000_step_episode_inside_the_headline_batch(hip_lib, scheme):
    """4096 slots, default modes (lazy, two slices, multi-env front end, fused group norm, pre-split projections): the
    fixture's envs sit at both ends of both slices and in the middle; their fold phases (slot % 13) differ."""
    _run_xlstm(None, 4096, [0, 1, 2047, 2048, 2049, 3000, 4094, 4095], scheme=scheme)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("slots,where", [(4, [0, 1, 2, 3]), (512, [0, 255, 256, 511])])
def test_xlstm_206m_200_step_episode_vs_oracle_fixture(hip_lib, slots, where, scheme):
    """The 206M geometry (20 blocks, head dim 640: per-env front end, score kernel, several column slices per head in the read
    pass, sLSTM head dim 320) over a Meta-World-length episode: lazy at 4 slots, and the C4 per-GPU batch -- 512 slots, two
    slices -- with the fixture's envs at both ends of both slices."""
    _run_xlstm("lazy" if slots == 4 else None, slots, where, case="xlstm206m", scheme=scheme)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_mamba_48m_200_step_episode_vs_oracle_fixture(hip_lib, scheme):
    from lram_amd.engine import Engine
    c, fx, fx64 = _fixtures("mamba", scheme)
    close = _closer(fx, fx64)
    spec = preset(c["preset"])
    sd = init_state_dict(spec, seed=WEIGHT_SEED, scheme=scheme)
    assert abs(weight_checksum(sd) - float(fx["weight_checksum"])) <= 1e-9 * float(fx["weight_checksum"])
    obs, rtg, mask = horizon_inputs(spec, "mamba", scheme)
    FB = case_envs("mamba")
    for slots, where in ((FB, list(range(FB))), (2048, [0, 1, 1023, 1024, 1025, 1500, 2046, 2047])):
        eng = Engine(spec, sd, slots, device="cuda:0")
        idx = torch.as_tensor(where, device="cuda")
        g = torch.Generator(device="cuda").manual_seed(12)
        d_obs = torch.zeros(slots, spec.state_dim, device="cuda")
        d_rtg = torch.full((slots,), 6.5, device="cuda")
        d_rew = torch.zeros(slots, device="cuda")
        d_mask = torch.zeros(slots, dtype=torch.uint8, device="cuda")
        relaxed_rows_reset()
        ties = 0
        name = f"mamba48m_{slots}" + ("" if scheme == "exercise" else "_" + scheme)
        for t in range(obs.shape[0]):
            if slots > FB:
                d_obs[:, :39] = torch.rand(slots, 39, generator=g, device="cuda") * 2 - 1
                if scheme == "trained_like":
                    d_obs[:, 3] *= 30.0
                d_mask.copy_((torch.rand(slots, generator=g, device="cuda") < (1.0 if t == 0 else 0.01)).to(torch.uint8))
                d_rtg.copy_(torch.where(d_mask.bool(), torch.full_like(d_rtg, 6.5), d_rtg - 0.02))
            d_obs[idx], d_rtg[idx], d_mask[idx] = obs[t].cuda(), rtg[t].cuda(), mask[t].cuda()
            a, _ = eng.step(d_obs, d_rtg, d_rew, d_mask)
            if t + 1 in c["marks"]:
                torch.cuda.synchronize()
                _, hidden, logits = eng.taps()
                try:
                    close(hidden[idx], f"hidden_{t + 1}", f"{name} step {t + 1} hidden")
                    close(logits[idx], f"logits_{t + 1}", f"{name} step {t + 1} logits")
                    ties += assert_actions_match(a[idx], torch.from_numpy(fx[f"actions_{t + 1}"]),
                                                 torch.from_numpy(fx[f"logits_{t + 1}"]), spec, what=f"{name} step {t + 1}")
                except AssertionError as ex:
                    raise AssertionError(f"first diverging mark: step {t + 1}: {ex}") from None
            if t + 1 == c["episode"] or t + 1 == obs.shape[0]:
                tag = "ep" if t + 1 == c["episode"] else "end"
                for i in c["blocks"]:
                    close(eng.export_state_tensor(i, 0)[idx[list(SSM_ENVS)]], f"{tag}_l{i}_ssm", f"{name} {tag} layer {i} ssm")
                    close(eng.export_state_tensor(i, 3)[idx], f"{tag}_l{i}_conv", f"{name} {tag} layer {i} conv")
        REPORT[name] = {"steps": int(obs.shape[0]), "action_ties_below_2e-4": ties,
                        "rows_on_the_float64_rule": round(relaxed_rows_fraction(), 4)}
        _write_report()
        assert relaxed_rows_fraction() <= 0.05
        eng.close()
        torch.cuda.empty_cache()
