"""CPU checks of the long-horizon fixtures (tests/golden/make_horizon_fixture.py): the committed files are what the
generator writes (its first marks are recomputed here by the oracle: same actions, logits to 2e-6), they hold every key the GPU test
(tests/test_gpu_horizon.py) reads, and the float64 companion sits where the fp32 oracle can be expected to be."""
import os

import numpy as np
import pytest
import torch

from lram_amd import init_state_dict, preset
from oracle.dt_ref import OraclePolicy
from tests.golden.make_horizon_fixture import (CASES, SCHEMES, SSM_ENVS, WEIGHT_SEED, case_envs, fixture_name, horizon_inputs,
                                               weight_checksum)
from tests.helpers import assert_actions_match

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("case", ["xlstm", "mamba", "xlstm206m"])
def test_horizon_fixture_is_what_the_oracle_computes(case, scheme):
    c = CASES[case]
    B = case_envs(case)
    fx = np.load(os.path.join(GOLD, fixture_name(case, scheme)))
    fx64 = np.load(os.path.join(GOLD, fixture_name(case, scheme, fp64=True)))
    spec = preset(c["preset"])
    sd = init_state_dict(spec, seed=WEIGHT_SEED, scheme=scheme)
    assert abs(weight_checksum(sd) - float(fx["weight_checksum"])) <= 1e-9 * float(fx["weight_checksum"])
    obs, rtg, mask = horizon_inputs(spec, case, scheme)
    assert obs.shape[0] == c["episode"] + c["tail"] and int(mask.sum()) == B + B // 2
    assert bool((rtg[c["episode"], : B // 2] == c["rtg0"]).all()) and bool((rtg[c["episode"], B // 2:] < rtg[0, 0]).all())
    ora = OraclePolicy(spec, sd)
    # (the generator ran on 4 threads; the thread count changes matmul summation order, hence the last bits: logits to 2e-6,
    # actions exact.  Not set here: it is process-global and other tests compare bit for bit with their own fixtures.)
    for t in range(10 if case != "xlstm206m" else 1):   # (206M: ~1 s per oracle step)
        act, dbg = ora.step(obs[t], rtg[t], torch.zeros(B), mask[t] if mask[t].any() else None, return_debug=True)
        if t + 1 in (1, 10):
            scale = max(1.0, float(np.abs(fx[f"logits_{t + 1}"]).max()))
            np.testing.assert_allclose(dbg["logits"].numpy(), fx[f"logits_{t + 1}"], rtol=0, atol=2e-6 * scale)
            # (through the tie rule: a host with another core count sums its matmuls in another order, and an action whose
            # top-2 logits sit within 2e-4 of each other may then legitimately flip)
            assert_actions_match(act, torch.from_numpy(fx[f"actions_{t + 1}"]), torch.from_numpy(fx[f"logits_{t + 1}"]), spec,
                                 what=f"{case} {scheme} step {t + 1}")
    for s in c["marks"]:
        for k in ("actions", "logits", "hidden"):
            assert f"{k}_{s}" in fx.files and f"{k}_{s}" in fx64.files
        # the two precisions tell the same story at every mark: fp32 within 5e-3 of float64 relative to the largest logit
        d = np.abs(fx[f"logits_{s}"].astype(np.float64) - fx64[f"logits_{s}"]).max() / np.abs(fx64[f"logits_{s}"]).max()
        assert d < 5e-3, (s, d)
    for tag in ("ep", "end"):
        for i in c["blocks"]:
            keys = [f"{tag}_b{i}_{x}" for x in ("n", "m", "conv", "Cr", "rC", "Cabsmax")] if case != "mamba" else \
                [f"{tag}_l{i}_conv", f"{tag}_l{i}_ssm"]
            for k in keys:
                assert k in fx.files and fx[k].shape == fx64[k].shape, k
    if case == "mamba":
        assert fx[f"ep_l{c['blocks'][0]}_ssm"].shape[0] == len(SSM_ENVS)
    else:
        lo, hi = fx[f"m_range_b{c['blocks'][0]}"]
        assert np.isfinite(lo) and np.isfinite(hi) and lo < hi
        if scheme == "trained_like":   # the regime the scheme exists for: the stabiliser leaves [-8, 8]
            assert max(float(np.abs(fx[f"m_range_b{i}"]).max()) for i in c["blocks"]) > 8.0
