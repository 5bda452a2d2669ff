#!/usr/bin/env python3
"""Generate tests/golden/horizon_{xlstm16m,mamba48m}[_fp64].npz: the step path at the reference's real episode length.

    python tests/golden/make_horizon_fixture.py [--fp64] [--model xlstm|mamba|xlstm206m] [--scheme exercise|reference|trained_like]
                                                                                    (about 5 + 10 minutes on 8 cores)

The reference loop runs an episode to `done` (src/callbacks/evaluation.py:130-177: DMControl episodes are 1000 steps,
Meta-World 200) and, unless `reset_inf_cache_freq` fires, never clears the cache in between
(src/algos/decision_transformer_sb3.py:663-666): the recurrent state integrates 3000 (600) tokens.  The oracle follows
that order -- one `layers.step` per token (decision_xlstm.py:155-166 / decision_mamba.py:130-147) through
OraclePolicy.step -- for

  xlstm  xLSTM[7:1] 16M, 8 envs, DMControl-shaped inputs (17 native dims, rtg falling by r/scale = 0.01 per step as
         evaluation.py:165 does), 1000 env-steps without a reset, a reset of envs 0..3 at the 1001st, 60 more;
  mamba  Mamba 48M, 8 envs, Meta-World-shaped inputs (39 native dims), 200 env-steps, reset of envs 0..3, 20 more;
  xlstm206m  xLSTM[7:1] 206M (20 blocks, head dim 640), 4 envs, 200 env-steps, reset of envs 0..1, 20 more,

far too slow to repeat inside the GPU test run, so the outputs are committed as fixtures: inputs are regenerated from
the seed by `horizon_inputs`, weights by init_state_dict(seed) (checksum stored and asserted); expected values =
actions / logits / hidden at the listed steps, the recurrent state at the end of the long episode and at the end of
the run (n, m, conv in full; the matrix memory C through its products with a fixed probe vector), and the range the
stabiliser m covered.  `--fp64` evaluates the same trajectory in float64 (tests/helpers.py::Fp64Oracle): the reference
point of the conditioning-aware comparison.  tests/test_gpu_horizon.py drives lram_step over the same inputs."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

B, WEIGHT_SEED, INPUT_SEED = 8, 0, 90125
SSM_ENVS = (0, 3, 4, 7)   # Mamba: envs whose ssm state is stored (two that restart at the episode end, two that do not)
CASES = {
    # preset, native obs dims, episode length, steps after the reset, rtg start, rtg decrement, stored steps (1-based)
    "xlstm": dict(preset="xlstm_16m", native=17, episode=1000, tail=60, rtg0=4.51274, drtg=0.01,
                  marks=(1, 10, 100, 250, 500, 750, 1000, 1001, 1060), blocks=(0, 7), slstm=1, file="horizon_xlstm16m"),
    "mamba": dict(preset="mamba_48m", native=39, episode=200, tail=20, rtg0=6.50346, drtg=0.02,
                  marks=(1, 10, 50, 100, 150, 200, 201, 220), blocks=(0, 11), slstm=None, file="horizon_mamba48m"),
    # the 206M geometry (head dim 640: score kernel + several column slices per head + per-env front end, none of which the
    # 16M case touches) over a Meta-World-length episode; 4 envs (the oracle takes ~1 s per step at this size)
    "xlstm206m": dict(preset="xlstm_206m", native=39, episode=200, tail=20, rtg0=6.50346, drtg=0.02, envs=4,
                      marks=(1, 10, 50, 100, 150, 200, 201, 220), blocks=(0, 19), slstm=3, file="horizon_xlstm206m"),
}


SCHEMES = ("exercise", "reference", "trained_like")
OUTLIER_CHANNEL, OUTLIER_GAIN = 3, 30.0   # scheme "trained_like": one native observation channel carries 30 x the range


def case_envs(case):
    return CASES[case].get("envs", B)


def fixture_name(case, scheme="exercise", fp64=False):
    return CASES[case]["file"] + ("" if scheme == "exercise" else "_" + scheme) + ("_fp64" if fp64 else "") + ".npz"


def horizon_inputs(spec, case, scheme="exercise"):
    """obs [steps, B, state_dim] (native dims U(-1,1), rest zero as after pad_inputs), rtg [steps, B], mask [steps, B]."""
    c = CASES[case]
    B = case_envs(case)
    n = c["episode"] + c["tail"]
    g = torch.Generator().manual_seed(INPUT_SEED)
    obs = torch.zeros(n, B, spec.state_dim)
    obs[:, :, : c["native"]] = torch.rand(n, B, c["native"], generator=g) * 2 - 1
    if scheme == "trained_like":
        obs[:, :, OUTLIER_CHANNEL] *= OUTLIER_GAIN
    mask = torch.zeros(n, B, dtype=torch.uint8)
    mask[0] = 1
    mask[c["episode"], : B // 2] = 1
    rtg = torch.empty(n, B)
    cur = torch.full((B,), c["rtg0"])
    for t in range(n):
        cur = torch.where(mask[t].bool(), torch.full_like(cur, c["rtg0"]), cur - c["drtg"]) if t else cur
        rtg[t] = cur
    return obs, rtg, mask


def probe(dh):
    return torch.linspace(-1.0, 1.0, dh).cos()


def weight_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def store_state(out, tag, ora, c, fp64):
    for i in c["blocks"]:
        if c["preset"].startswith("xlstm"):
            cm, n, m = ora.state[f"block_{i}"]["mlstm_state"]
            r = probe(cm.shape[-1]).to(cm.dtype)
            out[f"{tag}_b{i}_n"], out[f"{tag}_b{i}_m"] = n.numpy().copy(), m.numpy().copy()
            out[f"{tag}_b{i}_conv"] = ora.state[f"block_{i}"]["conv_state"][0].numpy().copy()
            out[f"{tag}_b{i}_Cr"] = (cm @ r).numpy()
            out[f"{tag}_b{i}_rC"] = (r @ cm).numpy()
            out[f"{tag}_b{i}_Cabsmax"] = cm.abs().amax(dim=(-1, -2)).numpy()
        else:
            conv, ssm = ora.state[i]
            out[f"{tag}_l{i}_conv"] = conv.numpy().copy()
            out[f"{tag}_l{i}_ssm"] = ssm[list(SSM_ENVS)].numpy().copy()   # (0.8 MB per layer and tag for all eight)
    if c["slstm"] is not None:
        out[f"{tag}_b{c['slstm']}_slstm"] = ora.state[f"block_{c['slstm']}"]["slstm_state"].numpy().copy()


def main(case, fp64=False, scheme="exercise"):
    from lram_amd import init_state_dict, preset
    from oracle.dt_ref import OraclePolicy
    torch.set_num_threads(int(os.environ.get("HORIZON_THREADS", os.cpu_count() or 1)))
    c = CASES[case]
    spec = preset(c["preset"])
    sd = init_state_dict(spec, seed=WEIGHT_SEED, scheme=scheme)
    obs, rtg, mask = horizon_inputs(spec, case, scheme)
    if fp64:
        from tests.helpers import Fp64Oracle
        f64 = Fp64Oracle(spec, sd)
        step, ora = f64.step, f64.ora
    else:
        ora = OraclePolicy(spec, sd)
        step = ora.step
    zero = torch.zeros(case_envs(case))
    out = {"weight_checksum": np.float64(weight_checksum(sd))}
    m_lo, m_hi = {i: float("inf") for i in c["blocks"]}, {i: float("-inf") for i in c["blocks"]}
    n_hi = {i: 0.0 for i in c["blocks"]}
    t0 = time.time()
    for t in range(c["episode"] + c["tail"]):
        act, dbg = step(obs[t], rtg[t], zero, mask[t] if mask[t].any() else None, return_debug=True)
        if t + 1 in c["marks"]:
            out[f"actions_{t + 1}"] = act.numpy()
            out[f"logits_{t + 1}"] = dbg["logits"].numpy()
            out[f"hidden_{t + 1}"] = dbg["hidden"].numpy()
        if c["preset"].startswith("xlstm"):
            for i in c["blocks"]:
                m = ora.state[f"block_{i}"]["mlstm_state"][2]
                m_lo[i], m_hi[i] = min(m_lo[i], float(m.min())), max(m_hi[i], float(m.max()))
                n_hi[i] = max(n_hi[i], float(ora.state[f"block_{i}"]["mlstm_state"][1].abs().max()))
        if t + 1 == c["episode"]:
            store_state(out, "ep", ora, c, fp64)
        if t % 20 == 0:
            print(f"{case} step {t} ({time.time() - t0:.0f} s)", flush=True)
    store_state(out, "end", ora, c, fp64)
    if c["preset"].startswith("xlstm"):
        for i in c["blocks"]:
            out[f"m_range_b{i}"] = np.array([m_lo[i], m_hi[i]])
            out[f"n_absmax_b{i}"] = np.float64(n_hi[i])
    if fp64:   # float64 RESULTS stored as float32 (6e-8 relative, against comparison bars of 2e-4): halves the fixture
        out = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 and k != "weight_checksum" else v)
               for k, v in out.items()}
    name = fixture_name(case, scheme, fp64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    which = [sys.argv[sys.argv.index("--model") + 1]] if "--model" in sys.argv else list(CASES)
    scheme = sys.argv[sys.argv.index("--scheme") + 1] if "--scheme" in sys.argv else "exercise"
    assert scheme in SCHEMES, scheme
    for w in which:
        main(w, fp64="--fp64" in sys.argv, scheme=scheme)
