#!/usr/bin/env python3
"""Generate known-answer trajectories (KATs) from the CPU oracle: tests/golden/kat_<name>.npz.

    python tests/golden/make_kat.py
Each KAT holds, for a seeded model (weights regenerated from the seed; a sha256 of the weight bytes guards
against generator drift) and a seeded 8-step trajectory with per-env resets: the inputs (obs, rtg, reward,
reset mask), the oracle's actions, argmax logits margins, encoder hidden states, and the final recurrent
state of block 0.  tests/test_oracle_kat.py re-runs the oracle against them (CPU); tests/test_gpu_golden.py
runs the HIP engine against them (GPU).
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from lram_amd import init_state_dict, preset  # noqa: E402
from oracle.dt_ref import OraclePolicy  # noqa: E402
from tests.helpers import make_inputs  # noqa: E402

KATS = {"xlstm_tiny": dict(B=4, steps=8, seed=21, discrete=False),
        "xlstm_tiny_discrete": dict(B=4, steps=8, seed=22, discrete=True, preset="xlstm_tiny"),
        "xlstm_c1": dict(B=4, steps=8, seed=23, discrete=False),
        "mamba_tiny": dict(B=4, steps=8, seed=24, discrete=False)}


def weights_digest(sd):
    """sha256 over names and bytes.  init_state_dict draws from PCG64 with exact integer->float arithmetic, so
    the digest is host-independent (transcendental inits are computed in float64 and rounded once)."""
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def weights_l1(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def run(name, B, steps, seed, discrete, preset_name=None):
    spec = preset(preset_name or name)
    sd = init_state_dict(spec, seed=seed)
    ora = OraclePolicy(spec, sd)
    rec = {k: [] for k in ("obs", "rtg", "rew", "mask", "actions", "hidden", "gap")}
    for obs, rtg, rew, mask in make_inputs(spec, B, steps, seed=seed + 1000):
        a, dbg = ora.step(obs, rtg, rew, mask, discrete=discrete, return_debug=True)
        lg = dbg["logits"][..., : spec.n_discrete] if discrete else dbg["logits"]
        top2 = lg.topk(2, dim=-1).values
        rec["obs"].append(obs.numpy()), rec["rtg"].append(rtg.numpy()), rec["rew"].append(rew.numpy())
        rec["mask"].append(mask.numpy()), rec["actions"].append(a.numpy().astype(np.float32))
        rec["hidden"].append(dbg["hidden"].numpy()), rec["gap"].append((top2[..., 0] - top2[..., 1]).numpy())
    out = {k: np.stack(v) for k, v in rec.items()}
    if spec.backbone == "mamba":
        out["state0_a"] = ora.state[0][0].numpy()
        out["state0_b"] = ora.state[0][1].numpy()
    else:
        c, n, m = ora.state["block_0"]["mlstm_state"]
        out["state0_a"] = c.numpy()[:, :, :8, :8].copy()  # corner of the matrix memory (keeps the file small)
        out["state0_b"] = n.numpy()
    out["weights_sha256"] = np.array(weights_digest(sd))
    out["weights_l1"] = np.array(weights_l1(sd))
    out["meta"] = np.array(f"preset={preset_name or name} B={B} steps={steps} seed={seed} discrete={discrete}")
    return out


if __name__ == "__main__":
    for name, kw in KATS.items():
        kw = dict(kw)
        pn = kw.pop("preset", None)
        out = run(name, preset_name=pn, **kw)
        path = os.path.join(HERE, f"kat_{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "min action gap", float(out["gap"].min()), os.path.getsize(path), "bytes")
