#!/usr/bin/env python3
"""Generate tests/golden/c5_prefill_206m.npz: BASELINE config 5 at its real depth, computed by the CPU oracle.

    python tests/golden/make_c5_fixture.py          (about 15 minutes on 8 cores: 520 x 3 token steps through 206M)

C5 = xLSTM[7:1] 206M (xlstm_huge.yaml + slstm_at=[1,3,5], 20 blocks), a stored context of 512 timesteps
(1536 tokens) followed by single-step decode, Mimicgen-shaped inputs (168 of 204 state dims, 7 of 8 action dims).
The oracle runs the reference's order -- one `layers.step` per token (decision_xlstm.py:155-166) through
OraclePolicy.step, 2 envs -- far too slow to repeat inside the GPU test run, so its outputs are committed as a
fixture: inputs are regenerated from the seed by `c5_inputs`, weights by init_state_dict(seed) (a checksum of them is
stored and asserted), expected values = actions / logits / hidden of the last context step and of 8 decode steps, and
the recurrent state after the context (n, m, conv in full for three blocks; the matrix memory C through its product
with a fixed probe vector).  tests/test_gpu_configs.py compares lram_prefill + lram_step (hipGraph decode) with it."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

B, L, N_DECODE, WEIGHT_SEED, INPUT_SEED = 2, 512, 8, 0, 4242
STATE_BLOCKS = (0, 9, 19)    # mLSTM blocks whose state is stored (first, middle, last)
SLSTM_BLOCK = 3


def c5_inputs(spec):
    """obs [B, L + N_DECODE, state_dim] (Mimicgen: 168 native dims, zero-padded), rtg [B, L + N_DECODE]."""
    g = torch.Generator().manual_seed(INPUT_SEED)
    n = L + N_DECODE
    obs = torch.zeros(B, n, spec.state_dim)
    obs[:, :, :168] = torch.rand(B, n, 168, generator=g) * 2 - 1
    rtg = 5.0 - 0.005 * torch.arange(n).float().view(1, -1).expand(B, n).contiguous()
    return obs, rtg


def probe(dh):
    return torch.linspace(-1.0, 1.0, dh).cos()


def weight_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def main(fp64=False):
    """fp64=True: the same trajectory evaluated in float64 (tests/helpers.py::Fp64Oracle) -> c5_prefill_206m_fp64.npz,
    the reference point of the conditioning-aware comparison (after 1536 tokens x 20 blocks the fp32 oracle itself is
    up to ~1e-3 from the exact result on some rows; the engine must be as close to float64 as the fp32 oracle is)."""
    from lram_amd import init_state_dict, preset
    from oracle.dt_ref import OraclePolicy
    torch.set_num_threads(int(os.environ.get("C5_THREADS", os.cpu_count() or 1)))
    spec = preset("xlstm_206m")
    sd = init_state_dict(spec, seed=WEIGHT_SEED)
    obs, rtg = c5_inputs(spec)
    if fp64:
        from tests.helpers import Fp64Oracle
        f64 = Fp64Oracle(spec, sd)
        ora = f64.ora

        class _Wrap:   # .step in float64, .state of the wrapped policy
            def step(self, *a, **kw):
                return f64.step(*a, **kw)

            @property
            def state(self):
                return f64.ora.state
        ora = _Wrap()
    else:
        ora = OraclePolicy(spec, sd)
    zero = torch.zeros(B)
    out = {"weight_checksum": np.float64(weight_checksum(sd))}
    t0 = time.time()
    for t in range(L + N_DECODE):
        mask = torch.ones(B, dtype=torch.uint8) if t == 0 else None
        act, dbg = ora.step(obs[:, t], rtg[:, t], zero, mask, return_debug=True)
        if t >= L - 1:
            k = t - (L - 1)   # 0 = last context step, 1.. = decode steps
            out[f"actions_{k}"] = act.numpy()
            out[f"logits_{k}"] = dbg["logits"].numpy()
            out[f"hidden_{k}"] = dbg["hidden"].numpy()
        if t == L - 1:
            r = probe(spec.head_dim).to(torch.float64 if fp64 else torch.float32)
            for i in STATE_BLOCKS:
                c, n, m = ora.state[f"block_{i}"]["mlstm_state"]
                out[f"b{i}_n"], out[f"b{i}_m"] = n.numpy(), m.numpy()
                out[f"b{i}_conv"] = ora.state[f"block_{i}"]["conv_state"][0].numpy()
                out[f"b{i}_Cr"] = (c @ r).numpy()            # [B, NH, DH]: rows of C against the probe
                out[f"b{i}_rC"] = (r @ c).numpy()            # columns
                out[f"b{i}_Cabsmax"] = c.abs().amax(dim=(-1, -2)).numpy()
            out[f"b{SLSTM_BLOCK}_slstm"] = ora.state[f"block_{SLSTM_BLOCK}"]["slstm_state"].numpy()
        if t % 32 == 0:
            print(f"step {t} / {L + N_DECODE}  ({time.time() - t0:.0f} s)", flush=True)
    name = "c5_prefill_206m_fp64.npz" if fp64 else "c5_prefill_206m.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, time.time() - t0)


if __name__ == "__main__":
    main(fp64="--fp64" in sys.argv)
