"""CPU oracle: Mamba-1 block stack, recurrent step form.  TEST INFRASTRUCTURE.

Restates the algorithm of `mamba_ssm==2.1.0` + `causal-conv1d==1.3.0.post1` (un-vendored third-party
dependencies of the reference, /root/reference/README.md:99-103) as wired by the reference at
  src/algos/models/decision_mamba.py:52-107  (MambaEncoder: create_block x n_layer + norm_f)
  src/algos/models/decision_mamba.py:109-166 (forward: per-layer (per-token) loop, fused add+norm)
  src/algos/decision_mamba.py:9-25           (InferenceParams: {layer_idx: (conv_state, ssm_state)})
[3P] mamba_ssm/modules/mamba_simple.py Mamba.step, mamba_ssm/modules/block.py Block.forward and
mamba_ssm/ops/triton/layer_norm.py (rms_norm_fn with residual, prenorm) are the functions restated.
Package source is not available here; the independent pure-torch `transformers.models.mamba.MambaMixer`
(installed) is used by tests/test_oracle_mamba.py as a cross-check of the mixer math.

State layout (as in the reference's `key_value_memory_dict`): per layer
  conv_state (B, d_inner, d_conv)   ssm_state (B, d_inner, d_state)      fp32.
Semantics: a FRESH InferenceParams (zero states).  The reference's stale-state behaviour after
`InferenceParams.reset()` is documented, not reproduced.
"""
import torch
import torch.nn.functional as F


def rms_norm(x, weight, eps):
    """[3P] mamba_ssm RMSNorm / rms_norm_fn core: x * rsqrt(mean(x^2) + eps) * weight."""
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * weight


def mamba_step(sd, p, hidden, conv_state, ssm_state, dt_rank, d_state):
    """[3P] mamba_ssm Mamba.step (pure-torch branch, identical math to the fused kernels).

    hidden: (B, D); conv_state: (B, d_inner, d_conv); ssm_state: (B, d_inner, d_state).
    Returns out (B, D) and the new states (the package updates them in place)."""
    xz = F.linear(hidden, sd[p + "in_proj.weight"], sd.get(p + "in_proj.bias"))
    x, z = xz.chunk(2, dim=-1)
    conv_new = torch.roll(conv_state, shifts=-1, dims=-1).clone()
    conv_new[:, :, -1] = x
    x = torch.sum(conv_new * sd[p + "conv1d.weight"][:, 0, :], dim=-1)
    if (p + "conv1d.bias") in sd:
        x = x + sd[p + "conv1d.bias"]
    x = F.silu(x)
    x_db = F.linear(x, sd[p + "x_proj.weight"])
    dt, Bm, Cm = torch.split(x_db, [dt_rank, d_state, d_state], dim=-1)
    dt = F.linear(dt, sd[p + "dt_proj.weight"])
    A = -torch.exp(sd[p + "A_log"].float())
    dt = F.softplus(dt + sd[p + "dt_proj.bias"])
    dA = torch.exp(torch.einsum("bd,dn->bdn", dt, A))
    dB = torch.einsum("bd,bn->bdn", dt, Bm)
    ssm_new = ssm_state * dA + x.unsqueeze(-1) * dB
    y = torch.einsum("bdn,bn->bd", ssm_new, Cm)
    y = y + sd[p + "D"] * x
    y = y * F.silu(z)
    out = F.linear(y, sd[p + "out_proj.weight"], sd.get(p + "out_proj.bias"))
    return out, conv_new, ssm_new


def zero_state(spec, B):
    return {i: (torch.zeros(B, spec.d_inner, spec.d_conv), torch.zeros(B, spec.d_inner, spec.d_state))
            for i in range(spec.n_blocks)}


def reset_state_rows(state, mask):
    keep = (~mask).to(torch.float32).view(-1, 1, 1)
    return {i: (c * keep, s * keep) for i, (c, s) in state.items()}


def encoder_forward_cached(spec, sd, inputs_embeds, state=None, prefix="encoder."):
    """Reference src/algos/models/decision_mamba.py:124-165 for a fresh cache: every layer consumes the
    T tokens of the timestep sequentially (layer 0 through the full scan, layers >= 1 through the
    per-token step loop -- both equal T sequential `Mamba.step`s from the stored state), Block =
    add -> RMSNorm -> mixer with an fp32 residual stream, then the final fused add + norm_f."""
    B, T, D = inputs_embeds.shape
    state = zero_state(spec, B) if state is None else dict(state)
    hidden = inputs_embeds
    residual = None
    for i in range(spec.n_blocks):
        p = f"{prefix}layers.{i}."
        residual = hidden if residual is None else hidden + residual
        normed = rms_norm(residual, sd[p + "norm.weight"], spec.norm_eps)
        conv_s, ssm_s = state[i]
        outs = []
        for t in range(T):
            o, conv_s, ssm_s = mamba_step(sd, p + "mixer.", normed[:, t], conv_s, ssm_s, spec.dt_rank, spec.d_state)
            outs.append(o)
        state[i] = (conv_s, ssm_s)
        hidden = torch.stack(outs, dim=1)
    residual = hidden + residual
    return rms_norm(residual, sd[prefix + "norm_f.weight"], spec.norm_eps), state
