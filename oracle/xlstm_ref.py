"""CPU oracle: xLSTM block stack, recurrent (`step`) and parallel forms.  TEST INFRASTRUCTURE.

Restates the algorithm of the third-party `xlstm` package (PyPI, un-vendored and unpinned in the
reference: /root/reference/README.md:94-97; API generation 1.0.x) as used by the reference at
  src/algos/models/decision_xlstm.py:8-12   (imports xLSTMBlockStack, LayerNorm, MultiHeadLayerNorm,
                                              LinearHeadwiseExpand, sLSTMCell_cuda, mLSTMCell)
  src/algos/models/decision_xlstm.py:130-133 (xLSTMBlockStackConfig via dacite, xLSTMBlockStack(cfg))
  src/algos/models/decision_xlstm.py:155-167 (per-token `self.layers.step` loop / parallel forward)
  src/algos/models/decision_xlstm.py:186-191 (rms_norm swap of the non-multihead LayerNorms)
The package source is NOT available in this container => "parity unpinned" (see oracle/__init__.py).
Module names quoted below ([3P] xlstm/...) are the public package layout the restatement follows.

All tensors fp32, CPU.  Weights are read from a flat state-dict `sd` that uses the reference's
checkpoint key names below a prefix, e.g. `encoder.layers.`.

State layout (identical to what the reference stores in `past_key_values`,
SURVEY.md section 3.4):
  state[f"block_{i}"] = {"mlstm_state": (C[B,NH,DH,DH], n[B,NH,DH,1], m[B,NH,1,1]),
                         "conv_state": (conv[B,K,inner],)}                       # mLSTM block
  state[f"block_{i}"] = {"slstm_state": S[4,B,D] (= y,c,n,m), "conv_state": (conv[B,K,D],)}  # sLSTM
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# components
# ----------------------------------------------------------------------------------------------
def layer_norm(x, weight, bias=None, eps=1e-5, residual_weight=True):
    """[3P] xlstm/components/ln.py LayerNorm.forward: F.layer_norm with weight_proxy = 1 + weight."""
    w = (1.0 + weight) if residual_weight else weight
    return F.layer_norm(x, (x.shape[-1],), weight=w, bias=bias, eps=eps)


def rms_norm(x, weight, eps):
    """Reference src/algos/models/rms_norm.py:17-22 (LlamaRMSNorm.forward), fp32 input."""
    variance = x.pow(2).mean(-1, keepdim=True)
    return weight * (x * torch.rsqrt(variance + eps))


def multihead_layer_norm(x, weight, bias=None, eps=1e-5):
    """[3P] xlstm/components/ln.py MultiHeadLayerNorm.forward.  x: (B, NH, S, DH) -> (B, NH, S, DH).

    group_norm over (B*S, NH*DH) with NH groups, affine weight_proxy = 1 + weight of size NH*DH."""
    B, NH, S, DH = x.shape
    gn_in = x.transpose(1, 2).reshape(B * S, NH * DH)
    out = F.group_norm(gn_in, num_groups=NH, weight=1.0 + weight, bias=bias, eps=eps)
    return out.view(B, S, NH, DH).transpose(1, 2)


def linear_headwise(x, weight):
    """[3P] xlstm/components/linear_headwise.py LinearHeadwiseExpand.forward (no bias).

    weight: (num_heads, out_per_head, in_per_head); einsum '...hd,hod->...ho'."""
    shape = x.shape
    nh = weight.shape[0]
    xh = x.view(*shape[:-1], nh, -1)
    y = torch.einsum("...hd,hod->...ho", xh, weight)
    return y.reshape(*shape[:-1], -1)


def conv1d_step(x, conv_state, conv_weight, conv_bias):
    """[3P] xlstm/components/conv.py conv1d_step + CausalConv1d.step.

    x: (B,1,D); conv_state: (B,K,D) or None; conv_weight: nn.Conv1d weight (D,1,K); returns y (B,1,D),
    new state.  The reference mutates the state in place (copy_); the oracle returns a fresh tensor."""
    B, S, D = x.shape
    assert S == 1
    K = conv_weight.shape[-1]
    if conv_state is None:
        conv_state = torch.zeros(B, K, D, dtype=x.dtype)
    new_state = torch.roll(conv_state, shifts=-1, dims=1).clone()
    new_state[:, -1:, :] = x
    w = conv_weight[:, 0, :].transpose(0, 1)  # (K, D)
    y = torch.sum(new_state * w, dim=1, keepdim=True)
    if conv_bias is not None:
        y = y + conv_bias
    return y, new_state


def conv1d_full(x, conv_weight, conv_bias):
    """[3P] CausalConv1d.forward: depthwise Conv1d with left padding K-1 (causal), x: (B,S,D)."""
    K = conv_weight.shape[-1]
    y = F.conv1d(x.transpose(1, 2), conv_weight, conv_bias, padding=K - 1, groups=x.shape[-1])
    return y[:, :, : -(K - 1)].transpose(1, 2) if K > 1 else y.transpose(1, 2)


# ----------------------------------------------------------------------------------------------
# mLSTM
# ----------------------------------------------------------------------------------------------
def mlstm_recurrent_step(c_state, n_state, m_state, q, k, v, igate_preact, fgate_preact, eps=1e-6):
    """[3P] xlstm/blocks/mlstm/backends.py recurrent_step_stabilized_simple.

    c (B,NH,DH,DH), n (B,NH,DH,1), m (B,NH,1,1); q,k,v (B,NH,1,DH); gates (B,NH,1,1)."""
    B, NH, S, DH = q.shape
    q, k, v = q.squeeze(2).unsqueeze(-1), k.squeeze(2).unsqueeze(-1), v.squeeze(2).unsqueeze(-1)
    log_fg_act = F.logsigmoid(fgate_preact)
    m_state_new = torch.max(log_fg_act + m_state, igate_preact)
    fg_act = torch.exp(log_fg_act + m_state - m_state_new)
    ig_act = torch.exp(igate_preact - m_state_new)
    k_scaled = k / math.sqrt(DH)
    c_state_new = fg_act * c_state + ig_act * (k_scaled @ v.transpose(-1, -2))
    n_state_new = fg_act * n_state + ig_act * k_scaled
    h_num = q.transpose(-1, -2) @ c_state_new
    qn_dotproduct = q.transpose(-1, -2) @ n_state_new
    max_val = torch.exp(-m_state_new)
    h_denom = torch.maximum(qn_dotproduct.abs(), max_val) + eps
    h = h_num / h_denom
    return h, (c_state_new, n_state_new, m_state_new)


def mlstm_parallel(q, k, v, igate_preact, fgate_preact, eps=1e-6):
    """[3P] xlstm/blocks/mlstm/backends.py parallel_stabilized_simple (row-wise stabilisation).

    q,k,v (B,NH,S,DH); gates (B,NH,S,1) -> h_tilde (B,NH,S,DH).  Used only for the step<->parallel
    self-check (reference call site: decision_xlstm.py:167, the no-cache path)."""
    B, NH, S, DH = q.shape
    log_fgates = F.logsigmoid(fgate_preact)
    ltr = torch.tril(torch.ones((S, S), dtype=torch.bool))
    log_fgates_cumsum = torch.cat([torch.zeros((B, NH, 1, 1), dtype=q.dtype), torch.cumsum(log_fgates, dim=-2)], dim=-2)
    rep = log_fgates_cumsum.repeat(1, 1, 1, S + 1)
    _log_fg_matrix = rep - rep.transpose(-2, -1)
    log_fg_matrix = torch.where(ltr, _log_fg_matrix[:, :, 1:, 1:], -float("inf"))
    log_D_matrix = log_fg_matrix + igate_preact.transpose(-2, -1)
    max_log_D, _ = torch.max(log_D_matrix, dim=-1, keepdim=True)
    D_matrix = torch.exp(log_D_matrix - max_log_D)
    keys_scaled = k / math.sqrt(DH)
    qk_matrix = q @ keys_scaled.transpose(-2, -1)
    C_matrix = qk_matrix * D_matrix
    normalizer = torch.maximum(C_matrix.sum(dim=-1, keepdim=True).abs(), torch.exp(-max_log_D))
    C_matrix_normalized = C_matrix / (normalizer + eps)
    return C_matrix_normalized @ v


def _mlstm_qkv_gates(sd, p, x_mlstm, x_conv_act, NH):
    """Shared by step and parallel: q,k from the conv branch, v from the pre-conv branch; gate
    pre-activations from cat[q,k,v] ([3P] mLSTMLayer.forward/step + mLSTMCell.forward/step)."""
    B, S, _ = x_mlstm.shape
    q = linear_headwise(x_conv_act, sd[p + "q_proj.weight"])
    k = linear_headwise(x_conv_act, sd[p + "k_proj.weight"])
    v = linear_headwise(x_mlstm, sd[p + "v_proj.weight"])
    if_gate_input = torch.cat([q, k, v], dim=-1)
    ig = F.linear(if_gate_input, sd[p + "mlstm_cell.igate.weight"], sd[p + "mlstm_cell.igate.bias"])
    fg = F.linear(if_gate_input, sd[p + "mlstm_cell.fgate.weight"], sd[p + "mlstm_cell.fgate.bias"])
    ig = ig.transpose(-1, -2).unsqueeze(-1)  # (B,NH,S,1)
    fg = fg.transpose(-1, -2).unsqueeze(-1)
    q = q.view(B, S, NH, -1).transpose(1, 2)
    k = k.view(B, S, NH, -1).transpose(1, 2)
    v = v.view(B, S, NH, -1).transpose(1, 2)
    return q, k, v, ig, fg


def mlstm_layer_step(sd, p, x, NH, mlstm_state=None, conv_state=None, ln_bias=False):
    """[3P] xlstm/blocks/mlstm/layer.py mLSTMLayer.step + cell.py mLSTMCell.step.  x: (B,1,D)."""
    B, S, _ = x.shape
    inner = sd[p + "proj_down.weight"].shape[1]
    x_inner = F.linear(x, sd[p + "proj_up.weight"])
    x_mlstm, z = torch.split(x_inner, inner, dim=-1)
    x_conv, conv_new = conv1d_step(x_mlstm, None if conv_state is None else conv_state[0],
                                   sd[p + "conv1d.conv.weight"], sd[p + "conv1d.conv.bias"])
    x_conv_act = F.silu(x_conv)
    q, k, v, ig, fg = _mlstm_qkv_gates(sd, p, x_mlstm, x_conv_act, NH)
    DH = inner // NH
    if mlstm_state is None:
        c = torch.zeros(B, NH, DH, DH)
        n = torch.zeros(B, NH, DH, 1)
        m = torch.zeros(B, NH, 1, 1)
    else:
        c, n, m = mlstm_state
    h_state, new_state = mlstm_recurrent_step(c, n, m, q, k, v, ig, fg)
    h_norm = multihead_layer_norm(h_state, sd[p + "mlstm_cell.outnorm.weight"],
                                  sd.get(p + "mlstm_cell.outnorm.bias") if ln_bias else None)
    h_norm = h_norm.transpose(1, 2).reshape(B, S, -1)
    h_skip = h_norm + sd[p + "learnable_skip"] * x_conv_act
    h_out = h_skip * F.silu(z)
    y = F.linear(h_out, sd[p + "proj_down.weight"])
    return y, {"mlstm_state": new_state, "conv_state": (conv_new,)}


def mlstm_layer_forward(sd, p, x, NH, ln_bias=False):
    """[3P] mLSTMLayer.forward (parallel form), x: (B,S,D)."""
    B, S, _ = x.shape
    inner = sd[p + "proj_down.weight"].shape[1]
    x_inner = F.linear(x, sd[p + "proj_up.weight"])
    x_mlstm, z = torch.split(x_inner, inner, dim=-1)
    x_conv_act = F.silu(conv1d_full(x_mlstm, sd[p + "conv1d.conv.weight"], sd[p + "conv1d.conv.bias"]))
    q, k, v, ig, fg = _mlstm_qkv_gates(sd, p, x_mlstm, x_conv_act, NH)
    h_state = mlstm_parallel(q, k, v, ig, fg)
    h_norm = multihead_layer_norm(h_state, sd[p + "mlstm_cell.outnorm.weight"],
                                  sd.get(p + "mlstm_cell.outnorm.bias") if ln_bias else None)
    h_norm = h_norm.transpose(1, 2).reshape(B, S, -1)
    h_out = (h_norm + sd[p + "learnable_skip"] * x_conv_act) * F.silu(z)
    return F.linear(h_out, sd[p + "proj_down.weight"])


# ----------------------------------------------------------------------------------------------
# sLSTM
# ----------------------------------------------------------------------------------------------
def slstm_pointwise(Wx, Ry, b, states, per_env_first_step=True):
    """[3P] xlstm/blocks/slstm/src/vanilla/slstm.py slstm_forward_pointwise (the CPU 'vanilla' backend).

    Wx, Ry: (B, 4, H) gate-major (i,f,z,o); b: (4, H); states: (4, B, H) = y,c,n,m.
    The package's CPU path tests `torch.all(n == 0.0)` over the whole state tensor; the reference always
    runs batch 1 (src/callbacks/evaluation.py:80), so the faithful batched semantics is per env
    (`per_env_first_step=True`).  The CUDA kernel ([3P] slstm_pointwise.cuh) tests per element; the three
    coincide because n is 0 everywhere before an env's first step and > 0 everywhere after it."""
    raw = Wx + Ry + b
    y, c, n, m = torch.unbind(states, dim=0)
    iraw, fraw, zraw, oraw = torch.unbind(raw, dim=1)
    logfplusm = m + F.logsigmoid(fraw)
    if per_env_first_step:
        first = torch.all(n == 0.0, dim=-1, keepdim=True)
        mnew = torch.where(first, iraw, torch.max(iraw, logfplusm))
    else:
        mnew = iraw if torch.all(n == 0.0) else torch.max(iraw, logfplusm)
    ogate = torch.sigmoid(oraw)
    igate = torch.minimum(torch.exp(iraw - mnew), torch.ones_like(iraw))
    fgate = torch.minimum(torch.exp(logfplusm - mnew), torch.ones_like(iraw))
    cnew = fgate * c + igate * torch.tanh(zraw)
    nnew = fgate * n + igate
    ynew = ogate * cnew / nnew
    return torch.stack((ynew, cnew, nnew, mnew), dim=0)


def slstm_cell_step(gates_in, states, R, bias, NH):
    """[3P] xlstm/blocks/slstm/cell.py sLSTMCellBase.forward for S=1 + vanilla slstm_forward.

    gates_in: (B, 4*H) laid out as cat[i,f,z,o] (gate-major, each H = NH*DH);
    states: (4,B,H); R: `_recurrent_kernel_` (NH, DH_in, 4, DH_out); bias: `_bias_` (NH, 4, DH)."""
    B = gates_in.shape[0]
    H = gates_in.shape[1] // 4
    DH = H // NH
    Wx = gates_in.view(B, 4, H)
    y_prev = states[0].view(B, NH, DH)
    # Ry[b, g, h, o] = sum_i y_prev[b,h,i] * R[h,i,g,o]
    Ry = torch.einsum("bhi,higo->bgho", y_prev, R).reshape(B, 4, H)
    b = bias.permute(1, 0, 2).reshape(4, H)
    return slstm_pointwise(Wx, Ry, b, states)


def slstm_layer_step(sd, p, x, NH, slstm_state=None, conv_state=None, ln_bias=False):
    """[3P] xlstm/blocks/slstm/layer.py sLSTMLayer.step.  x: (B,1,D).

    NOTE the package's gate wiring, kept verbatim because checkpoints depend on it:
        i, f, z, o = (self.fgate(x_conv), self.igate(x_conv), self.zgate(x), self.ogate(x))
    i.e. the tensor fed to the cell's *input*-gate slot is produced by the module named `fgate`
    and the *forget*-gate slot by `igate`."""
    B, S, D = x.shape
    x_conv, conv_new = conv1d_step(x, None if conv_state is None else conv_state[0],
                                   sd[p + "conv1d.conv.weight"], sd[p + "conv1d.conv.bias"])
    x_conv = F.silu(x_conv)
    i = linear_headwise(x_conv, sd[p + "fgate.weight"])
    f = linear_headwise(x_conv, sd[p + "igate.weight"])
    z = linear_headwise(x, sd[p + "zgate.weight"])
    o = linear_headwise(x, sd[p + "ogate.weight"])
    if slstm_state is None:
        slstm_state = torch.zeros(4, B, D)
    gates_in = torch.cat([i, f, z, o], dim=-1)[:, 0]
    new_states = slstm_cell_step(gates_in, slstm_state, sd[p + "slstm_cell._recurrent_kernel_"],
                                 sd[p + "slstm_cell._bias_"], NH)
    y = new_states[0].view(B, 1, NH, D // NH).permute(0, 2, 1, 3)  # output_shape "BNSH"
    out = multihead_layer_norm(y, sd[p + "group_norm.weight"],
                               sd.get(p + "group_norm.bias") if ln_bias else None)
    out = out.transpose(1, 2).reshape(B, S, -1)
    return out, {"slstm_state": new_states, "conv_state": (conv_new,)}


def feedforward(sd, p, x):
    """[3P] xlstm/components/feedforward.py GatedFeedForward.forward (act_fn 'gelu' = exact erf GELU,
    reference configs/agent_params/huggingface/xlstm_medium.yaml:19-21)."""
    F_dim = sd[p + "proj_down.weight"].shape[1]
    gate_preact, up_proj = F.linear(x, sd[p + "proj_up.weight"]).split(F_dim, dim=-1)
    return F.linear(F.gelu(gate_preact) * up_proj, sd[p + "proj_down.weight"])


# ----------------------------------------------------------------------------------------------
# block stack
# ----------------------------------------------------------------------------------------------
def _norm(spec, sd, key, x):
    """Block / post-stack norm: xLSTM LayerNorm (gamma = 1 + w) or, when the HF config carries
    `rms_norm`, LlamaRMSNorm(ndim, eps=module.eps) with a plain weight (decision_xlstm.py:190-191,236-241)."""
    if getattr(spec, "rms_norm", False):
        return rms_norm(x, sd[key + ".weight"], spec.ln_eps)
    bias = sd.get(key + ".bias") if getattr(spec, "ln_bias", False) else None
    return layer_norm(x, sd[key + ".weight"], bias, eps=spec.ln_eps)


def block_step(spec, sd, prefix, i, x, block_state):
    """[3P] xlstm/blocks/xlstm_block.py xLSTMBlock.step."""
    p = f"{prefix}blocks.{i}."
    xn = _norm(spec, sd, p + "xlstm_norm", x)
    lb = getattr(spec, "ln_bias", False)
    if i in spec.slstm_at:
        y, new_state = slstm_layer_step(sd, p + "xlstm.", xn, spec.n_heads, ln_bias=lb, **block_state)
    else:
        y, new_state = mlstm_layer_step(sd, p + "xlstm.", xn, spec.n_heads, ln_bias=lb, **block_state)
    x = x + y
    if i in spec.slstm_at:
        x = x + feedforward(sd, p + "ffn.", _norm(spec, sd, p + "ffn_norm", x))
    return x, new_state


def stack_step(spec, sd, x, state=None, prefix="encoder.layers."):
    """[3P] xlstm/xlstm_block_stack.py xLSTMBlockStack.step.  x: (B,1,D) -> (B,1,D), state dict.
    post_blocks_norm is applied on every call, i.e. per token (decision_xlstm.py:162-164)."""
    state = {} if state is None else dict(state)
    for i in range(spec.n_blocks):
        x, state[f"block_{i}"] = block_step(spec, sd, prefix, i, x, state.get(f"block_{i}", {}))
    x = _norm(spec, sd, prefix + "post_blocks_norm", x)
    return x, state


def encoder_forward_cached(spec, sd, inputs_embeds, past_key_values=None, prefix="encoder.layers."):
    """Reference src/algos/models/decision_xlstm.py:155-166 (use_cache, chunkwise_step=False):
    one `layers.step` per token, hidden states concatenated, last state returned."""
    hs = []
    for i in range(inputs_embeds.shape[1]):
        h, past_key_values = stack_step(spec, sd, inputs_embeds[:, i].unsqueeze(1), past_key_values, prefix)
        hs.append(h)
    return torch.cat(hs, dim=1), past_key_values


def stack_forward_parallel(spec, sd, x, prefix="encoder.layers."):
    """[3P] xLSTMBlockStack.forward restricted to mLSTM blocks (parallel form, from zero state).
    Only for the step<->parallel self-check; reference call site decision_xlstm.py:167."""
    assert len(spec.slstm_at) == 0, "parallel self-check covers mLSTM-only stacks"
    lb = getattr(spec, "ln_bias", False)
    for i in range(spec.n_blocks):
        p = f"{prefix}blocks.{i}."
        x = x + mlstm_layer_forward(sd, p + "xlstm.", _norm(spec, sd, p + "xlstm_norm", x), spec.n_heads, lb)
    return _norm(spec, sd, prefix + "post_blocks_norm", x)


def zero_state(spec, B):
    """Explicit zero state in the reference's `past_key_values` layout (what `None` means upstream)."""
    st = {}
    D = spec.d_model
    for i in range(spec.n_blocks):
        if i in spec.slstm_at:
            st[f"block_{i}"] = {"slstm_state": torch.zeros(4, B, D),
                                "conv_state": (torch.zeros(B, spec.conv_k, D),)}
        else:
            inner, NH = spec.inner, spec.n_heads
            DH = inner // NH
            st[f"block_{i}"] = {"mlstm_state": (torch.zeros(B, NH, DH, DH), torch.zeros(B, NH, DH, 1),
                                                torch.zeros(B, NH, 1, 1)),
                                "conv_state": (torch.zeros(B, spec.conv_k, inner),)}
    return st


def reset_state_rows(state, mask):
    """Zero the recurrent state of every env whose mask entry is True (what the rollout loop does by
    setting `model.past_key_values = None` for its single env, src/callbacks/evaluation.py:238-251)."""
    keep = (~mask).to(torch.float32)
    out = {}
    for name, blk in state.items():
        nb = {}
        if "mlstm_state" in blk:
            c, n, m = blk["mlstm_state"]
            nb["mlstm_state"] = (c * keep.view(-1, 1, 1, 1), n * keep.view(-1, 1, 1, 1), m * keep.view(-1, 1, 1, 1))
        if "slstm_state" in blk:
            nb["slstm_state"] = blk["slstm_state"] * keep.view(1, -1, 1)
        nb["conv_state"] = (blk["conv_state"][0] * keep.view(-1, 1, 1),)
        out[name] = nb
    return out
