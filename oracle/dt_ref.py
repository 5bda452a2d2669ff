"""CPU oracle: decision-transformer token front end, action head, tokenizer, batched policy step.
TEST INFRASTRUCTURE.

Restates, for the multi-domain recurrent models (MDDXLSTM / MDDMamba, model kwargs of
/root/reference/configs/agent_params/model_kwargs/multi_domain.yaml: reward_condition, rtg_condition,
action_condition=False, shared_a_head, use_time_embds=False, tokenize_a), the reference functions
  src/algos/models/online_decision_transformer_model.py:463-530  compute_inputs / embed_inputs
  src/algos/models/online_decision_transformer_model.py:588-612  prepare_inputs_and_masks (stack, embed_ln)
  src/algos/models/discrete_decision_transformer_model.py:236-316 construct_inputs_and_masks
        (reward_condition & rtg_condition & !action_condition: inputs = (s, rtg, r); tok_to_pred_pos["a"] = 1)
  src/algos/models/discrete_decision_transformer_model.py:368-383 get_predictions
  src/algos/models/multi_domain_discrete_dt_model.py:83-108       get_action_from_logits / prepare_action_logits
  src/tokenizers_custom/minmax_tokenizer.py:14-47                 MinMaxTokenizer.tokenize / inv_tokenize
  src/algos/models/image_encoders.py:10-131                       ImpalaCNN
  src/algos/decision_xlstm.py:11-28, src/algos/decision_transformer_sb3.py:621-667  pad_inputs / predict
and the per-timestep loop body of src/callbacks/evaluation.py:130-177 in batched form.
"""
import torch
import torch.nn.functional as F

from . import mamba_ref, xlstm_ref


# ----------------------------------------------------------------------------------------------
# tokenizer (reference: src/tokenizers_custom/minmax_tokenizer.py)
# ----------------------------------------------------------------------------------------------
def minmax_tokenize(x, vocab_size=256, shift=18, min_val=-1.0, max_val=1.0):
    """MinMaxTokenizer.tokenize (:14-29): ((x-min)/bin_width).long().clamp(0, V-1) + shift."""
    bin_width = (max_val - min_val) / vocab_size
    tokens = ((x - min_val) / bin_width).long().clamp(min=0, max=vocab_size - 1)
    return tokens + shift


def minmax_inv_tokenize(tok, vocab_size=256, shift=18, min_val=-1.0, max_val=1.0):
    """MinMaxTokenizer.inv_tokenize (:31-47): x = tok - shift; x[x<0] = 0; x.float()*bin_width + min."""
    bin_width = (max_val - min_val) / vocab_size
    x = tok - shift
    x = torch.where(x < 0, torch.zeros_like(x), x)
    return x.float() * bin_width + min_val


# ----------------------------------------------------------------------------------------------
# ImpalaCNN (reference: src/algos/models/image_encoders.py:10-131, model_size 1, out_relu True)
# ----------------------------------------------------------------------------------------------
def impala_cnn(sd, p, x):
    """x: float (B,3,64,64) already divided by 255 (online_decision_transformer_model.py:523-525)."""
    def conv(key, t):
        return F.conv2d(t, sd[key + ".weight"], sd[key + ".bias"], stride=1, padding=1)

    for b in range(3):
        q = f"{p}cnn.{b}."
        x = conv(q + "conv", x)
        x = F.max_pool2d(x, 3, 2, padding=1)
        for r in range(2):
            y = conv(f"{q}residual_{r}.conv_0", F.relu(x))
            y = conv(f"{q}residual_{r}.conv_1", F.relu(y))
            x = x + y
    x = F.relu(x).flatten(1)
    return F.relu(F.linear(x, sd[p + "linear.0.weight"], sd[p + "linear.0.bias"]))


# ----------------------------------------------------------------------------------------------
# front end + head
# ----------------------------------------------------------------------------------------------
def embed_tokens(spec, sd, obs, rtg, reward, state_mean=None, state_std=None):
    """obs: (B, state_dim) float32 already zero-padded to max_state_dim (decision_xlstm.py:16-19), or
    uint8 (B,3,64,64); rtg, reward: (B,).  Returns embed_ln(stack(s, rtg, r)) of shape (B, 3, D)."""
    if obs.dim() == 4:
        s = impala_cnn(sd, "embed_image.", obs.float() / 255.0)
    else:
        if state_mean is not None:
            obs = (obs - state_mean) / state_std  # decision_transformer_sb3.py:650-651
        s = F.linear(obs, sd["embed_state.weight"], sd["embed_state.bias"])
    g = F.linear(rtg.view(-1, 1), sd["embed_return.weight"], sd["embed_return.bias"])
    r = F.linear(reward.view(-1, 1), sd["embed_rewards.weight"], sd["embed_rewards.bias"])
    x = torch.stack((s, g, r), dim=1)
    return F.layer_norm(x, (x.shape[-1],), sd["embed_ln.weight"], sd.get("embed_ln.bias"), eps=1e-5)


def actions_from_logits(spec, logits, discrete):
    """action_net output (B, act_dim * n_vocab) -> (action, logits as the reference shapes them).
    multi_domain_discrete_dt_model.py:83-108 (prepare_action_logits + get_action_from_logits), shared head;
    pinned against the reference's own methods by tests/golden/reference_vectors.json `action_from_logits`.

    continuous: logits (B, act_dim, n_vocab) -> argmax -> inv_tokenize -> float (B, act_dim)
    discrete  : logits[:, :n_vocab][:, :n_discrete] -> argmax -> int64 (B, 1)"""
    B = logits.shape[0]
    if discrete:
        lg = logits[:, : spec.n_vocab]
        act = torch.argmax(lg[:, : spec.n_discrete], dim=-1).view(B, 1)
        return act, lg.view(B, 1, spec.n_vocab)
    lg = logits.view(B, spec.act_dim, spec.n_vocab)
    tok = torch.argmax(lg, dim=-1)
    return minmax_inv_tokenize(tok, spec.action_channels, spec.n_discrete), lg


def action_head(spec, sd, x_a, discrete):
    """x_a: (B, D) hidden at the rtg-token position.  Returns (action, logits)."""
    return actions_from_logits(spec, F.linear(x_a, sd["action_net.0.weight"], sd["action_net.0.bias"]), discrete)


class OraclePolicy:
    """Batched CPU oracle of one env-step of `agent.predict` with `use_inference_cache=True`.

    step(obs, rtg, reward, reset_mask) == for every env independently: (reset its cache if masked,)
    embed (s, rtg, r), run the 3 tokens through the recurrent stack, read the action at the rtg token.
    """

    def __init__(self, spec, sd, state_mean=None, state_std=None, mamba_repeat=1, stale_state=False):
        """mamba_repeat / stale_state: the trajectory the reference's Mamba agent actually produces:
        Q2  DiscreteDecisionMamba.get_action_pred (src/algos/decision_mamba.py:107-122) runs `policy(**inputs)` once
            per action dim with the cache on -- the same (s, rtg, r) tokens advance the conv / ssm state
            `mamba_repeat` = env_act_dim times per env-step, and action dim i is `action_preds[0, -1, i]` of forward i
            (columns >= mamba_repeat, which the reference slices away, hold the last forward here);
        Q1  InferenceParams.reset() (:20-25) only zeroes seqlen_offset; MambaEncoder.forward bumps the offset inside
            its layer loop (src/algos/models/decision_mamba.py:130-149), so after a reset layer 0 takes the
            full-sequence path from an empty state (overwriting its cache) while layers >= 1 take the step path on the
            previous episode's conv / ssm state: `stale_state` resets layer 0 only."""
        if (mamba_repeat != 1 or stale_state) and spec.backbone != "mamba":
            raise ValueError("mamba_repeat / stale_state are quirks of the reference's Mamba agent")
        self.mamba_repeat, self.stale_state = int(mamba_repeat), bool(stale_state)
        self.spec = spec
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}
        self.state_mean, self.state_std = state_mean, state_std
        self.state = None
        self.B = None

    def reset(self, B):
        self.B = B
        mod = mamba_ref if self.spec.backbone == "mamba" else xlstm_ref
        self.state = mod.zero_state(self.spec, B)

    @torch.no_grad()
    def step(self, obs, rtg, reward, reset_mask=None, discrete=False, return_debug=False):
        spec, sd = self.spec, self.sd
        if self.state is None:
            self.reset(obs.shape[0])
        mod = mamba_ref if spec.backbone == "mamba" else xlstm_ref
        if reset_mask is not None and bool(reset_mask.any()):
            if self.stale_state:
                fresh = mod.reset_state_rows({0: self.state[0]}, reset_mask.bool())
                self.state = {**self.state, 0: fresh[0]}
            else:
                self.state = mod.reset_state_rows(self.state, reset_mask.bool())
        x = embed_tokens(spec, sd, obs, rtg, reward, self.state_mean, self.state_std)
        passes = 1 if (discrete or spec.backbone != "mamba") else max(1, min(self.mamba_repeat, spec.act_dim))
        act = logits = hidden = None
        for p in range(passes):
            if spec.backbone == "mamba":
                hidden, self.state = mamba_ref.encoder_forward_cached(spec, sd, x, self.state)
            else:
                hidden, self.state = xlstm_ref.encoder_forward_cached(spec, sd, x, self.state)
            act_p, logits_p = action_head(spec, sd, hidden[:, 1], discrete)
            if p == 0:
                act, logits = act_p.clone(), logits_p.clone()
            else:  # action dim i comes from forward i (columns beyond the last forward follow it)
                act[:, p:] = act_p[:, p:]
                logits[:, p:] = logits_p[:, p:]
        if return_debug:
            return act, {"tokens": x, "hidden": hidden, "logits": logits}
        return act
