"""Rollout surface of the reference's recurrent agents on top of the HIP engine.

Keeps the attribute / method surface that `custom_evaluate_policy` touches on the agent
(src/callbacks/evaluation.py:97-129,134-139,238-251):

    predict(policy, observation, actions, rewards, returns_to_go, timesteps, state=None, episode_start=None,
            deterministic=True, context_len=5, prompt=None, task_id=None, is_eval=False, env_act_dim=None)
                                                   (src/algos/decision_transformer_sb3.py:621-667)
    get_action_pred(policy, states, actions, rewards, returns_to_go, timesteps, attention_mask,
                    deterministic, prompt, is_eval=False, task_id=None, env_act_dim=None)
                                                   (src/algos/discrete_decision_transformer_sb3.py:13-72,
                                                    src/algos/decision_mamba.py:76-127)
    get_action(...)  alias named by BASELINE.json's north_star
    attributes: policy, device, eval_context_len, use_inference_cache, past_key_values,
                inference_params.reset(), persist_context, compile, replay_buffer.{seqs_per_sample,
                max_state_dim, max_act_dim}, target_return_type, compute_target_return_val(),
                get_reward_scale_for_env()

The single-env methods are thin views (batch slot 0 of a B=1 engine, exactly the reference's operating
point); `predict_batch` is the native entry: one call advances all B envs by one timestep.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .config import ModelSpec
from .engine import Engine


class _InferenceParams:
    """Stand-in for the reference's InferenceParams (src/algos/decision_mamba.py:9-25): reset() clears the
    cache -- every layer by default; with the agent's `compat_stale_state` only layer 0, as the reference does
    (SURVEY.md 3.5 Q1)."""

    def __init__(self, agent: "RecurrentAgent"):
        self._agent = agent
        self.seqlen_offset = 0

    def reset(self):
        self.seqlen_offset = 0
        self._agent.engine.reset()


class RecurrentAgent:
    def __init__(self, spec: ModelSpec, state_dict: Dict[str, torch.Tensor], n_envs: int = 1, device=None,
                 discrete: bool = False, state_mean: Optional[torch.Tensor] = None,
                 state_std: Optional[torch.Tensor] = None, target_return: float = 0.0, reward_scale: float = 1.0,
                 graph: bool = False, reprime_context: bool = False, persist_context: bool = False,
                 compat_mamba_repeat: bool = False,
                 compat_stale_state: bool = False):
        self.spec = spec
        # host copy of the weights: lets the agent cross a process boundary (make_pickleable / reinit_cuda_kernels)
        self._state_dict = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        self._graph = bool(graph)
        self.engine = Engine(spec, state_dict, n_envs, device)
        self.device = self.engine.device
        self.n_envs = n_envs
        self.is_discrete = bool(discrete)
        self.policy = self  # `model.predict(model.policy, ...)`: the policy argument is accepted and ignored
        self.state_mean = None if state_mean is None else state_mean.to(self.device, torch.float32)
        self.state_std = None if state_std is None else state_std.to(self.device, torch.float32)
        # image observations go through the engine's own IMPALA-CNN kernels (lram_embed_images): one backend in the package
        # (the PyTorch / MIOpen module that cross-checks them lives with the tests: tests/torch_image_encoder.py)
        self.has_image_encoder = any(k.startswith("embed_image.") for k in state_dict) and spec.image_shape is not None
        # attributes read by the evaluation loop
        self.eval_context_len = spec.max_length
        self.use_inference_cache = True
        self.reset_inf_cache_freq = spec.reset_inf_cache_freq
        # reference behaviour (False): the context is dropped when the cache is reset;
        # True: the last eval_context_len stored timesteps are fed back through Engine.prefill
        self.reprime_context = bool(reprime_context)
        # evaluation.py:213-236: keep the context (here: the recurrent cache) across episode ends
        self.persist_context = bool(persist_context)
        self.compile = False
        self.target_return_type = "predefined"
        self.target_return = float(target_return) / float(reward_scale)
        self.reward_scale = float(reward_scale)
        self.replay_buffer = SimpleNamespace(seqs_per_sample=1, max_state_dim=spec.state_dim, max_act_dim=spec.act_dim)
        self.inference_params = _InferenceParams(self)
        self.inf_dummy_batch_size = None
        if graph:
            self.engine.set_graph_mode(True)
        self._zero_reward = torch.zeros(n_envs, dtype=torch.float32, device=self.device)
        # Reference-trajectory modes of the Mamba agent (lram_set_compat_mode).  Off: one state
        # advance per env-step and a reset empties every layer.  On: the trajectory the reference's
        # DiscreteDecisionMamba.get_action_pred / InferenceParams.reset() actually produce.
        if (compat_mamba_repeat or compat_stale_state) and spec.backbone != "mamba":
            raise ValueError("compat_mamba_repeat / compat_stale_state reproduce quirks of the reference's Mamba agent "
                             "(src/algos/decision_mamba.py); the xLSTM agent has neither")
        self.compat_mamba_repeat = bool(compat_mamba_repeat)
        self.compat_stale_state = bool(compat_stale_state)
        self._compat_repeat_now = 1
        if self.compat_stale_state:
            self.engine.set_compat_mode(1, True)

    @property
    def trajectory_mode(self) -> dict:
        """Which trajectory semantics the rollout uses (logged by rollout / bench)."""
        return {"compat_mamba_repeat": self.compat_mamba_repeat, "compat_stale_state": self.compat_stale_state}

    # ---- cache handle: `model.past_key_values = None` resets, reading exports the reference layout ----
    @property
    def past_key_values(self):
        return self.engine.export_past_key_values()

    @past_key_values.setter
    def past_key_values(self, value):
        if value is None:
            self.engine.reset()
        else:
            self.engine.import_past_key_values(value)

    # ---- multiprocess evaluation (src/callbacks/custom_eval_callback.py:22-33, decision_xlstm.py:243-267) ----
    def make_pickleable(self, replace_cell: bool = False):
        """The native engine handle cannot be serialised: release it (spec and host weights stay), as the reference
        unsets its sLSTM CUDA kernels before handing the model to loky workers."""
        if self.engine is not None:
            self.engine.close()
            self.engine = None

    def reinit_cuda_kernels(self, replace_cell: bool = False):
        """Worker-side counterpart of make_pickleable: build a fresh engine (recurrent state starts empty)."""
        if self.engine is None:
            self.engine = Engine(self.spec, self._state_dict, self.n_envs, self.device)
            if self._graph:
                self.engine.set_graph_mode(True)
            if self.compat_stale_state or self._compat_repeat_now != 1:
                self.engine.set_compat_mode(self._compat_repeat_now, self.compat_stale_state)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["engine"] = None          # never pickled; reinit_cuda_kernels() rebuilds it
        d["policy"] = None          # self-reference, restored below
        d["inference_params"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.policy = self
        self.inference_params = _InferenceParams(self)

    def compute_target_return_val(self, env=None, task_id=0):
        return self.target_return

    def get_reward_scale_for_env(self, envid=None):
        return self.reward_scale

    # ---- native batched entry ---------------------------------------------------------------------
    def _prepare_obs(self, obs: torch.Tensor):
        """pad_inputs + normalisation (src/algos/decision_xlstm.py:11-28, decision_transformer_sb3.py:650-651)
        or the image encoder; returns (tensor, is_embedding)."""
        obs = obs.to(self.device)
        if obs.dim() == 4:
            if not self.has_image_encoder:
                raise RuntimeError("image observation given but the state dict has no embed_image.* weights")
            return self.engine.embed_images(obs.to(torch.uint8).contiguous()), True
        obs = obs.to(torch.float32)
        pad = self.spec.state_dim - obs.shape[-1]
        if pad < 0:
            raise ValueError(f"observation dim {obs.shape[-1]} exceeds max_state_dim {self.spec.state_dim}")
        if pad > 0:
            obs = torch.cat([obs, torch.zeros(*obs.shape[:-1], pad, device=self.device)], dim=-1)
        if self.state_mean is not None and self.state_std is not None:
            obs = (obs - self.state_mean) / self.state_std
        return obs.contiguous(), False

    @torch.no_grad()
    def predict_batch(self, observation: torch.Tensor, returns_to_go: torch.Tensor,
                      rewards: Optional[torch.Tensor] = None, reset_mask: Optional[torch.Tensor] = None,
                      env_act_dim: Optional[int] = None) -> torch.Tensor:
        """observation [B, obs_dim] (or uint8 [B,3,64,64]), returns_to_go [B] -> actions [B, env_act_dim]
        (float32; for discrete agents int64 [B, 1]).  The returned tensor is a view of an engine-owned
        buffer that the next call overwrites."""
        images = observation.dim() == 4
        if images:   # frames go to the engine as they are: lram_step_images runs the CNN inside the step (per env slice)
            if not self.has_image_encoder:
                raise RuntimeError("image observation given but the state dict has no embed_image.* weights")
            obs, is_emb = observation.to(self.device).to(torch.uint8).contiguous(), True
        else:
            obs, is_emb = self._prepare_obs(observation)
        rtg = returns_to_go.to(self.device, torch.float32).reshape(-1).contiguous()
        rew = self._zero_reward if rewards is None else rewards.to(self.device, torch.float32).reshape(-1).contiguous()
        if reset_mask is not None:
            reset_mask = reset_mask.to(self.device, torch.uint8).contiguous()
        if getattr(self, "compat_mamba_repeat", False):
            # one forward per action dim of the env (decision_mamba.py:107: env_act_dim, else the padded action width)
            rep = 1 if self.is_discrete else int(self.spec.act_dim if env_act_dim is None else env_act_dim)
            if rep != self._compat_repeat_now:
                self.engine.set_compat_mode(rep, self.compat_stale_state)
                self._compat_repeat_now = rep
        if images:
            actions, _ = self.engine.step_images(obs, rtg, rew, reset_mask, discrete=self.is_discrete)
        else:
            actions, _ = self.engine.step(obs, rtg, rew, reset_mask, discrete=self.is_discrete, obs_is_embedding=is_emb)
        if self.is_discrete:
            return actions[:, :1].to(torch.int64)
        return actions if env_act_dim is None else actions[:, :env_act_dim]

    # ---- reference single-env surface -------------------------------------------------------------
    @torch.no_grad()
    def predict(self, policy, observation, actions, rewards, returns_to_go, timesteps, state=None,
                episode_start=None, deterministic=True, context_len=5, prompt=None, task_id=None, is_eval=False,
                env_act_dim=None):
        if self.n_envs != 1:
            raise RuntimeError("predict() is the reference's single-env entry; use predict_batch() for n_envs > 1")
        obs_shape = observation.shape[1:]
        states = observation.reshape(1, -1, *obs_shape)
        returns_to_go = returns_to_go.reshape(1, -1, 1)
        timesteps = timesteps.reshape(1, -1)
        if rewards is not None:
            rewards = rewards.reshape(1, -1, 1)
        a1, a2 = self.get_action_pred(policy, states, actions, rewards, returns_to_go, timesteps, None,
                                      deterministic, prompt, is_eval=is_eval, task_id=task_id,
                                      env_act_dim=env_act_dim)
        if self.reset_inf_cache_freq is not None:
            current_step = int(timesteps[0, -1])
            if current_step > 0 and current_step % self.reset_inf_cache_freq == 0:
                self.past_key_values = None  # context is dropped, not re-primed
                if self.reprime_context and observation.dim() == 2:
                    n = min(int(self.eval_context_len), states.shape[1])
                    obs_seq, _ = self._prepare_obs(states[0, -n:])
                    rew_seq = torch.zeros(1, n, device=self.device) if rewards is None else \
                        rewards[:, -n:, 0].to(self.device, torch.float32)
                    self.engine.prefill(obs_seq.view(1, n, -1).contiguous(),
                                        returns_to_go[:, -n:, 0].to(self.device, torch.float32).contiguous(),
                                        rew_seq.contiguous(), want_action=False)
        return a1, a2

    @torch.no_grad()
    def get_action_pred(self, policy, states, actions, rewards, returns_to_go, timesteps, attention_mask,
                        deterministic, prompt, is_eval=False, task_id=None, env_act_dim=None):
        """With the inference cache on, only the last timestep's (state, rtg, reward) reach the encoder
        (online_decision_transformer_model.py:466-470)."""
        obs = states[:, -1]
        rtg = returns_to_go[:, -1].reshape(1)
        rew = None if rewards is None else rewards[:, -1].reshape(1)
        act = self.predict_batch(obs, rtg, rew, None, env_act_dim).clone()
        a = act[0]
        if self.is_discrete:
            a = a[: (1 if env_act_dim is None else env_act_dim)]
        return a, a

    get_action = get_action_pred
