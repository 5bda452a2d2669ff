"""Batched multi-env rollout driver (the caller of the hot path) and a synthetic vector env.

Batched counterpart of the reference's single-env loop `custom_evaluate_policy`
(src/callbacks/evaluation.py:130-257): per timestep it feeds (state, rtg, reward-token 0) to the agent,
steps the envs, updates `rtg -= r / reward_scale` (:165), and on `done` resets that env's context
(:238-251).  All bookkeeping is vectorised device tensors; there is no per-step host sync except the one
the caller asks for.  Timing keys follow src/callbacks/custom_eval_callback.py:468-475.

`SyntheticVecEnv` is the batched DummyEnv (src/envs/dummy_env_utils.py:8-35): observations U(-1, 1),
reward 1 every step, episode ends after `ep_len` steps.
"""
from __future__ import annotations

import time
from typing import Dict, Optional, Sequence

import torch

# DMControl full observation space offsets used by cheetah-run (src/envs/dmcontrol_utils.py:44-49):
# 'position' (8 dims) starts at 41, 'velocity' (9 dims) at 14.
CHEETAH_RUN_OBS_INDEX = tuple(range(41, 49)) + tuple(range(14, 23))


class SyntheticVecEnv:
    def __init__(self, n_envs: int, obs_dim: int = 10, act_dim: int = 1, ep_len: int = 1000, device="cpu",
                 seed: int = 1234, stagger: bool = True, obs_index: Optional[Sequence[int]] = None,
                 full_dim: Optional[int] = None, image_shape: Optional[Sequence[int]] = None):
        self.n_envs, self.obs_dim, self.act_dim, self.ep_len = n_envs, obs_dim, act_dim, ep_len
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.image_shape = tuple(image_shape) if image_shape is not None else None
        self.full_dim = full_dim
        self.obs_index = None if obs_index is None else torch.tensor(list(obs_index), device=self.device)
        # env e is `e mod ep_len` steps into its episode at t = 0: resets are staggered over time
        self.t = (torch.arange(n_envs, device=self.device) % ep_len) if stagger else \
            torch.zeros(n_envs, dtype=torch.long, device=self.device)

    def _sample(self):
        if self.image_shape is not None:
            return torch.randint(0, 256, (self.n_envs, *self.image_shape), generator=self.gen, device=self.device,
                                 dtype=torch.uint8)
        raw = torch.rand(self.n_envs, self.obs_dim, generator=self.gen, device=self.device) * 2.0 - 1.0
        if self.obs_index is None:
            return raw
        full = torch.zeros(self.n_envs, self.full_dim, device=self.device)
        full[:, self.obs_index] = raw
        return full

    def reset(self):
        return self._sample()

    def step(self, actions):
        self.t = self.t + 1
        done = self.t >= self.ep_len
        self.t = torch.where(done, torch.zeros_like(self.t), self.t)
        reward = torch.ones(self.n_envs, device=self.device)
        return self._sample(), reward, done


class BatchedRollout:
    """agent: object with predict_batch(obs, rtg, rewards, reset_mask, env_act_dim) (lram_amd.agent.RecurrentAgent)."""

    def __init__(self, agent, env: SyntheticVecEnv, target_return: float, reward_scale: float,
                 env_act_dim: Optional[int] = None):
        self.agent, self.env = agent, env
        self.reward_scale = float(reward_scale)
        self.rtg0 = float(target_return) / float(reward_scale)
        self.env_act_dim = env_act_dim
        dev = env.device
        self.obs = env.reset()
        self.rtg = torch.full((env.n_envs,), self.rtg0, device=dev)
        self.reset_mask = torch.ones(env.n_envs, dtype=torch.uint8, device=dev)  # every env starts an episode
        self.timestep = torch.zeros(env.n_envs, dtype=torch.long, device=dev)
        self.ep_return = torch.zeros(env.n_envs, device=dev)
        self.finished_returns = []
        self.finished_lengths = []

    @torch.no_grad()
    def step(self):
        actions = self.agent.predict_batch(self.obs, self.rtg, None, self.reset_mask, self.env_act_dim)
        obs, reward, done = self.env.step(actions)
        self.ep_return += reward
        self.timestep += 1
        if bool(done.any()):
            self.finished_returns.append(self.ep_return[done].clone())
            self.finished_lengths.append(self.timestep[done].clone())
        # evaluation.py:163-169 (not done) / :238-246 (done: fresh target return, timestep 0)
        self.rtg = torch.where(done, torch.full_like(self.rtg, self.rtg0), self.rtg - reward / self.reward_scale)
        self.timestep = torch.where(done, torch.zeros_like(self.timestep), self.timestep)
        self.ep_return = torch.where(done, torch.zeros_like(self.ep_return), self.ep_return)
        self.reset_mask = done.to(torch.uint8)
        self.obs = obs
        return actions

    def run(self, n_steps: int, sync=None) -> Dict[str, float]:
        if sync is not None:
            sync()
        t0 = time.time()
        for _ in range(n_steps):
            self.step()
        if sync is not None:
            sync()
        wall = time.time() - t0
        n = self.env.n_envs
        out = {
            "time_per_step": wall / max(n_steps, 1),
            "steps_per_second": n_steps / max(wall, 1e-12),               # per env, as the reference logs it
            "total_steps_per_second": n_steps * n / max(wall, 1e-12),     # x batch (inf_dummy_batch_size analogue)
            "n_envs": n, "n_steps": n_steps, "wall_s": wall,
        }
        if self.finished_returns:
            out["mean_reward"] = float(torch.cat(self.finished_returns).float().mean())
            out["mean_ep_length"] = float(torch.cat(self.finished_lengths).float().mean())
        return out
