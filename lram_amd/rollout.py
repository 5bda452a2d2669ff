"""Batched multi-env rollout driver (the caller of the hot path) and a synthetic vector env.

Batched counterpart of the reference's single-env loop `custom_evaluate_policy`
(src/callbacks/evaluation.py:130-257): per timestep it feeds (state, rtg, reward-token 0) to the agent,
steps the envs, updates `rtg -= r / reward_scale` (:165), and on `done` resets that env's context
(:238-251).  All bookkeeping is vectorised device tensors; there is no per-step host sync except the one
the caller asks for.  Timing keys follow src/callbacks/custom_eval_callback.py:468-475.

`SyntheticVecEnv` is the batched DummyEnv (src/envs/dummy_env_utils.py:8-35): observations U(-1, 1),
reward 1 every step, episode ends after `ep_len` steps.

`evaluate_policy_batched` keeps the call contract of `custom_evaluate_policy` (evaluation.py:14-271: episode
targets per sub-env, return tuple, `reward_threshold`, per-step `callback`, `persist_context`) for a vector env of
any width, and `eval_log_record` produces the keys `_on_step_logging` records
(src/callbacks/custom_eval_callback.py:439-515, 563-588).
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

# DMControl full observation space offsets used by cheetah-run (src/envs/dmcontrol_utils.py:44-49):
# 'position' (8 dims) starts at 41, 'velocity' (9 dims) at 14.
CHEETAH_RUN_OBS_INDEX = tuple(range(41, 49)) + tuple(range(14, 23))


class SyntheticVecEnv:
    def __init__(self, n_envs: int, obs_dim: int = 10, act_dim: int = 1, ep_len: int = 1000, device="cpu",
                 seed: int = 1234, stagger: bool = True, obs_index: Optional[Sequence[int]] = None,
                 full_dim: Optional[int] = None, image_shape: Optional[Sequence[int]] = None,
                 success_every: Optional[int] = None):
        self.n_envs, self.obs_dim, self.act_dim, self.ep_len = n_envs, obs_dim, act_dim, ep_len
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.image_shape = tuple(image_shape) if image_shape is not None else None
        self.full_dim = full_dim
        self.obs_index = None if obs_index is None else torch.tensor(list(obs_index), device=self.device)
        # env e is `e mod ep_len` steps into its episode at t = 0: resets are staggered over time
        self.t = (torch.arange(n_envs, device=self.device) % ep_len) if stagger else \
            torch.zeros(n_envs, dtype=torch.long, device=self.device)
        # Meta-World-style `is_success` info at episode end (every `success_every`-th episode of an env succeeds)
        self.success_every = success_every
        self.episodes = torch.zeros(n_envs, dtype=torch.long, device=self.device)
        self.last_info: Dict[str, torch.Tensor] = {}

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
        self.episodes = self.episodes + done.long()
        if self.success_every is not None:
            self.last_info = {"is_success": done & (self.episodes % self.success_every == 0)}
        return self._sample(), reward, done


class BatchedRollout:
    """agent: object with predict_batch(obs, rtg, rewards, reset_mask, env_act_dim) (lram_amd.agent.RecurrentAgent)."""

    def __init__(self, agent, env: SyntheticVecEnv, target_return: float, reward_scale: float,
                 env_act_dim: Optional[int] = None, persist_context: bool = False):
        self.agent, self.env = agent, env
        # evaluation.py:213-236: with persist_context the (cached) context survives episode ends -- only the
        # target return and the timestep restart; without it the env's cache is reset (:238-251)
        self.persist_context = bool(persist_context)
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
        self.reset_mask = torch.zeros_like(self.reset_mask) if self.persist_context else done.to(torch.uint8)
        self.obs = obs
        self.last_reward, self.last_done = reward, done
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


def evaluate_policy_batched(agent, env, n_eval_episodes: int = 10, target_return: Optional[float] = None,
                            reward_scale: Optional[float] = None, env_act_dim: Optional[int] = None,
                            callback: Optional[Callable[[dict, dict], None]] = None,
                            reward_threshold: Optional[float] = None, return_episode_rewards: bool = False,
                            task_id: int = 0, max_steps: Optional[int] = None,
                            is_success_buffer: Optional[List[float]] = None):
    """`custom_evaluate_policy` (src/callbacks/evaluation.py:14-271) over a vector env of any width.

    Episodes are divided over the sub-envs as evenly as possible (:94-96) and an env stops contributing once it
    has finished its share (:184,205-212); returns (mean_reward, std_reward, mean_episode_time), or the
    (episode_rewards, episode_lengths, episode_times) lists with `return_episode_rewards` (:266-271).
    `target_return` / `reward_scale` default to the agent's `compute_target_return_val` /
    `get_reward_scale_for_env` (:111-122); `agent.persist_context` selects the cross-episode mode; an env's
    `is_success` info at episode end is appended to `is_success_buffer` (the reference collects it through its
    `_log_success_callback`, custom_eval_callback.py:36-52)."""
    n = env.n_envs
    if target_return is None:
        target_return = agent.compute_target_return_val(env=env, task_id=task_id) * agent.get_reward_scale_for_env(None)
    if reward_scale is None:
        reward_scale = agent.get_reward_scale_for_env(None)
    ro = BatchedRollout(agent, env, target_return, reward_scale, env_act_dim,
                        persist_context=bool(getattr(agent, "persist_context", False)))
    targets = np.array([(n_eval_episodes + i) // n for i in range(n)], dtype=int)
    counts = np.zeros(n, dtype=int)
    cur_r = np.zeros(n)
    cur_l = np.zeros(n, dtype=int)
    start = [time.time()] * n
    episode_rewards: List[float] = []
    episode_lengths: List[int] = []
    episode_times: List[float] = []
    is_success: List[float] = [] if is_success_buffer is None else is_success_buffer
    steps = 0
    while (counts < targets).any():
        ro.step()
        steps += 1
        reward = ro.last_reward.detach().cpu().numpy()
        done = ro.last_done.detach().cpu().numpy().astype(bool)
        info = getattr(env, "last_info", {}) or {}
        succ = info["is_success"].detach().cpu().numpy() if "is_success" in info else None
        cur_r += reward
        cur_l += 1
        for i in range(n):
            if counts[i] < targets[i]:
                if callback is not None:
                    callback(locals(), globals())
                if done[i]:
                    episode_rewards.append(float(cur_r[i]))
                    episode_lengths.append(int(cur_l[i]))
                    episode_times.append(time.time() - start[i])
                    if succ is not None:
                        is_success.append(float(succ[i]))
                    counts[i] += 1
            if done[i]:
                cur_r[i], cur_l[i], start[i] = 0.0, 0, time.time()
        if max_steps is not None and steps >= max_steps:
            break
    # evaluation.py:258-261: the cache is dropped when the evaluation ends
    if hasattr(agent, "inference_params"):
        agent.inference_params.reset()
    mean_reward = float(np.mean(episode_rewards)) if episode_rewards else float("nan")
    std_reward = float(np.std(episode_rewards)) if episode_rewards else float("nan")
    if reward_threshold is not None:
        assert mean_reward > reward_threshold, f"Mean reward below threshold: {mean_reward:.2f} < {reward_threshold:.2f}"
    if return_episode_rewards:
        return episode_rewards, episode_lengths, episode_times
    return mean_reward, std_reward, float(np.mean(episode_times)) if episode_times else float("nan")


def eval_log_record(prefix: str, env_name: str, idx: int, episode_rewards, episode_lengths, episode_times=None,
                    is_success=None, inf_batch: Optional[int] = None, num_envs: int = 1,
                    score_ref: Optional[Sequence[float]] = None, score_type: str = "dns") -> Dict[str, float]:
    """The keys `_on_step_logging` records for one evaluated env (custom_eval_callback.py:446-508).
    `score_ref` = (random, data-or-human) reference returns of that env for the normalised score
    (src/envs/dn_scores.py:484-488, hn_scores.py:129-133); `inf_batch` = env slots advanced per step."""
    env_id = f"{env_name}_{idx}"
    mean_reward, mean_len = float(np.mean(episode_rewards)), float(np.mean(episode_lengths))
    rec = {f"{prefix}/{env_id}/mean_reward": mean_reward, f"{prefix}/{env_id}/mean_ep_length": mean_len}
    if episode_times is not None and len(episode_times) > 0:
        t = float(np.mean(episode_times))
        rec[f"{prefix}/{env_id}/mean_ep_time"] = t
        rec[f"{prefix}/{env_id}/time_per_step"] = t / (mean_len + 1e-8)
        rec[f"{prefix}/{env_id}/steps_per_second"] = mean_len / (t + 1e-8)
        if inf_batch is not None:
            rec[f"{prefix}/{env_id}/total_steps_per_second"] = mean_len * inf_batch / (t + 1e-8)
    if num_envs == 1:
        rec[f"{prefix}/mean_reward"] = mean_reward
        rec[f"{prefix}/mean_ep_length"] = mean_len
    if is_success is not None and len(is_success) > 0:
        sr = float(np.mean(is_success))
        rec[f"{prefix}/{env_id}/success_rate"] = sr
        if num_envs == 1:
            rec[f"{prefix}/success_rate"] = sr
    if score_ref is not None:
        rnd, ref = float(score_ref[0]), float(score_ref[1])
        score = float(np.mean((np.asarray(episode_rewards, dtype=np.float64) - rnd) / (ref - rnd)))
        rec[f"{prefix}/{env_id}/{score_type}"] = score
        if num_envs == 1:
            rec[f"{prefix}/{score_type}"] = score
    return rec
