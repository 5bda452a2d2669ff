"""Env sharding over ranks (gloo, world_size 2, CPU) and the batched rollout bookkeeping."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lram_amd import dist as ldist
from lram_amd import init_state_dict, preset
from lram_amd.rollout import CHEETAH_RUN_OBS_INDEX, BatchedRollout, SyntheticVecEnv
from tests.helpers import make_inputs


def test_shard_bounds_cover_exactly():
    for total in (1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            b = [ldist.shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w, _ = ldist.init_distributed("gloo")
    from oracle.dt_ref import OraclePolicy
    spec = preset("xlstm_tiny")
    sd = init_state_dict(spec, seed=0)
    lo, hi = ldist.shard_bounds(total, r, w)
    pol = OraclePolicy(spec, sd)      # stands in for the per-rank engine: same sharding / gather code path
    outs = []
    for obs, rtg, rew, mask in make_inputs(spec, total, 3, seed=9):
        a = pol.step(obs[lo:hi], rtg[lo:hi], rew[lo:hi], mask[lo:hi])
        outs.append(ldist.all_gather_actions(a, total))
    ldist.barrier()
    mx = ldist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    if rank == 0:
        # plain nested lists: a tensor on an mp.Queue travels as a file descriptor that the receiver can only open
        # while the sender is still alive (ConnectionResetError otherwise)
        q.put((torch.stack(outs).tolist(), mx))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 7])
def test_env_sharded_rollout_equals_single_process(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:   # a free port chosen by the OS (fixed ports collide when suites run side by side)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, mx = q.get(timeout=120)
    gathered = torch.tensor(gathered)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from oracle.dt_ref import OraclePolicy
    spec = preset("xlstm_tiny")
    pol = OraclePolicy(spec, init_state_dict(spec, seed=0))
    ref = torch.stack([pol.step(*x) for x in make_inputs(spec, total, 3, seed=9)])
    assert torch.equal(gathered, ref)  # envs are independent: sharding changes nothing
    assert mx == 2.0


from tests.stub_engine import StubEngine as _StubEngine  # noqa: E402


def _bench_worker(rank, world, port, argv, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import bench
    engines = []

    def factory(spec, batch, device):
        engines.append(_StubEngine(spec, batch, device))
        return engines[-1]

    out = bench.main(argv, engine_factory=factory)
    q.put((rank, {k: v for k, v in out.items() if k != "last_actions"}, out["last_actions"].tolist(), engines[0].batch,
           engines[0].steps))


@pytest.mark.parametrize("argv,total", [(["--batch", "6"], 12), (["--global-batch", "7"], 7)])
def test_bench_multi_rank_code_path_on_gloo(argv, total):
    """bench.py's own N > 1 path (shard -> step -> all_gather_actions -> barrier -> max over ranks), weak scaling and
    the ragged strong-scaling split, executed on CPU over gloo with a stand-in engine: the first 8-GPU run is then not
    the first execution of that code."""
    import socket
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    full = ["--gpus", "2", "--steps", "3", "--warmup", "2", "--config", "xlstm_tiny"] + argv
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, full, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(2)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, line0, act0, b0, n0), (_, line1, act1, b1, n1) = got
    assert b0 + b1 == total and abs(b0 - b1) <= 1
    assert n0 == n1 == 2 + 3                                   # W warm-up + exactly K timed steps per rank
    assert line0["n_gpus"] == 2 and line0["steps"] == 3 and line0["warmup"] == 2
    assert line0["scaling"] == ("strong" if "--global-batch" in argv else "weak")
    assert line0["config"]["global_batch"] == total
    assert line0["value"] == line1["value"] > 0               # max over ranks: both ranks report the same wall
    assert abs(line0["value"] - total * 3 / (line0["ms_per_step"] * 3e-3)) < 1e-6 * line0["value"]
    a0, a1 = torch.tensor(act0), torch.tensor(act1)
    assert a0.shape == (total, 4) and torch.equal(a0, a1)     # every rank holds the full gathered action tensor


def test_bench_starts_its_own_ranks_from_a_plain_process(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment must run TWO ranks (round-2 review: it ran one and said
    so only in a warning).  The parent starts the children itself and relays rank 0's line; n_gpus and ranks_seen (an
    all-reduce of ones) both say 2.  Stand-in engine over gloo, through the same entry point the driver uses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    factory = os.path.join(root, "tests", "stub_engine.py") + ":factory"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--config",
           "xlstm_tiny", "--batch", "6", "--engine-factory", factory]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # ONE JSON line, rank 0's, relayed by the parent
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["config"]["global_batch"] == 12 and line["config"]["batch_per_gpu"] == 6
    assert torch.tensor(line["last_actions"]).shape == (12, 4)
    col = line["collective"]    # the one data-path collective explains itself: backend, version, measured latency
    assert col["backend"] == "gloo" and col["world_size"] == 2 and col["all_gather_us"] > 0 and col["iters"] == 20
    # same through bench.main from a plain process (no WORLD_SIZE): the parsed line comes back
    import bench
    saved = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        out = bench.main(cmd[2:])
    finally:
        os.environ.update(saved)
    assert out["n_gpus"] == 2 == out["ranks_seen"]


def test_bench_single_rank_collectives_mode_on_gloo():
    """LRAM_DIST_SINGLE_RANK=1 (how a one-GPU box puts RCCL through the N > 1 path: tests/test_gpu_dist_single_rank.py): a
    WORLD_SIZE == 1 job creates its process group and issues every collective; here the CPU twin over gloo with the stand-in engine."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LRAM_DIST_SINGLE_RANK="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "xlstm_tiny",
           "--batch", "6", "--engine-factory", os.path.join(root, "tests", "stub_engine.py") + ":factory"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1
    assert line["collective"]["backend"] == "gloo" and line["collective"]["world_size"] == 1
    assert torch.tensor(line["last_actions"]).shape == (6, 4)
    # without the switch a single rank creates no process group and reports no collective
    env.pop("LRAM_DIST_SINGLE_RANK")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "collective" not in json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "xlstm_tiny",
                          "--engine-factory", os.path.join(root, "tests", "stub_engine.py") + ":factory"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies takes the job down with a non-zero exit instead of leaving the others in the rendezvous."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "xlstm_tiny",
                          "--steps", "2", "--warmup", "1", "--batch", "4",
                          "--engine-factory", os.path.join(root, "tests", "stub_engine.py") + ":factory_rank1_dies"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "job aborted" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


class _FakeAgent:
    def __init__(self):
        self.calls = []

    def predict_batch(self, obs, rtg, rewards, reset_mask, env_act_dim):
        self.calls.append((obs.clone(), rtg.clone(), reset_mask.clone()))
        return torch.zeros(obs.shape[0], env_act_dim)


def test_rollout_bookkeeping_and_staggered_resets():
    env = SyntheticVecEnv(6, obs_dim=17, act_dim=6, ep_len=4, stagger=True, obs_index=CHEETAH_RUN_OBS_INDEX, full_dim=204)
    agent = _FakeAgent()
    ro = BatchedRollout(agent, env, target_return=450.0, reward_scale=100.0, env_act_dim=6)
    stats = ro.run(9)
    obs0, rtg0, m0 = agent.calls[0]
    assert obs0.shape == (6, 204) and bool(m0.all()) and torch.allclose(rtg0, torch.full((6,), 4.5))
    nz = obs0[0].nonzero().flatten().tolist()
    assert set(nz) <= set(CHEETAH_RUN_OBS_INDEX) and len(nz) >= 15
    # env e is e % 4 steps into its episode at t=0 -> done after 4 - e%4 steps; rtg drops 1/100 per step
    for t in range(1, 9):
        _, rtg, mask = agent.calls[t]
        for e in range(6):
            age = (e % 4 + t) % 4
            assert int(mask[e]) == int(age == 0), (t, e)
            steps_since_reset = age if t >= 4 - e % 4 else t
            assert abs(float(rtg[e]) - (4.5 - 0.01 * steps_since_reset)) < 1e-6, (t, e)
    assert stats["n_envs"] == 6 and stats["total_steps_per_second"] > 0 and "mean_ep_length" in stats


class _FakeEvalAgent(_FakeAgent):
    persist_context = False

    def __init__(self):
        super().__init__()
        self.resets = 0
        self.inference_params = self

    def reset(self):
        self.resets += 1

    def compute_target_return_val(self, env=None, task_id=0):
        return 4.5

    def get_reward_scale_for_env(self, envid=None):
        return 100.0


def test_evaluate_policy_batched_follows_the_reference_contract():
    """Episode shares per sub-env, return tuples, reward threshold, success buffer (evaluation.py:94-96,184-212,262-271)."""
    from lram_amd.rollout import eval_log_record, evaluate_policy_batched
    env = SyntheticVecEnv(4, obs_dim=5, act_dim=2, ep_len=3, stagger=False, success_every=2)
    agent = _FakeEvalAgent()
    succ = []
    rewards, lengths, times = evaluate_policy_batched(agent, env, n_eval_episodes=10, env_act_dim=2,
                                                      return_episode_rewards=True, is_success_buffer=succ)
    # targets (10 + i) // 4 = 2, 2, 3, 3 episodes of 3 steps with reward 1
    assert len(rewards) == 10 and rewards == [3.0] * 10 and lengths == [3] * 10 and len(times) == 10
    assert len(agent.calls) == 9 and agent.resets == 1
    assert torch.allclose(agent.calls[0][1], torch.full((4,), 4.5))   # compute_target_return_val is the rtg token value
    assert succ == [0.0] * 4 + [1.0] * 4 + [0.0] * 2                             # every 2nd episode succeeds
    env2 = SyntheticVecEnv(3, obs_dim=5, act_dim=2, ep_len=2, stagger=True)
    mean_r, std_r, mean_t = evaluate_policy_batched(_FakeEvalAgent(), env2, n_eval_episodes=5, env_act_dim=2)
    assert 1.0 <= mean_r <= 2.0 and std_r >= 0.0 and mean_t >= 0.0
    with pytest.raises(AssertionError):
        evaluate_policy_batched(_FakeEvalAgent(), SyntheticVecEnv(2, ep_len=2, stagger=False), n_eval_episodes=2,
                                env_act_dim=1, reward_threshold=5.0)
    rec = eval_log_record("eval", "cheetah-run", 3, rewards, lengths, times, is_success=succ, inf_batch=4,
                          score_ref=(1.0, 5.0), score_type="dns")
    assert rec["eval/cheetah-run_3/mean_reward"] == 3.0 and rec["eval/mean_ep_length"] == 3.0
    assert abs(rec["eval/cheetah-run_3/dns"] - 0.5) < 1e-12 and abs(rec["eval/success_rate"] - 0.4) < 1e-12
    assert abs(rec["eval/cheetah-run_3/total_steps_per_second"] - 4 * rec["eval/cheetah-run_3/steps_per_second"]) < 1e-6


def test_persist_context_keeps_the_cache_across_episodes():
    """evaluation.py:213-251: with persist_context only the target return restarts at an episode end."""
    env = SyntheticVecEnv(2, obs_dim=3, act_dim=1, ep_len=2, stagger=False)
    agent = _FakeAgent()
    ro = BatchedRollout(agent, env, target_return=10.0, reward_scale=10.0, env_act_dim=1, persist_context=True)
    ro.run(5)
    assert bool(agent.calls[0][2].all())                                  # the very first step starts from a clean cache
    assert all(int(c[2].sum()) == 0 for c in agent.calls[1:])             # ... and nothing resets it afterwards
    assert abs(float(agent.calls[2][1][0]) - 1.0) < 1e-6                  # rtg restarts after the 2-step episode


def test_agent_predict_follows_the_reference_trace():
    """RecurrentAgent.predict / get_action_pred (host logic, engine replaced by a recorder) against the trace of the
    reference's own predict -> pad_inputs -> get_action_pred chain (tests/golden `agent_predict_trace`, generated by
    executing those methods): what reaches the network each step -- zero-padded, normalised last observation,
    return-to-go, reward token -- the action slice handed back, and when the cache is dropped
    (`reset_inf_cache_freq`)."""
    import dataclasses
    import json
    from types import SimpleNamespace
    from lram_amd.agent import RecurrentAgent, _InferenceParams
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["agent_predict_trace"]
    canned = torch.tensor(vec["canned_action_preds"])

    class Recorder:
        def __init__(self):
            self.calls, self.resets, self.device = [], [], torch.device("cpu")

        def step(self, obs, rtg, rew, reset_mask, discrete=False, obs_is_embedding=False):
            self.calls.append((obs.clone(), float(rtg[0]), float(rew[0])))
            return canned[len(self.calls) - 1].view(1, -1).clone(), None

        def reset(self, mask=None):
            self.resets.append(len(self.calls))

    agent = object.__new__(RecurrentAgent)
    agent.spec = dataclasses.replace(preset("xlstm_16m"), reset_inf_cache_freq=vec["reset_inf_cache_freq"])
    agent.engine, agent.device, agent.n_envs, agent.is_discrete = Recorder(), torch.device("cpu"), 1, False
    agent.policy, agent.image_encoder, agent.has_image_encoder = agent, None, False
    agent.state_mean, agent.state_std = torch.tensor(vec["state_mean"]), torch.tensor(vec["state_std"])
    agent.eval_context_len, agent.reset_inf_cache_freq = 1, vec["reset_inf_cache_freq"]
    agent.reprime_context, agent._zero_reward = False, torch.zeros(1)
    agent.inference_params = _InferenceParams(agent)
    obs_all, env_r, scale = torch.tensor(vec["obs"]), torch.tensor(vec["env_rewards"]), vec["reward_scale"]
    A = vec["env_act_dim"]
    states, actions, rewards = obs_all[:1].clone(), torch.zeros((0, A)), torch.zeros(0)
    rtg, ts = torch.tensor(vec["target_return0"]).reshape(1, 1), torch.tensor(0).reshape(1, 1)
    for t, saw in enumerate(vec["policy_saw"]):
        actions = torch.cat([actions, torch.zeros((1, A))])
        rewards = torch.cat([rewards, torch.zeros(1)])
        n_resets_before = len(agent.engine.resets)
        a, _ = agent.predict(agent.policy, states, actions, rewards, rtg, ts, deterministic=True, context_len=1,
                             is_eval=True, env_act_dim=A)
        obs_seen, rtg_seen, rew_seen = agent.engine.calls[t]
        assert torch.allclose(obs_seen[0], torch.tensor(saw["state_last"]), rtol=0, atol=1e-6), t
        assert abs(rtg_seen - saw["rtg_last"]) < 1e-6 and rew_seen == saw["reward_last"] == 0.0, t
        assert torch.allclose(a, torch.tensor(vec["returned_actions"][t])), t
        # the reference drops the cache after the step: the *next* call sees past_key_values None
        dropped_now = len(agent.engine.resets) > n_resets_before
        if t + 1 < len(vec["policy_saw"]):
            assert dropped_now == vec["policy_saw"][t + 1]["cache_is_none"], t
        actions[-1] = a
        rewards[-1] = env_r[t] / scale
        states = torch.cat([states, obs_all[t + 1: t + 2]])
        rtg = torch.cat([rtg, (rtg[0, -1] - env_r[t] / scale).reshape(1, 1)], dim=1)
        ts = torch.cat([ts, torch.full((1, 1), t + 1)], dim=1)


def test_batched_evaluator_follows_the_executed_reference_loop():
    """`evaluate_policy_trace` = custom_evaluate_policy (src/callbacks/evaluation.py:14-271) executed over three scripted
    episodes of one env (make_golden_from_reference.py).  The batched driver, given the same env as a 1-wide vector env
    and a recording agent, hands the agent the same observation and return-to-go at every step, resets the cache exactly
    where the reference drops it, and returns the same episode returns and lengths."""
    import json
    from lram_amd.rollout import evaluate_policy_batched
    v = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))["evaluate_policy_trace"]
    obs_all, rew_all = torch.tensor(v["obs"]), torch.tensor(v["rewards"])
    ends = set(int(x) for x in torch.tensor(v["episode_lengths_scripted"]).cumsum(0) - 1)

    class ScriptedEnv:
        n_envs, device, last_info = 1, torch.device("cpu"), {}

        def __init__(self):
            self.t = 0

        def reset(self):
            return obs_all[:1].clone()

        def step(self, actions):
            r, done = rew_all[self.t: self.t + 1].clone(), torch.tensor([self.t in ends])
            self.t += 1
            return obs_all[self.t: self.t + 1].clone(), r, done

    calls, final_resets = [], []

    class Agent:
        persist_context = False
        inference_params = type("IP", (), {"reset": staticmethod(lambda: final_resets.append(len(calls)))})()

        def compute_target_return_val(self, env=None, task_id=0):
            return v["target_return"] / v["reward_scale"]

        def get_reward_scale_for_env(self, envid=None):
            return v["reward_scale"]

        def predict_batch(self, obs, rtg, rewards, reset_mask, env_act_dim):
            calls.append((obs.clone(), float(rtg[0]), int(reset_mask[0]), env_act_dim))
            return torch.zeros(1, env_act_dim)

    rewards, lengths, _ = evaluate_policy_batched(Agent(), ScriptedEnv(), n_eval_episodes=3, env_act_dim=v["act_dim"],
                                                  return_episode_rewards=True)
    ref_calls = v["predict_calls"]
    assert len(calls) == len(ref_calls) == 9
    drops_before = {i for k, i in v["cache_drops"] if k == "past_key_values=None"}
    for k, ((obs, rtg, mask, ead), ref) in enumerate(zip(calls, ref_calls)):
        assert torch.allclose(obs[0], torch.tensor(ref["obs_last"]), atol=0), k
        assert abs(rtg - ref["rtg_last"]) < 1e-5, (k, rtg, ref["rtg_last"])
        assert mask == int(k in drops_before), k          # cache reset <=> the reference dropped it before this call
        assert ead == ref["env_act_dim"] and ref["reward_last"] == 0.0      # reward token 0
    assert lengths == v["episode_lengths"]
    assert all(abs(a - b) < 1e-4 for a, b in zip(rewards, v["episode_rewards"]))
    assert final_resets == [9] and 9 in drops_before     # and once more when the evaluation ends (evaluation.py:258-261)
