"""Env-sharded multi-GPU rollout: one process per GPU, independent env slices, one all-gather of actions.

The reference's only multi-GPU inference mechanism assigns whole eval tasks to ranks
(`idx % world_size != global_rank -> skip`, src/callbacks/custom_eval_callback.py:385,445) and gathers
pickled result dicts (src/utils/misc.py:159-191).  Here the batch of envs is partitioned into contiguous
slices, weights are replicated, recurrent state never leaves its GPU, and the single exchange per
timestep is an `all_gather` of the [B/N, act_dim] action tensor (RCCL over xGMI when the backend is
"nccl"; "gloo" on CPU for tests).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def dist_env() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def _single_rank_collectives() -> bool:
    """LRAM_DIST_SINGLE_RANK=1: a WORLD_SIZE == 1 job still creates its process group and issues every collective (each
    is the identity then).  Not a speed option: it is how a one-GPU box puts RCCL through this module's code --
    communicator creation bound to the device, all_gather_into_tensor, all_reduce, barrier -- before the first N-GPU run
    (tests/test_gpu_dist_single_rank.py)."""
    return os.environ.get("LRAM_DIST_SINGLE_RANK", "0") == "1"


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _single_rank_collectives())


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for WORLD_SIZE == 1)."""
    rank, world, local_rank = dist_env()
    if (world > 1 or _single_rank_collectives()) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)   # binds the communicator; barrier() need not guess
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous env slice [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_actions(local_actions: torch.Tensor, total: Optional[int] = None) -> torch.Tensor:
    """Concatenate every rank's [B_r, A] action slice in rank order -> [B, A] on every rank."""
    if not _active():
        return local_actions
    world = dist.get_world_size()
    local_actions = local_actions.contiguous()
    if total is None or total % world == 0:
        out = torch.empty((local_actions.shape[0] * world, *local_actions.shape[1:]), dtype=local_actions.dtype,
                          device=local_actions.device)
        dist.all_gather_into_tensor(out, local_actions)
        return out
    # ragged slices: pad to the largest shard
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    padded = torch.zeros((mx, *local_actions.shape[1:]), dtype=local_actions.dtype, device=local_actions.device)
    padded[: local_actions.shape[0]] = local_actions
    out = torch.empty((mx * world, *local_actions.shape[1:]), dtype=local_actions.dtype, device=local_actions.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def barrier():
    if _active():
        dist.barrier()


def count_ranks(device) -> int:
    """Number of ranks that actually take part: an all-reduce of one 1 per rank (1 without a process group)."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def collective_report(sample: torch.Tensor, total: Optional[int] = None, iters: int = 20) -> dict:
    """What the one data-path collective costs on this job, for the first multi-GPU run to explain itself: backend and its
    library version (RCCL reports through torch.cuda.nccl.version()), the all-gather of one action tensor timed over
    `iters` back-to-back calls between device synchronisations (latency-bound: 128 KB per rank at 4096 env slots).  Every rank
    must call it; {} without a process group."""
    if not _active():
        return {}
    import time
    backend = dist.get_backend()
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            version = None
    on_gpu = sample.device.type == "cuda"

    def sync():
        if on_gpu:
            torch.cuda.synchronize(sample.device)

    for _ in range(3):
        all_gather_actions(sample, total)
    sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        all_gather_actions(sample, total)
    sync()
    dt = (time.perf_counter() - t0) / iters
    return {"backend": backend, "library_version": version, "world_size": dist.get_world_size(),
            "all_gather_us": max_over_ranks(dt, sample.device) * 1e6,
            "all_gather_bytes_per_rank": sample.numel() * sample.element_size(), "iters": iters}
