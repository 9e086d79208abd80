"""One RGB-D stream per GPU: the only multi-GPU mode of this path (SURVEY.md §8e — "replicas only").

KinectFusion shards by stream, not within a stream: every rank owns a full pipeline and a full TSDF volume and
never exchanges per-frame data.  The collective (RCCL on GPUs, gloo in the CPU tests) is used for exactly two
things: a start/stop barrier so that all ranks run concurrently, and one all-reduce of a small stats vector at
the end (frames, wall seconds, ATE sums, untracked frames).  Aggregate throughput = sum(frames) / max(seconds).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

# 4 distinct trajectories, reused x2 on an 8-GPU node (BASELINE.json configs[4])
N_TRAJECTORIES = 4


def stream_variant(rank: int) -> int:
    return rank % N_TRAJECTORIES


@dataclass
class RankStats:
    frames: int
    seconds: float
    ate_sum: float = 0.0
    ate_max: float = 0.0
    untracked: int = 0
    failed: int = 0       # 1 = this rank hit an error: the job reports no result and every rank exits non-zero


@dataclass
class JobStats:
    world: int
    frames: int
    seconds: float        # max over ranks
    fps: float            # whole-job aggregate
    ate_mean: float
    ate_max: float
    untracked: int
    failed: int = 0       # ranks that reported an error


def env_rank():
    """(world, rank, local_rank) from the torch.distributed.run environment; (1, 0, 0) when launched plainly."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")))


def aggregate(stats: RankStats, dist=None, device="cpu") -> JobStats:
    """All-reduce the per-rank stats.  `dist` is torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        n = max(stats.frames, 1)
        return JobStats(1, stats.frames, stats.seconds, stats.frames / stats.seconds if stats.seconds > 0 else 0.0,
                        stats.ate_sum / n, stats.ate_max, stats.untracked, stats.failed)
    import torch
    sums = torch.tensor([float(stats.frames), stats.ate_sum, float(stats.untracked), float(stats.failed)], dtype=torch.float64, device=device)
    maxs = torch.tensor([stats.seconds, stats.ate_max], dtype=torch.float64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
    frames = int(round(sums[0].item()))
    seconds = float(maxs[0].item())
    return JobStats(dist.get_world_size(), frames, seconds, frames / seconds if seconds > 0 else 0.0,
                    float(sums[1].item()) / max(frames, 1), float(maxs[1].item()), int(round(sums[2].item())),
                    int(round(sums[3].item())))


def all_ok(ok: bool, dist=None, device="cpu") -> bool:
    """True when every rank says so (one all-reduce; a rank that failed must still call this, so that the others learn
    of it here instead of waiting in the next barrier until the watchdog kills the job)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    import torch
    t = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item() == 0.0


def ranks_seen(dist=None, device="cpu") -> int:
    """How many ranks the collective library really connected: an all-reduce (sum) of ones over the process group."""
    if dist is None or not dist.is_initialized():
        return 1
    import torch
    t = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(t.item()))


def gather_rows(row, dist=None, device="cpu"):
    """Every rank's small numeric row, in rank order, on every rank: one all-reduce (sum) of a world x len(row) matrix in which a rank
    fills its own line - the collective the bench already relies on, not a pickling all_gather_object."""
    row = [float(v) for v in row]
    if dist is None or not dist.is_initialized():
        return [row]
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.zeros((world, len(row)), dtype=torch.float64, device=device)
    t[rank] = torch.tensor(row, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().tolist()
