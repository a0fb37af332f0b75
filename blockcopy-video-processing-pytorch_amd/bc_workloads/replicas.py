"""Clip-parallel replicas: the only multi-GPU form of this path.

A clip is the unit of temporal state (``reset_temporal()`` per clip, reference test_swiftnet.py:181-182) and clips share
nothing but read-only weights, so N GPUs run N independent replicas: clip i of a job goes to rank i mod N, there is no
collective on the data path, and the job's throughput is (all frames) / (slowest rank's time).  ``torch.distributed``
is used only to bracket the clock (barrier + MAX / SUM all-reduce of two host scalars), over a host-side (gloo) group:
nothing on the data path needs RCCL, and the bracket must not depend on device-to-device plumbing."""
from __future__ import annotations

import os
from typing import List

import torch


def dist_env():
    """(rank, world_size, local_rank) from the torchrun environment (1 process when absent)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def clips_for_rank(n_clips: int, rank: int, world: int) -> List[int]:
    """Indices of the job's clips processed by ``rank`` (round-robin)."""
    return [i for i in range(n_clips) if i % world == rank]


def init_clock_group(timeout_s: float = 1800.0):
    """Process group used ONLY to bracket the clock (rendezvous from the torchrun environment, 127.0.0.1)."""
    import datetime

    if not torch.distributed.is_initialized():
        torch.distributed.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))


def barrier(world: int, device=None):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
        if device is not None and torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)


def job_throughput(frames_local: int, elapsed_local: float, world: int, device="cpu"):
    """(frames/s of the whole job, slowest rank's seconds, total frames): SUM of frames over ranks / MAX of time."""
    if world == 1:
        return frames_local / elapsed_local, elapsed_local, frames_local
    if torch.distributed.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    f = torch.tensor([float(frames_local)], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.SUM)
    return float(f.item()) / float(t.item()), float(t.item()), int(f.item())


def gather_scalars(value: float, world: int) -> List[float]:
    """One host scalar per rank, on every rank (clock-bracket group only; used for the per-rank fps of the report)."""
    if world == 1:
        return [float(value)]
    bufs = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    torch.distributed.all_gather(bufs, torch.tensor([float(value)], dtype=torch.float64))
    return [float(b.item()) for b in bufs]
