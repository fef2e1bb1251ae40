"""Multi-GPU side of the RX inner path: frames shard embarrassingly (one process per GPU, each
decoding its own frames); the ONLY collective is the sum of the monitor counters
{FRA, BE, FE} -- what tools::Monitor_reduction does across threads in the reference
(/root/reference src/mains/TX_RX_BB/main.cpp:123-125,155-161) -- plus a max over ranks for
timings.  24-byte messages: latency-bound, never in the per-frame path.
Backend: torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist


def _dev_for(device):
    if device is not None:
        return device
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def reduce_counters(ctr: Sequence[int], device=None):
    """Sum of {FRA, BE, FE} over all ranks (identity when not initialised / world 1)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(c) for c in ctr]
    t = torch.tensor([int(c) for c in ctr], dtype=torch.int64, device=_dev_for(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def reduce_max(x: float, device=None) -> float:
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_dev_for(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_frames(n_frames: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of a batch of frames owned by `rank`."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_stream(n_samples: int, rank: int, world: int, halo: int):
    """Contiguous block of a sample STREAM for the matched filter (a5): [lo, hi) plus the `halo`
    = T-1 samples before lo that the rank must also read (overlap-save; rank 0's halo is the
    filter state of the previous call).  -> (lo, hi, halo_lo)"""
    lo, hi = shard_frames(n_samples, rank, world)
    return lo, hi, max(0, lo - halo)


def all_done(local_fe_total: int, max_fe: int, device=None) -> bool:
    """Stop criterion of the Monte-Carlo loop: reduced FE >= max_fe (Monitor_reduction::is_done_all)."""
    return reduce_counters([0, 0, local_fe_total], device)[2] >= max_fe
