"""Multi-GPU layout of the hot path: replicas only.

The online phase is one sequential chain over the points of a timepoint, so events of one stream are never
sharded; N GPUs process N independent event streams (samples), one per rank, with no data-path collective.
`torch.distributed` is used by bench.py only for the barrier and the max-over-ranks time."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def stream_seed(base_seed, rank):
    """Every rank clusters its own synthetic stream (same shape, different seed)."""
    return int(base_seed) + int(rank)


def max_over_ranks(value, dist=None, device=None):
    """The job's step time is the slowest rank's.  `dist` is torch.distributed (or None for one process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(points_per_rank, steps, world, seconds):
    """value of bench.py: units all ranks processed / max-over-ranks time."""
    return world * points_per_rank * steps / seconds
