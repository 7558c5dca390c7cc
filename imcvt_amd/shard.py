"""Frame sharding for multi-GPU runs (one process per GPU).  Frames are independent units: rank r of
`world` owns a contiguous block of the global frame list; nothing is exchanged on the data path.  The only
collectives are bookkeeping: a MAX-reduce of the wall time and, if a driver wants it, a gather of stream lengths."""


def frame_range(rank: int, world: int, frames_per_rank: int):
    """Weak scaling: every rank encodes `frames_per_rank` frames; global ids are rank-major."""
    assert 0 <= rank < world and frames_per_rank >= 0
    return range(rank * frames_per_rank, (rank + 1) * frames_per_rank)


def split_frames(n_frames: int, rank: int, world: int):
    """Strong scaling: partition a fixed list of n_frames as evenly as possible (first ranks get the remainder)."""
    q, r = divmod(n_frames, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def max_over_ranks(seconds: float, device=None) -> float:
    """Wall time of the slowest rank (the job's time).  Works on gloo (CPU) and nccl/RCCL (GPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_lengths(lengths, device=None):
    """All ranks' per-frame stream lengths, rank-major (list of lists)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(lengths)]
    t = torch.tensor(list(lengths), dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().tolist() for o in out]
