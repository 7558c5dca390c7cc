"""Frame sharding for multi-GPU runs (one process per GPU).  Frames are independent units: rank r of
`world` owns a contiguous block of the global frame list (the reference's seam is its serial file loop,
src/main.c:162-211); nothing is exchanged while encoding.  Cross-rank traffic is the bookkeeping (a MAX-reduce of the
wall time) and the one real exchange step the job has: the gather of the encoded streams to rank 0 — an all-gather of
the per-frame lengths, then one point-to-point send per peer of its packed streams (variable size; over RCCL each
peer uses its own xGMI link to the root)."""


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
    """All ranks' per-frame stream lengths, rank-major (list of lists).  Ranks may own different numbers of frames."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(lengths)]
    dev = device if device is not None else "cpu"
    world = dist.get_world_size()
    lengths = list(lengths)
    cnt = torch.tensor([len(lengths)], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cnts = [int(c.item()) for c in cnts]
    m = max(cnts + [1])
    t = torch.tensor(lengths + [0] * (m - len(lengths)), dtype=torch.int64, device=dev)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [o.cpu().tolist()[:cnts[r]] for r, o in enumerate(out)]


def pack_streams(outs, lengths):
    """One contiguous uint8 tensor holding this rank's streams back to back (outs[i][:lengths[i]], frame order)."""
    import torch
    if not outs:
        return torch.empty(0, dtype=torch.uint8)
    return torch.cat([o[:n] for o, n in zip(outs, lengths)])


def gather_streams(packed, lengths, device=None):
    """The job's exchange step: every rank's packed streams arrive on rank 0.

    packed: this rank's streams back to back (pack_streams), on `device` for RCCL or on the CPU for gloo.
    Returns on rank 0 a list over ranks of (lengths, packed tensor) — rank 0's own entry is its input — and None elsewhere."""
    import torch
    import torch.distributed as dist
    lengths = list(lengths)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [(lengths, packed)]
    all_lens = gather_lengths(lengths, device)
    rank, world = dist.get_rank(), dist.get_world_size()
    # every transfer of the step is posted at once (on RCCL: one group, each peer on its own xGMI link to the root); a rank without
    # frames sends nothing and nothing is posted for it
    if rank != 0:
        if packed.numel():
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed, 0)]):
                w.wait()
        return None
    res, ops = [(lengths, packed)], []
    for r in range(1, world):
        buf = torch.empty(sum(all_lens[r]), dtype=torch.uint8, device=packed.device)
        if buf.numel():
            ops.append(dist.P2POp(dist.irecv, buf, r))
        res.append((all_lens[r], buf))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return res


def ranks_observed(device=None) -> int:
    """How many ranks the collective backend itself sees: an all-reduce (SUM) of ones.  bench.py prints it so that an N-GPU line
    certifies that N ranks took part over RCCL (1 without a process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def unpack_streams(lengths, packed):
    """Inverse of pack_streams: list of 1-D uint8 views."""
    out, pos = [], 0
    for n in lengths:
        out.append(packed[pos:pos + n])
        pos += n
    return out
