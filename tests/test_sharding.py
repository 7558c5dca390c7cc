"""The N>1 path on CPU: two gloo ranks shard a frame list, agree on the job time (MAX), gather per-frame stream
lengths and — the job's one exchange step — gather the encoded streams themselves to rank 0 (imcvt_amd/shard.py, the same
code bench.py runs over RCCL).  Frames are independent, so nothing else crosses ranks.
The per-rank "encoder" here is the CPU checker, used as test infrastructure to give the shards real payloads."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from imcvt_amd import shard
    from oracle import oracle, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import torch
    mine = shard.frame_range(rank, world, 3)
    lens = [len(oracle.port_encode(synth.syn(40, 24, s), 2)[0]) for s in mine]
    t = shard.max_over_ranks(1.0 + rank)                 # slowest rank defines the job time
    all_lens = shard.gather_lengths(lens)
    # strong-scaled job of 7 frames (4 + 3): encoded streams gathered to rank 0, as bench.py does over RCCL
    own = list(shard.split_frames(7, rank, world))
    streams = [oracle.port_encode(synth.syn(40, 24, s), 1)[0] for s in own]
    outs = [torch.from_numpy(np.frombuffer(b + bytes(5), dtype=np.uint8).copy()) for b in streams]   # buffers longer than the streams, like the device ones
    slens = [len(b) for b in streams]
    got = shard.gather_streams(shard.pack_streams(outs, slens), slens)
    gathered = None
    if rank == 0:
        gathered = [bytes(v.numpy().tobytes()) for ls, packed in got for v in shard.unpack_streams(ls, packed)]
    else:
        assert got is None
    dist.barrier()
    q.put((rank, list(mine), lens, t, all_lens, gathered))
    dist.destroy_process_group()


def test_two_rank_frame_sharding(built):
    from imcvt_amd import shard
    from oracle import oracle, synth
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(timeout=60)
    assert [r[1] for r in res] == [[0, 1, 2], [3, 4, 5]]                      # disjoint, rank-major cover
    assert all(abs(r[3] - 2.0) < 1e-9 for r in res)                            # MAX over ranks
    want = [len(oracle.port_encode(synth.syn(40, 24, s), 2)[0]) for s in range(6)]
    for r in res:
        assert sum(r[4], []) == want                                           # every rank sees all lengths, in frame order
    # rank 0 holds all 7 streams of the strong-scaled job, byte-equal to a single-process run, in frame order
    assert res[0][5] == [oracle.port_encode(synth.syn(40, 24, s), 1)[0] for s in range(7)] and res[1][5] is None
    # strong-scaling partition covers a list exactly once
    for n in (0, 1, 7, 512):
        got = [i for k in range(3) for i in shard.split_frames(n, k, 3)]
        assert got == list(range(n))


def _worker4(rank, world, port, q, n_frames):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from imcvt_amd import shard
    from oracle import oracle, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import torch
    seen = shard.ranks_observed()
    own = list(shard.split_frames(n_frames, rank, world))
    streams = [oracle.port_encode(synth.syn(24 + 8 * (s % 3), 24, s), 3)[0] for s in own]      # (streams of different lengths)
    outs = [torch.from_numpy(np.frombuffer(b + bytes(3), dtype=np.uint8).copy()) for b in streams]
    slens = [len(b) for b in streams]
    got = shard.gather_streams(shard.pack_streams(outs, slens), slens)
    gathered = [[bytes(v.numpy().tobytes()) for v in shard.unpack_streams(ls, packed)] for ls, packed in got] if rank == 0 else None
    assert (got is None) == (rank != 0)
    dist.barrier()
    q.put((rank, seen, own, gathered))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [6, 3])      # 4 ranks: shares 2,2,1,1 — and 1,1,1,0 (a rank with nothing to send)
def test_four_ranks_unequal_shares_and_an_empty_rank(built, n_frames):
    from oracle import oracle, synth
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q, n_frames)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs: p.join(timeout=60)
    assert [r[1] for r in res] == [world] * world                              # the backend itself saw four ranks
    assert sum((r[2] for r in res), []) == list(range(n_frames))               # contiguous blocks, rank-major
    per_rank = res[0][3]
    assert [len(v) for v in per_rank] == [len(r[2]) for r in res]              # one list per rank, empty for the rank without frames
    assert sum(per_rank, []) == [oracle.port_encode(synth.syn(24 + 8 * (s % 3), 24, s), 3)[0] for s in range(n_frames)]
