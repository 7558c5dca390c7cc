"""The N>1 path on CPU: two gloo ranks shard a frame list, agree on the job time (MAX) and gather per-frame
stream lengths — the only cross-rank traffic the multi-GPU bench has (frames are independent).
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
    mine = shard.frame_range(rank, world, 3)
    lens = [len(oracle.port_encode(synth.syn(40, 24, s), 2)[0]) for s in mine]
    t = shard.max_over_ranks(1.0 + rank)                 # slowest rank defines the job time
    all_lens = shard.gather_lengths(lens)
    dist.barrier()
    q.put((rank, list(mine), lens, t, all_lens))
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
    # strong-scaling partition covers a list exactly once
    for n in (0, 1, 7, 512):
        got = [i for k in range(3) for i in shard.split_frames(n, k, 3)]
        assert got == list(range(n))
