"""CPU, world_size 2 over gloo: the -M sharding rule and the timing reductions bench.py uses for N > 1."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from emsar_amd import dist as D


def test_shard_partitions_samples():
    for n in (0, 1, 7, 8, 20):
        for world in (1, 2, 4, 8):
            parts = [D.shard(n, r, world) for r in range(world)]
            assert sorted(x for p in parts for x in p) == list(range(n))          # every sample exactly once
            assert all(all(i % world == r for i in p) for r, p in enumerate(parts))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    g = D.Group("gloo")
    g.barrier()
    mine = D.shard(5, g.rank, g.world)
    # each rank "solves" its own samples; the job's time is the slowest rank's, its work the sum
    t = 1.0 + g.rank
    mx = g.max([t, float(len(mine))])
    sm = g.sum([float(len(mine))])
    g.close()
    q.put((rank, mine, mx, sm))


def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for _, _, mx, sm in res:
        assert mx == [2.0, 3.0] and sm == [5.0]
