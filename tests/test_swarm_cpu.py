"""world_size-2 gloo test of the N>1 exchange path used by bench.py (d2slam_amd/swarm.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, F, cap, q):
    sys.path.insert(0, ROOT)
    from d2slam_amd import swarm
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NP = swarm.pool_rows(F, world)
    desc = torch.zeros(NP, cap, 256); cnt = torch.zeros(NP, dtype=torch.int32)
    # current frames of this rank: row r filled with value 1000*rank + r
    for r in range(2 * F):
        desc[r] = 1000 * rank + r; cnt[r] = 10 * rank + r + 1
    gd = torch.zeros(world, F, cap, 256); gc = torch.zeros(world, F, dtype=torch.int32)
    swarm.exchange_blocks(desc, cnt, F, rank, world, gd, gc)
    q.put((rank, desc[:, 0, 0].numpy().copy(), cnt.numpy().copy()))
    dist.destroy_process_group()


def test_exchange_blocks_world2():
    F, cap, world = 3, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, F, cap, q)) for r in range(world)]
    [p.start() for p in ps]
    res = {}
    for _ in range(world):
        r, d, c = q.get(timeout=120)
        res[r] = (d, c)
    [p.join(60) for p in ps]
    for rank in range(world):
        other = 1 - rank
        d, c = res[rank]
        # remote region holds the OTHER rank's left frames (rows 0,2,4 of that rank)
        assert d[3 * F:].tolist() == [1000 * other + 2 * f for f in range(F)]
        assert c[3 * F:].tolist() == [10 * other + 2 * f + 1 for f in range(F)]
        # own rows untouched
        assert d[:2 * F].tolist() == [1000 * rank + r for r in range(2 * F)]


def test_build_pairs_layout():
    from d2slam_amd import swarm
    a, b = swarm.build_pairs(2, 1)
    assert (a, b) == ([0, 0, 2, 2], [1, 4, 3, 5])
    a, b = swarm.build_pairs(2, 3)
    assert len(a) == 2 * 2 + 2 * 2 and swarm.pool_rows(2, 3) == 10
    assert a[4:] == [0, 2, 0, 2] and b[4:] == [6, 7, 8, 9]
