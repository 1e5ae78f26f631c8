"""world_size-2 gloo test of the N>1 exchange path used by bench.py (d2slam_amd/swarm.py): block layout, the single all-gather,
and the pair list's addressing of remote descriptors inside the gathered buffer."""
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


def _pack_host(desc, kps, scores, n, gdesc, cap, G):
    """Host restatement of d2fe_pack_blocks_device (csrc/swarm.hip) for the CPU test."""
    from d2slam_amd import swarm
    BLK = swarm.block_words(cap, G)
    b = np.zeros(BLK, np.float32)
    o = lambda f: swarm.block_field_offset(cap, G, f)
    b[o("desc"):o("desc") + n * 256] = desc[:n].reshape(-1)
    b[o("kps"):o("kps") + n * 2] = kps[:n].reshape(-1)
    b[o("scores"):o("scores") + n] = scores[:n]
    b[o("netvlad"):o("netvlad") + G] = gdesc
    b.view(np.int32)[o("n")] = n
    return b


def _worker(rank, world, port, F, cap, G, q):
    sys.path.insert(0, ROOT)
    from d2slam_amd import swarm
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    BLK = swarm.block_words(cap, G)
    rng = np.random.RandomState(100 + rank)
    blocks = np.zeros((F, BLK), np.float32)
    for f in range(F):
        n = 1 + (rank * 3 + f) % cap
        blocks[f] = _pack_host(rng.randn(cap, 256).astype(np.float32) + 1000 * rank + f, rng.rand(cap, 2).astype(np.float32),
                               rng.rand(cap).astype(np.float32), n, np.full(G, 10 * rank + f, np.float32), cap, G)
    # the pool: local descriptor rows, then the gathered blocks (as bench.py lays it out)
    pool = torch.zeros(3 * F * cap * 256 + world * F * BLK)
    gath = pool[3 * F * cap * 256:].view(world, F, BLK)
    swarm.all_gather_blocks(gath, torch.from_numpy(blocks))
    pl = swarm.PairList(F, cap, world, rank, BLK)
    rows = pool.view(-1, 256)
    rem = [(int(pl.b_off[pl.n_local + i]), int(pl.remote_block[i])) for i in range(pl.n_remote)]
    first_rows = [float(rows[off][0]) for off, _ in rem]
    n_field = [int(gath.view(torch.int32).view(world * F, BLK)[blk, swarm.block_field_offset(cap, G, "n")]) for _, blk in rem]
    g_field = [float(gath.view(world * F, BLK)[blk, swarm.block_field_offset(cap, G, "netvlad")]) for _, blk in rem]
    q.put((rank, blocks[:, :4].copy(), first_rows, n_field, g_field, pl.n_local, pl.n_remote))
    dist.destroy_process_group()


def test_exchange_blocks_world2():
    F, cap, G, world = 3, 4, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, F, cap, G, q)) for r in range(world)]
    [p.start() for p in ps]
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    [p.join(60) for p in ps]
    for rank in range(world):
        other = 1 - rank
        blocks_other = res[other][0]
        _, first_rows, n_field, g_field, n_local, n_remote = res[rank]
        assert n_local == 2 * F and n_remote == (world - 1) * F
        # remote pair f addresses the first descriptor row of the OTHER rank's block f, in place inside the gathered buffer
        assert first_rows == [float(blocks_other[f, 0]) for f in range(F)]
        assert n_field == [1 + (other * 3 + f) % cap for f in range(F)]
        assert g_field == [float(10 * other + f) for f in range(F)]


def test_pair_list_layout():
    from d2slam_amd import swarm
    pl = swarm.PairList(2, 10, 1, 0, 0)
    assert (pl.a_off, pl.b_off) == ([0, 0, 10, 10], [20, 40, 30, 50]) and pl.n_remote == 0
    assert pl.a_cnt_row == [0, 0, 1, 1] and pl.b_cnt_row == [2, 4, 3, 5]
    BLK = swarm.block_words(10, 16)
    assert BLK % 256 == 0 and BLK >= 10 * 259 + 16 + 1
    pl = swarm.PairList(2, 10, 3, 1, BLK)
    assert pl.n_local == 4 and pl.n_remote == 4 and pl.remote_block == [0, 1, 4, 5] and pl.remote_q_frame == [0, 1, 0, 1]
    assert pl.b_off[4:] == [60 + b * (BLK // 256) for b in (0, 1, 4, 5)]


def test_remote_pair_layout_of_the_exchange_behind_the_pipe():
    """swarm.remote_pair_layout (what PipeExchange matches per submit): the same problems, in the same order, as PairList's remote part -- local left frame f against
    frame f of every other rank, rank-major -- with b-side rows relative to the gathered buffer instead of the device-API pool; loopback adds the rank's own blocks."""
    from d2slam_amd import swarm
    F, cap, G = 3, 50, 512
    BLK = swarm.block_words(cap, G)
    for world, rank in ((2, 0), (2, 1), (4, 2), (8, 7)):
        a, b, qf, rb = swarm.remote_pair_layout(world, rank, F, cap, BLK)
        pl = swarm.PairList(F, cap, world, rank, BLK)
        assert len(a) == pl.n_remote == (world - 1) * F
        assert a == pl.a_off[pl.n_local:] and qf == pl.remote_q_frame and rb == pl.remote_block
        assert [x + 3 * F * cap for x in b] == pl.b_off[pl.n_local:]
        assert all(r // F != rank for r in rb) and all(0 <= x < world * F * (BLK // 256) for x in b)
    a, b, qf, rb = swarm.remote_pair_layout(1, 0, F, cap, BLK, loopback=True)
    assert qf == [0, 1, 2] and rb == [0, 1, 2] and b == [f * (BLK // 256) for f in range(F)]
    assert swarm.remote_pair_layout(1, 0, F, cap, BLK) == ([], [], [], [])


def test_block_layout_matches_the_abi():
    """swarm.block_words / block_field_offset (pure Python, used by the CPU tests) == the C ABI's (d2fe_block_words / _field_offset)."""
    from d2slam_amd import api, swarm
    for cap, G in [(200, 4096), (100, 1024), (4, 8), (150, 0)]:
        assert api.block_words(cap, G) == swarm.block_words(cap, G)
        for f in ("desc", "kps", "scores", "netvlad", "n"):
            assert api.block_field_offset(cap, G, f) == swarm.block_field_offset(cap, G, f)


def _quad_worker(rank, world, port, q):
    """QuadSwarm's pair layout and exchange on CPU tensors (world 2, gloo): block (r, view, q) addressing inside the gathered buffer."""
    sys.path.insert(0, ROOT)
    from d2slam_amd import swarm
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Chain:      # the fields QuadSwarm reads of a QuadcamChain
        Q, NI, cap = 2, 8, 4
    G = 8
    qs = swarm.QuadSwarm(Chain, torch, torch.device("cpu"), world, rank, G, 0.5)
    BLK = qs.BLK
    blocks = torch.zeros((Chain.NI, BLK))
    for i in range(Chain.NI):
        blocks[i, 0] = 1000 * rank + i                       # first descriptor word identifies (rank, view*Q + q)
        blocks[i].view(torch.int32)[qs.n_off] = 1 + i % Chain.cap
    swarm.all_gather_blocks(qs.gath, blocks)
    rows = qs.gath.view(-1, 256)
    first = [float(rows[int(o)][0]) for o in qs.b_off]
    ncnt = [int(x) for x in qs.gath_i32[qs.rem_blk, qs.n_off]]
    q.put((rank, qs.jobs, qs.a_off.tolist(), first, ncnt, qs.job_loc.tolist(), qs.job_rem.tolist()))
    dist.destroy_process_group()


def test_quad_swarm_pairs_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_quad_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    [p.join(60) for p in ps]
    Q, cap = 2, 4
    for rank in range(world):
        other = 1 - rank
        jobs, a_off, first, ncnt, job_loc, job_rem = res[rank]
        assert jobs == [(other, 0), (other, 1)] and job_loc == [0, 1] and job_rem == [other * 8 + 0, other * 8 + 1]
        i = 0
        for (r, qq) in jobs:
            for lv in range(4):
                for rv in range(4):
                    assert a_off[i] == (lv * Q + qq) * cap                     # local view lv of quad frame qq
                    assert first[i] == 1000 * r + rv * Q + qq                  # remote view rv of the same time index, in place
                    assert ncnt[i] == 1 + (rv * Q + qq) % cap
                    i += 1
        assert i == 32


def test_pick_consumer_stream_ranks_candidates_without_a_gpu():
    """StereoPipe.pick_consumer_stream (the exchange's choice of ITS stream, d2slam_amd/swarm.py: PipeExchange): candidates are created one at a time and the first
    that takes turns with none of the lanes' streams -- or only with second (NetVLAD) streams -- is taken; after `tries` candidates the best seen; with no measured
    placement the first.  The pipe's two C calls are replaced by a table here (the measurement itself: tests/test_pipe.py on the GPU)."""
    from d2slam_amd import api

    class Fake(api.StereoPipe):
        def __init__(self, placement, n, classes):
            self._placement, self._n, self._classes, self.made, self.asked = placement, n, classes, 0, []

        def stream_placement(self):
            return self._placement, self._n

        def classify_stream(self, st):
            self.asked.append(st)
            return self._classes[st]

        def __del__(self):
            pass

    def run(placement, n, classes, tries=4):
        p = Fake(placement, n, classes)

        def make():
            p.made += 1
            return p.made - 1
        return p.pick_consumer_stream(make, tries=tries), p.made

    two = [(0, 2), (1, 3)]                                   # two lanes: classes 0 / 1 carry SuperPoint streams, 2 / 3 only NetVLAD streams
    assert run(two, 4, {0: 2}) == (0, 1)                     # first candidate beside a NetVLAD stream: taken at once
    assert run(two, 4, {0: 0, 1: 1, 2: 3}) == (2, 3)         # two candidates beside SuperPoint streams are passed over
    assert run(two, 4, {0: 0, 1: -1}) == (1, 2)              # a class no lane uses beats everything
    assert run(two, 4, {0: 1, 1: 0, 2: 1, 3: 0}) == (0, 4)   # nothing better within four tries: the first of the best rank
    four = [(0, 2), (1, 3), (2, 0), (3, 1)]                  # four lanes: every class carries a SuperPoint stream
    assert run(four, 4, {0: 2, 1: 3, 2: 0, 3: 1}) == (0, 4)
    assert run(two, 0, {}) == (0, 1)                         # no measured placement: the first candidate, nothing asked
