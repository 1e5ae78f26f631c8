"""Cross-agent exchange for N>1 GPUs: the MI355X-native stand-in for the reference's LCM broadcast of keyframe
descriptors (d2frontend/src/loop_net.cpp:24-87).

Every agent (rank) owns F stereo frames per step.  After extraction each rank packs ONE fixed-capacity block per left frame
(include/d2fe.h, d2fe_pack_blocks_device):

    desc[cap][256] | kps[cap][2] | scores[cap] | netvlad[G] | n (int32) | zero padding to a multiple of 256 words

and ONE all-gather (RCCL over xGMI with backend "nccl"; gloo in the tests) delivers every agent's blocks to every rank.  The
gathered buffer lives directly behind the rank's own descriptor rows in one pool of 256-float rows, so the batched matcher
addresses remote descriptors in place; each rank matches its own left frames against every remote left frame (row-block
decomposition of the all-to-all match matrix, no second exchange).  The reference only tracks a remote frame whose NetVLAD similarity
reaches track_remote_netvlad_thres (d2featuretracker.cpp:185-203): d2fe_gate_pairs_device evaluates that gate for the pair list on
the device and reports how many pairs pass.

Device-agnostic on purpose: works on whatever device the tensors live on (the block layout helpers are pure Python).
"""
from typing import List

import torch
import torch.distributed as dist


def block_words(cap: int, netvlad_dim: int) -> int:
    """Mirror of d2fe_block_words (api.hip): float words of one block."""
    return (cap * 259 + netvlad_dim + 1 + 255) // 256 * 256


def block_bytes_int8(cap: int, netvlad_dim: int) -> int:
    """Mirror of d2fe_block_bytes_int8: desc_q[cap][256] | netvlad_q[G] | kps f32[cap][2] | n int32, padded to 64 bytes."""
    return (cap * 256 + netvlad_dim + cap * 8 + 4 + 63) // 64 * 64


def block_field_offset(cap: int, netvlad_dim: int, field: str) -> int:
    return {"desc": 0, "kps": cap * 256, "scores": cap * 258, "netvlad": cap * 259, "n": cap * 259 + netvlad_dim}[field]


class PairList:
    """The matchKNN problems of one step on one rank, as row offsets into the pool of 256-float rows
       [0, F*cap)            left descriptors of the current step (frame f at row f*cap)
       [F*cap, 2F*cap)       right descriptors
       [2F*cap, 3F*cap)      left descriptors of the previous step
       [3F*cap, ...)         gathered blocks, block (r, f) at row 3F*cap + (r*F + f) * blk_words/256
    local pairs first: L_f<->R_f and L_f<->prevL_f (D2FeatureTracker::trackLocalFrames, d2featuretracker.cpp:403-456,658-695),
    then L_f <-> the left frame with the same time index f of every OTHER rank (trackRemoteFrames, d2featuretracker.cpp:237-310: an
    agent tracks the frame a remote agent has just broadcast against its own current keyframe).  The F frames of a step are F
    consecutive time steps of one agent, batched for throughput."""

    def __init__(self, F: int, cap: int, world: int, rank: int, blk_words: int):
        self.a_off: List[int] = []; self.b_off: List[int] = []
        self.a_cnt_row: List[int] = []; self.b_cnt_row: List[int] = []     # rows of the local count array [3F] (local pairs)
        self.remote_block: List[int] = []; self.remote_q_frame: List[int] = []
        for f in range(F):
            self.a_off += [f * cap, f * cap]
            self.b_off += [(F + f) * cap, (2 * F + f) * cap]
            self.a_cnt_row += [f, f]
            self.b_cnt_row += [F + f, 2 * F + f]
        self.n_local = len(self.a_off)
        if world > 1:
            assert blk_words % 256 == 0
            rows_per_block = blk_words // 256
            for r in range(world):
                if r == rank:
                    continue
                for f in range(F):
                    self.a_off.append(f * cap)
                    self.b_off.append(3 * F * cap + (r * F + f) * rows_per_block)
                    self.a_cnt_row.append(f)
                    self.remote_block.append(r * F + f)
                    self.remote_q_frame.append(f)
        self.npairs = len(self.a_off)
        self.n_remote = self.npairs - self.n_local


def all_gather_blocks(gath: torch.Tensor, blocks: torch.Tensor, group=None) -> None:
    """ONE collective per step: gath [world][F][BLK] <- every rank's blocks [F][BLK].  gloo (tests, 1-GPU debugging) cannot gather
    device tensors: stage through the host there."""
    if dist.get_backend(group) == "gloo" and blocks.is_cuda:
        g = torch.empty(gath.shape, dtype=gath.dtype)
        dist.all_gather_into_tensor(g.view(-1), blocks.detach().cpu().contiguous().view(-1), group=group)
        gath.copy_(g)
    else:
        dist.all_gather_into_tensor(gath.view(-1), blocks.view(-1), group=group)


class QuadSwarm:
    """BASELINE configs[4]: one quadcam (FOURCORNER_FISHEYE) agent per rank.  After the per-agent chain (d2slam_amd.quadcam.QuadcamChain:
    undistort, SuperPoint + NetVLAD on the 4 views, neighbour + temporal matching) every rank packs ONE block per VIEW (4 per quad
    frame, each with its own NetVLAD vector -- the reference broadcasts the whole VisualImageDescArray, loop_net.cpp:24-87), ONE
    all-gather ships them, and each rank matches its own views against every remote agent's views of the same time index: the 16 view
    pairs (local view lv, remote view rv) of a job (local quad frame q, remote agent r) are 16 consecutive matcher problems, index
    lv*4 + rv, whose b-side descriptors are read in place inside the gathered blocks (row-block decomposition, no second exchange).

    The reference's gate for this camera configuration (getMatchedPrevKeyframe, d2featuretracker.cpp:212-233, and the view pairing of
    trackRemoteFrames, :282-297) is evaluated on the device by d2fe_quad_gate_device: mode "all2all" matches all 16 view pairs and
    reports how many jobs pass the gate and which rotation the reference would pick; mode "gated" zeroes the a-side count of the 12
    (or 16) view pairs the reference would not track, so that exactly its four pairs are matched.  WHOLE_IMG_MATCH, no radius gate
    (trackRemote passes search_radius*2 but matchLocalFeatures only uses it with motion prediction, :1100-1115)."""

    def __init__(self, chain, torch, dev, world, rank, netvlad_dim, thres, mode="all2all", knn_ratio=0.8, exchange="fp32"):
        assert mode in ("all2all", "gated") and exchange in ("fp32", "int8", "int8-renorm256")
        self.chain, self.torch, self.world, self.rank, self.G, self.thres, self.mode, self.ratio = chain, torch, world, rank, netvlad_dim, thres, mode, knn_ratio
        self.exchange = exchange
        Q, NI, cap = chain.Q, chain.NI, chain.cap
        self.BLK = block_words(cap, netvlad_dim)
        self.block_bytes = 4 * self.BLK
        self.renorm = 1 if exchange == "int8-renorm256" else 0      # decode: 0 = as the reference's LCM constructor, 1 = every descriptor over its 256 floats
        if exchange.startswith("int8"):      # the reference's wire precision (include/d2fe.h): quantise, ONE all-gather of int8 blocks, decode into `gath`
            self.block_bytes = block_bytes_int8(cap, netvlad_dim)
            self.blocks_q = torch.zeros((NI, self.block_bytes), dtype=torch.int8, device=dev)
            self.gath_q = torch.zeros((world, NI, self.block_bytes), dtype=torch.int8, device=dev)
        rows_per_block = self.BLK // 256
        f32, i32 = torch.float32, torch.int32
        self.blocks = torch.zeros((NI, self.BLK), dtype=f32, device=dev)
        self.gath = torch.zeros((world, NI, self.BLK), dtype=f32, device=dev)
        self.gath_i32 = self.gath.view(i32).view(world * NI, self.BLK)
        job_loc, job_rem, a_off, b_off, a_row, rem_blk = [], [], [], [], [], []
        self.jobs = []                                       # (remote rank, quad frame)
        for r in range(world):
            if r == rank:
                continue
            for q in range(Q):
                self.jobs.append((r, q))
                job_loc.append(q); job_rem.append(r * NI + q)
                for lv in range(4):
                    for rv in range(4):
                        a_off.append((lv * Q + q) * cap); a_row.append(lv * Q + q)
                        blk = r * NI + rv * Q + q
                        b_off.append(blk * rows_per_block); rem_blk.append(blk)
        self.njobs, self.NP = len(self.jobs), len(a_off)
        t = lambda x, dt: torch.tensor(x, dtype=dt, device=dev)
        self.job_loc, self.job_rem = t(job_loc, i32), t(job_rem, i32)
        self.a_off, self.b_off = t(a_off, i32), t(b_off, i32)
        self.a_row, self.rem_blk = t(a_row, torch.int64), t(rem_blk, torch.int64)
        n = max(self.NP, 1)
        self.a_cnt = torch.zeros(n, dtype=i32, device=dev); self.b_cnt = torch.zeros(n, dtype=i32, device=dev)
        self.mq = torch.zeros((n, cap), dtype=i32, device=dev); self.mt = torch.zeros((n, cap), dtype=i32, device=dev)
        self.md = torch.zeros((n, cap), dtype=f32, device=dev); self.mn = torch.zeros(n, dtype=i32, device=dev)
        nj = max(self.njobs, 1)
        self.dir_prev = torch.full((nj,), -1, dtype=i32, device=dev); self.sims = torch.zeros((nj, 4), dtype=f32, device=dev)
        self.n_pass = torch.zeros(1, dtype=i32, device=dev)
        self.n_off = block_field_offset(cap, netvlad_dim, "n"); self.g_off = block_field_offset(cap, netvlad_dim, "netvlad")

    def step(self, st, group=None, src=None):
        """pack -> ONE all-gather -> gate -> cross-agent matching, all ordered on the current torch stream (raw handle `st`).  src: the (desc, pts, scores, cnt,
        gdesc) tensors the exchange reads -- the chain's own (default) or a snapshot of them (step_overlapped)."""
        c, fe, torch = self.chain, self.chain.fe, self.torch
        Q, NI, cap, G = c.Q, c.NI, c.cap, self.G
        desc, pts, scores, cnt, gdesc = src or (c.desc, c.pts, c.scores, c.cnt, c.gdesc)
        if self.exchange.startswith("int8"):
            fe.pack_blocks_int8_device(desc.data_ptr(), pts.data_ptr(), cnt.data_ptr(), gdesc.data_ptr(), 0, 1, NI, cap, G,
                                       self.blocks_q.data_ptr(), stream=st)
            all_gather_blocks(self.gath_q, self.blocks_q, group)
            fe.unpack_blocks_int8_device(self.gath_q.data_ptr(), self.world * NI, cap, G, self.gath.data_ptr(), renorm=self.renorm, stream=st)
        else:
            fe.pack_blocks_device(desc.data_ptr(), pts.data_ptr(), scores.data_ptr(), cnt.data_ptr(), gdesc.data_ptr(), 0, 1, NI, cap, G,
                                  self.blocks.data_ptr(), stream=st)
            all_gather_blocks(self.gath, self.blocks, group)
        if self.NP == 0:
            return
        torch.index_select(cnt, 0, self.a_row, out=self.a_cnt)
        self.b_cnt.copy_(self.gath_i32[self.rem_blk, self.n_off])
        self.n_pass.zero_()
        fe.quad_gate_device(gdesc.data_ptr(), G, self.gath.data_ptr() + 4 * self.g_off, self.BLK, G, self.job_loc.data_ptr(),
                            self.job_rem.data_ptr(), Q, Q, self.njobs, self.thres, d_dir_prev=self.dir_prev.data_ptr(), d_sims=self.sims.data_ptr(),
                            d_cnt_inout=self.a_cnt.data_ptr() if self.mode == "gated" else None, d_n_pass=self.n_pass.data_ptr(), stream=st)
        fe.match_batch_device(desc.data_ptr(), self.gath.data_ptr(), self.a_off.data_ptr(), self.b_off.data_ptr(), self.a_cnt.data_ptr(),
                              self.b_cnt.data_ptr(), self.NP, 256, cap, self.mq.data_ptr(), self.mt.data_ptr(), self.md.data_ptr(), self.mn.data_ptr(),
                              mode=0, ratio=self.ratio, radius=-1.0, stream=st)

    def step_overlapped(self, main, side, group=None):
        """The exchange of the chain step just queued on `main`, on the stream `side`: a snapshot of what the exchange reads (current views' descriptors, points,
        scores, counts, NetVLAD descriptors: 1.7 MB at 4 x 4 views) is taken on `side` as soon as the chain step is through, `main` waits for THAT only, and pack ->
        all-gather -> gate -> view x view matchKNN run on `side` beside the next chain step's convolutions (with RCCL; a host-blocking backend serialises)."""
        c, torch, NI = self.chain, self.torch, self.chain.NI
        if not hasattr(self, "snap"):
            self.snap = (torch.empty_like(c.desc[:NI]), torch.empty_like(c.pts[:NI]), torch.empty_like(c.scores), torch.empty_like(c.cnt[:NI]), torch.empty_like(c.gdesc))
            self.ev_chain, self.ev_snap = torch.cuda.Event(), torch.cuda.Event()
        self.ev_chain.record(main)
        with torch.cuda.stream(side):
            side.wait_event(self.ev_chain)
            for dst, srct in zip(self.snap, (c.desc[:NI], c.pts[:NI], c.scores, c.cnt[:NI], c.gdesc)):
                dst.copy_(srct, non_blocking=True)
            self.ev_snap.record(side)
            main.wait_event(self.ev_snap)          # the next chain step overwrites the chain's buffers: it may start once the snapshot exists
            self.step(side.cuda_stream, group, src=self.snap)


def remote_pair_layout(world: int, rank: int, F: int, cap: int, blk_words: int, loopback: bool = False):
    """The cross-agent matchKNN problems of one submit on one rank (pure host logic; PipeExchange and its CPU test share it): local left frame f against the frame
    with the same time index of every OTHER rank (trackRemoteFrames, d2featuretracker.cpp:237-310), rank-major.  Returns (a_off, b_off, q_frame, remote_block):
    a_off = row of the local frame's descriptors in the lane's result block (rows of 256 floats, left frames first), b_off = row of the remote block's descriptors in
    the gathered buffer [world][F][blk_words] (blocks are multiples of 256 words and start with their descriptors), q_frame = f, remote_block = r * F + f."""
    assert blk_words % 256 == 0
    rows = blk_words // 256
    a_off, b_off, qf, rb = [], [], [], []
    for r in range(world):
        if r == rank and not loopback:
            continue
        for f in range(F):
            a_off.append(f * cap); b_off.append((r * F + f) * rows); qf.append(f); rb.append(r * F + f)
    return a_off, b_off, qf, rb


class PipeExchange:
    """The cross-agent exchange of one rank BEHIND a frames-in-flight pipe, as a thin caller of the C ABI (include/d2fe.h d2fe_exchange_*, csrc/exchange.hip; round 6):
    the stand-in for the reference's broadcast of the frame it has just extracted (loop_net.cpp:24-87) and trackRemoteFrames on the receivers
    (d2featuretracker.cpp:237-310).  Per ticket the LIBRARY queues, on ONE stream of the exchange's own (own_stream=True, the measured best: profiles/r06_exchange_placement_ab.txt)
    or on the stream of the lane that produced the ticket (behind that lane's D2H; that lane's next pass then waits for the sequence):

        device view -> pack_blocks(_int8) -> ONE all-gather -> [int8: decode] -> counts -> NetVLAD gate -> ONE matcher launch -> release -> ONE D2H into a pinned slot

    Backend "nccl": the collective is ncclAllGather on an RCCL communicator of the library's own (d2fe_rccl_*: librccl through dlopen; the 128-byte id travels over
    the torch.distributed group once, at creation) -- neither torch's stream wrapper nor Python is in the per-ticket path.  Any other backend (gloo in the tests, two
    ranks on one GPU): the library calls back into all_gather_host below, which stages the blocks through the host, from a worker thread (a host-blocking collective
    on the submitting thread would starve the pipe; the reference's LCM handler runs beside the front-end thread too, loop_net.cpp).
    enqueue() is called one submit BEHIND the pipe (after submit(i): enqueue(ticket i - 1)).  collect() returns torch views of the pinned slot."""

    def __init__(self, torch, fe, pipe, dev, world, rank, F, cap, netvlad_dim, exchange="fp32", gate_thres=0.8, ratio=0.8, slots=4, group=None, loopback=False, own_stream=True,
                 timing=True):
        from . import api
        assert exchange in ("fp32", "int8", "int8-renorm256") and (world > 1 or loopback)
        self.torch, self.pipe, self.world, self.rank, self.F, self.cap, self.G, self.group, self.dev = torch, pipe, world, rank, F, cap, netvlad_dim, group, dev
        self.exchange = exchange
        self.comm = None
        self.worker = None
        self.err = None
        self.timeline = []
        self.backend = dist.get_backend(group)
        cb = None
        if self.backend == "nccl":
            uid = [api.rccl_unique_id() if dist.get_rank(group) == 0 else None]
            dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            self.comm = api.rccl_comm_init_rank(uid[0], world, rank, dev.index if dev.index is not None else 0)
        else:
            cb = self._all_gather_host
        self.x = api.Exchange(pipe, comm=self.comm, world=world, rank=rank, wire=exchange, loopback=loopback, slots=slots, own_stream=own_stream, timing=timing,
                              gate_thres=gate_thres, ratio=ratio, all_gather=cb)
        self.NR = self.x.npairs
        self.block_bytes = self.x.block_bytes
        self.d2h_bytes = 4 * (3 * self.NR * cap + 2 * self.NR + 1)
        self.stream = None          # the sequence runs on the lanes' own streams (own_stream: a hipStream_t address in self.x.stream)
        if self.backend != "nccl":
            import queue, threading
            self.q = queue.Queue()
            self.posted = [threading.Event() for _ in range(slots)]
            self.worker = threading.Thread(target=self._work, daemon=True)
            self.worker.start()

    def _all_gather_host(self, user, d_send, d_recv, nbytes, stream):
        """d2fe_all_gather_fn for backends without device collectives: stream -> host -> torch.distributed.all_gather -> device; complete on return"""
        try:
            import ctypes as C
            torch, hip = self.torch, _hip_runtime()
            if hip.hipStreamSynchronize(C.c_void_p(stream)):
                return 1
            send = torch.empty(nbytes, dtype=torch.uint8)
            if hip.hipMemcpy(C.c_void_p(send.data_ptr()), C.c_void_p(d_send), C.c_size_t(nbytes), 2):      # hipMemcpyDeviceToHost
                return 2
            recv = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(recv, send, group=self.group)
            allb = torch.cat(recv)
            if hip.hipMemcpy(C.c_void_p(d_recv), C.c_void_p(allb.data_ptr()), C.c_size_t(nbytes * self.world), 1):      # hipMemcpyHostToDevice
                return 3
            return 0
        except Exception as e:      # noqa: BLE001 -- never raise through the C frame
            self.err = e
            return 9

    def _work(self):
        self.torch.cuda.set_device(self.dev)
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                if self.err is None:        # after a failure no later ticket is processed (collect() surfaces the error)
                    self.x.enqueue(*item)
            except Exception as e:      # noqa: BLE001
                self.err = e
            self.posted[item[1]].set()

    def close(self):
        if self.worker:
            self.q.put(None); self.worker.join(timeout=30); self.worker = None
        if getattr(self, "x", None):
            self.x.close(); self.x = None
        if self.comm:
            from . import api
            api.rccl_comm_destroy(self.comm); self.comm = None

    def enqueue(self, ticket, slot):
        if self.worker:
            self.posted[slot].clear()
            self.q.put((ticket, slot))
        else:
            self.x.enqueue(ticket, slot)

    def collect(self, slot):
        """blocks until the slot's results are in host memory; torch views of the pinned slot (valid until the slot is enqueued again)"""
        if self.worker:
            self.posted[slot].wait()
        if self.err:
            raise self.err
        r = self.x.collect(slot)
        if r["phase_ms"] is not None and len(self.timeline) < 4096:
            self.timeline.append(r["phase_ms"])
        t = self.torch
        S = {k: t.from_numpy(r[k]) for k in ("mq", "mt", "md", "mn") if r[k] is not None}
        S["gate_pass"] = t.from_numpy(r["gate_pass"]) if r["gate_pass"] is not None else t.zeros(self.NR, dtype=t.int32)
        S["gate_n"] = t.tensor([r["gate_n"]], dtype=t.int32)
        S["ticket"] = r["ticket"]
        return S

    def timeline_ms(self):
        import numpy as np
        from . import api
        if not self.timeline:
            return None
        tl = np.array(self.timeline)
        out = {n: round(float(np.median(tl[:, i])), 4) for i, n in enumerate(api.EXCHANGE_PHASES)}
        out["all_gather_max"] = round(float(tl[:, 1].max()), 4)
        out["exchange_stream_busy_ms_per_submit"] = round(float(np.median(tl.sum(1))), 4)
        return out


_HIP = None


def _hip_runtime():
    """the HIP runtime the process already holds (the library's and torch's), for the host-staged collective of the test backends"""
    global _HIP
    if _HIP is None:
        import ctypes as C
        path = "libamdhip64.so"
        try:        # the very file that is mapped already (a second copy of the runtime would not see the first one's streams)
            with open("/proc/self/maps") as f:
                path = next((l.split()[-1] for l in f if "libamdhip64.so" in l), path)
        except OSError:
            pass
        _HIP = C.CDLL(path)
    return _HIP


class TorchPipeExchange:
    """Round 5's form of the same exchange (kept as the fallback `bench.py --exchange-impl torch`): the sequence driven from Python, torch.distributed for the
    collective, ONE stream of its own.  The cross-agent exchange of one rank BEHIND a frames-in-flight pipe (d2slam_amd.api.StereoPipe, include/d2fe.h d2fe_pipe_*): the stand-in for the reference's
    broadcast of the frame it has just extracted (loop_net.cpp:24-87) and trackRemoteFrames on the receivers (d2featuretracker.cpp:237-310), as ONE sequence per
    submit on a stream of its own --

        d2fe_pipe_device_view(ticket)         this stream waits for the ticket's SuperPoint + NetVLAD; device pointers into the lane's result block
        pack_blocks(_int8)                    one fixed-capacity block per left frame
        ONE all-gather                        RCCL over xGMI (backend "nccl"); gloo in the tests
        [int8: decode]  counts  NetVLAD gate  getMatchedPrevKeyframe's threshold on the device (d2featuretracker.cpp:185-203)
        ONE matcher launch                    L_f of this rank against the frame with the same time index of every other rank; the a side is read in place in the
                                              lane's block, the b side in place in the gathered blocks (row-block decomposition, no second exchange)
        d2fe_pipe_device_release(ticket)      the lane may write that block again once this stream got here
        D2H of the cross-agent match lists    into a ring of pinned slots; an event per slot

    -- so the N > 1 step IS the N = 1 step (the pipe: host frames in, host results out, `lanes` submits in flight) plus this sequence beside it; the lanes'
    convolutions never wait for the collective.  enqueue() is called one submit BEHIND the pipe (after submit(i): enqueue(ticket i-1)), which keeps a host-blocking
    collective (gloo) from starving the device."""

    def __init__(self, torch, fe, pipe, dev, world, rank, F, cap, netvlad_dim, exchange="fp32", gate_thres=0.8, ratio=0.8, slots=4, group=None, loopback=False):
        # loopback: the rank's OWN gathered blocks count as a remote agent too (a one-rank communicator then exercises the whole sequence: how the RCCL path is
        # checked on a 1-GPU box, tools/check_rccl_1rank.py)
        assert exchange in ("fp32", "int8", "int8-renorm256") and (world > 1 or loopback)
        self.torch, self.fe, self.pipe, self.world, self.rank, self.F, self.cap, self.G = torch, fe, pipe, world, rank, F, cap, netvlad_dim
        self.exchange, self.group, self.thres, self.ratio = exchange, group, gate_thres, ratio
        self.int8 = exchange.startswith("int8")
        self.renorm = 1 if exchange == "int8-renorm256" else 0
        self.BLK = block_words(cap, netvlad_dim)
        self.BLKB = block_bytes_int8(cap, netvlad_dim)
        self.block_bytes = self.BLKB if self.int8 else 4 * self.BLK
        f32, i32, i64 = torch.float32, torch.int32, torch.int64
        # a pool stream that takes turns with none of the lanes' SuperPoint streams on the device (d2fe_pipe_classify_stream; candidates one at a time): a stream picked
        # blindly shares a hardware pipe with some lane's SuperPoint stream every other time
        mk = lambda: torch.cuda.Stream(device=dev)
        self.stream = pipe.pick_consumer_stream(mk, handle=lambda s: s.cuda_stream) if hasattr(pipe, "pick_consumer_stream") else mk()
        self.blocks = torch.zeros((F, self.BLK), dtype=f32, device=dev)
        self.gath = torch.zeros((world, F, self.BLK), dtype=f32, device=dev)
        self.gath_i32 = self.gath.view(i32).view(world * F, self.BLK)
        if self.int8:
            assert (cap * 256 + netvlad_dim + cap * 8) % 4 == 0
            self.blocks_q = torch.zeros((F, self.BLKB), dtype=torch.int8, device=dev)
            self.gath_q = torch.zeros((world, F, self.BLKB), dtype=torch.int8, device=dev)
            self.own_n = self.blocks_q.view(i32).view(F, self.BLKB // 4)[:, (cap * 256 + netvlad_dim + cap * 8) // 4]
        else:
            self.own_n = self.blocks.view(i32).view(F, self.BLK)[:, block_field_offset(cap, netvlad_dim, "n")]
        self.n_off = block_field_offset(cap, netvlad_dim, "n"); self.g_off = block_field_offset(cap, netvlad_dim, "netvlad")
        a_off, b_off, qf, rb = remote_pair_layout(world, rank, F, cap, self.BLK, loopback)
        self.NR = len(a_off)
        t = lambda x, dt: torch.tensor(x, dtype=dt, device=dev)
        self.a_off, self.b_off = t(a_off, i32), t(b_off, i32)
        self.q_frame64, self.rem_blk64 = t(qf, i64), t(rb, i64)
        self.gate_q, self.gate_db = t(qf, i32), t(rb, i32)
        NR = self.NR
        self.a_cnt = torch.zeros(NR, dtype=i32, device=dev); self.b_cnt = torch.zeros(NR, dtype=i32, device=dev)
        self.mq = torch.zeros((NR, cap), dtype=i32, device=dev); self.mt = torch.zeros((NR, cap), dtype=i32, device=dev)
        self.md = torch.zeros((NR, cap), dtype=f32, device=dev); self.mn = torch.zeros(NR, dtype=i32, device=dev)
        self.gate_pass = torch.zeros(NR, dtype=i32, device=dev); self.gate_n = torch.zeros(1, dtype=i32, device=dev)
        self.sims = torch.zeros(NR, dtype=f32, device=dev)
        # ring of pinned result slots: cross-agent match lists, counts, gate decisions
        self.slots = []
        for _ in range(slots):
            self.slots.append({"mq": torch.empty((NR, cap), dtype=i32).pin_memory(), "mt": torch.empty((NR, cap), dtype=i32).pin_memory(),
                               "md": torch.empty((NR, cap), dtype=f32).pin_memory(), "mn": torch.empty(NR, dtype=i32).pin_memory(),
                               "gate_pass": torch.empty(NR, dtype=i32).pin_memory(), "gate_n": torch.empty(1, dtype=i32).pin_memory(),
                               "done": torch.cuda.Event(), "ev": None})
        self.d2h_bytes = 4 * (3 * NR * cap + 2 * NR + 1)
        self.timing = True
        self.timeline = []          # per enqueue: 6 timing events on the exchange stream
        # A host-blocking backend (gloo: the all-gather is staged through the host and returns when it is complete) would hold the SUBMITTING thread for a full
        # extraction + host round trip per submit and starve the pipe.  The sequence then runs on a worker thread of its own -- as the reference's LCM handler does
        # (loop_net.cpp runs beside the front-end thread) -- and enqueue() only posts the ticket.  RCCL collectives are asynchronous: no thread, enqueue() queues
        # the work directly.  Collectives of the caller's thread (barriers) must not interleave with posted tickets: collect() every slot first (bench.py does)
        self.dev = dev
        self.worker = None
        self.err = None
        if dist.get_backend(group) != "nccl":
            import queue, threading
            self.q = queue.Queue()
            self.posted = [threading.Event() for _ in range(slots)]
            self.worker = threading.Thread(target=self._work, daemon=True)
            self.worker.start()

    def _work(self):
        self.torch.cuda.set_device(self.dev)
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                self._enqueue(*item)
            except Exception as e:      # noqa: BLE001 -- surfaced by collect()
                self.err = e
            self.posted[item[1]].set()

    def close(self):
        if self.worker:
            self.q.put(None); self.worker.join(timeout=30); self.worker = None

    def enqueue(self, ticket, slot):
        """the whole sequence for one submitted ticket, asynchronous on the exchange stream (host-blocking backend: posted to the worker thread)"""
        if self.worker:
            self.posted[slot].clear()
            self.q.put((ticket, slot))
        else:
            self._enqueue(ticket, slot)

    def _enqueue(self, ticket, slot):
        torch, fe, F, cap, G, X = self.torch, self.fe, self.F, self.cap, self.G, self.stream
        st = X.cuda_stream
        S = self.slots[slot]
        with torch.cuda.stream(X):
            v = self.pipe.device_view(ticket, st)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if self.timing else None
            mark = (lambda i: ev[i].record(X)) if ev else (lambda i: None)
            mark(0)
            if self.int8:
                fe.pack_blocks_int8_device(v.d_desc, v.d_kps_xy, v.d_n_kp, v.d_netvlad, 0, 1, F, cap, G, self.blocks_q.data_ptr(), stream=st)
                mark(1)
                all_gather_blocks(self.gath_q, self.blocks_q, self.group)
                mark(2)
                fe.unpack_blocks_int8_device(self.gath_q.data_ptr(), self.world * F, cap, G, self.gath.data_ptr(), renorm=self.renorm, stream=st)
            else:
                fe.pack_blocks_device(v.d_desc, v.d_kps_xy, v.d_scores, v.d_n_kp, v.d_netvlad, 0, 1, F, cap, G, self.blocks.data_ptr(), stream=st)
                mark(1)
                all_gather_blocks(self.gath, self.blocks, self.group)
                mark(2)
            torch.index_select(self.own_n, 0, self.q_frame64, out=self.a_cnt)
            self.b_cnt.copy_(self.gath_i32[self.rem_blk64, self.n_off])
            if G:
                self.gate_n.zero_()
                # all-to-all mode (BASELINE configs[3]/[4]): every pair is matched; the reference's gate is evaluated and counted
                fe.gate_pairs_device(v.d_netvlad, G, self.gath.data_ptr() + 4 * self.g_off, self.BLK, G, self.gate_q.data_ptr(), self.gate_db.data_ptr(),
                                     self.NR, self.thres, d_pass=self.gate_pass.data_ptr(), d_sims=self.sims.data_ptr(), d_n_pass=self.gate_n.data_ptr(), stream=st)
            mark(3)
            fe.match_batch_device(v.d_desc, self.gath.data_ptr(), self.a_off.data_ptr(), self.b_off.data_ptr(), self.a_cnt.data_ptr(), self.b_cnt.data_ptr(),
                                  self.NR, 256, cap, self.mq.data_ptr(), self.mt.data_ptr(), self.md.data_ptr(), self.mn.data_ptr(),
                                  mode=0, ratio=self.ratio, radius=-1.0, stream=st)
            mark(4)
            self.pipe.device_release(ticket, st)
            for k in ("mq", "mt", "md", "mn", "gate_pass", "gate_n"):
                S[k].copy_(getattr(self, k), non_blocking=True)
            mark(5)
            S["done"].record(X)
        S["ev"] = ev
        S["ticket"] = ticket

    def collect(self, slot):
        """blocks until the slot's results are in host memory; returns the slot (pinned tensors) and appends its timeline"""
        S = self.slots[slot]
        if self.worker:
            self.posted[slot].wait()
        if self.err:
            raise self.err
        S["done"].synchronize()
        if S["ev"]:
            e = S["ev"]
            self.timeline.append([e[i].elapsed_time(e[i + 1]) for i in range(5)])
            S["ev"] = None
        return S

    def timeline_ms(self):
        import numpy as np
        if not self.timeline:
            return None
        t = np.array(self.timeline)
        names = ("pack_blocks", "all_gather", "decode_counts_gate", "match_remote", "release_and_d2h")
        out = {n: round(float(np.median(t[:, i])), 4) for i, n in enumerate(names)}
        out["all_gather_max"] = round(float(t[:, 1].max()), 4)
        out["exchange_stream_busy_ms_per_submit"] = round(float(np.median(t.sum(1))), 4)
        return out
