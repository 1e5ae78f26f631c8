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

    def step(self, st, group=None):
        """pack -> ONE all-gather -> gate -> cross-agent matching, all ordered on the current torch stream (raw handle `st`)."""
        c, fe, torch = self.chain, self.chain.fe, self.torch
        Q, NI, cap, G = c.Q, c.NI, c.cap, self.G
        if self.exchange.startswith("int8"):
            fe.pack_blocks_int8_device(c.desc.data_ptr(), c.pts.data_ptr(), c.cnt.data_ptr(), c.gdesc.data_ptr(), 0, 1, NI, cap, G,
                                       self.blocks_q.data_ptr(), stream=st)
            all_gather_blocks(self.gath_q, self.blocks_q, group)
            fe.unpack_blocks_int8_device(self.gath_q.data_ptr(), self.world * NI, cap, G, self.gath.data_ptr(), renorm=self.renorm, stream=st)
        else:
            fe.pack_blocks_device(c.desc.data_ptr(), c.pts.data_ptr(), c.scores.data_ptr(), c.cnt.data_ptr(), c.gdesc.data_ptr(), 0, 1, NI, cap, G,
                                  self.blocks.data_ptr(), stream=st)
            all_gather_blocks(self.gath, self.blocks, group)
        if self.NP == 0:
            return
        torch.index_select(c.cnt, 0, self.a_row, out=self.a_cnt)
        self.b_cnt.copy_(self.gath_i32[self.rem_blk, self.n_off])
        self.n_pass.zero_()
        fe.quad_gate_device(c.gdesc.data_ptr(), G, self.gath.data_ptr() + 4 * self.g_off, self.BLK, G, self.job_loc.data_ptr(),
                            self.job_rem.data_ptr(), Q, Q, self.njobs, self.thres, d_dir_prev=self.dir_prev.data_ptr(), d_sims=self.sims.data_ptr(),
                            d_cnt_inout=self.a_cnt.data_ptr() if self.mode == "gated" else None, d_n_pass=self.n_pass.data_ptr(), stream=st)
        fe.match_batch_device(c.desc.data_ptr(), self.gath.data_ptr(), self.a_off.data_ptr(), self.b_off.data_ptr(), self.a_cnt.data_ptr(),
                              self.b_cnt.data_ptr(), self.NP, 256, cap, self.mq.data_ptr(), self.mt.data_ptr(), self.md.data_ptr(), self.mn.data_ptr(),
                              mode=0, ratio=self.ratio, radius=-1.0, stream=st)
