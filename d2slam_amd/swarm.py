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
