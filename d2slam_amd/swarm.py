"""Cross-agent exchange for N>1 GPUs: the MI355X-native stand-in for the reference's LCM broadcast of keyframe
descriptors (d2frontend/src/loop_net.cpp:24-87).

Every agent (rank) owns F frames per step.  After extraction each rank contributes one fixed-capacity block per
frame -- descriptors [cap,256] and a keypoint count -- and ONE all-gather (RCCL over xGMI with backend "nccl",
gloo in the CPU tests) delivers every agent's blocks to every rank; each rank then matches its own frames against
all remote ones (row-block decomposition of the all-to-all match matrix, no second exchange).

Device-agnostic on purpose: works on whatever device the tensors live on.
"""
from typing import List, Tuple

import torch
import torch.distributed as dist


def pool_rows(F: int, world: int) -> int:
    """Rows of the descriptor pool: [0,2F) current L/R interleaved | [2F,3F) previous L | remote L of other ranks."""
    return 3 * F + (world - 1) * F


def build_pairs(F: int, world: int) -> Tuple[List[int], List[int]]:
    """(a_row, b_row) pool rows of every matchKNN problem of one step on one rank:
    L_f<->R_f and L_f<->prevL_f (the two calls of D2FeatureTracker::trackLocalFrames, d2featuretracker.cpp:403-456,
    658-695), then L_f<->remoteL_f for each other agent (trackRemoteFrames, d2featuretracker.cpp:237-310)."""
    NI = 2 * F
    a_rows, b_rows = [], []
    for f in range(F):
        a_rows += [2 * f, 2 * f]
        b_rows += [2 * f + 1, NI + f]
    for o in range(world - 1):
        for f in range(F):
            a_rows.append(2 * f)
            b_rows.append(NI + F + o * F + f)
    return a_rows, b_rows


def exchange_blocks(desc_pool: torch.Tensor, cnt_pool: torch.Tensor, F: int, rank: int, world: int,
                    gath_desc: torch.Tensor, gath_cnt: torch.Tensor, group=None) -> None:
    """All-gather the left-image blocks of this rank's F frames and scatter the OTHER ranks' blocks into the pool
    rows [3F, 3F+(world-1)F) in ascending rank order (own rank skipped)."""
    if world == 1:
        return
    NI = 2 * F
    left = torch.arange(0, NI, 2, device=desc_pool.device)
    dist.all_gather_into_tensor(gath_desc.view(-1), desc_pool[left].contiguous().view(-1), group=group)
    dist.all_gather_into_tensor(gath_cnt.view(-1), cnt_pool[left].contiguous(), group=group)
    others = [r for r in range(world) if r != rank]
    desc_pool[NI + F:] = gath_desc[others].reshape(-1, desc_pool.shape[1], desc_pool.shape[2])
    cnt_pool[NI + F:] = gath_cnt[others].reshape(-1)
