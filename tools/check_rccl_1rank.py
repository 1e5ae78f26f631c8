"""One-rank RCCL check of exactly the collective calls bench.py's --gpus N path makes (init with device_id, all_gather_into_tensor of the
exchange blocks on a side stream, barrier, all_reduce MAX): what can be verified of the nccl backend on a 1-GPU box."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d2slam_amd import swarm  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
F, BLK = 4, swarm.block_words(200, 4096)
blocks = torch.arange(F * BLK, dtype=torch.float32, device=dev).view(F, BLK)
gath = torch.zeros(1, F, BLK, device=dev)
tail = torch.cuda.Stream()
tail.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(tail):
    swarm.all_gather_blocks(gath, blocks)
    s = gath.sum()
torch.cuda.current_stream().wait_stream(tail)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(gath[0], blocks)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("RCCL 1-rank OK: init(device_id), all_gather_into_tensor on a side stream, barrier, all_reduce", float(s.item()) > 0)
