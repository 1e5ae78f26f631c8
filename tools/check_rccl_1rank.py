"""One-rank RCCL check of exactly the collective calls bench.py's --gpus N path makes (init with device_id, all_gather_into_tensor of the
exchange blocks on a side stream, barrier, all_reduce MAX), then the whole per-submit exchange sequence behind the frames-in-flight pipe
(swarm.PipeExchange in loopback mode) over that communicator: what can be verified of the nccl backend on a 1-GPU box."""
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

# ---- the exchange BEHIND the pipe over this one-rank RCCL communicator (swarm.PipeExchange, what `bench.py --gpus N` runs per submit): no worker thread (RCCL
# collectives are asynchronous), the all-gather on the exchange stream between d2fe_pipe_device_view and _release, loopback = the rank's own blocks as the remote
# agent: every keypoint of a left frame must match itself (distance 0) in the "cross-agent" list
import numpy as np  # noqa: E402
from d2slam_amd import api, netvlad as nvm  # noqa: E402
from d2slam_amd.synth import synth_stereo  # noqa: E402
from d2slam_amd.weights import synthetic_superpoint_weights  # noqa: E402
H, W, CAP, FR, LANES = 120, 160, 60, 2, int(os.environ.get("D2FE_RCCL_CHECK_LANES", "2"))
STEPS = int(os.environ.get("D2FE_RCCL_CHECK_STEPS", "5"))      # e.g. 4000: a soak of the view / release hand-over between the lanes and the exchange stream
fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2 * FR, precision=api.PREC_F32_WINO))
fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
pipe = api.StereoPipe(fe, lanes=LANES, frames=FR, width=W, height=H, cap=CAP, netvlad=True)
NS = LANES + 2
x = swarm.PipeExchange(torch, fe, pipe, dev, 1, 0, FR, CAP, fe.netvlad_dim, exchange="fp32", slots=NS, loopback=True)
assert x.worker is None and x.NR == FR
tk, enq, checked = [], 0, 0
sets = []
for i in range(6):      # a ring of frame sets (the soak cycles over them: every result block of the pipe is rewritten hundreds of times)
    fr = [synth_stereo(H, W, seed=900 + 3 * i + f) for f in range(FR)]
    sets.append((np.stack([p[0] for p in fr]), np.stack([p[1] for p in fr])))
for i in range(STEPS):
    tk.append(pipe.submit(*sets[i % len(sets)]))
    while enq <= i - 1:
        x.enqueue(tk[enq], enq % NS); enq += 1
    if i >= LANES:
        j = i - LANES
        o = pipe.wait(tk[j]); S = x.collect(j % NS)
        for f in range(FR):
            n = int(o["n_kp"][f])
            assert int(S["mn"][f]) == n and n > 10, (j, f, int(S["mn"][f]), n)
            assert np.array_equal(S["mq"][f, :n].numpy(), np.arange(n)) and np.array_equal(S["mt"][f, :n].numpy(), np.arange(n)) and float(S["md"][f, :n].abs().max()) == 0.0
            assert int(S["gate_pass"][f]) == 1          # a frame against itself: NetVLAD similarity 1
            checked += 1
while enq < STEPS:
    x.enqueue(tk[enq], enq % NS); enq += 1
for j in range(STEPS - LANES, STEPS):
    pipe.wait_raw(tk[j]); x.collect(j % NS)
tl = x.timeline_ms()
assert checked == (STEPS - LANES) * FR and tl["all_gather"] > 0
x.close(); pipe.close(); fe.close()
dist.destroy_process_group()
print("RCCL 1-rank exchange behind the pipe OK: %d frames checked, all-gather %.3f ms on the exchange stream" % (checked, tl["all_gather"]))
print("RCCL 1-rank OK: init(device_id), all_gather_into_tensor on a side stream, barrier, all_reduce", float(s.item()) > 0)
