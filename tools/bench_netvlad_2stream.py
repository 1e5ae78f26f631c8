"""Experiment: NetVLAD on 32 images as ONE call vs TWO concurrent half-batches (two handles, two HIP streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from d2slam_amd import api, netvlad as nvm
from d2slam_amd.synth import synth_image
H, W, N = 480, 640, 32
dev = torch.device("cuda", 0)
imgs = torch.from_numpy(np.stack([synth_image(H, W, s % 4) for s in range(N)])).to(dev)
nv = nvm.synthetic_netvlad_weights()
for lanes in (1, 2, 4):
    n = N // lanes
    fes = [api.FrontEnd(api.SuperPointConfig(input_width=W, input_height=H, max_batch=n)) for _ in range(lanes)]
    for fe in fes: fe.load_netvlad(nv)
    outs = [torch.zeros((n, 4096), device=dev) for _ in range(lanes)]
    sts = [torch.cuda.Stream() for _ in range(lanes)]
    main = torch.cuda.Stream()
    def run():
        ev = torch.cuda.Event(); ev.record(main)
        for i in range(lanes):
            sts[i].wait_event(ev)
            fes[i].netvlad_device(imgs[i * n:].data_ptr(), n, W, H, outs[i].data_ptr(), stream=sts[i].cuda_stream)
            e2 = torch.cuda.Event(); e2.record(sts[i]); main.wait_event(e2)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(30): run()
    e1.record(main)
    torch.cuda.synchronize()
    print("NetVLAD %d images as %d concurrent call(s) of %d: %.3f ms" % (N, lanes, n, e0.elapsed_time(e1) / 30), flush=True)
    for fe in fes: fe.close()
