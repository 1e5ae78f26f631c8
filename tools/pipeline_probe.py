#!/usr/bin/env python3
"""Probe: does running two independent extract(+match) sequences on two handles / two HIP streams raise throughput over one
stream (tails of the persistent conv kernels and the latency-bound post-processing of one step under the other step's convs)?
Usage: python tools/pipeline_probe.py [frames_per_step]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from d2slam_amd import api
from d2slam_amd.synth import synth_stereo
from d2slam_amd.weights import synthetic_superpoint_weights

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, W, CAP, NI = 480, 640, 200, 2 * F
dev = torch.device("cuda", 0)
w = synthetic_superpoint_weights(dustbin_bias=7.5)
host = np.stack([im for f in range(F) for im in synth_stereo(H, W, seed=f)])


def make():
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=NI))
    fe.load_superpoint(w)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        b = dict(imgs=torch.from_numpy(host).to(dev), kps=torch.zeros((NI, CAP, 2), device=dev), sc=torch.zeros((NI, CAP), device=dev),
                 desc=torch.zeros((NI, CAP, 256), device=dev), idx=torch.zeros((NI, CAP), dtype=torch.int32, device=dev),
                 cnt=torch.zeros(NI, dtype=torch.int32, device=dev))
        b["a_off"] = (torch.arange(0, NI, 2, device=dev) * CAP).to(torch.int32); b["b_off"] = b["a_off"] + CAP
        b["a_cnt"] = torch.zeros(F, dtype=torch.int32, device=dev); b["b_cnt"] = torch.zeros(F, dtype=torch.int32, device=dev)
        b["mq"] = torch.zeros((F, CAP), dtype=torch.int32, device=dev); b["mt"] = torch.zeros((F, CAP), dtype=torch.int32, device=dev)
        b["md"] = torch.zeros((F, CAP), device=dev); b["mn"] = torch.zeros(F, dtype=torch.int32, device=dev)
    return fe, st, b


def step(fe, st, b):
    s = st.cuda_stream
    fe.extract_device(b["imgs"].data_ptr(), NI, W, H, b["kps"].data_ptr(), b["sc"].data_ptr(), b["desc"].data_ptr(), b["idx"].data_ptr(), CAP,
                      b["cnt"].data_ptr(), stream=s)
    with torch.cuda.stream(st):
        b["a_cnt"].copy_(b["cnt"][0::2]); b["b_cnt"].copy_(b["cnt"][1::2])
    fe.match_batch_device(b["desc"].data_ptr(), b["desc"].data_ptr(), b["a_off"].data_ptr(), b["b_off"].data_ptr(), b["a_cnt"].data_ptr(),
                          b["b_cnt"].data_ptr(), F, 256, CAP, b["mq"].data_ptr(), b["mt"].data_ptr(), b["md"].data_ptr(), b["mn"].data_ptr(),
                          mode=0, ratio=0.8, radius=-1.0, stream=s)


def run(lanes, steps=40):
    for _ in range(4):
        for l in lanes:
            step(*l)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        step(*lanes[i % len(lanes)])
    torch.cuda.synchronize()
    return steps * F / (time.perf_counter() - t)


A, B = make(), make()
r1 = run([A]); r2 = run([A, B]); r1b = run([A]); r2b = run([A, B])
print("one stream: %.1f / %.1f stereo fps   two handles on two streams: %.1f / %.1f stereo fps" % (r1, r1b, r2, r2b))
