#!/usr/bin/env python3
"""Latency of the host-pointer entry points (what a D2SLAM adapter calls): one stereo extract, matchKNN, codec, undistort.
Usage: python tools/bench_host_api.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from d2slam_amd import api
from d2slam_amd.synth import synth_stereo, synth_descriptor_pair
from d2slam_amd.weights import synthetic_superpoint_weights


def timeit(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n * 1e3


fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=200, input_width=640, input_height=480, max_batch=2))
fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5))
l, r = synth_stereo(480, 640, seed=1)
pair = np.stack([l, r])
a, b, pa, pb = synth_descriptor_pair(200, 200, 256, seed=2)
x = np.random.RandomState(0).randn(200 * 256).astype(np.float32)
print("extract_batch (stereo pair, host pointers, exact fp32): %.3f ms" % timeit(lambda: fe.extract_batch(pair)))
print("matchKNN 200x200x256 (host pointers):                  %.3f ms" % timeit(lambda: fe.match_knn(a, b, 0.8, pa, pb, 32.0)))
print("match_crosscheck 200x200x256:                          %.3f ms" % timeit(lambda: fe.match_crosscheck(a, b)))
print("quantize_int8 200x256:                                 %.3f ms" % timeit(lambda: fe.quantize_int8(x)))
fe.close()
