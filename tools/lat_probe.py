#!/usr/bin/env python3
"""Single-frame latency probe: N calls of d2fe_superpoint_extract_batch on one stereo pair (host pointers) + 2 d2fe_match_knn, wino mode.
Run under `rocprofv3 --kernel-trace` and feed the trace to tools/lat_timeline.py to see where a call's wall time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from d2slam_amd import api
from d2slam_amd.synth import synth_stereo
from d2slam_amd.weights import synthetic_superpoint_weights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=200, input_width=640, input_height=480, max_batch=2, precision=api.PREC_F32_WINO))
fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5))
l, r = synth_stereo(480, 640, seed=3)
pair = np.stack([l, r])
for _ in range(5):
    out = fe.extract_batch(pair)
(_, _, d0), (_, _, d1) = out
t = []
for _ in range(N):
    t0 = time.perf_counter()
    out = fe.extract_batch(pair)
    t1 = time.perf_counter()
    fe.match_knn(d0, d1, 0.8)
    t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t1))
t = np.array(t) * 1e3
print("extract_batch(2): p50 %.3f ms   match_knn: p50 %.3f ms" % (np.median(t[:, 0]), np.median(t[:, 1])))
fe.profile_enable(2)
for _ in range(10):
    fe.extract_batch(pair)
print({k: round(v[0] / max(v[1], 1), 4) for k, v in fe.profile_read().items() if v[1]})
fe.close()
