#!/usr/bin/env python3
"""Timing of the section 8(f)-4 LK-tracker entry points on the GPU (host-pointer API, so H2D/D2H of the points is included),
with the oracle timed beside them.  Usage: python tools/bench_lk.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from d2slam_amd import api
from d2slam_amd.synth import synth_stereo
from oracle import oracle as orc


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n * 1e3


def main():
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    for (h, w) in ((480, 640), (400, 800)):
        l, r = synth_stereo(h, w, seed=5)
        fl, fr = api.buildImagePyramid(fe, l), api.buildImagePyramid(fe, r)
        pts = api.detectFastByRegion(fe, fl, 150, 3, 4)
        res = {
            "pyramid (upload + 2 pyrDown)": timeit(lambda: api.buildImagePyramid(fe, l).close()),
            "lk_track fwd+rev, %d pts" % len(pts): timeit(lambda: api.lk_track(fe, fl, fr, pts, pts)),
            "lk_track_batch, 8 pairs x %d pts (quadcam)" % len(pts): timeit(lambda: api.lk_track_batch(fe, [(fl, fr, pts, pts, 0, 0.0)] * 8)),
            "detectFastByRegion 150 (3x4)": timeit(lambda: api.detectFastByRegion(fe, fl, 150, 3, 4)),
            "goodFeaturesToTrack 150": timeit(lambda: api.goodFeaturesToTrack(fe, fl, 150, 0.01, 20.0)),
        }
        pl, pr = orc.pyr_build(l), orc.pyr_build(r)
        cpu = {
            "pyramid (upload + 2 pyrDown)": timeit(lambda: orc.pyr_build(l), 5, 1),
            "lk_track fwd+rev, %d pts" % len(pts): timeit(lambda: orc.lk_track(pl, pr, w, h, pts, pts), 5, 1),
            "lk_track_batch, 8 pairs x %d pts (quadcam)" % len(pts): 8 * timeit(lambda: orc.lk_track(pl, pr, w, h, pts, pts), 3, 1),
            "detectFastByRegion 150 (3x4)": timeit(lambda: orc.fast_by_region(l, 150, 3, 4), 5, 1),
            "goodFeaturesToTrack 150": timeit(lambda: orc.good_features(l, 150, 0.01, 20.0), 5, 1),
        }
        for k in res:
            print("%dx%d  %-44s GPU %8.3f ms   oracle (CPU, OpenMP where parallel) %8.3f ms" % (w, h, k, res[k], cpu[k]))
    fe.close()


if __name__ == "__main__":
    main()
