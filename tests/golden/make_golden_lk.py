"""Generates tests/golden/lk_golden.npz: regression pins of the LK-tracker oracle (oracle/d2fe_oracle_lk.c) on a small seeded
stereo pair.  The reference ships no vectors for this path and OpenCV is not installed here, so these are known-answer pins of
the restated algorithm, not outputs of the reference.  Run from the repo root:  python tests/golden/make_golden_lk.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
from d2slam_amd.synth import synth_stereo
from oracle import oracle as orc

img0, img1 = synth_stereo(240, 320, seed=77)
pyr0 = orc.pyr_build(img0, 2)
fast_xy, fast_resp = orc.fast_by_region(img0, 60, 3, 4)
gftt_xy = orc.good_features(img0, 50, 0.01, 15.0)
lk_pts, lk_status = orc.lk_track(pyr0, orc.pyr_build(img1, 2), 320, 240, fast_xy, fast_xy)
np.savez_compressed(os.path.join(HERE, "lk_golden.npz"), img0=img0, img1=img1, pyr0=pyr0, fast_xy=fast_xy, fast_resp=fast_resp,
                    gftt_xy=gftt_xy, lk_pts=lk_pts, lk_status=lk_status)
print("wrote lk_golden.npz:", len(fast_xy), "FAST points,", len(gftt_xy), "corners,", int(lk_status.sum()), "tracked")
