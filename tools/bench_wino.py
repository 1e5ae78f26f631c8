"""Per-layer timing of the Winograd kernels (d2fe_debug_conv3x3_wino) at the SuperPoint layer shapes, batch of `--imgs` images."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d2slam_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--imgs", type=int, default=16)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--layers", default="")
a = ap.parse_args()
LAYERS = [("conv1b", 480, 640, 64, 64, True), ("conv2a", 240, 320, 64, 64, False), ("conv2b", 240, 320, 64, 64, True),
          ("conv3a", 120, 160, 64, 128, False), ("conv3b", 120, 160, 128, 128, True), ("conv4a", 60, 80, 128, 128, False),
          ("convPa", 60, 80, 128, 256, False)]
fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=16, input_width=64, input_height=64, max_batch=1))
rng = np.random.default_rng(0)
for name, H, W, cin, cout, pool in LAYERS:
    if a.layers and name not in a.layers.split(","):
        continue
    n = a.imgs if H < 480 else max(1, a.imgs // 4)
    x = np.maximum(rng.standard_normal((n, H, W, cin)).astype(np.float32), 0)
    wg = (rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32)
    b = np.zeros(cout, np.float32)
    _, ms = fe.debug_conv3x3_wino(x, wg, b, pool=pool, iters=a.iters)
    fl = 2.0 * 9 * cin * cout * H * W * n
    print("%-7s n=%2d %3dx%3d %3d->%3d pool=%d  %.3f ms/launch  direct-equivalent %.1f TF/s  (MFMA %.1f TF/s)" %
          (name, n, H, W, cin, cout, pool, ms, fl / ms / 1e9, fl / 2.25 / ms / 1e9), flush=True)
fe.close()
