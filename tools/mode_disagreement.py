#!/usr/bin/env python3
"""Index-parity evidence for the arithmetic modes: keypoint and match index sets of the Winograd fp32 mode (`value`) and of the
fp16 hi/lo mode against the bitwise-exact direct-convolution fp32 mode, over >= 1000 images of 640x480 -- 496 synthetic stereo pairs plus frames
derived from the three real crops of the reference's sample_data/fisheye.jpg that tests/golden/reference_headline.npz carries (flips, mirror-padded
re-crops, gain changes: real image statistics, no new reference content) -- at N = 100 / 150 / 200 keypoints and both thresholds the reference's
configurations use (0.015: superpoint_onnx.h:19; 0.15: quadcam_single.yaml:117).  Pairs: synthetic L<->R, real frame <-> the frame shifted by (3, 5) px.
Writes one JSON object (rates per 10^4 keypoints / matches) to stdout; run on the GPU box:  python tools/mode_disagreement.py > gpurun_out/mode_disagreement.json
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=496, help="synthetic stereo pairs (2 images each)")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from d2slam_amd import api, parity_study as ps
    t0 = time.time()
    imgs, pairs, n_syn = ps.frames(args.pairs)
    NI = len(imgs)
    out = {"images": NI, "synthetic_images": n_syn, "real_derived_images": NI - n_syn, "pairs": len(pairs), "geometry": "640x480", "configs": []}
    for thr in (0.015, 0.15):
        for N in (100, 150, 200):
            rec = ps.study(api, imgs, pairs, n_syn, thr, N, args.batch)
            out["configs"].append(rec)
            print("thr %.3f N %d: wino %s  f16x2 %s" % (thr, N, rec["wino_vs_f32_all"], rec["f16x2_vs_f32_all"]), file=sys.stderr, flush=True)
    out["seconds"] = round(time.time() - t0, 1)
    out["note"] = ("symmetric differences of raster-index sets (keypoints per image; matches per pair as (index, index) pairs) against the exact fp32 mode; seeded random-init "
                   "weights compress the score distribution, so near-ties at the top-K cut are far more frequent than with a trained network")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
