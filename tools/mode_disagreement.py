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
H, W = 480, 640


def real_frames():
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_headline.npz"))
    d435, quad, tum = z["img_d435"], z["img_quad"], z["img_tum"]
    out = []
    for g in (1.0, 0.8, 1.25):
        base = np.clip(np.rint(d435.astype(np.float32) * g), 0, 255).astype(np.uint8)
        out += [base, base[:, ::-1].copy(), base[::-1].copy(), base[::-1, ::-1].copy()]
    q = np.pad(quad, ((40, 40), (0, 0)), mode="reflect")                      # 480 x 800
    for x0 in (0, 40, 80, 120, 160):
        out += [q[:, x0:x0 + W].copy(), q[::-1, x0:x0 + W].copy()]
    t = np.pad(tum, ((0, 0), (64, 64)), mode="reflect")                       # 512 x 640
    for y0 in (0, 8, 16, 24, 32):
        out += [t[y0:y0 + H].copy(), t[y0:y0 + H, ::-1].copy()]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=496, help="synthetic stereo pairs (2 images each)")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    from d2slam_amd import api
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights
    t0 = time.time()
    imgs, pairs = [], []
    for s in range(args.pairs):
        l, r = synth_stereo(H, W, seed=5000 + s)
        pairs.append((len(imgs), len(imgs) + 1)); imgs += [l, r]
    n_syn = len(imgs)
    for f in real_frames():
        pairs.append((len(imgs), len(imgs) + 1)); imgs += [f, np.roll(f, (3, 5), (0, 1))]
    imgs = np.ascontiguousarray(np.stack(imgs))
    NI = len(imgs)
    w015 = synthetic_superpoint_weights(dustbin_bias=7.5)
    w15 = dict(w015); Wt, b = w15["convPb"]; b = b.copy(); b[64] -= np.float32(3.5); w15["convPb"] = (Wt, b)     # as bench.py's quadcam leg: enough candidates above 0.15
    modes = {"f32": api.PREC_F32, "wino": api.PREC_F32_WINO, "f16x2": api.PREC_F16X2}
    out = {"images": NI, "synthetic_images": n_syn, "real_derived_images": NI - n_syn, "pairs": len(pairs), "geometry": "640x480", "configs": []}
    for thr, weights in ((0.015, w015), (0.15, w15)):
        for N in (100, 150, 200):
            sel = {}
            for name, prec in modes.items():
                fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=N, input_width=W, input_height=H, max_batch=args.batch, precision=prec, keypoint_threshold=thr))
                fe.load_superpoint(weights)
                kp, ds = [], []
                for i0 in range(0, NI, args.batch):
                    for k, s, d in fe.extract_batch(imgs[i0:i0 + args.batch], cap=N):
                        kp.append(k); ds.append(d)
                mt = []
                for ia, ib in pairs:
                    q, t, _ = fe.match_knn(ds[ia], ds[ib], 0.8)
                    ra = (kp[ia][:, 1].astype(np.int64) * W + kp[ia][:, 0].astype(np.int64)); rb = (kp[ib][:, 1].astype(np.int64) * W + kp[ib][:, 0].astype(np.int64))
                    mt.append(set(zip(ra[q].tolist(), rb[t].tolist())))
                sel[name] = ([set((k[:, 1].astype(np.int64) * W + k[:, 0].astype(np.int64)).tolist()) for k in kp], mt)
                fe.close()
            rec = {"threshold": thr, "max_keypoints": N}
            for name in ("wino", "f16x2"):
                for part, lo, hi in (("all", 0, NI), ("synthetic", 0, n_syn), ("real_derived", n_syn, NI)):
                    kt = sum(len(sel["f32"][0][i]) for i in range(lo, hi)); kd = sum(len(sel["f32"][0][i] ^ sel[name][0][i]) for i in range(lo, hi))
                    ki = sum(1 for i in range(lo, hi) if sel["f32"][0][i] ^ sel[name][0][i])
                    plo, phi = (0, len(pairs)) if part == "all" else ((0, n_syn // 2) if part == "synthetic" else (n_syn // 2, len(pairs)))
                    mtot = sum(len(sel["f32"][1][p]) for p in range(plo, phi)); md = sum(len(sel["f32"][1][p] ^ sel[name][1][p]) for p in range(plo, phi))
                    rec["%s_vs_f32_%s" % (name, part)] = {"keypoints": kt, "keypoints_in_one_mode_only": kd, "per_1e4_keypoints": round(1e4 * kd / max(kt, 1), 3),
                                                          "images_with_a_difference": ki, "matches": mtot, "matches_in_one_mode_only": md,
                                                          "per_1e4_matches": round(1e4 * md / max(mtot, 1), 3)}
            out["configs"].append(rec)
            print("thr %.3f N %d: wino %s  f16x2 %s" % (thr, N, rec["wino_vs_f32_all"], rec["f16x2_vs_f32_all"]), file=sys.stderr, flush=True)
    out["seconds"] = round(time.time() - t0, 1)
    out["note"] = ("symmetric differences of raster-index sets (keypoints per image; matches per pair as (index, index) pairs) against the exact fp32 mode; seeded random-init "
                   "weights compress the score distribution, so near-ties at the top-K cut are far more frequent than with a trained network")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
