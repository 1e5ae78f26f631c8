"""Index-parity study of the arithmetic modes (host logic shared by tools/mode_disagreement.py -- the 1056-image run behind profiles/r04_mode_disagreement.json -- and
bench.py, which runs a 128-image subset of it inside every default run: `index_parity_in_run`).

Keypoint and match index sets of the Winograd fp32 mode (`value`) and of the fp16 hi/lo mode against the bitwise-exact direct-convolution fp32 mode on 640x480 frames:
synthetic stereo pairs plus frames derived from the three real crops of the reference's sample_data/fisheye.jpg that tests/golden/reference_headline.npz carries (flips,
mirror-padded re-crops, gain changes: real image statistics, no new reference content).  Pairs: synthetic L<->R, real frame <-> the frame shifted by (3, 5) px.
Symmetric differences of raster-index sets (keypoints per image; matches per pair as (index, index) pairs)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def frames(n_synthetic_pairs, n_real=None, seed0=5000):
    """(images [NI,H,W] u8, pairs [(ia, ib)], number of synthetic images)"""
    from d2slam_amd.synth import synth_stereo
    imgs, pairs = [], []
    for s in range(n_synthetic_pairs):
        l, r = synth_stereo(H, W, seed=seed0 + s)
        pairs.append((len(imgs), len(imgs) + 1)); imgs += [l, r]
    n_syn = len(imgs)
    real = real_frames()
    for f in (real if n_real is None else real[:n_real]):
        pairs.append((len(imgs), len(imgs) + 1)); imgs += [f, np.roll(f, (3, 5), (0, 1))]
    return np.ascontiguousarray(np.stack(imgs)), pairs, n_syn


def weights_for(thr):
    """the seeded weights with a dustbin bias that leaves enough candidates above `thr` (0.15: as bench.py's quadcam leg)"""
    from d2slam_amd.weights import synthetic_superpoint_weights
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    if thr >= 0.1:
        Wt, b = w["convPb"]; b = b.copy(); b[64] -= np.float32(3.5); w["convPb"] = (Wt, b)
    return w


def select(api, imgs, pairs, thr, N, prec, batch=32, device_id=0):
    """(per-image raster-index sets, per-pair sets of (index, index) matches) of one arithmetic mode"""
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=N, input_width=W, input_height=H, max_batch=batch, precision=prec, keypoint_threshold=thr, device_id=device_id))
    fe.load_superpoint(weights_for(thr))
    kp, ds = [], []
    for i0 in range(0, len(imgs), batch):
        for k, s, d in fe.extract_batch(imgs[i0:i0 + batch], cap=N):
            kp.append(k); ds.append(d)
    mt = []
    for ia, ib in pairs:
        q, t, _ = fe.match_knn(ds[ia], ds[ib], 0.8)
        ra = (kp[ia][:, 1].astype(np.int64) * W + kp[ia][:, 0].astype(np.int64)); rb = (kp[ib][:, 1].astype(np.int64) * W + kp[ib][:, 0].astype(np.int64))
        mt.append(set(zip(ra[q].tolist(), rb[t].tolist())))
    fe.close()
    return [set((k[:, 1].astype(np.int64) * W + k[:, 0].astype(np.int64)).tolist()) for k in kp], mt


def compare(ref, other, lo, hi, plo, phi):
    kt = sum(len(ref[0][i]) for i in range(lo, hi)); kd = sum(len(ref[0][i] ^ other[0][i]) for i in range(lo, hi))
    ki = sum(1 for i in range(lo, hi) if ref[0][i] ^ other[0][i])
    mtot = sum(len(ref[1][p]) for p in range(plo, phi)); md = sum(len(ref[1][p] ^ other[1][p]) for p in range(plo, phi))
    return {"keypoints": kt, "keypoints_in_one_mode_only": kd, "per_1e4_keypoints": round(1e4 * kd / max(kt, 1), 3), "images_with_a_difference": ki,
            "matches": mtot, "matches_in_one_mode_only": md, "per_1e4_matches": round(1e4 * md / max(mtot, 1), 3)}


def study(api, imgs, pairs, n_syn, thr, N, batch=32, device_id=0):
    """one configuration: both fast modes against the exact mode, over all / synthetic / real-derived frames"""
    modes = {"f32": api.PREC_F32, "wino": api.PREC_F32_WINO, "f16x2": api.PREC_F16X2}
    sel = {name: select(api, imgs, pairs, thr, N, prec, batch, device_id) for name, prec in modes.items()}
    NI = len(imgs)
    rec = {"threshold": thr, "max_keypoints": N}
    for name in ("wino", "f16x2"):
        for part, lo, hi in (("all", 0, NI), ("synthetic", 0, n_syn), ("real_derived", n_syn, NI)):
            plo, phi = (0, len(pairs)) if part == "all" else ((0, n_syn // 2) if part == "synthetic" else (n_syn // 2, len(pairs)))
            rec["%s_vs_f32_%s" % (name, part)] = compare(sel["f32"], sel[name], lo, hi, plo, phi)
    return rec
