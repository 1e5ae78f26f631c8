"""the LK optical-flow tracker (opticaltrack_utils.cpp:173-279,375-493,526-542).
CPU part: the oracle (oracle/d2fe_oracle_lk.c) against independent numpy restatements and ground-truth motion.
GPU part (-m gpu): the HIP kernels behind include/d2fe.h against the oracle, bit for bit (u8 / int / fp32 bit patterns)."""
import os

import numpy as np
import pytest

from d2slam_amd.synth import synth_image, synth_stereo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden.npz")


# ---------------------------------------------------------------------------------------------------------------- helpers
def _reflect101(i, n):
    i = np.asarray(i)
    i = np.where(i < 0, -i, i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def _pyr_down_np(img):
    """5x5 binomial, reflect-101, exact integer sum / 256, round half to even -- written with numpy gathers."""
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)
    ys = _reflect101(2 * np.arange(dh)[:, None] + np.arange(-2, 3)[None, :], h)      # [dh,5]
    xs = _reflect101(2 * np.arange(dw)[:, None] + np.arange(-2, 3)[None, :], w)      # [dw,5]
    a = img.astype(np.int64)
    rows = (a[:, xs] * k[None, None, :]).sum(-1)                                     # [h,dw]
    s = (rows[ys, :] * k[None, :, None]).sum(1)                                      # [dh,dw]
    return np.rint(s / 256.0).astype(np.uint8)                                       # np.rint = half to even; s/256 exact in fp64


def _shift_image(img, dx, dy):
    """img(x - dx, y - dy) with fp64 bilinear interpolation (content moves by (+dx, +dy))."""
    h, w = img.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    sx, sy = np.clip(xx - dx, 0, w - 1), np.clip(yy - dy, 0, h - 1)
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    fx, fy = sx - x0, sy - y0
    a = img.astype(np.float64)
    v = a[y0, x0] * (1 - fx) * (1 - fy) + a[y0, x1] * fx * (1 - fy) + a[y1, x0] * (1 - fx) * fy + a[y1, x1] * fx * fy
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


_CIRC = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
         (-2, 2), (-1, 3)]


def _fast_strength_np(img):
    """max over 9-arcs of min(|q - v|) for all-brighter / all-darker arcs (0 where undefined), interior pixels only."""
    h, w = img.shape
    a = img.astype(np.int32)
    d = np.stack([np.roll(np.roll(a, -dy, 0), -dx, 1) - a for dx, dy in _CIRC])     # q - v, [16,h,w]
    best = np.full((h, w), -256, np.int32)
    for s in range(16):
        arc = d[[(s + k) % 16 for k in range(9)]]
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    return best


def _fast_by_region_np(img, features, cols, rows, thr=10):
    h, w = img.shape
    sw, sh = w // cols, h // rows
    kps = []
    for i in range(cols):
        for j in range(rows):
            roi = img[sh * j:sh * (j + 1), sw * i:sw * (i + 1)]
            m = _fast_strength_np(roi)
            score = np.zeros((sh, sw), np.int32)
            cnt = 0
            for y in range(3, sh - 3):
                for x in np.nonzero(m[y, 3:sw - 3] > thr)[0] + 3:
                    if cnt < features:
                        score[y, x] = m[y, x] - 1
                        cnt += 1
            for y in range(3, sh - 3):
                for x in np.nonzero(score[y] > 0)[0]:
                    nb = score[y - 1:y + 2, x - 1:x + 2].copy(); s = nb[1, 1]; nb[1, 1] = -1
                    if s > nb.max():
                        kps.append((-int(s), len(kps), x + sw * i, y + sh * j))
    kps.sort()
    return np.array([[k[2], k[3]] for k in kps[:features]], np.float32).reshape(-1, 2), np.array([-k[0] for k in kps[:features]], np.int32)


# -------------------------------------------------------------------------------------------------------------- CPU: oracle
def test_pyr_down_oracle_vs_numpy(orc):
    for (h, w, seed) in ((480, 640, 1), (101, 77, 2), (400, 800, 3)):
        img = synth_image(h, w, seed)
        pyr = orc.pyr_build(img, 2)
        total, off, ws, hs = orc.pyr_layout(w, h, 2)
        assert np.array_equal(pyr[:h * w].reshape(h, w), img)
        l1 = _pyr_down_np(img)
        assert (hs[1], ws[1]) == l1.shape and np.array_equal(pyr[off[1]:off[2]].reshape(hs[1], ws[1]), l1)
        assert np.array_equal(pyr[off[2]:total].reshape(hs[2], ws[2]), _pyr_down_np(l1))


def test_lk_oracle_recovers_motion(orc):
    img = synth_image(480, 640, 7)
    pts, _ = orc.fast_by_region(img, 120)
    p0 = orc.pyr_build(img)
    for dx, dy in ((5.0, 0.0), (-3.25, 2.5), (11.5, -7.75)):
        cur = _shift_image(img, dx, dy)
        out, st = orc.lk_track(p0, orc.pyr_build(cur), 640, 480, pts, pts)
        inside = (pts[:, 0] > 40) & (pts[:, 0] < 600) & (pts[:, 1] > 40) & (pts[:, 1] < 440)
        good = st.astype(bool) & inside
        assert good.sum() > 0.7 * inside.sum()
        err = np.abs(out[good] - pts[good] - np.array([dx, dy], np.float32))
        assert np.median(err) < 0.05 and err.max() < 0.5
    # a point outside the image, and one on a flat patch, fail
    flat = np.full((480, 640), 128, np.uint8)
    pf = orc.pyr_build(flat)
    out, st = orc.lk_track(pf, pf, 640, 480, [[320, 240], [-5, 10], [700, 100]], [[320, 240], [-5, 10], [700, 100]])
    assert st.tolist() == [0, 0, 0]


def test_lk_oracle_left_right_types(orc):
    """type 1/2: the reverse track starts from the forward result shifted back by move_cols (opticaltrack_utils.cpp:246-255)."""
    img = synth_image(400, 800, 9)
    move = np.float32(800 * 90.0 / 200.0)
    cur = _shift_image(img, float(move) + 2.0, 0.0)[:, :800]
    pts, _ = orc.fast_by_region(img[:, :400], 60, 2, 2)
    pts = pts[(pts[:, 0] < 330) & (pts[:, 0] > 30)]
    init = pts.copy(); init[:, 0] += move
    out, st = orc.lk_track(orc.pyr_build(img), orc.pyr_build(cur), 800, 400, pts, init, track_type=1, move_cols=float(move))
    assert st.sum() > 0.5 * len(pts)
    err = np.abs(out[st > 0] - pts[st > 0] - np.array([float(move) + 2.0, 0], np.float32))
    assert np.median(err) < 0.1
    # with type 0 the reverse track starts `move` columns away and the 0.5 px forward/backward test rejects (almost) all
    _, st0 = orc.lk_track(orc.pyr_build(img), orc.pyr_build(cur), 800, 400, pts, init, track_type=0, move_cols=float(move))
    assert st0.sum() < 0.2 * max(st.sum(), 1)


def test_fast_oracle_vs_numpy(orc):
    for (h, w, seed, feats, cols, rows) in ((120, 160, 1, 40, 2, 2), (96, 200, 2, 500, 3, 1), (240, 320, 3, 25, 3, 4)):
        img = synth_image(h, w, seed)
        xy, resp = orc.fast_by_region(img, feats, cols, rows)
        rxy, rresp = _fast_by_region_np(img, feats, cols, rows)
        assert len(xy) > 0 and np.array_equal(xy, rxy) and np.array_equal(resp, rresp)
        assert (np.diff(resp) <= 0).all() and resp.min() >= 10


def test_good_features_oracle_properties(orc):
    img = synth_image(240, 320, 5)
    eig = orc.min_eigen(img)
    # independent fp64 evaluation of the same definition
    a = img.astype(np.float64)
    P = np.pad(a, 1, mode="reflect")
    sc = 1.0 / (4 * 3 * 255.0)
    dx = sc * ((P[:-2, 2:] - P[:-2, :-2]) + 2 * (P[1:-1, 2:] - P[1:-1, :-2]) + (P[2:, 2:] - P[2:, :-2]))
    dy = sc * ((P[2:, :-2] + 2 * P[2:, 1:-1] + P[2:, 2:]) - (P[:-2, :-2] + 2 * P[:-2, 1:-1] + P[:-2, 2:]))

    def box(m):
        Q = np.pad(m, 1, mode="reflect")
        return sum(Q[i:i + 240, j:j + 320] for i in range(3) for j in range(3))
    A, B, Cc = box(dx * dx) * 0.5, box(dx * dy), box(dy * dy) * 0.5
    ref = (A + Cc) - np.sqrt((A - Cc) ** 2 + B ** 2)
    assert np.abs(eig - ref).max() < 1e-6 * max(ref.max(), 1e-3) + 1e-9
    pts = orc.good_features(img, 80, 0.01, 12.0)
    assert 10 < len(pts) <= 80
    d = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1) + np.eye(len(pts)) * 1e9
    assert d.min() >= 12.0
    v = eig[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    assert (np.diff(v) <= 0).all() and v.min() > 0.01 * eig.max()
    # every accepted corner is a 3x3 local maximum
    for x, y in pts.astype(int):
        assert eig[y, x] == eig[y - 1:y + 2, x - 1:x + 2].max()


def test_lk_golden(orc):
    """Known-answer vectors generated once by tests/golden/make_golden.py (the reference has none for this path)."""
    g = np.load(GOLD)
    img0, img1 = g["img0"], g["img1"]
    assert np.array_equal(orc.pyr_build(img0, 2), g["pyr0"])
    xy, resp = orc.fast_by_region(img0, 60, 3, 4)
    assert np.array_equal(xy, g["fast_xy"]) and np.array_equal(resp, g["fast_resp"])
    assert np.array_equal(orc.good_features(img0, 50, 0.01, 15.0), g["gftt_xy"])
    out, st = orc.lk_track(orc.pyr_build(img0), orc.pyr_build(img1), img0.shape[1], img0.shape[0], g["fast_xy"], g["fast_xy"])
    assert np.array_equal(st, g["lk_status"])
    assert np.array_equal(out.view(np.uint32), g["lk_pts"].view(np.uint32))


# ---------------------------------------------------------------------------------------------------------------- GPU
def _fe():
    from d2slam_amd import api
    return api, api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))


@pytest.mark.gpu
def test_pyramid_gpu(orc):
    api, fe = _fe()
    for (h, w, seed) in ((480, 640, 11), (400, 800, 12), (101, 77, 13)):
        img = synth_image(h, w, seed)
        f = api.buildImagePyramid(fe, img, 2)
        total, off, ws, hs = orc.pyr_layout(w, h, 2)
        ref = orc.pyr_build(img, 2)
        for l in range(3):
            end = off[l + 1] if l < 2 else total
            assert np.array_equal(f.level(l), ref[off[l]:end].reshape(hs[l], ws[l]))
        f.close()
    fe.close()


@pytest.mark.gpu
def test_lk_track_gpu_bitexact(orc):
    api, fe = _fe()
    for seed, (h, w) in enumerate(((480, 640), (400, 800), (480, 640))):
        l, r = synth_stereo(h, w, seed=20 + seed)
        pts, _ = orc.fast_by_region(l, 150)
        rng = np.random.RandomState(seed)
        extra = np.stack([rng.uniform(-20, w + 20, 40), rng.uniform(-20, h + 20, 40)], 1).astype(np.float32)   # incl. outside points
        pts = np.concatenate([pts, extra])
        init = pts + rng.uniform(-3, 3, pts.shape).astype(np.float32)
        fl, fr = api.buildImagePyramid(fe, l), api.buildImagePyramid(fe, r)
        pl, pr = orc.pyr_build(l), orc.pyr_build(r)
        for ttype, mv in ((0, 0.0), (1, 12.5), (2, 12.5)):
            got, st = api.lk_track(fe, fl, fr, pts, init, ttype, mv)
            ref, rst = orc.lk_track(pl, pr, w, h, pts, init, ttype, mv)
            assert np.array_equal(st, rst)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))          # fp32 bit patterns
            assert ttype != 0 or st.sum() > 20
        fl.close(); fr.close()
    fe.close()


@pytest.mark.gpu
def test_optical_flow_track_pyr_gpu(orc):
    """The reference-shaped wrapper: compaction by status, ids carried along (opticaltrack_utils.cpp:173-279)."""
    api, fe = _fe()
    img = synth_image(480, 640, 31)
    cur = _shift_image(img, 4.5, -2.0)
    prev = api.buildImagePyramid(fe, img)
    pts = api.detectFastByRegion(fe, prev, 150, 3, 4)
    lk = {"lk_pts": pts, "lk_ids": np.arange(100, 100 + len(pts)), "pyr": prev}
    ret = api.opticalflowTrackPyr(fe, cur, lk, api.WHOLE_IMG_MATCH)
    assert len(ret["lk_pts"]) > 0.6 * len(pts) and len(ret["lk_ids"]) == len(ret["lk_pts"])
    src = pts[ret["lk_ids"] - 100]
    assert np.median(np.abs(ret["lk_pts"] - src - np.array([4.5, -2.0], np.float32))) < 0.05
    ref, rst = orc.lk_track(orc.pyr_build(img), orc.pyr_build(cur), 640, 480, pts, pts)
    assert np.array_equal(ret["lk_pts"], ref[rst > 0])
    # empty input -> empty output, pyramid still built
    e = api.opticalflowTrackPyr(fe, cur, {"lk_pts": np.zeros((0, 2), np.float32), "pyr": prev})
    assert len(e["lk_pts"]) == 0 and e["pyr"].level(2).shape == (120, 160)
    fe.close()


@pytest.mark.gpu
def test_fast_gpu(orc):
    api, fe = _fe()
    for (h, w, seed, feats, cols, rows) in ((480, 640, 41, 150, 3, 4), (400, 800, 42, 300, 4, 3), (480, 640, 43, 20, 3, 4),
                                            (120, 160, 44, 1000, 1, 1)):
        img = synth_image(h, w, seed)
        f = api.buildImagePyramid(fe, img)
        xy, resp = api.detectFastByRegion(fe, f, feats, cols, rows, with_response=True)
        rxy, rresp = orc.fast_by_region(img, feats, cols, rows)
        assert len(rxy) > 0 and np.array_equal(xy, rxy) and np.array_equal(resp, rresp)
        f.close()
    fe.close()


@pytest.mark.gpu
def test_good_features_and_detect_points_gpu(orc):
    api, fe = _fe()
    for (h, w, seed, n, md) in ((480, 640, 51, 150, 20.0), (400, 800, 52, 60, 35.0), (240, 320, 53, 0, 0.0)):
        img = synth_image(h, w, seed)
        f = api.buildImagePyramid(fe, img)
        got = api.goodFeaturesToTrack(fe, f, n, 0.01, md)
        ref = orc.good_features(img, n, 0.01, md)
        assert len(ref) > 0 and np.array_equal(got, ref)
        f.close()
    # detectPoints: nothing when fewer than a quarter are missing; otherwise new points keep feature_min_dist to all others
    img = synth_image(480, 640, 54)
    f = api.buildImagePyramid(fe, img)
    have = api.goodFeaturesToTrack(fe, f, 130, 0.01, 20.0)
    assert len(api.detectPoints(fe, f, have[:120], 150)) == 0
    for use_fast in (False, True):
        new = api.detectPoints(fe, f, have[:50], 150, use_fast=use_fast)
        assert 0 < len(new) <= 100
        allp = np.concatenate([have[:50], new])
        d = np.linalg.norm(allp[:, None] - allp[None], axis=-1) + np.eye(len(allp)) * 1e9
        assert d[50:].min() >= 20.0
    f.close(); fe.close()


@pytest.mark.gpu
def test_lk_track_batch_gpu(orc):
    """d2fe_lk_track_batch: several (prev, cur) pairs of different sizes and types in one launch == the single calls == the oracle."""
    api, fe = _fe()
    jobs, refs = [], []
    frames = []
    for k, (h, w, ttype, mv) in enumerate(((480, 640, 0, 0.0), (400, 800, 1, 9.0), (400, 800, 2, 9.0), (240, 320, 0, 0.0))):
        l, r = synth_stereo(h, w, seed=60 + k)
        pts, _ = orc.fast_by_region(l, 90 + 10 * k)
        rng = np.random.RandomState(k)
        init = pts + rng.uniform(-2, 2, pts.shape).astype(np.float32)
        fl, fr = api.buildImagePyramid(fe, l), api.buildImagePyramid(fe, r)
        frames += [fl, fr]
        jobs.append((fl, fr, pts, init, ttype, mv))
        refs.append(orc.lk_track(orc.pyr_build(l), orc.pyr_build(r), w, h, pts, init, ttype, mv))
    jobs.append((frames[0], frames[1], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), 0, 0.0))   # an empty pair is fine
    out = api.lk_track_batch(fe, jobs)
    assert len(out) == len(jobs) and len(out[-1][0]) == 0
    for (got, st), (ref, rst), job in zip(out, refs, jobs):
        assert np.array_equal(st, rst) and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        one, ost = api.lk_track(fe, job[0], job[1], job[2], job[3], job[4], job[5])
        assert np.array_equal(ost, st) and np.array_equal(one.view(np.uint32), got.view(np.uint32))
    with pytest.raises(api.D2FEError):      # geometry mismatch between the frames of a pair
        api.lk_track_batch(fe, [(frames[0], frames[2], jobs[0][2], jobs[0][3], 0, 0.0)])
    for f in frames:
        f.close()
    fe.close()
