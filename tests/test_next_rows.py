"""SURVEY.md section 8(f) 'next' rows: fisheye undistort + gain, NetVLAD database search/gate, int8 wire codec.
CPU part pins the oracle against independent numpy; GPU part (-m gpu) checks the HIP kernels against the oracle."""
import numpy as np
import pytest

from d2slam_amd.synth import synth_image


def _maps(dh, dw, sh, sw, seed):
    """Cylinder-like synthetic maps incl. coordinates outside the source (constant-0 border)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:dh, 0:dw].astype(np.float32)
    mx = (xx / dw * (sw + 40) - 20 + 6 * np.sin(yy / 17.0)).astype(np.float32)
    my = (yy / dh * (sh + 30) - 15 + 5 * np.cos(xx / 23.0)).astype(np.float32)
    gain = rng.uniform(0.8, 1.6, size=(dh, dw)).astype(np.float32)
    return mx, my, gain


def test_undistort_oracle_vs_numpy(orc):
    src = synth_image(200, 320, 1)
    mx, my, gain = _maps(100, 200, 200, 320, 0)
    got = orc.undistort(src, mx, my, gain).astype(np.int32)
    x1 = np.floor(mx).astype(int); y1 = np.floor(my).astype(int)
    P = np.pad(src.astype(np.float64), 2)

    def S(y, x):
        ok = (y >= 0) & (y < 200) & (x >= 0) & (x < 320)
        return np.where(ok, P[np.clip(y, -2, 201) + 2, np.clip(x, -2, 321) + 2], 0.0)
    fx = mx.astype(np.float64) - x1; fy = my.astype(np.float64) - y1
    v = S(y1, x1) * (1 - fx) * (1 - fy) + S(y1, x1 + 1) * fx * (1 - fy) + S(y1 + 1, x1) * (1 - fx) * fy + S(y1 + 1, x1 + 1) * fx * fy
    u = np.clip(np.rint(v), 0, 255)
    ref = np.clip(np.rint(u * gain.astype(np.float64)), 0, 255).astype(np.int32)
    d = np.abs(got - ref)
    assert d.max() <= 2 and (d == 0).mean() > 0.99        # only rounding-boundary pixels may differ from the fp64 evaluation
    assert orc.undistort(src, mx, my, None).max() > 0


def test_db_query_oracle(orc):
    rng = np.random.RandomState(0)
    db = rng.randn(300, 64).astype(np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[120] + 0.05 * rng.randn(64).astype(np.float32); q /= np.linalg.norm(q)
    label, sim, labels, sims = orc.db_query(db, q, max_index=10, thres=0.5)
    s = db.astype(np.float64) @ q.astype(np.float64)
    assert label == 120 and abs(sim - s[120]) < 1e-5
    assert labels.tolist() == np.argsort(-s, kind="stable")[:15].tolist()
    # too recent (label > ntotal - max_index) is skipped by the gate
    q2 = db[295]
    label2, _, labels2, _ = orc.db_query(db, q2, max_index=10, thres=0.5)
    assert labels2[0] == 295 and label2 == -1


def test_int8_codec_oracle(orc):
    rng = np.random.RandomState(1)
    x = rng.randn(50 * 256).astype(np.float32); x /= 16
    q = orc.quant_int8(x)
    m = np.abs(x).max()
    assert np.array_equal(q, np.trunc(x / m * np.float32(127)).astype(np.int8))
    back = orc.dequant_int8(q, landmark_num=50)
    seg = (q.astype(np.float64) / 127.0).reshape(-1, 32)
    ref = seg.copy(); ref[:50] /= np.linalg.norm(ref[:50], axis=1, keepdims=True)     # only the first landmark_num segments
    assert np.abs(back.reshape(-1, 32) - ref).max() < 1e-6
    g = rng.randn(4096).astype(np.float32); g /= np.linalg.norm(g)
    gq = orc.quant_int8(g, double_max=True)
    gb = orc.dequant_int8(gq, -1)
    assert abs(np.linalg.norm(gb) - 1) < 1e-5 and float(gb @ g) > 0.99


@pytest.mark.gpu
def test_undistort_gpu(orc):
    from d2slam_amd import api
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    src = synth_image(800, 1280, 4)                      # quadcam raw frame size (quadcam_single.yaml:18-23)
    mx, my, gain = _maps(400, 800, 800, 1280, 2)
    for g in (gain, None):
        got = fe.undistort(src, mx, my, g)
        assert np.array_equal(got, orc.undistort(src, mx, my, g))      # u8, bit-exact
    fe.close()


@pytest.mark.gpu
def test_db_gpu(orc):
    from d2slam_amd import api
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    rng = np.random.RandomState(3)
    for dim, n in ((1024, 700), (4096, 257)):
        vec = rng.randn(n, dim).astype(np.float32); vec /= np.linalg.norm(vec, axis=1, keepdims=True)
        db = api.FlatIPDatabase(fe, dim, capacity=1024)
        assert db.add(vec[:100]) == 0 and db.add(vec[100:]) == 100 and db.ntotal == n
        for target, mi, thr in ((50, 10, 0.5), (n - 3, 10, 0.5), (20, 0, 0.999999)):
            q = vec[target] + 0.02 * rng.randn(dim).astype(np.float32); q /= np.linalg.norm(q)
            label, sim = db.query_gated(q, mi, thr)
            rl, rs, rlabels, rsims = orc.db_query(vec, q, mi, thr)
            assert label == rl and (label < 0 or abs(sim - rs) < 1e-5)
            sims, labels = db.search(q, len(rlabels))
            assert labels[0].tolist() == rlabels.tolist() and np.abs(sims[0] - rsims).max() < 1e-5
        with pytest.raises(api.D2FEError):
            db.add(np.zeros((2000, dim), np.float32))     # capacity exceeded -> loud
        db.close()
    fe.close()


@pytest.mark.gpu
def test_int8_codec_gpu(orc):
    from d2slam_amd import api
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    rng = np.random.RandomState(5)
    x = rng.randn(200 * 256).astype(np.float32); x /= np.linalg.norm(x.reshape(200, 256), axis=1).repeat(256)
    q = fe.quantize_int8(x)
    assert np.array_equal(q, orc.quant_int8(x))                             # bytes: bit-exact
    assert np.abs(fe.dequantize_int8(q, 200) - orc.dequant_int8(q, 200)).max() <= 1e-6
    g = rng.randn(4096).astype(np.float32); g /= np.linalg.norm(g)
    gq = fe.quantize_int8(g, double_max=True)
    assert np.array_equal(gq, orc.quant_int8(g, double_max=True))
    assert np.abs(fe.dequantize_int8(gq, -1) - orc.dequant_int8(gq, -1)).max() <= 1e-6
    # round trip property: cosine with the original stays high (what cross-agent matching relies on)
    back = fe.dequantize_int8(q, 200).reshape(-1, 32)[:200]          # the 200 re-normalised 32-float segments
    xs = x.reshape(-1, 32)[:200]
    cos = (back * xs).sum(1) / np.linalg.norm(xs, axis=1)
    assert cos.min() > 0.98
    fe.close()


# ---- (f)-1 map generation (fisheye_undistort.h:458-500,559-660) --------------------------------------------------------------------
# cam0 of config/quadcam/quad_cam_calib-camchain-imucam-7-inch-n3.yaml (camera_model omni, distortion radtan)
QUADCAM_MEI = dict(xi=2.2176903753419963, k1=-0.17703529535292872, k2=0.7517933338735744, p1=-0.0008911425891703079,
                   p2=2.1653595535258756e-05, gamma1=1162.5434300524314, gamma2=1161.839362615319, u0=660.6393183718625,
                   v0=386.1663300322095)
_MEI_KEYS = ("xi", "k1", "k2", "p1", "p2", "gamma1", "gamma2", "u0", "v0")


def _mei_np(cam, X, Y, Z):
    n = np.sqrt(X * X + Y * Y + Z * Z)
    z = Z + cam["xi"] * n
    x, y = X / z, Y / z
    r2 = x * x + y * y
    rad = cam["k1"] * r2 + cam["k2"] * r2 * r2
    dx = x * rad + 2 * cam["p1"] * x * y + cam["p2"] * (r2 + 2 * x * x)
    dy = y * rad + 2 * cam["p2"] * x * y + cam["p1"] * (r2 + 2 * y * y)
    return cam["gamma1"] * (x + dx) + cam["u0"], cam["gamma2"] * (y + dy) + cam["v0"]


def test_gen_maps_oracle_vs_numpy(orc):
    cam9 = [QUADCAM_MEI[k] for k in _MEI_KEYS]
    W, H, fov = 800, 400, 200.0                                       # quadcam_single.yaml:18-23
    mx, my = orc.gen_cylinder_map(cam9, W, H, fov)
    f = W / np.deg2rad(fov)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    phi = (xx - W // 2) / f
    z = np.where(np.abs(phi) > np.pi / 2, -1.0, 1.0)
    X = z * np.tan(phi); rho = np.sqrt(X * X + z * z)
    u, v = _mei_np(QUADCAM_MEI, X, (yy - H // 2) / f * rho, z)
    assert np.abs(mx - u).max() < 2e-4 and np.abs(my - v).max() < 2e-4
    assert abs(mx[H // 2, W // 2] - QUADCAM_MEI["u0"]) < 1e-3 and abs(my[H // 2, W // 2] - QUADCAM_MEI["v0"]) < 1e-3
    assert (np.diff(mx[H // 2]) > 0).all() and (np.diff(my[:, W // 2]) > 0).all()          # monotone along the axes
    # pinhole virtual camera rotated -45 deg about y (a side camera of generateAllUndistMap, :429-437)
    a = -np.pi / 4
    q = [np.cos(a / 2), 0.0, np.sin(a / 2), 0.0]
    px, py = orc.gen_pinhole_map(cam9, q, 600, 300, 300.0)
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    yy, xx = np.mgrid[0:300, 0:600].astype(np.float64)
    P = np.stack([xx - 300, yy - 150, np.full_like(xx, 300.0)], -1) @ R.T
    u, v = _mei_np(QUADCAM_MEI, P[..., 0], P[..., 1], P[..., 2])
    assert np.abs(px - u).max() < 2e-4 and np.abs(py - v).max() < 2e-4


@pytest.mark.gpu
def test_gen_maps_gpu(orc):
    from d2slam_amd import api
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    cam9 = [QUADCAM_MEI[k] for k in _MEI_KEYS]
    gx, gy = fe.gen_cylinder_map(QUADCAM_MEI, 800, 400, 200.0)
    rx, ry = orc.gen_cylinder_map(cam9, 800, 400, 200.0)
    # fp64 evaluation rounded to fp32: the device tan()/sqrt() may differ from glibc's in the last fp64 bit -> at most one fp32 ulp
    assert np.abs(gx - rx).max() <= 1.3e-4 and np.abs(gy - ry).max() <= 1.3e-4
    assert (gx == rx).mean() > 0.999 and (gy == ry).mean() > 0.999
    q = [np.cos(np.pi / 8), 0.0, np.sin(np.pi / 8), 0.0]
    px, py = fe.gen_pinhole_map(cam9, q, 600, 300, 300.0)
    ox, oy = orc.gen_pinhole_map(cam9, q, 600, 300, 300.0)
    assert np.abs(px - ox).max() <= 1.3e-4 and np.abs(py - oy).max() <= 1.3e-4
    # end to end: raw 1280x800 frame -> cylinder image with device-generated maps == oracle with oracle maps (u8)
    src = synth_image(800, 1280, 9)
    got = fe.undistort(src, gx, gy, None)
    ref = orc.undistort(src, rx, ry, None)
    assert (got != ref).mean() < 1e-4 and np.abs(got.astype(int) - ref.astype(int)).max() <= 1
    fe.close()


# ---- A1 variant A / NetVLAD prep: BGR -> gray + resize (superpoint_onnx.cpp:76-83, mobilenetvlad_onnx.h:51-59) -------------------
def test_prepare_gray_oracle_vs_float(orc):
    bgr = np.stack([synth_image(300, 400, s) for s in (1, 2, 3)], -1)
    g = orc.bgr2gray(bgr)
    assert np.abs(g.astype(int) - np.rint(bgr[..., 0] * 0.114 + bgr[..., 1] * 0.587 + bgr[..., 2] * 0.299)).max() <= 1
    img = synth_image(400, 800, 1)
    for (dw, dh) in ((640, 480), (1000, 500), (123, 77)):
        r = orc.resize_linear_u8(img, dw, dh)
        yy, xx = np.mgrid[0:dh, 0:dw]
        fx = (xx + 0.5) * (800 / dw) - 0.5; fy = (yy + 0.5) * (400 / dh) - 0.5
        x0 = np.floor(fx).astype(int); y0 = np.floor(fy).astype(int); ax = fx - x0; ay = fy - y0
        c = lambda v, n: np.clip(v, 0, n - 1)
        a = img.astype(np.float64)
        ref = (a[c(y0, 400), c(x0, 800)] * (1 - ax) * (1 - ay) + a[c(y0, 400), c(x0 + 1, 800)] * ax * (1 - ay)
               + a[c(y0 + 1, 400), c(x0, 800)] * (1 - ax) * ay + a[c(y0 + 1, 400), c(x0 + 1, 800)] * ax * ay)
        assert np.abs(r.astype(int) - np.rint(ref).astype(int)).max() <= 1      # 11-bit fixed point vs exact bilinear
    # exact 2x decimation = INTER_AREA
    half = orc.resize_linear_u8(img, 400, 200)
    assert np.array_equal(half, ((img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2).astype(np.uint8))


@pytest.mark.gpu
def test_prepare_gray_gpu(orc):
    from d2slam_amd import api
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    bgr = np.stack([synth_image(400, 800, s) for s in (4, 5, 6)], -1)
    gray = orc.bgr2gray(bgr)
    assert np.array_equal(fe.prepare_gray(bgr, 800, 400), gray)                                   # colour only
    for (dw, dh) in ((640, 480), (400, 200), (1280, 800), (123, 77)):
        assert np.array_equal(fe.prepare_gray(bgr, dw, dh), orc.resize_linear_u8(gray, dw, dh))   # colour + resize, fused
        assert np.array_equal(fe.prepare_gray(gray, dw, dh), orc.resize_linear_u8(gray, dw, dh))
    fe.close()
