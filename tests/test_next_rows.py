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
