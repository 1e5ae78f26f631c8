"""Winograd F(4,3) x F(2,3) (4 x 2 output tiles: 3 multiplies per output where F(2x2,3x3) has 4) -- the experimental kernel conv_wino43.hip of the development
library and the oracle's restatement of its evaluation order (orc_conv3x3_wino43).

CPU: the restatement against the direct-convolution chain (orc_conv) and an fp64 convolution.  GPU: the kernel against the restatement, bit for bit."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    return o


def _layer(rng, h, w, cin, cout, n=1):
    x = np.maximum(rng.standard_normal((n, h, w, cin)).astype(np.float32), 0.0)    # post-ReLU activations
    wg = (rng.standard_normal((cout, cin, 3, 3)) * (0.6 / np.sqrt(cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    return x, wg, b


def _conv64(x, wg, b, relu):
    h, w, cin = x.shape
    xp = np.zeros((h + 2, w + 2, cin), np.float64); xp[1:-1, 1:-1] = x
    out = np.zeros((h, w, wg.shape[0]), np.float64) + b.astype(np.float64)
    for ky in range(3):
        for kx in range(3):
            out += xp[ky:ky + h, kx:kx + w] @ wg[:, :, ky, kx].astype(np.float64).T
    return np.maximum(out, 0) if relu else out


@pytest.mark.parametrize("h,w,cin,cout", [(13, 21, 64, 64), (16, 40, 128, 96), (8, 32, 64, 128), (2, 2, 64, 8), (5, 3, 64, 32), (60, 80, 128, 32)])
def test_oracle_wino43_close_to_direct(orc, h, w, cin, cout):
    x, wg, b = _layer(np.random.default_rng(h * w + cin), h, w, cin, cout)
    t = _conv64(x[0], wg, b, True)
    d = orc.conv(x[0], wg, b, True)
    v2 = orc.conv_wino(x[0], wg, b, True)
    v = orc.conv_wino43(x[0], wg, b, True)
    scale = max(1.0, float(np.abs(t).max()))
    e_d, e_2, e_43 = (float(np.abs(a - t).max()) / scale for a in (d, v2, v))
    assert e_43 <= 4e-6 and e_43 <= 4.0 * max(e_d, e_2, 2e-7), (e_d, e_2, e_43)     # same error class as the direct chain and F(2x2)
    assert np.abs(d - v).max() <= 6e-6 * scale
    z = orc.conv_wino43(np.zeros_like(x[0]), wg, b, True)                         # zero input: exact zeros + bias
    assert np.array_equal(z, np.broadcast_to(np.maximum(b, 0), z.shape))


W43_LAYERS = [  # n, H, W, Cin, Cout, pool
    (1, 16, 16, 64, 32, False), (2, 16, 64, 64, 64, False), (1, 24, 96, 64, 64, True), (1, 16, 32, 64, 128, False), (2, 8, 32, 128, 128, True),
    (1, 60, 80, 128, 256, False), (1, 30, 46, 64, 64, False), (1, 22, 34, 128, 128, True), (3, 8, 32, 64, 65, False), (1, 10, 12, 128, 64, False)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,W,cin,cout,pool", W43_LAYERS)
def test_wino43_layer_bitwise(orc, monkeypatch, n, H, W, cin, cout, pool):
    """conv_wino43.hip through the development library's layer hook (D2FE_WINO43=1) against orc_conv3x3_wino43, bit for bit"""
    from d2slam_amd import api
    monkeypatch.setenv("D2FE_WINO43", "1")
    x, wg, b = _layer(np.random.default_rng(H * W + cin + cout), H, W, cin, cout, n)
    fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=16, input_width=64, input_height=64, max_batch=1))
    out, _ = fe.debug_conv3x3_wino(x, wg, b, pool=pool)
    fe.close()
    for i in range(n):
        ref = orc.conv_wino43(x[i], wg, b, True)
        if pool:
            ref = orc.maxpool2(ref)
        assert out[i].shape == ref.shape
        assert np.array_equal(out[i], ref), "image %d: max |diff| %g, %d of %d differ" % (i, np.abs(out[i] - ref).max(), int((out[i] != ref).sum()), ref.size)
