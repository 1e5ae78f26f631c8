"""Precision mode 2 (D2FE_PREC_F32_WINO): the eight 3x3 layers as Winograd F(2x2,3x3) on the fp32 matrix pipe.

CPU: the oracle's restatement of that evaluation order (orc_conv3x3_wino) against the direct-convolution chain (orc_conv).
GPU: the HIP kernels against the restatement, bit for bit, layer by layer and through the whole extractor; and the mode's
outputs against the direct oracle at the north-star tolerances (descriptors 1e-4, keypoint indices exact)."""
import numpy as np
import pytest

from d2slam_amd.synth import synth_stereo
from d2slam_amd.weights import synthetic_superpoint_weights


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    return o


@pytest.fixture(scope="module")
def api():
    from d2slam_amd import api as a
    return a


def _layer(rng, h, w, cin, cout, n=1):
    x = np.maximum(rng.standard_normal((n, h, w, cin)).astype(np.float32), 0.0)    # post-ReLU activations
    wg = (rng.standard_normal((cout, cin, 3, 3)) * (0.6 / np.sqrt(cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    return x, wg, b


@pytest.mark.parametrize("h,w,cin,cout", [(13, 21, 64, 64), (16, 40, 128, 96), (8, 32, 64, 128), (2, 2, 64, 8)])
def test_oracle_wino_close_to_direct(orc, h, w, cin, cout):
    x, wg, b = _layer(np.random.default_rng(h * w + cin), h, w, cin, cout)
    d = orc.conv(x[0], wg, b, True)
    v = orc.conv_wino(x[0], wg, b, True)
    assert np.abs(d - v).max() <= 4e-6 * max(1.0, float(np.abs(d).max()))
    # zero weights / zero input: exact zeros + bias
    z = orc.conv_wino(np.zeros_like(x[0]), wg, b, True)
    assert np.array_equal(z, np.broadcast_to(np.maximum(b, 0), z.shape))


def test_oracle_forward_wino_close_to_direct(orc):
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    img = synth_stereo(64, 96, seed=5)[0]
    a = orc.superpoint_forward(img, w)
    b = orc.superpoint_forward(img, w, wino=True)
    assert np.abs(a["semi"] - b["semi"]).max() <= 1e-6
    assert np.abs(a["desc"] - b["desc"]).max() <= 1e-5


def test_weight_transform_packing_cpu():
    """pack_weights_wino (host code of the library, no GPU needed): U = G g G^T in double, rounded once, in MFMA fragment order."""
    import ctypes as C
    from d2slam_amd import build
    lib = C.CDLL(build.build(dev=True))      # a test hook of the development library (include/d2fe_debug.h)
    lib.d2fe_debug_pack_wino.restype = C.c_long
    rng = np.random.default_rng(3)
    cout, cin = 72, 64                       # 72 -> padded to 128 channels: the padding must come out as zeros
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    out = np.full(16 * cin * 128, np.nan, np.float32)
    n = lib.d2fe_debug_pack_wino(w.ctypes.data_as(C.c_void_p), cout, cin, out.ctypes.data_as(C.c_void_p), C.c_long(out.size))
    assert n == out.size
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    U = np.einsum("ia,ocab,jb->ijoc", G, w.astype(np.float64), G)          # [i][j][co][ci]
    pk = out.reshape(128 // 32, cin // 2, 4, 64, 4)                           # [group][k-step][row i][lane][j]
    for grp in range(4):
        for ks in (0, 1, 5, 31):
            for lane in (0, 17, 32, 63):
                co, ci = grp * 32 + (lane & 31), 8 * (ks // 4) + 4 * (lane >> 5) + ks % 4
                ref = U[:, :, co, ci].astype(np.float32) if co < cout else np.zeros((4, 4), np.float32)
                assert np.allclose(pk[grp, ks, :, lane, :], ref, rtol=0, atol=1e-7 * max(1.0, float(np.abs(ref).max())))
    assert not np.isnan(out).any()


WINO_LAYERS = [  # n, H, W, Cin, Cout, pool
    (2, 16, 64, 64, 64, False), (1, 24, 96, 64, 64, True), (1, 16, 32, 64, 128, False), (2, 8, 32, 128, 128, True),
    (1, 60, 80, 128, 256, False), (1, 30, 46, 64, 64, False), (1, 22, 34, 128, 128, True), (3, 8, 32, 64, 65, False),
    (1, 10, 12, 128, 64, False)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,W,cin,cout,pool", WINO_LAYERS)
def test_wino_layer_bitwise(api, orc, n, H, W, cin, cout, pool):
    x, wg, b = _layer(np.random.default_rng(H * W + cin + cout), H, W, cin, cout, n)
    fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=16, input_width=64, input_height=64, max_batch=1))
    out, _ = fe.debug_conv3x3_wino(x, wg, b, pool=pool)
    fe.close()
    for i in range(n):
        ref = orc.conv_wino(x[i], wg, b, True)
        if pool:
            ref = orc.maxpool2(ref)
        assert out[i].shape == ref.shape
        assert np.array_equal(out[i], ref), "image %d: max |diff| %g" % (i, np.abs(out[i] - ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,n", [(96, 128, 2), (480, 640, 4), (120, 168, 1), (400, 800, 2), (512, 512, 2)])   # incl. every BASELINE geometry
def test_wino_extract_vs_oracles(api, orc, H, W, n):
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    imgs = np.stack([synth_stereo(H, W, seed=11 + i)[i & 1] for i in range(n)])
    cap = 200
    fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=cap, input_width=W, input_height=H, max_batch=n,
                                           precision=api.PREC_F32_WINO, keep_score_map=True, dense_descriptors=True))
    fe.load_superpoint(w)
    res = fe.extract_batch(imgs, cap=cap)
    semi = fe.debug_read("semi", (n, H, W))
    draw = fe.debug_read("desc_raw", (n, H // 8, W // 8, 256))
    fe.close()
    for i in range(n):
        fw = orc.superpoint_forward(imgs[i], w, wino=True)
        # bit for bit against the restatement of the mode's evaluation order
        assert np.array_equal(semi[i], fw["semi"]) and np.array_equal(draw[i], fw["desc_raw"])
        rk, rs, ri = orc.select_b(fw["semi"], 0.015, 1, cap)
        kps, sc, desc = res[i]
        assert np.array_equal(kps, rk) and np.array_equal(sc, rs)
        assert np.abs(desc - orc.sample_b(fw["desc"], rk)).max() <= 1e-6
        # north-star bars against the direct-convolution oracle: descriptors 1e-4, indices exact unless a score sits within
        # the rounding distance of the K-th score / the threshold
        fd = orc.superpoint_forward(imgs[i], w)
        eps = float(np.abs(fd["semi"] - fw["semi"]).max())
        assert eps <= 3e-6
        dk, ds, di = orc.select_b(fd["semi"], 0.015, 1, cap)
        gi = (kps[:, 1] * W + kps[:, 0]).astype(np.int64)
        flat = fd["semi"].reshape(-1)
        kth = ds[-1] if len(ds) == cap else 0.015
        for j in np.setxor1d(gi, di):
            assert abs(flat[j] - kth) <= 2 * eps + 1e-9 or abs(flat[j] - 0.015) <= 2 * eps + 1e-9
        common, ia, ib = np.intersect1d(gi, di, return_indices=True)
        assert len(common) >= len(di) - 2
        dd = orc.sample_b(fd["desc"], dk)
        assert np.abs(desc[ia] - dd[ib]).max() <= 1e-4
        assert np.abs(desc[ia] - dd[ib]).max() <= 1e-5


@pytest.mark.gpu
def test_wino_sparse_head_matches_dense_within_tolerance(api):
    # in this mode the sparse descriptor head (direct chains at the keypoint cells) and the dense Winograd head differ by rounding only
    H, W, n, cap = 120, 160, 4, 100
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    imgs = np.stack([synth_stereo(H, W, seed=21 + i)[i & 1] for i in range(n)])
    outs = []
    for dense in (True, False):
        fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=cap, input_width=W, input_height=H, max_batch=n,
                                               precision=api.PREC_F32_WINO, dense_descriptors=dense))
        fe.load_superpoint(w)
        outs.append(fe.extract_batch(imgs, cap=cap))
        fe.close()
    for (k0, s0, d0), (k1, s1, d1) in zip(*outs):
        assert np.array_equal(k0, k1) and np.array_equal(s0, s1)
        assert np.abs(d0 - d1).max() <= 1e-5


@pytest.mark.gpu
def test_wino_variant_a_matches_its_oracle(api, orc):
    # post-processing variant A (NMS2 + grid_sampler sampling) on top of the Winograd network: keypoints exact, descriptors 1e-6
    H, W, n, cap = 96, 128, 4, 60
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    imgs = np.stack([synth_stereo(H, W, seed=31 + i)[i & 1] for i in range(n)])
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=cap, input_width=W, input_height=H, max_batch=n, precision=api.PREC_F32_WINO,
                                           postproc=api.POSTPROC_A, nms_dist=4))
    fe.load_superpoint(w)
    res = fe.extract_batch(imgs, cap=cap)
    fe.close()
    for i in range(n):
        f = orc.superpoint_forward(imgs[i], w, wino=True)
        rk, rs = orc.nms2_a(f["semi"], 0.015, 4, cap)[:2]
        kps, sc, desc = res[i]
        assert np.array_equal(kps, rk) and np.array_equal(sc, rs)
        rd = orc.sample_a(f["desc"], rk, W, H)
        assert np.abs(desc - rd).max() <= 1e-5   # sparse head: direct chains at the keypoint cells vs the dense Winograd map


@pytest.mark.gpu
def test_wino_fused_conv1a_is_bit_identical(api, monkeypatch):
    # D2FE_FUSE1A=1: conv1a evaluated on the matrix pipe inside the Winograd conv1b's staging -- same bits as the two-kernel form
    H, W, n, cap = 120, 168, 4, 100
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    imgs = np.stack([synth_stereo(H, W, seed=41 + i)[i & 1] for i in range(n)])
    outs, trunks = [], []
    for fuse in ("0", "1"):
        monkeypatch.setenv("D2FE_FUSE1A", fuse)
        fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=cap, input_width=W, input_height=H, max_batch=n, precision=api.PREC_F32_WINO))
        fe.load_superpoint(w)
        outs.append(fe.extract_batch(imgs, cap=cap))
        trunks.append(fe.debug_read("conv1b", (n, H // 2, W // 2, 64)))
        fe.close()
    assert np.array_equal(trunks[0], trunks[1])
    for (k0, s0, d0), (k1, s1, d1) in zip(*outs):
        assert np.array_equal(k0, k1) and np.array_equal(s0, s1) and np.array_equal(d0, d1)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(100, 150), (24, 40), (75, 133)])
def test_wino_fused_staging_borders_and_stride(api, monkeypatch, H, W):
    """The fused conv1a staging reads the frame through a 12 x 20 byte patch in LDS: sizes that are no multiple of the 8 x 16 work item
    (items cut by the right / bottom border, patches hanging over every edge) and a row stride > width whose padding is poisoned
    must give the bits of the two-kernel form (which reads the frame in a different kernel altogether)."""
    import ctypes as C
    n, cap = 3, 60
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    rng = np.random.RandomState(H * 1000 + W)
    imgs = rng.randint(0, 256, size=(n, H, W)).astype(np.uint8)
    imgs[0, :2, :] = 255; imgs[0, -2:, :] = 255; imgs[0, :, :2] = 255; imgs[0, :, -2:] = 255      # bright frame border: padding errors show
    trunks, outs, strided = [], [], []
    for fuse in ("0", "1"):
        monkeypatch.setenv("D2FE_FUSE1A", fuse)
        fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=cap, input_width=W, input_height=H, max_batch=n, precision=api.PREC_F32_WINO))
        fe.load_superpoint(w)
        outs.append(fe.extract_batch(imgs, cap=cap))
        trunks.append(fe.debug_read("conv1b", (n, H // 2, W // 2, 64)))
        stride = W + 23
        buf = np.full((H, stride), 255, np.uint8); buf[:, :W] = imgs[1]
        kps = np.zeros((cap, 2), np.float32); sc = np.zeros(cap, np.float32); desc = np.zeros((cap, 256), np.float32); k = C.c_int(0)
        rc = api.load_library(dev=True).d2fe_superpoint_extract(fe.handle, buf.ctypes.data, W, H, stride, kps.ctypes.data, sc.ctypes.data,
                                                        desc.ctypes.data, cap, C.byref(k))
        assert rc == 0
        strided.append((kps[:k.value].copy(), sc[:k.value].copy(), desc[:k.value].copy()))
        fe.close()
    assert np.array_equal(trunks[0], trunks[1])
    for (k0, s0, d0), (k1, s1, d1) in zip(*outs):
        assert np.array_equal(k0, k1) and np.array_equal(s0, s1) and np.array_equal(d0, d1)
    for a, b in zip(strided[0], strided[1]):
        assert np.array_equal(a, b)
    for a, b in zip(strided[1], outs[1][1]):        # the strided call sees the same image as batch entry 1
        assert np.array_equal(a, b)
