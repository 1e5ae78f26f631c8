"""The oracle's variant-B selection/sampling, variant-A NMS2, matchKNN glue and the quadcam half-image chain held to the
REFERENCE'S OWN C++ (oracle/_ref/libspref.so: line ranges of superpoint_tensorrt.cpp, superpoint_common.cpp,
feature_matcher.cpp and d2featuretracker.cpp compiled from /root/reference against stand-in Eigen/OpenCV headers,
oracle/build_ref.py).  CPU tests: oracle vs reference; the `gpu` tests at the bottom hold the HIP path to it directly.

Unstable std::sort: the reference's order among EQUAL scores is libstdc++'s; the oracle breaks ties by raster index.
Comparisons are exact, tie groups are compared as sets (the helper below); on the natural inputs used here no ties occur."""
import numpy as np
import pytest

from d2slam_amd.synth import synth_descriptor_pair, synth_image, synth_stereo
from oracle import ref as spref

pytestmark = pytest.mark.skipif(not spref.available(), reason="oracle/_ref/libspref.so absent and /root/reference not present")

# BASELINE.json configs: [1] d435 640x480 N=200 thr 0.015; [2] quadcam 800x400 N=100 thr 0.15; [0] TUM 512x512 N=150
CONFIGS = [(480, 640, 200, 0.015, 11), (400, 800, 100, 0.15, 12), (512, 512, 150, 0.015, 13), (120, 160, 100, 0.015, 14)]


def assert_same_selection(kps, sc, rk, rs, what=""):
    """Exact equality, allowing a different order/choice only inside groups of exactly equal scores (std::sort is unstable)."""
    assert len(kps) == len(rk), (what, len(kps), len(rk))
    if np.array_equal(kps, rk) and np.array_equal(sc, rs):
        return
    assert np.array_equal(sc, rs), what + ": score sequences differ"          # equal even when ties are permuted
    for s in np.unique(sc):
        g = sc == s
        a = {tuple(p) for p in kps[g]}; b = {tuple(p) for p in rk[g]}
        if s == sc.min():
            continue        # a tie group cut by the top-K boundary may keep different members
        assert a == b, what + ": keypoints differ outside a tie group"


@pytest.fixture(scope="module")
def forwards(orc, sp_weights):
    cache = {}

    def get(H, W, seed, dustbin_shift=0.0):
        key = (H, W, seed, dustbin_shift)
        if key not in cache:
            w = sp_weights
            if dustbin_shift:
                w = dict(w); Wt, b = w["convPb"]; b = b.copy(); b[64] -= np.float32(dustbin_shift); w["convPb"] = (Wt, b)
            cache[key] = orc.superpoint_forward(synth_image(H, W, seed), w)
        return cache[key]
    return get


@pytest.mark.parametrize("H,W,N,thr,seed", CONFIGS)
def test_variant_b_vs_reference_cpp(orc, forwards, H, W, N, thr, seed):
    """A3-A5: findHighScoreIndex / removeBorders / topKeypoints / sampleDescriptors as compiled from the reference."""
    f = forwards(H, W, seed, 3.5 if thr > 0.1 else 0.0)
    rk, rs, rd = spref.superpoint_post(f["semi"], f["desc"], thr, 1, N)
    k, s, idx = orc.select_b(f["semi"], thr, 1, N)
    assert len(rk) == N                                    # the top-K branch is the one exercised
    assert_same_selection(k, s, rk, rs, "variant B")
    d = orc.sample_b(f["desc"], rk)
    # descriptors: the stand-in Eigen norm() sums sequentially like the oracle -> bitwise; bound kept at 1e-6 for real Eigen
    assert np.abs(d - rd).max() <= 1e-6
    assert np.array_equal(d, rd)


@pytest.mark.parametrize("N", [-1, 100000])
def test_variant_b_keep_all_is_raster_order(orc, forwards, N):
    """max_keypoints = -1 or >= candidates: nothing is sorted, raster order is kept (superpoint_tensorrt.cpp:241-253)."""
    f = forwards(120, 160, 14)
    rk, rs, rd = spref.superpoint_post(f["semi"], f["desc"], 0.015, 1, N)
    k, s, idx = orc.select_b(f["semi"], 0.015, 1, N, cap=120 * 160)
    assert len(rk) > 300 and np.array_equal(k, rk) and np.array_equal(s, rs)
    ras = rk[:, 1] * 160 + rk[:, 0]
    assert np.all(np.diff(ras) > 0)


def test_variant_b_borders_and_empty(orc, forwards):
    f = forwards(120, 160, 14)
    for border in (0, 1, 4):
        rk, rs, _ = spref.superpoint_post(f["semi"], f["desc"], 0.015, border, 50)
        k, s, _ = orc.select_b(f["semi"], 0.015, border, 50)
        assert_same_selection(k, s, rk, rs, "border %d" % border)
    rk, rs, rd = spref.superpoint_post(f["semi"], f["desc"], 2.0, 1, 50)       # nothing passes
    assert len(rk) == 0 and len(orc.select_b(f["semi"], 2.0, 1, 50)[0]) == 0


@pytest.mark.parametrize("H,W,N,thr,seed", CONFIGS)
@pytest.mark.parametrize("d", [4, 10])
def test_nms2_vs_reference_cpp(orc, forwards, H, W, N, thr, seed, d):
    """A6: getKeyPoints + NMS2 as compiled from the reference (raster sweep, CV_16UC1 index map, sort, max_num)."""
    f = forwards(H, W, seed, 3.5 if thr > 0.1 else 0.0)
    rk, rs = spref.get_keypoints(f["semi"], thr, d, N)
    k, s = orc.nms2_a(f["semi"], thr, d, N)
    assert len(rk) > 5
    assert_same_selection(k, s, rk, rs, "NMS2")


def test_nms2_u16_wrap_vs_reference_cpp(orc, forwards):
    """More than 65 536 candidates: the reference's CV_16UC1 index map wraps; the oracle reproduces exactly that."""
    f = forwards(256, 320, 5)
    assert int((f["semi"] > 1e-6).sum()) > 70000
    rk, rs = spref.get_keypoints(f["semi"], 1e-6, 3, 400)
    k, s = orc.nms2_a(f["semi"], 1e-6, 3, 400)
    assert_same_selection(k, s, rk, rs, "NMS2 wrap")
    assert sum(1 for (x, y), v in zip(rk.astype(int), rs) if f["semi"][y, x] != v) > 0


MATCH_CASES = [(200, 200, 256, 0.8, -1.0, 0.05), (150, 97, 256, 0.7, 32.0, 0.2), (33, 200, 64, 0.9, -1.0, 0.05),
               (100, 100, 256, 0.9, 19.2, 0.05), (1, 5, 256, 0.8, -1.0, 0.05), (5, 1, 256, 0.8, -1.0, 0.05), (2, 2, 256, 0.8, -1.0, 0.05)]


@pytest.mark.parametrize("na,nb,dim,ratio,radius,sigma", MATCH_CASES)
def test_match_knn_vs_reference_cpp(orc, na, nb, dim, ratio, radius, sigma):
    """A10: the reference's matchKNN (ratio test in double, inverse dictionary, radius gate) over the stand-in BFMatcher."""
    a, b, pa, pb = synth_descriptor_pair(na, nb, dim, seed=na * 7 + nb, sigma=sigma)
    rq, rt, rd = spref.match_knn(a, b, ratio, pa, pb, radius)
    q, t, d = orc.match_knn(a, b, ratio, pa, pb, radius)
    assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd)


def test_match_knn_real_descriptors_vs_reference_cpp(orc, sp_weights):
    l, r = synth_stereo(120, 160, 3)
    kl, sl, dl, _, _ = orc.extract_b(l, sp_weights, 0.015, 1, 100)
    kr, sr, dr, _, _ = orc.extract_b(r, sp_weights, 0.015, 1, 100)
    for radius in (-1.0, 32.0):
        rq, rt, rd = spref.match_knn(dl, dr, 0.8, kl, kr, radius)
        q, t, d = orc.match_knn(dl, dr, 0.8, kl, kr, radius)
        assert len(rq) > 3 or radius > 0
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd)


@pytest.mark.parametrize("fov,width", [(200.0, 800), (190.0, 640), (235.0, 1280)])
def test_half_image_vs_reference_cpp(orc, fov, width):
    """A12: getFeatureHalfImg (d2featuretracker.cpp:1051-1075)."""
    rng = np.random.RandomState(4)
    pts = np.stack([rng.randint(0, width, 300), rng.randint(0, 400, 300)], 1).astype(np.float32)
    mc = width * 90.0 / fov
    pts[:6, 0] = [np.float32(mc), np.floor(mc), np.ceil(mc), np.float32(width - mc), np.floor(width - mc), np.ceil(width - mc)]
    for left in (True, False):
        assert np.array_equal(orc.half_img(pts, left, width, fov), spref.half_image(pts, left, width, fov))


@pytest.mark.parametrize("type_lr", [1, 2])
@pytest.mark.parametrize("local", [True, False])
def test_neighbour_chain_vs_reference_cpp(orc, type_lr, local):
    """The quadcam neighbour branch of matchLocalFeatures (d2featuretracker.cpp:1146-1181) end to end: half-image filter on
    both sides, +-move_cols shift of the a-side points, matchKNN with the radius gate, index remap."""
    W_u, fov = 800, 200.0
    a, _, pa, _ = synth_descriptor_pair(100, 100, 256, seed=21 + type_lr, sigma=0.05)
    rng = np.random.RandomState(7)
    b = a + rng.normal(0, 0.05, a.shape).astype(np.float32); b /= np.linalg.norm(b, axis=1, keepdims=True)
    pa[:, 0] = rng.uniform(0, W_u, 100); pa[:, 1] = rng.uniform(0, 400, 100)
    mc = np.float32(W_u * 90.0 / fov)
    pb = pa.copy(); pb[:, 0] += (mc if type_lr == 1 else -mc) + rng.normal(0, 5, 100).astype(np.float32)   # the same scene seen by the neighbour
    radius = 0.05 * W_u
    ref = spref.match_neighbour(pa, a, pb, b, type_lr, 0.8, local, radius, W_u, fov)
    got = orc_neighbour_chain(orc, pa, a, pb, b, type_lr, 0.8, local, radius, W_u, fov)
    assert ref is not None and got is not None and len(ref[0]) > 3
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)
    # an empty half makes the reference's branch return false
    far = pa.copy(); far[:, 0] = W_u - 1.0 if type_lr == 1 else 0.0
    assert spref.match_neighbour(far, a, pb, b, type_lr, 0.8, local, radius, W_u, fov) is None
    assert orc_neighbour_chain(orc, far, a, pb, b, type_lr, 0.8, local, radius, W_u, fov) is None


def orc_neighbour_chain(orc, pts_a, desc_a, pts_b, desc_b, type_lr, ratio, enable_search_in_local, search_radius, W_u, fov):
    """The oracle's statement of d2featuretracker.cpp:1146-1181 (also what the HIP chain is compared with)."""
    ma = orc.half_img(pts_a, type_lr == 1, W_u, fov)
    mb = orc.half_img(pts_b, type_lr == 2, W_u, fov)
    if len(ma) == 0 or len(mb) == 0:
        return None
    pa = pts_a[ma].copy(); pb = pts_b[mb]
    if enable_search_in_local:
        mc = np.float32(W_u * 90.0 / fov)
        pa[:, 0] = pa[:, 0] + (mc if type_lr == 1 else -mc)
    q, t, d = orc.match_knn(desc_a[ma], desc_b[mb], ratio, pa, pb, search_radius if enable_search_in_local else -1.0)
    return ma[q].astype(np.int32), mb[t].astype(np.int32), d


# ---- the HIP path against the reference's own C++ (GPU box: the prebuilt libspref.so travels with the snapshot) -------------------
@pytest.fixture(scope="module")
def api():
    from d2slam_amd import api as a
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,N,thr,seed", CONFIGS[:3])
@pytest.mark.parametrize("prec", ["f32", "wino"])
def test_hip_variant_b_vs_reference_cpp(api, orc, sp_weights, H, W, N, thr, seed, prec):
    """HIP extract (network + selection + sampling) vs the reference's processOutput applied to the HIP path's OWN network
    outputs (read back with debug_read), at the BASELINE geometries, in both fp32 modes: indices exact, descriptors <= 1e-6."""
    w = sp_weights
    if thr > 0.1:
        w = dict(w); Wt, b = w["convPb"]; b = b.copy(); b[64] -= np.float32(3.5); w["convPb"] = (Wt, b)
    img = synth_image(H, W, seed)
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=N, input_width=W, input_height=H, max_batch=1, keypoint_threshold=thr,
                                           precision=api.PREC_F32 if prec == "f32" else api.PREC_F32_WINO, keep_score_map=True,
                                           dense_descriptors=True))
    fe.load_superpoint(w)
    (kps, sc, desc), = fe.extract_batch(img[None], cap=N)
    semi = fe.debug_read("semi", (1, H, W))[0]
    draw = fe.debug_read("desc_raw", (1, H // 8, W // 8, 256))[0]
    dmap = orc.l2norm_rows(draw)                       # channel L2 of the raw descriptor map (in-graph in the reference)
    rk, rs, rd = spref.superpoint_post(semi, dmap, thr, 1, N)
    assert len(rk) == N
    assert_same_selection(kps, sc, rk, rs, "HIP variant B")
    if np.array_equal(kps, rk):
        assert np.abs(desc - rd).max() <= 1e-6
    fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,N,thr,seed", CONFIGS[:3])
def test_hip_nms2_vs_reference_cpp(api, sp_weights, H, W, N, thr, seed):
    w = sp_weights
    if thr > 0.1:
        w = dict(w); Wt, b = w["convPb"]; b = b.copy(); b[64] -= np.float32(3.5); w["convPb"] = (Wt, b)
    img = synth_image(H, W, seed)
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=N, input_width=W, input_height=H, max_batch=1, keypoint_threshold=thr,
                                           postproc=api.POSTPROC_A, nms_dist=10, keep_score_map=True))
    fe.load_superpoint(w)
    (kps, sc, desc), = fe.extract_batch(img[None], cap=N)
    semi = fe.debug_read("semi", (1, H, W))[0]
    rk, rs = spref.get_keypoints(semi, thr, 10, N)
    assert len(rk) > 5
    assert_same_selection(kps, sc, rk, rs, "HIP NMS2")
    fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("na,nb,dim,ratio,radius,sigma", MATCH_CASES)
def test_hip_match_knn_vs_reference_cpp(api, na, nb, dim, ratio, radius, sigma):
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=100, input_width=64, input_height=64, max_batch=1))
    a, b, pa, pb = synth_descriptor_pair(na, nb, dim, seed=na * 7 + nb, sigma=sigma)
    q, t, d = fe.match_knn(a, b, ratio, pa, pb, radius)
    rq, rt, rd = spref.match_knn(a, b, ratio, pa, pb, radius)
    assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rd)
    fe.close()


@pytest.mark.gpu
def test_hip_keep_all_vs_reference_cpp(api, orc, sp_weights):
    """max_keypoints = -1 (SuperPoint::topKeypoints keeps everything when k == -1, superpoint_tensorrt.cpp:241-253): all keypoints above
    the threshold, in raster order, like the reference; more than the call's capacity -> D2FE_ERR_TRUNCATED with the strongest kept."""
    H, W = 120, 160
    img = synth_image(H, W, 14)
    f = orc.superpoint_forward(img, sp_weights)
    thr = float(np.sort(f["semi"].reshape(-1))[-700])                        # ~700 candidates
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=-1, input_width=W, input_height=H, max_batch=1, keypoint_threshold=thr,
                                           keep_score_map=True, dense_descriptors=True))
    fe.load_superpoint(sp_weights)
    (kps, sc, desc), = fe.extract_batch(img[None], cap=1024)
    rk, rs, rd = spref.superpoint_post(f["semi"], f["desc"], thr, 1, -1)
    assert 600 < len(rk) < 1024 and np.array_equal(kps, rk) and np.array_equal(sc, rs)
    assert np.abs(desc - rd).max() <= 1e-6
    with pytest.raises(api.D2FEError) as e:
        fe.extract_batch(img[None], cap=256)
    assert e.value.code == -4                                                 # D2FE_ERR_TRUNCATED
    fe.close()
