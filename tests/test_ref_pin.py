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
    fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=N, input_width=W, input_height=H, max_batch=1, keypoint_threshold=thr,
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
    # descriptors: always, on every keypoint both lists hold (all of them unless a tie group was cut differently by the top-K boundary)
    pos = {(float(x), float(y)): i for i, (x, y) in enumerate(rk)}
    both = [(i, pos[(float(x), float(y))]) for i, (x, y) in enumerate(kps) if (float(x), float(y)) in pos]
    assert len(both) >= N - 4, (len(both), N)
    gi = np.array([a for a, _ in both]); ri = np.array([b for _, b in both])
    assert np.abs(desc[gi] - rd[ri]).max() <= 1e-6
    fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,N,thr,seed", CONFIGS[:3])
def test_hip_nms2_vs_reference_cpp(api, sp_weights, H, W, N, thr, seed):
    w = sp_weights
    if thr > 0.1:
        w = dict(w); Wt, b = w["convPb"]; b = b.copy(); b[64] -= np.float32(3.5); w["convPb"] = (Wt, b)
    img = synth_image(H, W, seed)
    fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=N, input_width=W, input_height=H, max_batch=1, keypoint_threshold=thr,
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
    # a capacity below the keypoint count: D2FE_ERR_TRUNCATED, reported as a flag, with the 256 STRONGEST keypoints kept
    (k2, s2, d2), = fe.extract_batch(img[None], cap=256)
    assert fe.last_truncated and len(k2) == 256
    order = np.lexsort((np.arange(len(rs)), -rs))[:256]
    assert {tuple(p) for p in k2} == {tuple(p) for p in rk[order]}
    (k3, _, _), = fe.extract_batch(img[None], cap=1024)
    assert not fe.last_truncated and len(k3) == len(rk)
    fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "wino"])
def test_hip_keep_all_thousands_of_keypoints_vs_reference_cpp(api, orc, sp_weights, prec):
    """Keep-all (topKeypoints with k = -1, superpoint_tensorrt.cpp:241-253) is unbounded in the reference: with a low threshold a 240x320 image
    yields tens of thousands of keypoints, all of which come back, in raster order, with their descriptors -- far beyond the 1024 the
    interface used to stop at, and beyond the 16384 the in-LDS sort takes (the raster compaction of the dense score map serves those)."""
    H, W = 240, 320
    img = synth_image(H, W, 15)
    for target in (5000, 30000):
        fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=-1, input_width=W, input_height=H, max_batch=1, keypoint_threshold=0.0,
                                               precision=api.PREC_F32 if prec == "f32" else api.PREC_F32_WINO, keep_score_map=True, dense_descriptors=True))
        fe.load_superpoint(sp_weights)
        fe.extract_batch(img[None], cap=16)
        semi = fe.debug_read("semi", (1, H, W))[0]
        fe.close()
        thr = float(np.sort(semi.reshape(-1))[-target])
        fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=-1, input_width=W, input_height=H, max_batch=1, keypoint_threshold=thr,
                                               precision=api.PREC_F32 if prec == "f32" else api.PREC_F32_WINO, keep_score_map=True, dense_descriptors=True))
        fe.load_superpoint(sp_weights)
        (kps, sc, desc), = fe.extract_batch(img[None], cap=H * W)
        draw = fe.debug_read("desc_raw", (1, H // 8, W // 8, 256))[0]
        rk, rs, rd = spref.superpoint_post(semi, orc.l2norm_rows(draw), thr, 1, -1)
        assert not fe.last_truncated and 0.8 * target < len(rk) < target
        assert np.array_equal(kps, rk) and np.array_equal(sc, rs)
        assert np.abs(desc - rd).max() <= 1e-6
        fe.close()


# ---- round 3: the reference-owned code either side of the hot path (oracle/ref_shim/spref_api2.cpp) ---------------------------------
def _unit_rows(x):
    return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)


def _codec_inputs():
    rng = np.random.RandomState(21)
    lm = _unit_rows(rng.randn(120, 256)).reshape(-1)
    lm[5 * 256:5 * 256 + 32] *= np.float32(1e-4)          # a 32-float segment that quantises to all zeros (Eigen's z > 0 guard)
    nv = _unit_rows(rng.randn(4096))
    return lm, nv


def test_int8_codec_vs_reference_cpp(orc):
    """(f)-3: VisualImageDesc::toLCM's int8 quantisation (float max for landmark descriptors, double max for NetVLAD) and the LCM
    constructor's decode (q/127.0, the hard-coded 32-float renormalisation of the first landmark_num segments, whole-vector NetVLAD
    normalisation), d2frontend_types.h:230-237,262-268,319-341, compiled in place: the oracle's restatement is bitwise equal."""
    lm, nv = _codec_inputs()
    ql, qn = spref.quant_landmarks(lm), spref.quant_netvlad(nv)
    assert np.array_equal(orc.quant_int8(lm), ql) and np.array_equal(orc.quant_int8(nv, double_max=True), qn)
    assert not ql[5 * 256:5 * 256 + 32].any()
    for landmark_num in (120, 7, 0, 960):           # 960 = every 32-float segment of the 120 descriptors
        rl, rn = spref.dequant(ql, landmark_num, qn)
        assert np.array_equal(orc.dequant_int8(ql, landmark_num), rl) and not np.isnan(rl).any()
        assert np.array_equal(orc.dequant_int8(qn, -1), rn)
    # and a descriptor set whose maximum is negative / at the last element
    x = -np.abs(lm); x[-1] = -2.0
    assert np.array_equal(orc.quant_int8(x), spref.quant_landmarks(x))


@pytest.mark.parametrize("n,dim,max_index,thres", [(300, 64, 10, 0.5), (40, 1024, 0, 0.9), (3, 4096, 10, 0.2), (700, 1024, 100, 0.5)])
def test_db_gate_vs_reference_cpp(orc, n, dim, max_index, thres):
    """(f)-2 / A14: LoopDetector::queryIndexFromDatabase (loop_detector.cpp:300-350) compiled in place over a stand-in IndexFlatIP."""
    rng = np.random.RandomState(n + dim)
    db = _unit_rows(rng.randn(n, dim))
    for target in (0, n // 2, n - 1, min(max(n - max_index, 0), n - 1), max(n - max_index - 1, 0)):
        q = _unit_rows(db[target] + (0.3 / np.sqrt(dim)) * rng.randn(dim).astype(np.float32))
        label, sim, _, _ = orc.db_query(db, q, max_index, thres)
        rl, rs = spref.db_query(db, q, max_index, thres)
        assert label == rl, (target, label, rl)
        if label >= 0:
            assert abs(sim - rs) <= 2e-6
    q = _unit_rows(rng.randn(dim))                     # unrelated query: nothing above the threshold
    assert orc.db_query(db, q, max_index, 0.9)[0] == spref.db_query(db, q, max_index, 0.9)[0] == -1


def _gate_cases():
    rng = np.random.RandomState(5)
    G = 512
    kf = _unit_rows(rng.randn(4, 4, G))
    cases = []
    for rot in range(4):                                # the remote drone sees keyframe 2 turned by `rot` quarter turns
        rem = _unit_rows(kf[2][[(v + rot) % 4 for v in range(4)]] + (0.4 / np.sqrt(G)) * rng.randn(4, G).astype(np.float32))
        cases.append((rem, kf, 0.6))
    cases.append((_unit_rows(rng.randn(4, G)), kf, 0.6))            # no keyframe matches
    cases.append((cases[0][0], kf[:0].reshape(0, 4, G), 0.6))       # no keyframes at all
    return cases


def test_tracker_gate_vs_reference_cpp(orc):
    """A14: D2FeatureTracker::getMatchedPrevKeyframe (d2featuretracker.cpp:166-235), both camera-configuration branches, and the
    FOURCORNER_FISHEYE view pairing of trackRemoteFrames (:270-284), compiled in place."""
    for rem, kf, thres in _gate_cases():
        for quad in (True, False):
            a, b = orc.tracker_gate(rem, kf, thres, quad), spref.tracker_gate(rem, kf, thres, quad)
            assert (a is None) == (b is None)
            if a is not None:
                assert (a["kf"], a["dir_a"], a["dir_b"], a["pairs"]) == (b["kf"], b["dir_a"], b["dir_b"], b["pairs"])
                if quad:
                    assert len(a["pairs"]) == 4 and a["dir_a"] == 2
    # views without SuperPoint landmarks are skipped by the pairing
    rem, kf, thres = _gate_cases()[1]
    spr = [1, 0, 3, 2]; spk = np.ones((4, 4), np.int32); spk[2, 3] = 0
    a, b = orc.tracker_gate(rem, kf, thres, True, spr, spk), spref.tracker_gate(rem, kf, thres, True, spr, spk)
    assert a["pairs"] == b["pairs"] and len(a["pairs"]) < 4


def test_loopcam_match_vs_reference_cpp(orc):
    """A11: matchLocalFeatures of loop_cam.cpp:156-191 (BFMatcher(NORM_L2, crossCheck).match, every match accepted) = the oracle's
    cross-check matcher."""
    for na, nb, seed in ((150, 130, 1), (40, 200, 2), (1, 1, 3), (64, 64, 4)):
        a, b, pa, pb = synth_descriptor_pair(na, nb, 256, seed=seed)
        iu, idn, pu, pd = spref.loopcam_match(pa, a, pb, b)
        q, t, _ = orc.match_crosscheck(a, b)
        assert np.array_equal(iu, q) and np.array_equal(idn, t)
        assert np.array_equal(pu, pa[q]) and np.array_equal(pd, pb[t])


@pytest.mark.gpu
def test_hip_int8_codec_vs_reference_cpp(api):
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    lm, nv = _codec_inputs()
    ql, qn = spref.quant_landmarks(lm), spref.quant_netvlad(nv)
    assert np.array_equal(fe.quantize_int8(lm), ql) and np.array_equal(fe.quantize_int8(nv, double_max=True), qn)      # bytes: exact
    for landmark_num in (120, 7, 0, 960):
        rl, rn = spref.dequant(ql, landmark_num, qn)
        got = fe.dequantize_int8(ql, landmark_num)
        assert not np.isnan(got).any() and np.abs(got - rl).max() <= 1e-6         # Eigen's squaredNorm order is unspecified: 1 ulp
        assert np.abs(fe.dequantize_int8(qn, -1) - rn).max() <= 1e-6
    fe.close()


@pytest.mark.gpu
def test_hip_db_gate_vs_reference_cpp(api):
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    rng = np.random.RandomState(9)
    for n, dim, max_index, thres in ((300, 1024, 10, 0.5), (700, 4096, 100, 0.5), (3, 4096, 10, 0.2)):
        vec = _unit_rows(rng.randn(n, dim))
        db = api.FlatIPDatabase(fe, dim, capacity=1024)
        db.add(vec)
        for target in (0, n // 2, n - 1, max(n - max_index, 0)):
            q = _unit_rows(vec[target] + (0.3 / np.sqrt(dim)) * rng.randn(dim).astype(np.float32))
            label, sim = db.query_gated(q, max_index, thres)
            rl, rs = spref.db_query(vec, q, max_index, thres)
            assert label == rl and (label < 0 or abs(sim - rs) <= 2e-5)
        db.close()
    fe.close()


@pytest.mark.gpu
def test_hip_tracker_gates_vs_reference_cpp(api, orc):
    """d2fe_gate_pairs_device (stereo branch) and d2fe_quad_gate_device (FOURCORNER_FISHEYE branch + view pairing) against the
    reference's getMatchedPrevKeyframe / trackRemoteFrames compiled in place."""
    import torch
    dev = torch.device("cuda", 0)
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    cases = [c for c in _gate_cases() if len(c[1])]
    G = cases[0][0].shape[1]
    # one job per (case, keyframe): local = that keyframe's 4 views, remote = the case's remote frame; view-major layouts with a stride
    jobs = [(ci, k) for ci in range(len(cases)) for k in range(4)]
    loc = np.zeros((len(jobs) * 4, G + 4), np.float32); rem = np.zeros((len(jobs) * 4, 2 * G), np.float32)
    for j, (ci, k) in enumerate(jobs):
        loc[4 * j:4 * j + 4, :G] = cases[ci][1][k]; rem[4 * j:4 * j + 4, :G] = cases[ci][0]
    t = lambda a: torch.from_numpy(a).to(dev)
    d_loc, d_rem = t(loc), t(rem)
    rows = t(np.arange(len(jobs), dtype=np.int32) * 4)
    dirp = torch.zeros(len(jobs), dtype=torch.int32, device=dev); sims = torch.zeros((len(jobs), 4), device=dev)
    cnt = torch.full((len(jobs) * 16,), 9, dtype=torch.int32, device=dev); npass = torch.zeros(1, dtype=torch.int32, device=dev)
    fe.quad_gate_device(d_loc.data_ptr(), G + 4, d_rem.data_ptr(), 2 * G, G, rows.data_ptr(), rows.data_ptr(), 1, 1, len(jobs), 0.6,
                        d_dir_prev=dirp.data_ptr(), d_sims=sims.data_ptr(), d_cnt_inout=cnt.data_ptr(), d_n_pass=npass.data_ptr())
    fe.sync(); torch.cuda.synchronize()
    dirp, sims, cnt = dirp.cpu().numpy(), sims.cpu().numpy(), cnt.cpu().numpy().reshape(-1, 16)
    n_ok = 0
    for j, (ci, k) in enumerate(jobs):
        remote, kf, thres = cases[ci]
        r = spref.tracker_gate(remote, kf[k:k + 1], thres, True)
        o = orc.tracker_gate(remote, kf[k:k + 1], thres, True)
        assert (r is None) == (dirp[j] < 0), (j, dirp[j])
        keep = np.zeros(16, bool)
        if r is not None:
            n_ok += 1
            assert dirp[j] == r["dir_b"] == o["dir_b"]
            for a_view, b_view in r["pairs"]:                 # (remote view, local view)
                keep[b_view * 4 + a_view] = True
            nj = [2, 3, 0, 1].index(r["dir_b"]) + 1           # the oracle stops at the first passing view
            assert np.abs(sims[j][:nj] - o["sims"][:nj]).max() <= 2e-5
        assert np.array_equal(cnt[j], np.where(keep, 9, 0)), (j, cnt[j])
    assert int(npass.item()) == n_ok and n_ok >= 4
    fe.close()


@pytest.mark.gpu
def test_hip_crosscheck_vs_reference_loopcam_match(api):
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    for na, nb, seed in ((150, 130, 1), (40, 200, 2), (1, 1, 3), (64, 64, 4)):
        a, b, pa, pb = synth_descriptor_pair(na, nb, 256, seed=seed)
        iu, idn, _, _ = spref.loopcam_match(pa, a, pb, b)
        q, t, _ = fe.match_crosscheck(a, b)
        assert np.array_equal(iu, q) and np.array_equal(idn, t)
    fe.close()


# ---- (f)-1 map generation: camodocal (vendored camera_models/) + FisheyeUndist::genOneUndistMap compiled in place ---------------------
# cam0 of config/quadcam/quad_cam_calib-camchain-imucam-7-inch-n3.yaml (camera_model omni, distortion radtan)
_MEI9 = [2.2176903753419963, -0.17703529535292872, 0.7517933338735744, -0.0008911425891703079, 2.1653595535258756e-05,
         1162.5434300524314, 1161.839362615319, 660.6393183718625, 386.1663300322095]
_MAP_CASES = [(800, 400, 200.0), (640, 480, 190.0), (401, 203, 235.0)]      # the quadcam geometry, another one, odd sizes (unsigned width / 2)


@pytest.mark.parametrize("W,H,fov", _MAP_CASES)
def test_cylinder_map_vs_reference_cpp(orc, W, H, fov):
    """generateCylinderMap / genOneUndistMap (fisheye_undistort.h:458-500,559-613) with CataCamera::spaceToPlane / distortion and
    CylindricalCamera::liftProjective (camera_models/, CataCamera.cc:495-515,617-633, CylindricalCamera.cc:144-147,207-220) compiled where they lie:
    the oracle's restatement is bitwise equal (fp64 evaluation, float store)."""
    mx, my = orc.gen_cylinder_map(_MEI9, W, H, fov)
    rx, ry = spref.gen_cylinder_map(_MEI9, W, H, fov)
    assert np.array_equal(mx, rx) and np.array_equal(my, ry)


@pytest.mark.parametrize("angle,axis", [(-np.pi / 4, 1), (np.pi / 4, 1), (0.3, 0), (0.0, 1)])
def test_pinhole_map_vs_reference_cpp(orc, angle, axis):
    """The rotated-pinhole genOneUndistMap (fisheye_undistort.h:615-660) incl. Eigen's Quaternion * Vector3 order."""
    q = [np.cos(angle / 2), 0.0, 0.0, 0.0]; q[1 + axis] = np.sin(angle / 2)
    px, py = orc.gen_pinhole_map(_MEI9, q, 600, 300, 300.0)
    rx, ry = spref.gen_pinhole_map(_MEI9, q, 600, 300, 300.0)
    assert np.array_equal(px, rx) and np.array_equal(py, ry)


@pytest.mark.gpu
def test_hip_maps_vs_reference_cpp(api):
    """d2fe_gen_cylinder_map / d2fe_gen_pinhole_map against the reference's own map generation: fp64 on the device, where tan()/sqrt() may
    differ from glibc's in the last fp64 bit -> at most one fp32 ulp on a handful of entries."""
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    cam = dict(zip(("xi", "k1", "k2", "p1", "p2", "gamma1", "gamma2", "u0", "v0"), _MEI9))
    for W, H, fov in _MAP_CASES[:2]:
        gx, gy = fe.gen_cylinder_map(cam, W, H, fov)
        rx, ry = spref.gen_cylinder_map(_MEI9, W, H, fov)
        assert np.abs(gx - rx).max() <= 1.3e-4 and np.abs(gy - ry).max() <= 1.3e-4
        assert (gx == rx).mean() > 0.999 and (gy == ry).mean() > 0.999
    q = [np.cos(np.pi / 8), 0.0, np.sin(np.pi / 8), 0.0]
    px, py = fe.gen_pinhole_map(_MEI9, q, 600, 300, 300.0)
    rx, ry = spref.gen_pinhole_map(_MEI9, q, 600, 300, 300.0)
    assert np.abs(px - rx).max() <= 1.3e-4 and np.abs(py - ry).max() <= 1.3e-4 and (px == rx).mean() > 0.999
    fe.close()


# ---- (f)-4 LK tracker glue: the reference's opticalflowTrackPyr compiled in place over the oracle's SparsePyrLK restatement ------------------
def _shift(img, dx, dy):
    """bilinear shift of a u8 image (content moves by +dx, +dy), borders replicated"""
    h, w = img.shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    sx = np.clip(xx - dx, 0, w - 1); sy = np.clip(yy - dy, 0, h - 1)
    x0 = np.floor(sx).astype(int); y0 = np.floor(sy).astype(int); x1 = np.minimum(x0 + 1, w - 1); y1 = np.minimum(y0 + 1, h - 1)
    fx = sx - x0; fy = sy - y0
    f = img.astype(np.float32)
    v = (f[y0, x0] * (1 - fx) + f[y0, x1] * fx) * (1 - fy) + (f[y1, x0] * (1 - fx) + f[y1, x1] * fx) * fy
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def _track_like_the_mirror(track, prev_pts, W, track_type, fov):
    """include/d2fe.hpp's opticalflowTrackPyr on top of a bidirectional track function (orc.lk_track / the HIP d2fe_lk_track)"""
    move = np.float32(W * 90.0 / fov)
    p = np.asarray(prev_pts, np.float32).reshape(-1, 2)
    ids = np.arange(len(p))
    init = p.copy()
    if track_type == 1:
        keep = p[:, 0] < W - move; p, ids = p[keep], ids[keep]; init = p.copy(); init[:, 0] += move
    elif track_type == 2:
        keep = p[:, 0] >= move; p, ids = p[keep], ids[keep]; init = p.copy(); init[:, 0] -= move
    if not len(p):
        return np.zeros((0, 2), np.float32), ids[:0]
    out, st = track(p, init, float(move))
    return out[st > 0], ids[st > 0]


_LK_CASES = [(0, 3.5, -2.25, 640, 480), (1, 2.0, 0.5, 800, 400), (2, -1.5, 1.0, 800, 400)]


@pytest.mark.parametrize("ttype,dx,dy,W,H", _LK_CASES)
def test_lk_glue_vs_reference_cpp(orc, ttype, dx, dy, W, H):
    """(f)-4: opticalflowTrackPyr (opticaltrack_utils.cpp:173-278: the half-image pre-filter and +-move_cols shift, the reverse track from the shifted
    result, the 0.5 px forward/backward test, inBorder, reduceVector) compiled where it lies, calling the oracle's SparsePyrLK through a stand-in
    cv::cuda class: the oracle's one-call bidirectional track + the mirror's pre-filter give the same surviving points, ids and coordinates."""
    fov = 200.0
    move = float(np.float32(W * 90.0 / fov))
    img = synth_image(H, W, 21 + ttype)
    shift = dx + (move if ttype == 1 else -move if ttype == 2 else 0.0)
    cur = _shift(img, shift, dy)
    pts, _ = orc.fast_by_region(img, 150)
    pts = np.concatenate([pts, [[0.4, 0.4], [W - 1.2, H - 1.3], [W / 2, H / 2]]]).astype(np.float32)       # border cases for inBorder
    rp, rid = spref.lk_track_pyr(img, cur, pts, ttype, fov)
    p0, p1 = orc.pyr_build(img), orc.pyr_build(cur)
    op, oid = _track_like_the_mirror(lambda p, init, mv: orc.lk_track(p0, p1, W, H, p, init, track_type=ttype, move_cols=mv), pts, W, ttype, fov)
    assert len(rid) > 20 and np.array_equal(rid, oid) and np.array_equal(rp, op)


@pytest.mark.gpu
@pytest.mark.parametrize("ttype,dx,dy,W,H", _LK_CASES)
def test_hip_lk_glue_vs_reference_cpp(api, ttype, dx, dy, W, H):
    fov = 200.0
    move = float(np.float32(W * 90.0 / fov))
    img = synth_image(H, W, 21 + ttype)
    cur = _shift(img, dx + (move if ttype == 1 else -move if ttype == 2 else 0.0), dy)
    from oracle import oracle as orc
    orc.build()
    pts, _ = orc.fast_by_region(img, 150)
    pts = np.concatenate([pts, [[0.4, 0.4], [W - 1.2, H - 1.3], [W / 2, H / 2]]]).astype(np.float32)
    rp, rid = spref.lk_track_pyr(img, cur, pts, ttype, fov)
    fe = api.FrontEnd(api.SuperPointConfig(input_width=64, input_height=64, max_batch=1))
    f0, f1 = api.LKFrame(fe, img), api.LKFrame(fe, cur)
    gp, gid = _track_like_the_mirror(lambda p, init, mv: api.lk_track(fe, f0, f1, p, init, track_type=ttype, move_cols=mv), pts, W, ttype, fov)
    assert np.array_equal(rid, gid) and np.array_equal(rp, gp)
    f0.close(); f1.close(); fe.close()


def test_mirror_lift_vs_camodocal(tmp_path):
    """A8 (host): include/d2fe.hpp's liftProjectiveMEI -- what an adapter without camodocal calls where the reference calls camera->liftProjective
    (loop_cam.cpp:619-623) -- against camodocal's CataCamera::liftProjective (vendored camera_models/, CataCamera.cc:425-487: inverse K, the 8-step
    recursive distortion removal, the MEI ray) compiled in place: same fp64 operations in the same order, bitwise equal."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "liblift.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "lift_wrap.cpp"), "-o", so])
    L = C.CDLL(so)
    rng = np.random.RandomState(4)
    pts = np.concatenate([rng.rand(500, 2) * [1280, 800], [[660.64, 386.17], [0, 0], [1279, 799], [-50, 900]]]).astype(np.float32)
    cam = np.array(_MEI9, np.float64)
    out = np.zeros((len(pts), 3), np.float64)
    L.lift_mei(cam.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), len(pts), out.ctypes.data_as(C.c_void_p))
    ref = spref.cata_lift(_MEI9, pts)
    # far from the principal point the MEI ray does not exist (sqrt of a negative number): NaN in both -- the case extractorImgDescDeepnet skips
    assert np.array_equal(out, ref, equal_nan=True) and 0.2 < np.isfinite(ref).all(1).mean() < 1.0
    # no distortion (k1 = k2 = p1 = p2 = 0: m_noDistortion in camodocal) and xi = 1 (the parabolic branch)
    for cam2 in ([2.2, 0, 0, 0, 0, 1100.0, 1100.0, 640.0, 400.0], [1.0, -0.1, 0.05, 1e-3, -1e-3, 900.0, 905.0, 630.0, 410.0]):
        c2 = np.array(cam2, np.float64)
        L.lift_mei(c2.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), len(pts), out.ctypes.data_as(C.c_void_p))
        r2 = spref.cata_lift(cam2, pts)
        assert np.array_equal(np.isnan(out), np.isnan(r2)) and np.nanmax(np.abs(out - r2)) <= 1e-12, cam2


# ---- A7: the reference's computeDescriptors compiled against the image's real libtorch -------------------------------------------------------
@pytest.mark.skipif(not spref.torch_available(), reason="oracle/_ref/libspref_torch.so absent and not buildable here")
@pytest.mark.parametrize("H,W,N,thr,seed", CONFIGS[:3])
def test_variant_a_sampling_vs_reference_cpp_libtorch(orc, forwards, H, W, N, thr, seed):
    """computeDescriptors (superpoint_common.cpp:42-99: the (y, x) keypoint matrix, the grid 2x/W - 1, torch::grid_sampler(bilinear, zeros,
    align_corners = false), torch::norm over the KEYPOINT axis, the row normalisation, the optional PCA) compiled where it lies and run on ATen's own
    kernels: the oracle's restatement agrees to fp32 round-off, with and without PCA, at the BASELINE geometries."""
    f = forwards(H, W, seed, 3.5 if thr > 0.1 else 0.0)
    kps, _ = spref.get_keypoints(f["semi"], thr, 10, N)              # variant A's own keypoints (NMS2)
    assert len(kps) > 20
    ref = spref.compute_descriptors(f["desc"], kps, W, H)
    got = orc.sample_a(f["desc"], kps, W, H)
    assert ref.shape == got.shape == (len(kps), 256) and np.abs(got - ref).max() <= 2e-6
    rng = np.random.RandomState(seed)
    comp = np.linalg.qr(rng.randn(256, 64))[0].T.astype(np.float32); mean = (0.01 * rng.randn(256)).astype(np.float32)
    refp = spref.compute_descriptors(f["desc"], kps, W, H, comp, mean)
    gotp = orc.sample_a(f["desc"], kps, W, H, comp, mean)
    assert refp.shape == gotp.shape == (len(kps), 64) and np.abs(gotp - refp).max() <= 3e-6
    # keypoints on the image border sample outside the map (zeros padding)
    edge = np.array([[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1], [3, 5]], np.float32)
    assert np.abs(orc.sample_a(f["desc"], edge, W, H) - spref.compute_descriptors(f["desc"], edge, W, H)).max() <= 2e-6


@pytest.mark.gpu
@pytest.mark.skipif(not spref.torch_available(), reason="oracle/_ref/libspref_torch.so absent and not buildable here")
@pytest.mark.parametrize("H,W,N,thr,seed", CONFIGS[:3])
@pytest.mark.parametrize("pca", [False, True])
def test_hip_variant_a_descriptors_vs_reference_cpp_libtorch(api, orc, sp_weights, H, W, N, thr, seed, pca):
    """HIP variant A (NMS2 + sampling + PCA) against the reference's getKeyPoints AND computeDescriptors (real ATen grid_sampler), both applied to
    the HIP path's own network outputs."""
    w = sp_weights
    if thr > 0.1:
        w = dict(w); Wt, b = w["convPb"]; b = b.copy(); b[64] -= np.float32(3.5); w["convPb"] = (Wt, b)
    img = synth_image(H, W, seed)
    fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=N, input_width=W, input_height=H, max_batch=1, keypoint_threshold=thr,
                                           postproc=api.POSTPROC_A, nms_dist=10, keep_score_map=True, dense_descriptors=True))
    fe.load_superpoint(w)
    comp = mean = None
    if pca:
        rng = np.random.RandomState(seed)
        comp = np.linalg.qr(rng.randn(256, 64))[0].T.astype(np.float32); mean = (0.01 * rng.randn(256)).astype(np.float32)
        fe.set_pca(comp, mean)
    (kps, sc, desc), = fe.extract_batch(img[None], cap=N)
    semi = fe.debug_read("semi", (1, H, W))[0]
    dmap = orc.l2norm_rows(fe.debug_read("desc_raw", (1, H // 8, W // 8, 256))[0])
    rk, rs = spref.get_keypoints(semi, thr, 10, N)
    assert_same_selection(kps, sc, rk, rs, "HIP NMS2")
    ref = spref.compute_descriptors(dmap, kps, W, H, comp, mean)          # on the HIP path's own keypoint list (the channel norms depend on it)
    assert desc.shape == ref.shape and np.abs(desc - ref).max() <= 3e-6
    fe.close()
