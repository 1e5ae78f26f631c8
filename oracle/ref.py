"""ctypes front-end of oracle/_ref/libspref.so: the reference's OWN C++ for the post-processing and matching glue,
compiled from /root/reference by oracle/build_ref.py against the stand-in headers of oracle/ref_shim/.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__, bench.py's cpu_baseline leg).  Nothing under d2slam_amd/ imports this.
"""
import ctypes as C
import os

import numpy as np

from . import build_ref

_LIB = None


def available():
    """True when the library exists (prebuilt, e.g. on the GPU box) or can be built (this container)."""
    return os.path.exists(build_ref.LIB) or build_ref.available()


def lib():
    global _LIB
    if _LIB is None:
        so = build_ref.build()
        if so is None:
            raise RuntimeError("oracle/_ref/libspref.so is absent and /root/reference is not available to build it")
        _LIB = C.CDLL(so)
        for f in ("spref_superpoint_post", "spref_get_keypoints", "spref_match_knn", "spref_half_image", "spref_match_neighbour",
                  "spref_db_query", "spref_tracker_gate", "spref_loopcam_match"):
            getattr(_LIB, f).restype = C.c_int
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def superpoint_post(semi, desc_hwc, threshold, remove_borders, max_keypoints, cap=None):
    """SuperPoint::infer's CPU half (superpoint_tensorrt.cpp:161-183 -> processOutput :327-350) on given network outputs.
    semi [H,W]; desc_hwc [hc,wc,256] channel-normalised (transposed here to the CHW layout TensorRT hands the reference)."""
    semi = _f(semi); h, w = semi.shape
    hc, wc, dim = desc_hwc.shape
    chw = _f(np.transpose(desc_hwc, (2, 0, 1)))
    cap = cap or h * w
    kps = np.empty((cap, 2), np.float32); sc = np.empty(cap, np.float32); d = np.empty((cap, dim), np.float32)
    n = lib().spref_superpoint_post(_p(semi), h, w, _p(chw), dim, hc, wc, C.c_float(threshold), int(remove_borders), int(max_keypoints),
                                    _p(kps), _p(sc), _p(d), cap)
    assert n >= 0, n
    return kps[:n].copy(), sc[:n].copy(), d[:n].copy()


def get_keypoints(prob, threshold, nms_dist, max_num):
    """getKeyPoints + NMS2 (superpoint_common.cpp:12-40,107-177)."""
    prob = _f(prob); h, w = prob.shape
    cap = max(max_num, 1)
    kps = np.empty((cap, 2), np.float32); sc = np.empty(cap, np.float32)
    n = lib().spref_get_keypoints(_p(prob), h, w, C.c_float(threshold), int(nms_dist), int(max_num), _p(kps), _p(sc), cap)
    assert n >= 0, n
    return kps[:n].copy(), sc[:n].copy()


def match_knn(a, b, ratio=0.8, pts_a=None, pts_b=None, radius=-1.0):
    """matchKNN (feature_matcher.cpp:4-42) over the stand-in cv::BFMatcher."""
    a = _f(a); b = _f(b)
    na, dim = a.shape; nb = b.shape[0]
    pa = _f(pts_a) if pts_a is not None else None
    pb = _f(pts_b) if pts_b is not None else None
    cap = max(na, 1)
    q = np.empty(cap, np.int32); t = np.empty(cap, np.int32); d = np.empty(cap, np.float32)
    n = lib().spref_match_knn(_p(a), na, _p(b), nb, dim, C.c_double(ratio), _p(pa), _p(pb), C.c_double(radius), _p(q), _p(t), _p(d), cap)
    assert n >= 0, n
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


def half_image(pts, require_left, width_undistort, undistort_fov, dim=256):
    pts = _f(pts).reshape(-1, 2)
    m = np.empty(max(len(pts), 1), np.int32)
    n = lib().spref_half_image(_p(pts), len(pts), dim, int(require_left), int(width_undistort), C.c_double(undistort_fov), _p(m))
    return m[:n].copy()


def match_neighbour(pts_a, desc_a, pts_b, desc_b, type_lr, ratio, enable_search_in_local, search_radius, width_undistort,
                    undistort_fov, enable_knn_match=True):
    """The LEFT_RIGHT (type_lr=1) / RIGHT_LEFT (2) branch of matchLocalFeatures, d2featuretracker.cpp:1146-1181.
    Returns None when the reference's branch returns false (an empty half)."""
    pa = _f(pts_a).reshape(-1, 2); pb = _f(pts_b).reshape(-1, 2)
    da = _f(desc_a); db = _f(desc_b)
    dim = da.shape[1] if da.ndim == 2 and da.shape[0] else (db.shape[1] if db.ndim == 2 else 256)
    cap = max(len(pa), 1)
    q = np.empty(cap, np.int32); t = np.empty(cap, np.int32); d = np.empty(cap, np.float32)
    n = lib().spref_match_neighbour(_p(pa), _p(da), len(pa), _p(pb), _p(db), len(pb), dim, int(type_lr), int(enable_knn_match),
                                    C.c_double(ratio), int(enable_search_in_local), C.c_double(search_radius), int(width_undistort),
                                    C.c_double(undistort_fov), _p(q), _p(t), _p(d), cap)
    if n == -2:
        return None
    assert n >= 0, n
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


# ---- round 3: the reference-owned code either side of the hot path (oracle/ref_shim/spref_api2.cpp) ---------------------------------
def quant_landmarks(x):
    """VisualImageDesc::toLCM, landmark descriptors -> int8 (d2frontend_types.h:230-237: float max)."""
    x = _f(x).reshape(-1); out = np.empty(x.shape[0], np.int8)
    lib().spref_quant_landmarks(_p(x), x.shape[0], _p(out))
    return out


def quant_netvlad(x):
    """VisualImageDesc::toLCM, NetVLAD descriptor -> int8 (d2frontend_types.h:262-268: double max)."""
    x = _f(x).reshape(-1); out = np.empty(x.shape[0], np.int8)
    lib().spref_quant_netvlad(_p(x), x.shape[0], _p(out))
    return out


def dequant(lm_q, landmark_num, nv_q):
    """VisualImageDesc(const ImageDescriptor_t&), d2frontend_types.h:319-341.  Returns (landmark_descriptor, image_desc)."""
    lm_q = np.ascontiguousarray(lm_q, np.int8).reshape(-1); nv_q = np.ascontiguousarray(nv_q, np.int8).reshape(-1)
    lo = np.zeros(max(len(lm_q), 1), np.float32); no = np.zeros(max(len(nv_q), 1), np.float32)
    lib().spref_dequant(_p(lm_q), len(lm_q), int(landmark_num), _p(nv_q), len(nv_q), _p(lo), _p(no))
    return lo[:len(lm_q)].copy(), no[:len(nv_q)].copy()


def db_query(db, q, max_index, thres):
    """LoopDetector::queryIndexFromDatabase (loop_detector.cpp:300-350) over a stand-in faiss::IndexFlatIP.  Returns (label, similarity)."""
    db = _f(db); q = _f(q)
    sim = C.c_float(0)
    r = lib().spref_db_query(_p(db), db.shape[0], db.shape[1], _p(q), int(max_index), C.c_double(thres), C.byref(sim))
    return int(r), float(sim.value)


def tracker_gate(remote, keyframes, thres, quadcam, sp_remote=None, sp_kf=None):
    """getMatchedPrevKeyframe + trackRemoteFrames' view pairing (d2featuretracker.cpp:166-235,270-284); same return as oracle.tracker_gate."""
    remote = _f(remote); keyframes = _f(keyframes)
    n_kf, n_views, dim = keyframes.shape
    kf = C.c_int(); da = C.c_int(); db = C.c_int(); npairs = C.c_int()
    pc = np.zeros(4, np.int32); pp = np.zeros(4, np.int32)
    spr = np.ascontiguousarray(sp_remote, np.int32) if sp_remote is not None else None
    spk = np.ascontiguousarray(sp_kf, np.int32) if sp_kf is not None else None
    r = lib().spref_tracker_gate(3 if quadcam else 0, _p(remote), _p(spr) if spr is not None else None, _p(keyframes),
                                 _p(spk) if spk is not None else None, n_kf, n_views, dim, C.c_double(thres), C.byref(kf), C.byref(da),
                                 C.byref(db), _p(pc), _p(pp), C.byref(npairs))
    if not r:
        return None
    return dict(kf=kf.value, dir_a=da.value, dir_b=db.value, pairs=list(zip(pc[:npairs.value].tolist(), pp[:npairs.value].tolist())))


def loopcam_match(pts_up, desc_up, pts_down, desc_down):
    """matchLocalFeatures of loop_cam.cpp:156-191 (cross-check BFMatcher, ids and the compacted point lists)."""
    pu = _f(pts_up).reshape(-1, 2); pd = _f(pts_down).reshape(-1, 2); du = _f(desc_up); dd = _f(desc_down)
    n = max(len(pu), 1)
    iu = np.empty(n, np.int32); idn = np.empty(n, np.int32); po = np.empty((n, 2), np.float32); pdo = np.empty((n, 2), np.float32)
    k = lib().spref_loopcam_match(_p(pu), _p(du), len(pu), _p(pd), _p(dd), len(pd), du.shape[1], _p(iu), _p(idn), _p(po), _p(pdo))
    return iu[:k].copy(), idn[:k].copy(), po[:k].copy(), pdo[:k].copy()


def gen_cylinder_map(cam9, width, height, fov_deg):
    """FisheyeUndist::generateCylinderMap + genOneUndistMap (fisheye_undistort.h:458-500,559-613) over camodocal's CataCamera / CylindricalCamera
    (vendored camera_models/), compiled in place.  cam9 = (xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0).  Returns (mapx, mapy) float32 [h, w]."""
    c = np.ascontiguousarray(cam9, np.float64)
    mx = np.empty((height, width), np.float32); my = np.empty((height, width), np.float32)
    lib().spref_gen_cylinder_map(_p(c), int(width), int(height), C.c_double(fov_deg), _p(mx), _p(my))
    return mx, my


def gen_pinhole_map(cam9, q_wxyz, width, height, f):
    """The rotated-pinhole genOneUndistMap (fisheye_undistort.h:615-660)."""
    c = np.ascontiguousarray(cam9, np.float64); q = np.ascontiguousarray(q_wxyz, np.float64)
    mx = np.empty((height, width), np.float32); my = np.empty((height, width), np.float32)
    lib().spref_gen_pinhole_map(_p(c), _p(q), int(width), int(height), C.c_double(f), _p(mx), _p(my))
    return mx, my


_LK_CB = []      # the ctypes callbacks must outlive the calls


def lk_track_pyr(prev_img, cur_img, prev_pts, track_type, undistort_fov):
    """The reference's opticalflowTrackPyr (GPU form, opticaltrack_utils.cpp:173-278) + inBorder (:35-41) compiled in place, over a stand-in
    cv::cuda::SparsePyrLKOpticalFlow / buildImagePyramid that call the oracle's restatement (oracle/d2fe_oracle_lk.c).  Returns (cur_pts of the
    surviving points, their ids = indices into prev_pts)."""
    from . import oracle as orc
    L = lib()
    if not _LK_CB:
        ol = orc.lib()
        PB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
        PL = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
        LC = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p)
        _LK_CB.extend([C.cast(ol.orc_pyr_build, PB), C.cast(ol.orc_pyr_layout, PL), C.cast(ol.orc_lk_calc, LC)])
        L.spref_set_lk_backend(*_LK_CB)
    prev = np.ascontiguousarray(prev_img, np.uint8); cur = np.ascontiguousarray(cur_img, np.uint8)
    h, w = prev.shape
    pp = _f(prev_pts).reshape(-1, 2)
    n = len(pp)
    out = np.zeros((max(n, 1), 2), np.float32); ids = np.zeros(max(n, 1), np.int32)
    L.spref_lk_track_pyr.restype = C.c_int
    k = L.spref_lk_track_pyr(_p(prev), _p(cur), w, h, _p(pp), n, int(track_type), C.c_double(undistort_fov), _p(out), _p(ids))
    return out[:k].copy(), ids[:k].copy()


def cata_lift(cam9, pts):
    """camodocal CataCamera::liftProjective (camera_models/, CataCamera.cc:425-487) compiled in place: [n, 3] float64 rays."""
    c = np.ascontiguousarray(cam9, np.float64); p = _f(pts).reshape(-1, 2)
    out = np.zeros((len(p), 3), np.float64)
    lib().spref_cata_lift(_p(c), _p(p), len(p), _p(out))
    return out


_TLIB = None


def torch_available():
    return os.path.exists(build_ref.TORCH_LIB) or build_ref.available()


def compute_descriptors(desc_hwc, kps, img_w, img_h, pca_comp=None, pca_mean=None):
    """The reference's variant-A descriptor sampling, computeDescriptors (superpoint_common.cpp:42-99), compiled in place against the image's
    REAL libtorch (torch::grid_sampler, norm, div).  desc_hwc [hc, wc, 256] channel-normalised (the ONNX graph's `desc` output, transposed here
    to CHW); kps [n, 2] (x, y); pca_comp [pd, 256] (the CSV layout; the reference multiplies by its transpose).  Returns [n, pd or 256]."""
    global _TLIB
    if _TLIB is None:
        import torch  # noqa: F401  (libtorch_cpu / libc10 loaded first)
        so = build_ref.build_torch()
        if so is None:
            raise RuntimeError("oracle/_ref/libspref_torch.so is absent and cannot be built here")
        _TLIB = C.CDLL(so)
        _TLIB.spref_compute_descriptors.restype = C.c_int
    hc, wc, dim = desc_hwc.shape
    chw = _f(np.transpose(desc_hwc, (2, 0, 1)))
    k = _f(kps).reshape(-1, 2)
    n = len(k)
    pd = 0; ct = mean = None
    if pca_comp is not None:
        pc = _f(pca_comp); pd = pc.shape[0]
        ct = np.asfortranarray(pc.T)                 # pca_comp_T = comp^T, [dim, pd], column-major like Eigen::MatrixXf
        ct = np.ascontiguousarray(ct.T.reshape(-1))   # column-major storage of [dim, pd] == row-major storage of [pd, dim]
        mean = _f(pca_mean)
    out = np.zeros((max(n, 1), pd or dim), np.float32)
    m = _TLIB.spref_compute_descriptors(_p(chw), dim, hc, wc, _p(k), n, int(img_w), int(img_h), _p(ct) if ct is not None else None,
                                        _p(mean) if mean is not None else None, pd, _p(out))
    assert m == n * (pd or dim), (m, n)
    return out[:n].copy()


# ---- round 6: LoopCam::extractorImgDescDeepnet (loop_cam.cpp:589-648) compiled in place over the adapter / over the reference's own SuperPoint::infer ---------
class LoopCam:
    """One side of oracle/_ref/libspref_loopcam_{hip,ref}.so (oracle/ref_shim/spref_loopcam.cpp).  side "hip": superpoint_ptr / netvlad_onnx are
    include/d2fe_adapter.cpp over libd2fe_hip.so (needs a GPU; sp_path / nv_path = D2FW weight containers).  side "ref": the reference's own
    SuperPoint::infer + processOutput on network outputs handed in with set_network_outputs()."""

    def __init__(self, side, width, height, max_keypoints, threshold=0.015, remove_borders=1, self_id=0, camera_configuration=0, cams=((0, (400.0, 400.0, 320.0, 240.0)),),
                 sp_path=None, nv_path=None, precision=0):
        from oracle import build_ref
        libs = build_ref.build_loopcam()
        if side not in libs:
            raise RuntimeError("libspref_loopcam_%s.so is not available (no reference tree and no prebuilt library)" % side)
        self.lib = C.CDLL(libs[side])
        self.lib.spref_loopcam_create.restype = C.c_void_p
        self.side, self.W, self.H = side, width, height
        kinds = (C.c_int * len(cams))(*[k for k, _ in cams])
        prm = np.zeros((len(cams), 9), np.float64)
        for i, (_, p) in enumerate(cams):
            prm[i, :len(p)] = p
        self.h = self.lib.spref_loopcam_create(width, height, int(max_keypoints), C.c_float(threshold), int(remove_borders), int(self_id), int(camera_configuration),
                                               len(cams), kinds, _p(prm), (sp_path or "").encode(), (nv_path or "").encode(), int(precision))
        if not self.h:
            raise RuntimeError("spref_loopcam_create failed (side %s)" % side)
        self._keep = None

    def set_network_outputs(self, semi, desc_hwc, netvlad=None):
        semi = _f(semi); chw = _f(np.transpose(desc_hwc, (2, 0, 1)))
        g = _f(netvlad) if netvlad is not None else None
        self._keep = (semi, chw, g)
        self.lib.spref_loopcam_set_network_outputs(C.c_void_p(self.h), _p(semi), _p(chw), _p(g), 0 if g is None else int(g.size))

    def extract(self, img, stamp=12.5, camera_index=0, camera_id=7, superpoint_mode=False, cap_lm=20000, gdim=8192):
        """-> dict of every field the function sets; `img` (u8 [H][W]) is modified in place where the reference does (STEREO_FISHEYE mask)"""
        assert img.dtype == np.uint8 and img.flags.c_contiguous and img.shape == (self.H, self.W)
        head = np.zeros(8, np.float64); pt2d = np.zeros((cap_lm, 2), np.float32); pt3d = np.zeros((cap_lm, 3), np.float64)
        meta = np.zeros((cap_lm, 4), np.float64); color = np.zeros((cap_lm, 3), np.uint8)
        desc = np.zeros(cap_lm * 256, np.float32); sc = np.zeros(cap_lm, np.float32); g = np.zeros(gdim, np.float32)
        nan_w = C.c_int(0)
        n = self.lib.spref_loopcam_extract(C.c_void_p(self.h), C.c_double(stamp), _p(img), self.W, self.H, self.W, int(camera_index), int(camera_id), int(bool(superpoint_mode)),
                                           _p(head), _p(pt2d), _p(pt3d), _p(meta), _p(color), cap_lm, _p(desc), C.c_long(desc.size), _p(sc), C.c_long(sc.size), _p(g),
                                           C.c_long(g.size), C.byref(nan_w))
        assert n >= 0, n
        return dict(stamp=head[0], camera_index=int(head[1]), camera_id=int(head[2]), drone_id=int(head[3]), n_landmarks=int(head[4]),
                    pt2d=pt2d[:n].copy(), pt3d_norm=pt3d[:n].copy(), lm_camera_index=meta[:n, 0].copy(), lm_camera_id=meta[:n, 1].copy(), lm_stamp=meta[:n, 2].copy(),
                    lm_stamp_discover=meta[:n, 3].copy(), color=color[:n].copy(), landmark_descriptor=desc[:int(head[5])].copy(), landmark_scores=sc[:int(head[6])].copy(),
                    image_desc=g[:int(head[7])].copy(), nan_warnings=int(nan_w.value))

    def close(self):
        if self.h:
            self.lib.spref_loopcam_destroy(C.c_void_p(self.h)); self.h = None
