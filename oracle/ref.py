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
        for f in ("spref_superpoint_post", "spref_get_keypoints", "spref_match_knn", "spref_half_image", "spref_match_neighbour"):
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
