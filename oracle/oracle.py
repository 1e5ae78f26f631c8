"""ctypes front-end of the CPU oracle (oracle/d2fe_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under d2slam_amd/ imports this module.  Parity status: network + variant-A sampling
pinned to the reference's own Python modules (tests/golden/reference_notebook.npz), the rest unpinned -- see the
header of d2fe_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SP_LAYERS = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b",
             "convPa", "convPb", "convDa", "convDb"]


def build(force=False):
    so = os.path.join(_HERE, "libd2fe_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("d2fe_oracle.c", "d2fe_oracle_lk.c", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libd2fe_oracle.so")
        build()
        _LIB = C.CDLL(so)
        _LIB.orc_expf.restype = C.c_float
        _LIB.orc_expf.argtypes = [C.c_float]
        _LIB.orc_l2_dist.restype = C.c_float
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def prep_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h, w), np.float32)
    lib().orc_prep_u8(_p(img), h, w, w, _p(out))
    return out


def conv(x, wgt, bias, relu):
    """x: [H,W,Cin] NHWC; wgt: [Cout,Cin,K,K]; returns [H,W,Cout]."""
    x = _f(x); wgt = _f(wgt); bias = _f(bias)
    h, w, cin = x.shape
    cout, cin2, k, _ = wgt.shape
    assert cin == cin2
    out = np.empty((h, w, cout), np.float32)
    lib().orc_conv(_p(x), h, w, cin, _p(wgt), _p(bias), cout, k, int(relu), _p(out))
    return out


def conv_wino(x, wgt, bias, relu):
    """3x3 conv as Winograd F(2x2,3x3) in the evaluation order of the HIP library's precision mode 2 (orc_conv3x3_wino)."""
    x = _f(x); wgt = _f(wgt); bias = _f(bias)
    h, w, cin = x.shape
    cout, cin2, k, _ = wgt.shape
    assert cin == cin2 and k == 3 and cin % 8 == 0
    out = np.empty((h, w, cout), np.float32)
    lib().orc_conv3x3_wino(_p(x), h, w, cin, _p(wgt), _p(bias), cout, int(relu), _p(out))
    return out


def maxpool2(x):
    x = _f(x)
    h, w, c = x.shape
    out = np.empty((h // 2, w // 2, c), np.float32)
    lib().orc_maxpool2(_p(x), h, w, c, _p(out))
    return out


def expf(x):
    return float(lib().orc_expf(C.c_float(float(x))))


def softmax_semi(logits):
    logits = _f(logits)
    hc, wc, c = logits.shape
    assert c == 65
    out = np.empty((hc * 8, wc * 8), np.float32)
    lib().orc_softmax_semi(_p(logits), hc, wc, _p(out))
    return out


def l2norm_rows(x):
    x = _f(x)
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    out = np.empty_like(x2)
    lib().orc_l2norm_rows(_p(x2), x2.shape[0], x2.shape[1], _p(out))
    return out.reshape(shp)


def superpoint_forward(img_u8, weights, return_trunk=False, wino=False):
    """A1+A2 (d2frontend/superpoint.ipynb:300-374). Returns semi [H,W], desc_map [H/8,W/8,256] (normalised),
    plus raw logits / raw desc for finer-grained checks.  wino=True: the eight 3x3 layers with Cin >= 64 in the Winograd
    evaluation order of the HIP library's precision mode 2 (conv1a and the 1x1 heads are the same chains in every mode)."""
    x = prep_u8(img_u8)[:, :, None]
    w = weights
    c3 = conv_wino if wino else conv
    x = conv(x, *w["conv1a"], True)
    x = c3(x, *w["conv1b"], True)
    x = maxpool2(x)
    x = c3(x, *w["conv2a"], True)
    x = c3(x, *w["conv2b"], True)
    x = maxpool2(x)
    x = c3(x, *w["conv3a"], True)
    x = c3(x, *w["conv3b"], True)
    x = maxpool2(x)
    x = c3(x, *w["conv4a"], True)
    x = c3(x, *w["conv4b"], True)
    cpa = c3(x, *w["convPa"], True)
    logits = conv(cpa, *w["convPb"], False)
    cda = c3(x, *w["convDa"], True)
    draw = conv(cda, *w["convDb"], False)
    semi = softmax_semi(logits)
    desc = l2norm_rows(draw)
    out = dict(semi=semi, desc=desc, logits=logits, desc_raw=draw)
    if return_trunk:
        out["trunk"] = x
    return out


def select_b(semi, thr, border, max_kp, cap=None):
    semi = _f(semi)
    h, w = semi.shape
    cap = cap or h * w
    kps = np.empty((cap, 2), np.float32); sc = np.empty(cap, np.float32); idx = np.empty(cap, np.int32)
    n = lib().orc_select_b(_p(semi), h, w, C.c_float(thr), border, max_kp, _p(kps), _p(sc), _p(idx), cap)
    return kps[:n].copy(), sc[:n].copy(), idx[:n].copy()


def sample_b(desc_map, kps):
    desc_map = _f(desc_map); kps = _f(kps).reshape(-1, 2)
    hc, wc, dim = desc_map.shape
    n = kps.shape[0]
    out = np.empty((n, dim), np.float32)
    lib().orc_sample_b(_p(desc_map), hc, wc, dim, _p(kps), n, _p(out))
    return out


def nms2_a(prob, thr, dist_thresh, max_num, border=0, cap=None):
    prob = _f(prob)
    h, w = prob.shape
    cap = cap or max(max_num, 1)
    kps = np.empty((cap, 2), np.float32); sc = np.empty(cap, np.float32)
    n = lib().orc_nms2_a(_p(prob), h, w, C.c_float(thr), dist_thresh, border, max_num, _p(kps), _p(sc), cap)
    return kps[:n].copy(), sc[:n].copy()


def sample_a(desc_map, kps, img_w, img_h, pca_comp=None, pca_mean=None):
    desc_map = _f(desc_map); kps = _f(kps).reshape(-1, 2)
    hc, wc, dim = desc_map.shape
    n = kps.shape[0]
    if pca_comp is not None:
        pca_comp = _f(pca_comp); pca_mean = _f(pca_mean)
        pd = pca_comp.shape[0]
        out = np.empty((n, pd), np.float32)
        lib().orc_sample_a(_p(desc_map), hc, wc, dim, img_w, img_h, _p(kps), n, _p(pca_comp), _p(pca_mean), pd, _p(out))
    else:
        out = np.empty((n, dim), np.float32)
        lib().orc_sample_a(_p(desc_map), hc, wc, dim, img_w, img_h, _p(kps), n, None, None, 0, _p(out))
    return out


def extract_b(img_u8, weights, thr=0.015, border=1, max_kp=200, wino=False):
    """Full variant-B extractor == SuperPoint::infer (superpoint_tensorrt.cpp:161-183)."""
    f = superpoint_forward(img_u8, weights, wino=wino)
    kps, sc, idx = select_b(f["semi"], thr, border, max_kp)
    d = sample_b(f["desc"], kps)
    return kps, sc, d, idx, f


def l2_dist(a, b):
    a = _f(a); b = _f(b)
    return float(lib().orc_l2_dist(_p(a), _p(b), a.shape[0]))


def match_knn(a, b, ratio=0.8, pts_a=None, pts_b=None, radius=-1.0):
    a = _f(a); b = _f(b)
    na, dim = a.shape if a.ndim == 2 else (0, b.shape[1])
    nb = b.shape[0]
    cap = max(na, 1)
    q = np.empty(cap, np.int32); t = np.empty(cap, np.int32); d = np.empty(cap, np.float32)
    pa = _f(pts_a) if pts_a is not None else np.zeros((max(na, 1), 2), np.float32)
    pb = _f(pts_b) if pts_b is not None else np.zeros((max(nb, 1), 2), np.float32)
    n = lib().orc_match_knn(_p(a), na, _p(b), nb, dim, C.c_double(ratio), _p(pa), _p(pb),
                            C.c_double(radius if pts_a is not None else -1.0), _p(q), _p(t), _p(d), cap)
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


def match_crosscheck(a, b):
    a = _f(a); b = _f(b)
    na, dim = a.shape
    nb = b.shape[0]
    cap = max(na, 1)
    q = np.empty(cap, np.int32); t = np.empty(cap, np.int32); d = np.empty(cap, np.float32)
    n = lib().orc_match_crosscheck(_p(a), na, _p(b), nb, dim, _p(q), _p(t), _p(d), cap)
    return q[:n].copy(), t[:n].copy(), d[:n].copy()


def half_img(pts, require_left, width_undistort, undistort_fov):
    pts = _f(pts).reshape(-1, 2)
    m = np.empty(max(pts.shape[0], 1), np.int32)
    n = lib().orc_half_img(_p(pts), pts.shape[0], int(require_left), width_undistort, C.c_double(undistort_fov), _p(m))
    return m[:n].copy()


# ---- A9 NetVLAD stand-in (see d2slam_amd/netvlad.py and the A9 block of d2fe_oracle.c) -------------------------------
def conv2d_same(x, wgt, bias, stride, act):
    x = _f(x); wgt = _f(wgt); bias = _f(bias)
    h, w, cin = x.shape
    cout, _, k, _ = wgt.shape
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    out = np.empty((ho, wo, cout), np.float32)
    lib().orc_conv2d_same(_p(x), h, w, cin, _p(wgt), _p(bias), cout, k, stride, act, _p(out))
    return out


def dwconv3x3_same(x, wgt, bias, stride, act):
    x = _f(x); wgt = _f(wgt); bias = _f(bias)
    h, w, c = x.shape
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    out = np.empty((ho, wo, c), np.float32)
    lib().orc_dwconv3x3_same(_p(x), h, w, c, _p(wgt), _p(bias), stride, act, _p(out))
    return out


def netvlad_forward(img_u8, nv, pca=None, return_layers=False):
    """gray u8 [H,W] -> 4096-D (or PCA'd) global descriptor; mirrors MobileNetVLADONNX::inference."""
    x = ((np.ascontiguousarray(img_u8, np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0))[:, :, None]
    outs = []
    for l in nv["layers"]:
        if l["kind"] == "conv":
            y = conv2d_same(x, l["weight"], l["bias"], l["stride"], l["act"])
        elif l["kind"] == "dw":
            y = dwconv3x3_same(x, l["weight"], l["bias"], l["stride"], l["act"])
        else:
            y = conv2d_same(x, l["weight"][:, :, None, None], l["bias"], 1, l["act"])
        if l["res"] >= 0:
            y = (y + outs[l["res"]]).astype(np.float32)
        outs.append(y)
        x = y
    hd = nv["head"]
    feat = conv2d_same(x, hd["pre_w"][:, :, None, None], hd["pre_b"], 1, 0)
    hp, wp, d = feat.shape
    K = hd["assign_w"].shape[0]
    out = np.empty(K * d, np.float32)
    f2 = _f(feat.reshape(-1, d))
    lib().orc_netvlad_head(_p(f2), hp * wp, d, _p(_f(hd["assign_w"])), _p(_f(hd["assign_b"])), _p(_f(hd["centroids"])), K, _p(out))
    raw = out
    if pca is not None:
        comp, mean = _f(pca[0]), _f(pca[1])
        o2 = np.empty(comp.shape[0], np.float32)
        lib().orc_netvlad_pca(_p(raw), raw.shape[0], _p(comp), _p(mean), comp.shape[0], _p(o2))
        out = o2
    if return_layers:
        return out, outs, feat, raw
    return out


# ---- SURVEY.md section 8(f) next rows -----------------------------------------------------------------------------------
def undistort(src_u8, mapx, mapy, gain=None):
    src = np.ascontiguousarray(src_u8, np.uint8); mapx = _f(mapx); mapy = _f(mapy)
    sh, sw = src.shape; dh, dw = mapx.shape
    dst = np.empty((dh, dw), np.uint8)
    g = _f(gain) if gain is not None else None
    lib().orc_undistort(_p(src), sh, sw, sw, _p(mapx), _p(mapy), _p(g) if g is not None else None, dh, dw, _p(dst))
    return dst


def db_query(db, q, max_index, thres, search_nearest=5):
    db = _f(db); q = _f(q)
    ntotal, dim = db.shape
    k = C.c_int(0); sim = C.c_float(0)
    labels = np.empty(max(search_nearest + max_index, 1), np.int32); sims = np.empty_like(labels, dtype=np.float32)
    r = lib().orc_db_query(_p(db), ntotal, dim, _p(q), search_nearest, max_index, C.c_double(thres), _p(labels), _p(sims),
                           C.byref(k), C.byref(sim))
    return int(r), float(sim.value), labels[:k.value].copy(), sims[:k.value].copy()


def quant_int8(x, double_max=False):
    x = _f(x).reshape(-1)
    out = np.empty(x.shape[0], np.int8)
    lib().orc_quant_int8(_p(x), x.shape[0], int(double_max), _p(out))
    return out


def dequant_int8(q, landmark_num=-1):
    q = np.ascontiguousarray(q, np.int8).reshape(-1)
    out = np.empty(q.shape[0], np.float32)
    lib().orc_dequant_int8(_p(q), q.shape[0], landmark_num, _p(out))
    return out


def tracker_gate(remote, keyframes, thres, quadcam, sp_remote=None, sp_kf=None):
    """getMatchedPrevKeyframe (+ the FOURCORNER_FISHEYE view pairing of trackRemoteFrames), d2featuretracker.cpp:166-235,270-284.
    remote [n_views, dim]; keyframes [n_kf, n_views, dim] oldest first.  Returns None or dict(kf, dir_a, dir_b, pairs, sims)."""
    remote = _f(remote); keyframes = _f(keyframes)
    n_kf, n_views, dim = keyframes.shape
    kf = C.c_int(); da = C.c_int(); db = C.c_int(); npairs = C.c_int()
    pc = np.zeros(4, np.int32); pp = np.zeros(4, np.int32); sims = np.zeros(4, np.float32)
    spr = np.ascontiguousarray(sp_remote, np.int32) if sp_remote is not None else None
    spk = np.ascontiguousarray(sp_kf, np.int32) if sp_kf is not None else None
    lib().orc_tracker_gate.restype = C.c_int
    r = lib().orc_tracker_gate(int(bool(quadcam)), _p(remote), _p(spr) if spr is not None else None, _p(keyframes),
                               _p(spk) if spk is not None else None, n_kf, n_views, dim, C.c_double(thres), C.byref(kf), C.byref(da),
                               C.byref(db), _p(pc), _p(pp), C.byref(npairs), _p(sims))
    if not r:
        return None
    return dict(kf=kf.value, dir_a=da.value, dir_b=db.value, pairs=list(zip(pc[:npairs.value].tolist(), pp[:npairs.value].tolist())), sims=sims)


# ---- section 8(f)-4: LK optical-flow tracker (oracle/d2fe_oracle_lk.c) ---------------------------------------------------------
def pyr_layout(w, h, levels):
    off = (C.c_int * 16)(); ws = (C.c_int * 16)(); hs = (C.c_int * 16)()
    total = lib().orc_pyr_layout(int(w), int(h), int(levels), off, ws, hs)
    return total, list(off[:levels + 1]), list(ws[:levels + 1]), list(hs[:levels + 1])


def pyr_build(img_u8, levels=2):
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img.shape
    total, _, _, _ = pyr_layout(w, h, levels)
    pyr = np.zeros(total, np.uint8)
    lib().orc_pyr_build(_p(img), w, h, w, int(levels), _p(pyr))
    return pyr


def lk_track(prev_pyr, cur_pyr, w, h, prev_pts, cur_init, track_type=0, move_cols=0.0, levels=2, win=21, iters=30):
    pp = _f(prev_pts).reshape(-1, 2); ci = _f(cur_init).reshape(-1, 2)
    n = pp.shape[0]
    out = np.zeros((n, 2), np.float32); st = np.zeros(n, np.uint8)
    lib().orc_lk_track(_p(prev_pyr), _p(cur_pyr), int(w), int(h), int(levels), _p(pp), _p(ci), n, int(track_type),
                       C.c_float(move_cols), int(win), int(iters), _p(out), _p(st))
    return out, st


def fast_by_region(img_u8, features, cols=3, rows=4, threshold=10):
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img.shape
    cap = max(int(features), 1)
    xy = np.zeros((cap, 2), np.float32); resp = np.zeros(cap, np.int32)
    n = lib().orc_fast_by_region(_p(img), w, h, w, int(features), int(cols), int(rows), int(threshold), _p(xy), _p(resp), cap)
    return xy[:n].copy(), resp[:n].copy()


def min_eigen(img_u8):
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img.shape
    eig = np.zeros((h, w), np.float32)
    lib().orc_min_eigen(_p(img), w, h, w, _p(eig))
    return eig


def good_features(img_u8, max_corners, quality=0.01, min_dist=20.0):
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img.shape
    cap = max(int(max_corners), 1) if max_corners > 0 else w * h
    xy = np.zeros((cap, 2), np.float32)
    n = lib().orc_good_features(_p(img), w, h, w, int(max_corners), C.c_double(quality), C.c_double(min_dist), _p(xy), cap)
    return xy[:n].copy()


# ---- section 8(f)-1: undistortion map generation (MEI fisheye -> cylinder / pinhole virtual camera) -----------------------------
def gen_cylinder_map(cam9, width, height, fov_deg):
    cam = np.ascontiguousarray(cam9, np.float64)
    mx = np.zeros((height, width), np.float32); my = np.zeros((height, width), np.float32)
    lib().orc_gen_cylinder_map(_p(cam), int(width), int(height), C.c_double(fov_deg), _p(mx), _p(my))
    return mx, my


def gen_pinhole_map(cam9, q_wxyz, width, height, f):
    cam = np.ascontiguousarray(cam9, np.float64); q = np.ascontiguousarray(q_wxyz, np.float64)
    mx = np.zeros((height, width), np.float32); my = np.zeros((height, width), np.float32)
    lib().orc_gen_pinhole_map(_p(cam), _p(q), int(width), int(height), C.c_double(f), _p(mx), _p(my))
    return mx, my


# ---- A1 (variant A / NetVLAD): BGR -> gray, resize to the network size ------------------------------------------------------------
def bgr2gray(bgr_u8):
    img = np.ascontiguousarray(bgr_u8, np.uint8)
    h, w, _ = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_bgr2gray(_p(img), w, h, 3 * w, _p(out))
    return out


def resize_linear_u8(src_u8, dw, dh):
    src = np.ascontiguousarray(src_u8, np.uint8)
    sh, sw = src.shape
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), sw, sh, sw, _p(out), int(dw), int(dh))
    return out
