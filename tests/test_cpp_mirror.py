"""The C++ mirror of the reference interfaces (include/d2fe.hpp, the host side a D2SLAM call site compiles against) built with
g++ against the C ABI library: compiles and links everywhere; on a GPU box the driver program tests/cpp/mirror_test.cpp runs
SuperPoint::infer / matchKNN / cross-check / getFeatureHalfImg / detectPoints / opticalflowTrackPyr and every output is compared
with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from d2slam_amd.synth import synth_stereo
from d2slam_amd.weights import SP_LAYERS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from d2slam_amd import build as hipbuild
    lib = hipbuild.build()
    exe = str(tmp_path / "mirror_test")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mirror_test.cpp"),
           "-L", os.path.dirname(lib), "-ld2fe_hip", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    assert subprocess.run([exe], capture_output=True).returncode == 2          # usage error path: runs without touching the GPU


def _read_vecs(path, dtypes):
    out, data, pos = [], open(path, "rb").read(), 0
    for dt in dtypes:
        n = struct.unpack_from("<i", data, pos)[0]; pos += 4
        a = np.frombuffer(data, dt, n, pos).copy(); pos += n * np.dtype(dt).itemsize
        out.append(a)
    assert pos == len(data)
    return out


@pytest.mark.gpu
def test_cpp_mirror_matches_oracle(tmp_path, orc, sp_weights):
    exe = _build(tmp_path)
    H, W, maxkp = 240, 320, 150
    l, r = synth_stereo(H, W, seed=8)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<iii", H, W, maxkp))
        for n in SP_LAYERS:
            wt, b = sp_weights[n]
            f.write(struct.pack("<iii", wt.shape[0], wt.shape[1], wt.shape[2]))
            f.write(np.ascontiguousarray(wt, "<f4").tobytes()); f.write(np.ascontiguousarray(b, "<f4").tobytes())
        f.write(l.tobytes()); f.write(r.tobytes())
    env = dict(os.environ)
    res = subprocess.run([exe, fin, fout], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr
    k0, s0, d0, k1, s1, d1, m, mc, hidx, hdesc, fast, gftt, lkp, lkid, lifts = _read_vecs(
        fout, ["<f4"] * 8 + ["<i4", "<f4", "<f4", "<f4", "<f4", "<i8", "<f8"])
    for img, k, s, d in ((l, k0, s0, d0), (r, k1, s1, d1)):
        rk, rs, rd, _, _ = orc.extract_b(img, sp_weights, 0.015, 1, maxkp)
        assert np.array_equal(k.reshape(-1, 2), rk) and np.array_equal(s, rs) and np.abs(d.reshape(-1, 256) - rd).max() <= 1e-6
    A, B, ka, kb = d0.reshape(-1, 256), d1.reshape(-1, 256), k0.reshape(-1, 2), k1.reshape(-1, 2)
    rq, rt, rdist = orc.match_knn(A, B, 0.8, ka, kb, 0.2 * W)
    mm = m.reshape(-1, 3)
    assert np.array_equal(mm[:, 0], rq.astype(np.float32)) and np.array_equal(mm[:, 1], rt.astype(np.float32)) and np.array_equal(mm[:, 2], rdist)
    cq, ct, cd = orc.match_crosscheck(A, B)
    cc = mc.reshape(-1, 3)
    assert np.array_equal(cc[:, 0], cq.astype(np.float32)) and np.array_equal(cc[:, 1], ct.astype(np.float32)) and np.array_equal(cc[:, 2], cd)
    ridx = orc.half_img(ka, True, W, 200.0)
    assert np.array_equal(hidx, ridx) and np.array_equal(hdesc.reshape(-1, 256), A[ridx])
    # detectPoints with no existing points = the detector's list, thinned to feature_min_dist
    rfast, _ = orc.fast_by_region(l, 150, 3, 4)
    assert len(fast) > 0 and _thin(rfast, 20.0, 150).tobytes() == fast.reshape(-1, 2).tobytes()
    assert _thin(orc.good_features(l, 150, 0.01, 20.0), 20.0, 150).tobytes() == gftt.reshape(-1, 2).tobytes()
    fp = fast.reshape(-1, 2)
    rp, rst = orc.lk_track(orc.pyr_build(l), orc.pyr_build(r), W, H, fp, fp)
    assert np.array_equal(lkp.reshape(-1, 2), rp[rst > 0]) and np.array_equal(lkid, 1000 + np.nonzero(rst)[0])
    # A8 helpers of the C++ mirror: liftProjective (pinhole+radtan, MEI, cylindrical) + normalisation
    ref = _lift_reference(ka, H, W)
    assert lifts.shape == ref.shape and np.abs(lifts - ref).max() < 1e-12


@pytest.mark.gpu
def test_cpp_mirror_keep_all_does_not_throw(tmp_path, orc, sp_weights):
    """SuperPointConfig::max_keypoints = -1 through the C++ mirror: infer() used to size its buffers from -1 (std::length_error out of a
    function whose contract is "return false").  Now: a stated starting capacity, D2FE_ERR_TRUNCATED honoured by running again with room, and the result is
    the reference's -- every keypoint above the threshold in raster order (topKeypoints with k == -1, superpoint_tensorrt.cpp:241-253), here ~700 of them
    from a starting capacity of 64 (three growth steps)."""
    from d2slam_amd.synth import synth_image
    from oracle import ref as spref
    exe = _build(tmp_path)
    H, W = 120, 160
    img = synth_image(H, W, 14)
    f = orc.superpoint_forward(img, sp_weights)
    thr = float(np.sort(f["semi"].reshape(-1))[-700])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fh:
        fh.write(struct.pack("<iii", H, W, -1))
        for n in SP_LAYERS:
            wt, b = sp_weights[n]
            fh.write(struct.pack("<iii", wt.shape[0], wt.shape[1], wt.shape[2]))
            fh.write(np.ascontiguousarray(wt, "<f4").tobytes()); fh.write(np.ascontiguousarray(b, "<f4").tobytes())
        fh.write(img.tobytes()); fh.write(img.tobytes())
        fh.write(struct.pack("<fi", thr, 64))
    res = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.returncode, res.stderr)
    k, s, d = _read_vecs(fout, ["<f4"] * 3)
    rk, rs, rd, _, _ = orc.extract_b(img, sp_weights, thr, 1, -1)
    assert 600 < len(rk) < 1024
    assert np.array_equal(k.reshape(-1, 2), rk) and np.array_equal(s, rs) and np.abs(d.reshape(-1, 256) - rd).max() <= 1e-6
    if spref.available():      # the reference's own processOutput compiled in place, max_keypoints = -1
        pk, ps, pd = spref.superpoint_post(f["semi"], f["desc"], thr, 1, -1)
        assert np.array_equal(k.reshape(-1, 2), pk) and np.array_equal(s, ps) and np.abs(d.reshape(-1, 256) - pd).max() <= 1e-6


def _lift_reference(kps, H, W):
    """numpy fp64 restatement of the three liftProjective models (camera_models/src/camera_models/*.cc), normalised."""
    x, y = kps[:, 0].astype(np.float64), kps[:, 1].astype(np.float64)

    def undist(mx_d, my_d, k1, k2, p1, p2):
        mx, my = mx_d.copy(), my_d.copy()
        for i in range(8):
            ax, ay = (mx_d, my_d) if i == 0 else (mx, my)
            r2 = ax * ax + ay * ay
            rad = k1 * r2 + k2 * r2 * r2
            dx = ax * rad + 2 * p1 * ax * ay + p2 * (r2 + 2 * ax * ax)
            dy = ay * rad + 2 * p2 * ax * ay + p1 * (r2 + 2 * ay * ay)
            mx, my = mx_d - dx, my_d - dy
        return mx, my
    out = []
    mx, my = undist(x / 385.0 - 322.5 / 385.0, y / 386.0 - 241.0 / 386.0, 0.01, -0.02, 0.001, -0.0005)
    out.append(np.stack([mx, my, np.ones_like(mx)], 1))
    xi, k1, k2, p1, p2 = 2.2176903753419963, -0.17703529535292872, 0.7517933338735744, -0.0008911425891703079, 2.1653595535258756e-05
    for g1, g2, u0, v0 in ((0.9 * W, 0.9 * W, W // 2 + 0.3, H // 2 + 0.2),
                           (1162.5434300524314, 1161.839362615319, 660.6393183718625, 386.1663300322095)):
        mx, my = undist(x / g1 - u0 / g1, y / g2 - v0 / g2, k1, k2, p1, p2)
        r2 = mx * mx + my * my
        with np.errstate(invalid="ignore"):
            out.append(np.stack([mx, my, 1.0 - xi * (r2 + 1.0) / (xi + np.sqrt(1.0 + (1.0 - xi * xi) * r2))], 1))
    f = W / 3.4906585039886591
    phi = x / f - (W // 2) / f
    z = np.where(np.abs(phi) > np.pi / 2, -1.0, 1.0)
    X = z * np.tan(phi)
    out.append(np.stack([X, (y / f - (H // 2) / f) * np.sqrt(X * X + z * z), z], 1))
    out = [v / np.linalg.norm(v, axis=1, keepdims=True) for v in out]
    out = [v[~np.isnan(v).any(1)] for v in out]      # loop_cam.cpp:625-631: a NaN lift drops the landmark
    return np.concatenate(out).reshape(-1)


def _thin(pts, min_dist, lack):
    out = []
    for p in pts:
        if all(np.sqrt(np.float64(p[0] - q[0]) ** 2 + np.float64(p[1] - q[1]) ** 2) >= min_dist for q in out):
            out.append(p)
        if len(out) >= lack:
            break
    return np.asarray(out, np.float32).reshape(-1, 2)
