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
    k0, s0, d0, k1, s1, d1, m, mc, hidx, hdesc, fast, gftt, lkp, lkid = _read_vecs(
        fout, ["<f4"] * 8 + ["<i4", "<f4", "<f4", "<f4", "<f4", "<i8"])
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


def _thin(pts, min_dist, lack):
    out = []
    for p in pts:
        if all(np.sqrt(np.float64(p[0] - q[0]) ** 2 + np.float64(p[1] - q[1]) ** 2) >= min_dist for q in out):
            out.append(p)
        if len(out) >= lack:
            break
    return np.asarray(out, np.float32).reshape(-1, 2)
