"""The cross-agent exchange as a C++ call site writes it (tests/cpp/swarm_test.cpp): g++ against the C ABI, the HIP runtime
API and RCCL's ncclAllGather.  Compiles and links everywhere; on a GPU box it runs as a ONE-rank RCCL communicator (RCCL refuses two ranks per
device) and every output -- gathered blocks, gate decisions, similarities, match lists -- is compared with the Python path (d2slam_amd/api.py, the one
bench.py --gpus N drives) and the oracle.  Reference: loop_net.cpp:24-87 (broadcast), d2featuretracker.cpp:185-203 (gate), :237-310 (remote tracking)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from d2slam_amd import build as hipbuild
    lib = hipbuild.build()
    exe = str(tmp_path / "swarm_test")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "cpp", "swarm_test.cpp"), "-L", os.path.dirname(lib), "-ld2fe_hip", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_swarm_compiles_and_links_against_rccl(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    assert subprocess.run([exe], capture_output=True).returncode == 2          # usage error path: runs without touching the GPU
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "librccl" in ldd and "libd2fe_hip" in ldd


def _unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def _read_vecs(path, dtypes):
    out, data, pos = [], open(path, "rb").read(), 0
    for dt in dtypes:
        n = struct.unpack_from("<i", data, pos)[0]; pos += 4
        out.append(np.frombuffer(data, dt, n, pos).copy()); pos += n * np.dtype(dt).itemsize
    assert pos == len(data)
    return out


@pytest.mark.gpu
def test_cpp_swarm_two_devices_or_clean_skip(tmp_path):
    """`swarm_test --two-devices`: the exchange as TWO ranks of one process on devices 0 and 1 (ncclCommInitAll, grouped ncclAllGather) -- a real N > 1 RCCL
    collective between pack and gate, self-checked (gathered buffers identical on both devices and equal to both agents' packed blocks, symmetric gate
    similarities and match counts).  On a 1-GPU box the program reports that and exits 77: a skip, not a failure."""
    exe = _build(tmp_path)
    res = subprocess.run([exe, "--two-devices"], capture_output=True, text=True, timeout=300)
    if res.returncode == 77:
        assert "SKIP" in res.stdout
        pytest.skip(res.stdout.strip())
    assert res.returncode == 0, res.stdout + res.stderr
    assert "two devices" in res.stdout


@pytest.mark.gpu
def test_cpp_swarm_sequence_equals_python_path_and_oracle(tmp_path, orc):
    import torch
    from d2slam_amd import api
    exe = _build(tmp_path)
    F, cap, G, thres = 4, 60, 512, 0.55
    rng = np.random.RandomState(7)
    base = _unit(rng.randn(cap, 256))
    desc = np.stack([_unit(base[rng.permutation(cap)] + 0.02 * rng.randn(cap, 256)) for _ in range(F)]).astype(np.float32)     # the frames see the same scene
    cnt = np.array([cap, cap - 7, cap - 20, 33], np.int32)
    kps = (rng.rand(F, cap, 2) * 600).astype(np.float32); scores = rng.rand(F, cap).astype(np.float32)
    g0 = _unit(rng.randn(1, G))[0]
    nv = np.stack([_unit((g0 + s * rng.randn(G))[None])[0] for s in (0.01, 0.02, 0.05, 0.09)]).astype(np.float32)       # some pairs pass the gate, some do not
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<iii", F, cap, G))
        for a in (desc, kps, scores):
            f.write(np.ascontiguousarray(a, "<f4").tobytes())
        f.write(cnt.astype("<i4").tobytes()); f.write(np.ascontiguousarray(nv, "<f4").tobytes()); f.write(struct.pack("<d", thres))
    res = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    gath, cpass, csims, cnp, cq, ct, cd, cnm = _read_vecs(fout, ["<f4", "<i4", "<f4", "<i4", "<i4", "<i4", "<f4", "<i4"])
    # ---- the Python path on the same inputs
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=cap, input_width=64, input_height=64, max_batch=1))
    dev = torch.device("cuda", 0)
    BLK = api.block_words(cap, G)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_desc, d_kps, d_sc, d_cnt, d_nv = t(desc), t(kps), t(scores), t(cnt), t(nv)
    blocks = torch.zeros((F, BLK), dtype=torch.float32, device=dev)
    fe.pack_blocks_device(d_desc.data_ptr(), d_kps.data_ptr(), d_sc.data_ptr(), d_cnt.data_ptr(), d_nv.data_ptr(), 0, 1, F, cap, G, blocks.data_ptr())
    fe.sync(); torch.cuda.synchronize()
    assert np.array_equal(gath.view(np.int32), blocks.cpu().numpy().reshape(-1).view(np.int32)), "ncclAllGather of one rank == the packed blocks, bit for bit"
    pairs = [(f, g) for f in range(F) for g in range(F) if g != f]
    sims = np.array([np.dot(nv[f].astype(np.float32), nv[g].astype(np.float32)) for f, g in pairs], np.float32)
    assert np.abs(csims - sims).max() <= 1e-6
    exp_pass = (csims.astype(np.float64) >= thres).astype(np.int32)                 # `dot < thres` rejects (d2featuretracker.cpp:189-203), on the device's own sums
    assert np.array_equal(cpass, exp_pass) and int(cnp[0]) == int(exp_pass.sum()) and 0 < exp_pass.sum() < len(pairs)
    cq, ct, cd = cq.reshape(len(pairs), cap), ct.reshape(len(pairs), cap), cd.reshape(len(pairs), cap)
    nmatch = 0
    for p, (f, g) in enumerate(pairs):
        if not exp_pass[p]:
            assert cnm[p] == 0                                                       # a pair the reference would not have tracked
            continue
        rq, rt, rd = orc.match_knn(desc[f, :cnt[f]], desc[g, :cnt[g]], 0.8)
        n = int(cnm[p])
        assert n == len(rq) and np.array_equal(cq[p, :n], rq) and np.array_equal(ct[p, :n], rt) and np.array_equal(cd[p, :n], rd)
        q2, t2, d2 = fe.match_knn(desc[f, :cnt[f]], desc[g, :cnt[g]], 0.8)
        assert np.array_equal(cq[p, :n], q2) and np.array_equal(cd[p, :n], d2)
        nmatch += n
    assert nmatch > 50
    fe.close()
