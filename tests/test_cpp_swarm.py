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


def _build_exchange(tmp_path):
    """exchange_test.cpp links ONLY libd2fe_hip.so: RCCL is loaded by the library (dlopen) when d2fe_rccl_* / d2fe_exchange_* are used"""
    from d2slam_amd import build as hipbuild
    lib = hipbuild.build()
    exe = str(tmp_path / "exchange_test")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "exchange_test.cpp"),
           "-L", os.path.dirname(lib), "-ld2fe_hip", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--allow-shlib-undefined", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_exchange_compiles_against_the_c_abi_alone(tmp_path):
    exe = _build_exchange(tmp_path)
    assert subprocess.run([exe], capture_output=True).returncode == 2          # usage error path: runs without touching the GPU
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libd2fe_hip" in ldd and "librccl" not in ldd and "torch" not in ldd


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,wire,own_stream", [(4, 0, 0), (2, 0, 1), (4, 1, 1), (3, 2, 0)])
def test_cpp_exchange_entry_points_behind_the_pipe(tmp_path, orc, lanes, wire, own_stream):
    """d2fe_exchange_* + d2fe_rccl_* driven from g++ (tests/cpp/exchange_test.cpp: no Python, no torch in the process): a ONE-rank RCCL communicator made by the
    library itself, loopback, the sequence on the producing lane's stream (own_stream = 0) or on a stream of its own.  Per submit: the pipe's keypoint counts equal the
    single calls', every left frame against ITSELF matches keypoint i with keypoint i at distance 0 (fp32 wire) or -- int8 wire, the reference's LCM precision --
    equals the oracle's matchKNN of the frame's descriptors against their quantised-and-decoded copy; the gate passes (similarity 1)."""
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights, save_superpoint_d2fw, save_netvlad_d2fw
    exe = _build_exchange(tmp_path)
    H, W, cap, F, steps = 120, 160, 60, 2, 9
    w = synthetic_superpoint_weights(dustbin_bias=7.5); nv = nvm.synthetic_netvlad_weights()
    sp, nvp, fin, fout = (str(tmp_path / n) for n in ("sp.d2fw", "nv.d2fw", "frames.bin", "out.bin"))
    save_superpoint_d2fw(sp, w); save_netvlad_d2fw(nvp, nv)
    frames = []
    with open(fin, "wb") as f:
        f.write(struct.pack("<5i", steps, F, H, W, cap))
        for i in range(steps):
            fr = [synth_stereo(H, W, seed=500 + 5 * i + k) for k in range(F)]
            l, r = np.stack([p[0] for p in fr]), np.stack([p[1] for p in fr])
            frames.append((l, r)); f.write(l.tobytes()); f.write(r.tobytes())
    res = subprocess.run([exe, sp, nvp, fin, fout, str(lanes), str(wire), str(own_stream)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "exchange_test OK" in res.stdout and "rccl" in res.stdout.lower()
    data, pos = open(fout, "rb").read(), 0

    def take(dt, n):
        nonlocal pos
        a = np.frombuffer(data, dt, n, pos).copy(); pos += n * np.dtype(dt).itemsize
        return a
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=cap, input_width=W, input_height=H, max_batch=2 * F, precision=api.PREC_F32_WINO))
    fe.load_superpoint(w)
    total = 0
    for i in range(steps):
        nkp = take("<i4", 2 * F); npairs = int(take("<i4", 1)[0]); nm = take("<i4", npairs); gp = take("<i4", npairs)
        q = take("<i4", npairs * cap).reshape(npairs, cap); t = take("<i4", npairs * cap).reshape(npairs, cap); d = take("<f4", npairs * cap).reshape(npairs, cap)
        assert npairs == F and np.all(gp == 1)
        ext = fe.extract_batch(np.concatenate(frames[i]), cap=cap)
        for f in range(F):
            n = int(nkp[f])
            assert n == len(ext[f][0]) and n > 10
            if wire == 0:
                assert nm[f] == n and np.array_equal(q[f, :n], np.arange(n)) and np.array_equal(t[f, :n], np.arange(n)) and not d[f, :n].any()
            else:
                da = ext[f][2]
                qb = orc.quant_int8(da.reshape(-1))
                if wire == 1:
                    db = orc.dequant_int8(qb, n).reshape(n, 256)
                else:
                    xq = (qb.astype(np.float64) / 127.0).astype(np.float32).reshape(n, 256)
                    db = (xq / np.linalg.norm(xq, axis=1, keepdims=True)).astype(np.float32)
                rq, rt, rd = orc.match_knn(da, db, 0.8)
                k = int(nm[f])
                assert k == len(rq) and np.array_equal(q[f, :k], rq) and np.array_equal(t[f, :k], rt)
                assert np.abs(d[f, :k] - rd).max(initial=0) <= (0 if wire == 1 else 1e-6)
            total += int(nm[f])
    assert pos == len(data) and (total > 100 or wire == 1)
    fe.close()
