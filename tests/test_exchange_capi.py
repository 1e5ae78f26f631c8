"""d2fe_exchange_* (include/d2fe.h, csrc/exchange.hip) as an API: one process, one GPU, no torch.distributed -- the collective is a callback that copies the rank's
blocks into the gathered buffer on the given stream (world = 1, loopback), so that the whole sequence (device view -> pack -> all-gather -> counts -> gate -> remote
matchKNN -> release -> D2H into a pinned slot) and its error behaviour can be checked on their own.  Replaces loop_net.cpp:24-87 + d2featuretracker.cpp:237-310 (see the
header).  The RCCL form of the same entry points: tests/test_cpp_swarm.py (g++), tests/test_swarm_gpu.py (bench --force-dist), tools/check_rccl_1rank.py."""
import ctypes as C

import numpy as np
import pytest

from d2slam_amd.synth import synth_stereo
from d2slam_amd.weights import synthetic_superpoint_weights

H, W, CAP, F = 120, 160, 60, 2


def test_exchange_config_and_result_structs_match_the_header():
    """the ctypes mirrors of d2fe_exchange_config / d2fe_exchange_result against the field lists of include/d2fe.h (no GPU)"""
    import os
    import re
    from d2slam_amd import api
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "d2fe.h")).read()
    body = hdr[hdr.index("typedef struct {\n  int32_t struct_size;\n  int32_t world, rank;"):hdr.index("} d2fe_exchange_config;")]
    names = []
    body = re.sub(r"/\*.*?\*/", "", body.replace("typedef struct {", ""), flags=re.S)
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        decl = re.sub(r"^(int32_t|double|d2fe_all_gather_fn|void\s*\*)\s*", "", decl)
        names += [re.sub(r"\[.*\]|\*", "", n).strip() for n in decl.split(",")]
    assert names == [f[0] for f in api._ExchangeConfig._fields_], names
    lib = api.load_library()
    c = api._ExchangeConfig()
    lib.d2fe_exchange_default_config(C.byref(c))
    assert c.struct_size == C.sizeof(api._ExchangeConfig) and c.slots == 4 and c.own_stream == 1 and c.world == 1 and abs(c.gate_thres - 0.8) < 1e-12
    rbody = hdr[hdr.index("typedef struct {\n  int64_t ticket;"):hdr.index("} d2fe_exchange_result;")]
    for f in ("ticket", "npairs", "q_idx", "t_idx", "dist", "n_match", "gate_pass", "gate_sims", "gate_n", "phase_ms"):
        assert f in rbody and f in [x[0] for x in api._ExchangeResult._fields_]


def _hip():
    from d2slam_amd import swarm
    return swarm._hip_runtime()


@pytest.mark.gpu
@pytest.mark.parametrize("netvlad,own_stream,wire", [(True, True, "fp32"), (False, True, "fp32"), (True, False, "int8-renorm256")])
def test_exchange_capi_loopback_with_a_device_copy_as_the_collective(netvlad, own_stream, wire):
    from d2slam_amd import api, netvlad as nvm
    hip = _hip()
    calls = []

    def gather(user, d_send, d_recv, nbytes, stream):          # world = 1: the gathered buffer IS the rank's blocks; stream-ordered device copy
        calls.append(nbytes)
        return int(hip.hipMemcpyAsync(C.c_void_p(d_recv), C.c_void_p(d_send), C.c_size_t(nbytes), 3, C.c_void_p(stream)))      # hipMemcpyDeviceToDevice

    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2 * F, precision=api.PREC_F32_WINO))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5))
    if netvlad:
        fe.load_netvlad(nvm.synthetic_netvlad_weights())
    LANES = 3
    pipe = api.StereoPipe(fe, lanes=LANES, frames=F, width=W, height=H, cap=CAP, netvlad=netvlad)
    NS = LANES + 2
    x = api.Exchange(pipe, comm=None, world=1, rank=0, wire=wire, loopback=True, slots=NS, own_stream=own_stream, timing=True, all_gather=gather)
    assert x.npairs == F and (x.stream is not None) == own_stream
    assert x.block_bytes == (api.block_bytes_int8(CAP, fe.netvlad_dim if netvlad else 0) if wire != "fp32" else 4 * api.block_words(CAP, fe.netvlad_dim if netvlad else 0))
    frames = []
    for i in range(7):
        fr = [synth_stereo(H, W, seed=700 + 3 * i + k) for k in range(F)]
        frames.append((np.stack([p[0] for p in fr]), np.stack([p[1] for p in fr])))
    tk = [pipe.submit(*frames[i]) for i in range(LANES)]
    for j in range(LANES):
        x.enqueue(tk[j], j)
    with pytest.raises(api.D2FEError) as ei:                    # the slot's previous exchange has not been collected
        x.enqueue(tk[0], 0)
    assert ei.value.code == -3
    for j in range(LANES):
        o = pipe.wait(tk[j])
        r = x.collect(j)
        assert r["ticket"] == tk[j] and len(r["phase_ms"]) == 5 and r["phase_ms"][0] >= 0
        for f in range(F):
            n = int(o["n_kp"][f])
            assert n > 10
            if wire == "fp32":                                  # a frame against itself: keypoint i matches keypoint i at distance 0
                assert int(r["mn"][f]) == n and np.array_equal(r["mq"][f, :n], np.arange(n)) and np.array_equal(r["mt"][f, :n], np.arange(n)) and not r["md"][f, :n].any()
            else:                                               # against its int8 copy, re-normalised: the same pairing, distances of quantisation size
                k = int(r["mn"][f])
                assert k > 0.5 * n and np.array_equal(r["mq"][f, :k], r["mt"][f, :k]) and float(r["md"][f, :k].max()) < 0.05
        if netvlad:
            assert np.all(r["gate_pass"] == 1) and r["gate_n"] == F and np.all(r["sims"] > 0.99)
        else:
            assert r["gate_pass"] is None and r["gate_n"] == 0
    with pytest.raises(api.D2FEError):                          # nothing enqueued on this slot
        x.collect(NS - 1)
    assert len(calls) == LANES and all(c == F * x.block_bytes for c in calls)
    # the pipe goes on beside the exchange: more submits than result blocks, every view released
    for i in range(LANES, 7):
        t = pipe.submit(*frames[i]); x.enqueue(t, i % NS); pipe.wait(t); x.collect(i % NS)
    x.close(); pipe.close(); fe.close()


@pytest.mark.gpu
def test_exchange_capi_refuses_bad_configurations():
    from d2slam_amd import api, netvlad as nvm
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2 * F))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    pipe = api.StereoPipe(fe, lanes=2, frames=F, width=W, height=H, cap=CAP, netvlad=True)
    ok = lambda *a: 0
    for kw, what in ((dict(world=1, loopback=False, all_gather=ok), "nothing to exchange"), (dict(world=2, rank=2, all_gather=ok), "bad exchange configuration"),
                     (dict(world=2, rank=0), "neither an RCCL communicator nor"), (dict(world=2, rank=0, slots=0, all_gather=ok), "bad exchange configuration")):
        with pytest.raises(api.D2FEError, match=what):
            api.Exchange(pipe, comm=None, **kw)
    x = api.Exchange(pipe, comm=None, world=1, loopback=True, all_gather=lambda *a: 7)      # a failing collective: reported, the view released, the pipe unharmed
    t = pipe.submit(*[np.stack([synth_stereo(H, W, seed=k)[s] for k in range(F)]) for s in (0, 1)])
    with pytest.raises(api.D2FEError, match="all-gather callback failed"):
        x.enqueue(t, 0)
    assert int(pipe.wait(t)["n_kp"][0]) > 10
    for i in range(6):                                          # 2 * lanes + 2 more passes: an unreleased view would refuse one of these submits
        pipe.wait(pipe.submit(*[np.stack([synth_stereo(H, W, seed=10 * i + k)[s] for k in range(F)]) for s in (0, 1)]))
    x.close(); pipe.close(); fe.close()
