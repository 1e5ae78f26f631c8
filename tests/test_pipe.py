"""Frames in flight (d2fe_pipe_*, include/d2fe.h): the pipe's results equal the single-call entry points' bit for bit -- the reference's per-frame
sequence (loop_cam.cpp:589-648 infer + inference per image, d2featuretracker.cpp:403-456,658-695 matchKNN L<->R and L<->previous L)."""
import numpy as np
import pytest

from d2slam_amd.synth import synth_stereo

H, W, CAP = 120, 160, 80


def _frames(n):
    """consecutive frames of one scene under a small camera motion, so that the temporal matches exist"""
    l0, r0 = synth_stereo(H, W, seed=100)
    rng = np.random.RandomState(5)
    out = []
    for i in range(n):
        sh = (i % 4, (2 * i) % 5)
        noise = rng.randint(-2, 3, (2, H, W))
        out.append((np.clip(np.roll(l0, sh, (0, 1)).astype(np.int16) + noise[0], 0, 255).astype(np.uint8),
                    np.clip(np.roll(r0, sh, (0, 1)).astype(np.int16) + noise[1], 0, 255).astype(np.uint8)))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,group", [(2, 2), (4, 2), (4, 4), (6, 3)])
def test_pipe_netvlad_group_equals_single_calls(lanes, group):
    """netvlad_group: NetVLAD of `group` consecutive submits in one call on the pipe's own stream; same bits, whatever the order of the waits
    (a wait for a ticket whose group is still filling launches the part that is there)."""
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.weights import synthetic_superpoint_weights
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=api.PREC_F32_WINO, keypoint_threshold=0.005))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    nsub = 3 * lanes + 1
    fr = _frames(nsub)
    ref = [fe.extract_all_batch(np.stack(f), 1, cap=CAP) for f in fr]
    for order in ("lagged", "immediate"):
        pipe = api.StereoPipe(fe, lanes=lanes, frames=1, width=W, height=H, cap=CAP, netvlad=True, netvlad_group=group)
        got, tickets = {}, []
        for i in range(nsub):
            tickets.append(pipe.submit(fr[i][0][None], fr[i][1][None]))
            if order == "immediate":
                got[i] = {k: (None if v is None else v.copy()) for k, v in pipe.wait(tickets[i]).items()}
            elif i >= lanes - 1:
                j = i - (lanes - 1)
                got[j] = {k: (None if v is None else v.copy()) for k, v in pipe.wait(tickets[j]).items()}
        for j in range(nsub):
            if j not in got:
                got[j] = {k: (None if v is None else v.copy()) for k, v in pipe.wait(tickets[j]).items()}
        for i in range(nsub):
            ext, g = ref[i]
            np.testing.assert_array_equal(got[i]["netvlad"], g)
            for im in range(2):
                n = int(got[i]["n_kp"][im]); assert n == len(ext[im][0])
                np.testing.assert_array_equal(got[i]["desc"][im, :n], ext[im][2])
        pipe.close()
    fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,frames,netvlad,part,coal,depth", [(1, 1, True, False, 1, 0), (3, 1, True, False, 1, 0), (2, 2, True, False, 1, 0), (4, 1, False, False, 1, 0),
                                                                 (2, 3, True, False, 1, 0), (4, 1, True, True, 1, 0), (3, 2, True, True, 1, 0), (8, 1, True, True, 1, 0),
                                                                 (2, 1, True, False, 2, 0), (3, 1, True, False, 3, 0), (1, 1, False, False, 4, 0), (2, 1, True, True, 2, 0),
                                                                 (4, 1, True, False, 4, 2), (2, 1, True, False, 3, 1), (3, 1, False, False, 4, 3), (4, 1, True, False, 8, 1)])
def test_pipe_equals_single_calls(lanes, frames, netvlad, part, coal, depth):
    """depth > 0: dynamic batching (d2fe_pipe_config.coalesce_depth) -- how many submits share a pass then depends on the device's progress; the results do not"""
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.weights import synthetic_superpoint_weights
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2 * frames, precision=api.PREC_F32_WINO,
                                           keypoint_threshold=0.005))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5))
    if netvlad:
        fe.load_netvlad(nvm.synthetic_netvlad_weights())
    nsub = 2 * lanes * coal + 3              # an odd tail: the last pass is launched by wait() with fewer submits than `coalesce`
    inflight = lanes * coal
    fr = _frames(nsub * frames)
    radius_lr, radius_prev = 0.2 * W, 0.05 * W
    pipe = api.StereoPipe(fe, lanes=lanes, frames=frames, width=W, height=H, cap=CAP, netvlad=netvlad, ratio=0.8, radius_lr=radius_lr, radius_prev=radius_prev, cu_partition=part, coalesce=coal, coalesce_depth=depth)
    got = []
    tickets = []
    for sidx in range(nsub):
        L = np.stack([fr[sidx * frames + f][0] for f in range(frames)]); R = np.stack([fr[sidx * frames + f][1] for f in range(frames)])
        tickets.append(pipe.submit(L, R))
        if len(tickets) > inflight - 1:                    # keep `lanes` passes in flight
            t = tickets[len(got)]
            got.append({k: (None if v is None else v.copy()) for k, v in pipe.wait(t).items()})
    while len(got) < nsub:
        got.append({k: (None if v is None else v.copy()) for k, v in pipe.wait(tickets[len(got)]).items()})
    # single-call reference on the same handle
    prev = None
    nmatch = 0
    for sidx in range(nsub):
        imgs = np.stack([fr[sidx * frames + f][0] for f in range(frames)] + [fr[sidx * frames + f][1] for f in range(frames)])
        if netvlad:
            ext, g = fe.extract_all_batch(imgs, frames, cap=CAP)
        else:
            ext, g = fe.extract_batch(imgs, cap=CAP), None
        o = got[sidx]
        for i, (kps, sc, desc) in enumerate(ext):
            n = int(o["n_kp"][i])
            assert n == len(kps)
            np.testing.assert_array_equal(o["kps_xy"][i, :n], kps)
            np.testing.assert_array_equal(o["scores"][i, :n], sc)
            np.testing.assert_array_equal(o["desc"][i, :n], desc)
        if netvlad:
            np.testing.assert_array_equal(o["netvlad"], g)
        for f in range(frames):
            kl, _, dl = ext[f]; kr, _, dr = ext[frames + f]
            q, t, d = fe.match_knn(dl, dr, 0.8, kl, kr, radius_lr)
            n = int(o["lr_n"][f]); assert n == len(q)
            np.testing.assert_array_equal(o["lr_q"][f, :n], q); np.testing.assert_array_equal(o["lr_t"][f, :n], t); np.testing.assert_array_equal(o["lr_dist"][f, :n], d)
            pv = ext[f - 1] if f > 0 else prev
            n = int(o["prev_n"][f])
            if pv is None:
                assert n == 0
            else:
                q, t, d = fe.match_knn(dl, pv[2], 0.8, kl, pv[0], radius_prev)
                assert n == len(q)
                np.testing.assert_array_equal(o["prev_q"][f, :n], q); np.testing.assert_array_equal(o["prev_t"][f, :n], t); np.testing.assert_array_equal(o["prev_dist"][f, :n], d)
                nmatch += n
        prev = ext[frames - 1]
    assert nmatch > 0
    pipe.close(); fe.close()


@pytest.mark.gpu
def test_pipe_argument_errors():
    from d2slam_amd import api
    from d2slam_amd.weights import synthetic_superpoint_weights
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2))
    with pytest.raises(api.D2FEError):
        api.StereoPipe(fe, lanes=2, width=W, height=H, netvlad=False)          # weights not loaded
    fe.load_superpoint(synthetic_superpoint_weights())
    with pytest.raises(api.D2FEError):
        api.StereoPipe(fe, lanes=2, width=W, height=H, netvlad=True)           # no NetVLAD network
    with pytest.raises(api.D2FEError):
        api.StereoPipe(fe, lanes=0, width=W, height=H, netvlad=False)
    with pytest.raises(api.D2FEError):
        api.StereoPipe(fe, lanes=2, width=2 * W, height=H, netvlad=False)
    p = api.StereoPipe(fe, lanes=2, width=W, height=H, netvlad=False)
    with pytest.raises(api.D2FEError):
        p.wait(0)                                                               # nothing submitted
    p.close(); fe.close()


@pytest.mark.gpu
def test_pipe_stream_placement_is_measured_and_separates_a_lanes_streams():
    """d2fe_pipe_create measures which candidate streams take turns on the device (one hardware pipe) and hands every lane two streams of different classes, whatever the
    process has done to the runtime's queue pool before (here: streams created and destroyed in an irregular pattern first).  d2fe_pipe_stream_placement reports it."""
    import torch
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.weights import synthetic_superpoint_weights
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=api.PREC_F32_WINO, keypoint_threshold=0.005))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    keep = []
    for i in range(7):                 # holes in the pool: seven streams, every other one given back
        keep.append(torch.cuda.Stream())
    del keep[::2]
    shortfalls = []
    for lanes, inline in ((1, None), (2, None), (4, None), (4, True)):
        pipe = api.StereoPipe(fe, lanes=lanes, frames=1, width=W, height=H, cap=CAP, netvlad=True, netvlad_inline=inline)
        cls, n = pipe.stream_placement()
        assert len(cls) == lanes and (n == 0 or 2 <= n <= 8)
        short = []                                                  # the deal is best effort ("as far as the streams at hand allow"): what it should reach is checked,
        for k, (a, b) in enumerate(cls):                            # but a runtime that hands out no stream of some class is reported, not failed
            if inline:
                assert b == -1                                      # no second streams exist
            elif n >= 2:
                assert 0 <= a < n and 0 <= b < n, (lanes, cls)
                if a == b:
                    short.append("lane %d: both streams in class %d" % (k, a))
        if n >= 4 and lanes <= 4 and len({a for a, _ in cls}) != lanes:
            short.append("own streams share a class: %s" % (cls,))            # the lanes' own streams: all different
        if n >= 4 and lanes == 2 and not inline and len({c for ab in cls for c in ab}) != 4:
            short.append("two lanes, not four classes: %s" % (cls,))          # two lanes: four streams in four classes
        shortfalls.extend(short)
        l, r = _frames(1)[0]
        o = pipe.wait(pipe.submit(l[None], r[None]))                # the pipe works on the streams it chose
        assert int(o["n_kp"][0]) > 0
        pipe.close()
    fe.close()
    if shortfalls:
        pytest.skip("the pipes work, but the runtime offered no streams for the ideal deal in this process: " + "; ".join(shortfalls))


@pytest.mark.gpu
def test_pipe_netvlad_stream_modes_give_the_same_bits():
    """d2fe_pipe_config.netvlad_inline: 0 = NetVLAD on a second stream per lane, 1 = on the lane's one stream in front of SuperPoint, 2 (default) = auto (inline above two
    lanes: the device runs four busy streams side by side).  Where a kernel is queued does not change what it computes: every mode returns the same bits, and a bad value is refused."""
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.weights import synthetic_superpoint_weights
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=api.PREC_F32_WINO, keypoint_threshold=0.005))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    fr = _frames(9)
    outs = {}
    for lanes, mode in ((2, False), (2, True), (2, None), (4, False), (4, True), (4, None)):
        pipe = api.StereoPipe(fe, lanes=lanes, frames=1, width=W, height=H, cap=CAP, netvlad=True, netvlad_inline=mode)
        tk = [pipe.submit(l[None], r[None]) for l, r in fr[:lanes]]
        res = []
        for i, (l, r) in enumerate(fr):
            if i >= lanes:
                tk.append(pipe.submit(l[None], r[None]))
            o = pipe.wait(tk[i])
            res.append({k: (None if v is None else v.copy()) for k, v in o.items()})
        outs[(lanes, mode)] = res
        pipe.close()
    ref = outs[(2, False)]
    for key, res in outs.items():
        for a, b in zip(ref, res):
            for k in ("n_kp", "netvlad", "lr_n", "prev_n"):
                np.testing.assert_array_equal(a[k], b[k], err_msg="%s %s" % (key, k))
            for i in range(2):          # rows beyond a count are not part of the result
                n = int(a["n_kp"][i])
                for k in ("kps_xy", "scores", "desc"):
                    np.testing.assert_array_equal(a[k][i, :n], b[k][i, :n], err_msg="%s %s" % (key, k))
            for pre in ("lr", "prev"):
                n = int(a[pre + "_n"][0])
                for k in ("_q", "_t", "_dist"):
                    np.testing.assert_array_equal(a[pre + k][0, :n], b[pre + k][0, :n], err_msg="%s %s" % (key, pre + k))
    c = api._PipeConfig()
    fe._lib.d2fe_pipe_default_config(ctypes_byref(c))
    assert c.netvlad_inline == 2
    c.lanes, c.frames, c.width, c.height, c.cap, c.netvlad_inline = 2, 1, W, H, CAP, 3
    import ctypes as C
    p = C.c_void_p()
    assert fe._lib.d2fe_pipe_create(fe.handle, C.byref(c), C.byref(p)) == -1 and b"netvlad_inline" in fe._lib.d2fe_last_error()
    fe.close()


def ctypes_byref(x):
    import ctypes as C
    return C.byref(x)


@pytest.mark.gpu
def test_handle_with_live_pipes_refuses_destroy_and_reload():
    """A pipe's lanes read the parent handle's packed weights: while a pipe exists d2fe_destroy releases nothing and d2fe_load_* /
    d2fe_set_*_pca return D2FE_ERR_INVALID instead of leaving the lanes with dangling pointers; numpy frames through a pinned_input pipe are refused
    (the DMA would read memory nobody keeps alive); after the pipe is gone everything works again."""
    import ctypes as C
    from d2slam_amd import api
    from d2slam_amd.weights import synthetic_superpoint_weights
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, keypoint_threshold=0.005))
    fe.load_superpoint(w)
    fr = _frames(2)
    ref = fe.extract_batch(np.stack(fr[0]), cap=CAP)
    pipe = api.StereoPipe(fe, lanes=2, frames=1, width=W, height=H, cap=CAP, netvlad=False)
    with pytest.raises(api.D2FEError):
        fe.load_superpoint(w)
    o = pipe.wait(pipe.submit(fr[0][0][None], fr[0][1][None]))
    n = int(o["n_kp"][0]); np.testing.assert_array_equal(o["desc"][0, :n], ref[0][2])
    pipe.close()
    fe.load_superpoint(w)                                        # allowed again
    np.testing.assert_array_equal(fe.extract_batch(np.stack(fr[0]), cap=CAP)[0][2], ref[0][2])
    # d2fe_destroy under a live pipe is DEFERRED (round 6): the handle is marked, the pipe keeps working on its weights, and the last
    # d2fe_pipe_destroy releases the handle (a wrapper that destroys in the wrong order no longer leaks it)
    fe2 = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, keypoint_threshold=0.005))
    fe2.load_superpoint(w)
    p2 = api.StereoPipe(fe2, lanes=2, frames=1, width=W, height=H, cap=CAP, netvlad=False)
    fe2._lib.d2fe_destroy(fe2.handle)
    assert b"live pipes" in fe2._lib.d2fe_last_error() and b"released when the last" in fe2._lib.d2fe_last_error()
    o2 = p2.wait(p2.submit(fr[0][0][None], fr[0][1][None]))
    n2 = int(o2["n_kp"][0]); np.testing.assert_array_equal(o2["desc"][0, :n2], ref[0][2])
    p2.close()                                                   # releases the handle too
    import ctypes as C2
    fe2._h = C2.c_void_p()                                       # the wrapper must not destroy it a second time
    pin = api.StereoPipe(fe, lanes=2, frames=1, width=W, height=H, cap=CAP, netvlad=False, pinned_input=True)
    with pytest.raises(ValueError):
        pin.submit(fr[0][0][None], fr[0][1][None])
    fe.close()                                                   # closes `pin` first (FrontEnd.close), then the handle
    assert not pin._p.value


@pytest.mark.gpu
@pytest.mark.parametrize("coal", [1, 2])
def test_pipe_device_view_feeds_a_consumer_stream(coal):
    """d2fe_pipe_device_view / _release (the hook the cross-agent exchange hangs on): a consumer on its OWN stream packs exchange blocks straight out of the lane's
    result block -- no host trip, no wait on the host -- and what it packed equals the ticket's host results bit for bit; the lane's next write of that block waits
    for the release; a view that is never released fails the submit that would overwrite it (and that failure is final for the pipe)."""
    import torch
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.weights import synthetic_superpoint_weights
    dev = torch.device("cuda", 0)
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=api.PREC_F32_WINO, keypoint_threshold=0.005))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    G = fe.netvlad_dim
    BLK = api.block_words(CAP, G)
    off = {f: api.block_field_offset(CAP, G, f) for f in ("desc", "kps", "scores", "netvlad", "n")}
    fr = _frames(8)
    pipe = api.StereoPipe(fe, lanes=2, frames=1, width=W, height=H, cap=CAP, netvlad=True, coalesce=coal)
    # coal == 2: the consumer's stream is a pool stream the pipe ranks harmless (d2fe_pipe_classify_stream), coal == 1 the first that comes
    mk = lambda: torch.cuda.Stream(device=dev)
    X = pipe.pick_consumer_stream(mk, handle=lambda s: s.cuda_stream) if coal == 2 else mk()
    if coal == 2:            # the class of the picked stream is one the pipe knows (or -1: none of the lanes'); best effort: up to four candidates are tried
        assert -1 <= pipe.classify_stream(X.cuda_stream) < max(pipe.stream_placement()[1], 1)
    blocks = [torch.zeros((1, BLK), dtype=torch.float32, device=dev) for _ in fr]
    tk = []
    for i, (l, r) in enumerate(fr):
        tk.append(pipe.submit(l[None], r[None]))
        if i >= 1:        # one submit behind, as the exchange does it; nothing here waits on the host
            j = i - 1
            v = pipe.device_view(tk[j], X.cuda_stream)
            assert v.frames == 1 and v.cap == CAP and v.netvlad_dim == G
            fe.pack_blocks_device(v.d_desc, v.d_kps_xy, v.d_scores, v.d_n_kp, v.d_netvlad, 0, 1, 1, CAP, G, blocks[j].data_ptr(), stream=X.cuda_stream)
            pipe.device_release(tk[j], X.cuda_stream)
        if i >= 3:
            o = pipe.wait(tk[i - 3])
            X.synchronize()
            b = blocks[i - 3].cpu().numpy()[0]
            n = int(o["n_kp"][0])
            assert int(b.view(np.int32)[off["n"]]) == n and n > 10
            np.testing.assert_array_equal(b[off["desc"]:off["desc"] + n * 256].reshape(n, 256), o["desc"][0, :n])
            np.testing.assert_array_equal(b[off["kps"]:off["kps"] + 2 * n].reshape(n, 2), o["kps_xy"][0, :n])
            np.testing.assert_array_equal(b[off["netvlad"]:off["netvlad"] + G], o["netvlad"][0])
    with pytest.raises(api.D2FEError):
        pipe.device_release(tk[-1], X.cuda_stream)               # no view outstanding
    if coal == 1:
        with pytest.raises(api.D2FEError):
            pipe.device_view(tk[0], X.cuda_stream)               # 8 passes on a ring of 2 * lanes = 4: its block has been rewritten since
    # a view that is never released: the submit whose pass would overwrite its block (2 * lanes passes later) is REFUSED before anything is queued
    # (D2FE_ERR_NOT_READY, round 6: not sticky) -- the tickets in flight stay collectable, and after the release the pipe goes on, same bits as ever
    pipe.device_view(tk[-1], X.cuda_stream)
    more = []
    with pytest.raises(api.D2FEError) as ei:
        for l, r in fr + fr:
            more.append(pipe.submit(l[None], r[None]))
    assert ei.value.code == -3 and len(more) >= 1, (ei.value.code, len(more))
    with pytest.raises(api.D2FEError):
        pipe.submit(fr[0][0][None], fr[0][1][None])               # still refused: the view is still out
    o_last = pipe.wait(more[-1])                                     # finished work is not thrown away
    assert int(o_last["n_kp"][0]) > 10
    pipe.device_release(tk[-1], X.cuda_stream)
    o_next = pipe.wait(pipe.submit(fr[0][0][None], fr[0][1][None]))
    ref0 = fe.extract_batch(np.stack(fr[0]), cap=CAP)
    n0 = int(o_next["n_kp"][0]); np.testing.assert_array_equal(o_next["desc"][0, :n0], ref0[0][2])
    pipe.close(); fe.close()


@pytest.mark.gpu
def test_pipe_waits_in_any_order_twice_and_close_with_passes_in_flight():
    """Tickets may be waited for in any order and more than once (the result block stays valid for 2 * lanes passes); a pipe may be closed while passes are in
    flight; a ticket whose block has been reused is refused, not served stale."""
    from d2slam_amd import api
    from d2slam_amd.weights import synthetic_superpoint_weights
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=api.PREC_F32_WINO, keypoint_threshold=0.005))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5))
    fr = _frames(12)
    ref = [fe.extract_batch(np.stack(f), cap=CAP) for f in fr]
    pipe = api.StereoPipe(fe, lanes=3, frames=1, width=W, height=H, cap=CAP, netvlad=False)
    t = [pipe.submit(fr[i][0][None], fr[i][1][None]) for i in range(3)]
    for i in (2, 0, 1, 0, 2):                                   # out of order, repeated
        o = pipe.wait(t[i])
        for im in range(2):
            n = int(o["n_kp"][im]); assert n == len(ref[i][im][0])
            np.testing.assert_array_equal(o["desc"][im, :n], ref[i][im][2])
    for i in range(3, 10):
        t.append(pipe.submit(fr[i][0][None], fr[i][1][None]))  # blocks only on the lane's own previous pass
    with pytest.raises(api.D2FEError):
        pipe.wait(t[0])                                         # 9 passes later: its block has been written twice since
    o = pipe.wait(t[9])
    n = int(o["n_kp"][0]); np.testing.assert_array_equal(o["kps_xy"][0, :n], ref[9][0][0])
    t.append(pipe.submit(fr[10][0][None], fr[10][1][None])); t.append(pipe.submit(fr[11][0][None], fr[11][1][None]))
    pipe.close()                                                # two passes still in flight
    # the handle is still usable afterwards
    again = fe.extract_batch(np.stack(fr[0]), cap=CAP)
    np.testing.assert_array_equal(again[0][2], ref[0][0][2])
    fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,frames,coal,depth", [(2, 4, 1, 0), (4, 1, 4, 2)])
def test_pipe_at_the_baseline_geometry(lanes, frames, coal, depth):
    """BASELINE configs[1] / the metric configuration: 640 x 480 stereo, 200 keypoints, NetVLAD on the left image, L<->R and L<->previous-L matches -- the
    pipe (what bench.py times) against the single-call entry points, bit for bit."""
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.weights import synthetic_superpoint_weights
    Hb, Wb, capb = 480, 640, 200
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=capb, input_width=Wb, input_height=Hb, max_batch=2 * frames, precision=api.PREC_F32_WINO))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    nsub = 2 * lanes * coal + 1
    scenes = [synth_stereo(Hb, Wb, seed=300 + s) for s in range(4)]
    fr = []
    for i in range(nsub * frames):
        l, r = scenes[i % 4]; sh = ((i // 4) % 3, (2 * (i // 4)) % 5)
        fr.append((np.roll(l, sh, (0, 1)), np.roll(r, sh, (0, 1))))
    pipe = api.StereoPipe(fe, lanes=lanes, frames=frames, width=Wb, height=Hb, cap=capb, netvlad=True, ratio=0.8, coalesce=coal, coalesce_depth=depth)
    tickets = [pipe.submit(np.stack([fr[s * frames + f][0] for f in range(frames)]), np.stack([fr[s * frames + f][1] for f in range(frames)])) for s in range(min(nsub, lanes * coal))]
    got = []
    for s in range(nsub):
        got.append({k: (None if v is None else v.copy()) for k, v in pipe.wait(tickets[s]).items()})
        nxt = len(tickets)
        if nxt < nsub:
            tickets.append(pipe.submit(np.stack([fr[nxt * frames + f][0] for f in range(frames)]), np.stack([fr[nxt * frames + f][1] for f in range(frames)])))
    prev, nprev = None, 0
    for s in range(nsub):
        imgs = np.stack([fr[s * frames + f][0] for f in range(frames)] + [fr[s * frames + f][1] for f in range(frames)])
        ext, g = fe.extract_all_batch(imgs, frames, cap=capb)
        o = got[s]
        np.testing.assert_array_equal(o["netvlad"], g)
        for i, (kps, sc, desc) in enumerate(ext):
            n = int(o["n_kp"][i]); assert n == len(kps) == capb
            np.testing.assert_array_equal(o["kps_xy"][i, :n], kps); np.testing.assert_array_equal(o["scores"][i, :n], sc); np.testing.assert_array_equal(o["desc"][i, :n], desc)
        for f in range(frames):
            q, t, d = fe.match_knn(ext[f][2], ext[frames + f][2], 0.8)
            n = int(o["lr_n"][f]); assert n == len(q) and n > 10
            np.testing.assert_array_equal(o["lr_q"][f, :n], q); np.testing.assert_array_equal(o["lr_t"][f, :n], t); np.testing.assert_array_equal(o["lr_dist"][f, :n], d)
            pv = ext[f - 1] if f > 0 else prev
            if pv is not None:
                q, t, d = fe.match_knn(ext[f][2], pv[2], 0.8)
                n = int(o["prev_n"][f]); assert n == len(q)
                np.testing.assert_array_equal(o["prev_q"][f, :n], q); np.testing.assert_array_equal(o["prev_t"][f, :n], t); np.testing.assert_array_equal(o["prev_dist"][f, :n], d)
                nprev += n
        prev = ext[frames - 1]
    assert nprev > 0
    pipe.close(); fe.close()


@pytest.mark.gpu
def test_pipe_submit_and_wait_from_two_threads():
    """The reference's image callback and its tracker are two threads: one submits, the other waits (a bounded queue of tickets between them).  Results equal
    the single calls whatever the interleaving."""
    import queue
    import threading
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.weights import synthetic_superpoint_weights
    lanes, nsub = 3, 40
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=api.PREC_F32_WINO, keypoint_threshold=0.005))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5)); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    fr = _frames(nsub)
    ref = [fe.extract_all_batch(np.stack(f), 1, cap=CAP) for f in fr]
    pipe = api.StereoPipe(fe, lanes=lanes, frames=1, width=W, height=H, cap=CAP, netvlad=True, coalesce=2, coalesce_depth=1)
    q = queue.Queue(maxsize=lanes * 2)                    # lanes x coalesce tickets between the two threads: always inside the ring of 2 x lanes result passes
    got, errors = {}, []

    def producer():
        try:
            for i in range(nsub):
                q.put((i, pipe.submit(fr[i][0][None], fr[i][1][None])))
        except Exception as e:      # noqa: BLE001
            errors.append(e)
        q.put(None)

    def consumer():
        try:
            while True:
                it = q.get()
                if it is None:
                    return
                i, t = it
                got[i] = {k: (None if v is None else v.copy()) for k, v in pipe.wait(t).items()}
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=producer), threading.Thread(target=consumer)]
    for t in th: t.start()
    for t in th: t.join(120)
    assert not errors, errors
    assert len(got) == nsub
    for i in range(nsub):
        ext, g = ref[i]
        np.testing.assert_array_equal(got[i]["netvlad"], g)
        for im in range(2):
            n = int(got[i]["n_kp"][im]); assert n == len(ext[im][0])
            np.testing.assert_array_equal(got[i]["kps_xy"][im, :n], ext[im][0]); np.testing.assert_array_equal(got[i]["desc"][im, :n], ext[im][2])
        kl, _, dl = ext[0]; kr, _, dr = ext[1]
        mq, mt, md = fe.match_knn(dl, dr, 0.8)
        n = int(got[i]["lr_n"][0]); assert n == len(mq)
        np.testing.assert_array_equal(got[i]["lr_q"][0, :n], mq); np.testing.assert_array_equal(got[i]["lr_t"][0, :n], mt)
    pipe.close(); fe.close()
