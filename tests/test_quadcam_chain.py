"""BASELINE configs[2], the quadcam neighbour chain (A12 + A10 + the remap), device vs oracle vs the reference's own C++:
getFeatureHalfImg on both views, the a-side x shift by +-move_cols, matchKNN with the radius gate, index remap
(D2FeatureTracker::matchLocalFeatures, d2frontend/src/d2featuretracker.cpp:1144-1182), for the four neighbour pairs of a quad frame
(:121-133), plus the temporal pairs."""
import numpy as np
import pytest

from d2slam_amd.synth import synth_image
from oracle import ref as spref
from tests.test_ref_pin import orc_neighbour_chain


@pytest.mark.gpu
def test_half_compact_and_remap_kernels(orc):
    import torch
    from d2slam_amd import api
    dev = torch.device("cuda", 0)
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=100, input_width=64, input_height=64, max_batch=1))
    cap, rows, W_u, fov = 100, 3, 800, 200.0
    rng = np.random.RandomState(2)
    pts = np.stack([rng.randint(0, W_u, (rows, cap)), rng.randint(0, 400, (rows, cap))], -1).astype(np.float32)
    mc = fe.half_move_cols(W_u, fov)
    pts[0, :4, 0] = [mc, np.floor(mc), W_u - mc, np.ceil(W_u - mc)]
    desc = rng.randn(rows, cap, 256).astype(np.float32); n = np.array([100, 37, 0], np.int32)
    jobs = [(0, 1, mc), (0, 0, -mc), (1, 1, 0.0), (1, 0, 0.0), (2, 1, mc)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    od = torch.zeros((len(jobs), cap, 256), device=dev); op = torch.zeros((len(jobs), cap, 2), device=dev)
    om = torch.full((len(jobs), cap), -1, dtype=torch.int32, device=dev); on = torch.zeros(len(jobs), dtype=torch.int32, device=dev)
    d_desc, d_pts, d_n = t(desc), t(pts), t(n)               # keep the device tensors alive across the launch
    d_jr, d_jl, d_js = t(np.array([j[0] for j in jobs], np.int32)), t(np.array([j[1] for j in jobs], np.int32)), t(np.array([j[2] for j in jobs], np.float32))
    fe.half_image_compact_device(d_desc.data_ptr(), d_pts.data_ptr(), d_n.data_ptr(), d_jr.data_ptr(), d_jl.data_ptr(), d_js.data_ptr(),
                                 len(jobs), cap, 256, W_u, fov, od.data_ptr(), op.data_ptr(), om.data_ptr(), on.data_ptr())
    fe.sync(); torch.cuda.synchronize()
    for j, (row, left, shift) in enumerate(jobs):
        ref_map = orc.half_img(pts[row, :n[row]], bool(left), W_u, fov)
        if spref.available():
            assert np.array_equal(ref_map, spref.half_image(pts[row, :n[row]], bool(left), W_u, fov))
        k = int(on[j].item())
        assert k == len(ref_map) and np.array_equal(om[j, :k].cpu().numpy(), ref_map)
        assert np.array_equal(od[j, :k].cpu().numpy(), desc[row][ref_map])
        exp = pts[row][ref_map].copy(); exp[:, 0] = exp[:, 0] + np.float32(shift)
        assert np.array_equal(op[j, :k].cpu().numpy(), exp)
    fe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "wino"])
def test_quadcam_chain_vs_oracle_and_reference(orc, sp_weights, prec):
    """undistort -> SuperPoint -> neighbour chain -> matches, end to end on one quad frame x 2 steps, against the oracle chain run on the
    device's own keypoints/descriptors (which other tests hold to the oracle) and, when available, the reference's own branch."""
    import torch
    from d2slam_amd import api, netvlad as nvm, quadcam
    dev = torch.device("cuda", 0)
    RH, RW, UH, UW, CAPQ, Q = 400, 640, 200, 400, 100, 2
    w = dict(sp_weights); Wt, b = w["convPb"]; b = b.copy(); b[64] -= np.float32(3.5); w["convPb"] = (Wt, b)
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAPQ, input_width=UW, input_height=UH, max_batch=4 * Q, keypoint_threshold=0.15,
                                           precision=api.PREC_F32 if prec == "f32" else api.PREC_F32_WINO))
    fe.load_superpoint(w); fe.load_netvlad(nvm.synthetic_netvlad_weights())
    st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
    # neighbouring views see the same scene shifted by ~move_cols: crops of one wide panorama per quad frame
    # (view c+1 shows view c's content move_cols further right: its crop starts 2*move_cols raw pixels further left; the ring cannot be
    # closed on a flat panorama, so the (0,3) RIGHT_LEFT pair has no true correspondences -- it is compared all the same)
    mc = int(round(fe.half_move_cols(UW, 200.0)))
    step_raw = mc * (RW // UW)
    raws = []
    for c in range(4):
        for q in range(Q):
            pano = synth_image(RH, RW + 3 * step_raw, 900 + q)
            raws.append(np.ascontiguousarray(pano[:, (3 - c) * step_raw:(3 - c) * step_raw + RW]))
    raw = torch.from_numpy(np.stack(raws)).to(dev)
    maps_h = []
    for c in range(4):       # identity-like maps (a pure 2x downscale) so that the panorama shift survives the undistortion
        yy, xx = np.mgrid[0:UH, 0:UW].astype(np.float32)
        maps_h.append((xx * (RW / UW), yy * (RH / UH), np.ones((UH, UW), np.float32)))
    maps = [tuple(torch.from_numpy(np.ascontiguousarray(m)).to(dev) for m in mm) for mm in maps_h]
    chain = quadcam.QuadcamChain(fe, torch, dev, Q, UH, UW, CAPQ, undistort_fov=200.0, knn_ratio=0.8, search_local_max_dist=0.2)
    for it in range(2):
        chain.step(raw, RH, RW, maps, st.cuda_stream)
    torch.cuda.synchronize()
    NI = 4 * Q
    cnt = chain.cnt.cpu().numpy(); pts = chain.pts.cpu().numpy(); desc = chain.desc.cpu().numpy()
    mq, mt, md, mn = (x.cpu().numpy() for x in (chain.mq, chain.mt, chain.md, chain.mn))
    # the undistorted views and the extraction, against the oracle
    und = chain.und.cpu().numpy()
    for v in (0, NI - 1):
        c, q = divmod(v, Q)
        assert np.array_equal(und[v], orc.undistort(raws[v], *maps_h[c]))
        rk, rs, rd, _, _ = orc.extract_b(und[v], w, 0.15, 1, CAPQ, wino=(prec == "wino"))
        assert np.array_equal(pts[v, :cnt[v]], rk) and np.abs(desc[v, :cnt[v]] - rd).max() <= 1e-6
    p = 0
    total = 0
    for q in range(Q):
        for (ca, cb, typ) in quadcam.NEIGHBOURS:
            va, vb = ca * Q + q, cb * Q + q
            pa, da, pb, db = pts[va, :cnt[va]], desc[va, :cnt[va]], pts[vb, :cnt[vb]], desc[vb, :cnt[vb]]
            exp = orc_neighbour_chain(orc, pa, da, pb, db, typ, 0.8, True, 0.2 * UW, UW, 200.0)
            if spref.available():
                r2 = spref.match_neighbour(pa, da, pb, db, typ, 0.8, True, 0.2 * UW, UW, 200.0)
                assert (exp is None) == (r2 is None)
                if exp is not None:
                    assert all(np.array_equal(x, y) for x, y in zip(exp, r2))
            n = int(mn[p])
            if exp is None:
                assert n == 0
            else:
                assert n == len(exp[0]), (q, ca, cb, n, len(exp[0]))
                assert np.array_equal(mq[p, :n], exp[0]) and np.array_equal(mt[p, :n], exp[1]) and np.array_equal(md[p, :n], exp[2])
                total += n
            p += 1
    assert total >= 10, "the panorama crops must produce neighbour matches (%d)" % total
    # temporal pairs: the second step matched every view against itself (identical frames): every keypoint with a distinct descriptor matches
    for v in range(NI):
        n = int(mn[p + v]); k = int(cnt[v])
        rq, rt, rd = orc.match_knn(desc[v, :k], desc[NI + v, :cnt[NI + v]], 0.8)
        assert n == len(rq) and np.array_equal(mq[p + v, :n], rq) and np.array_equal(mt[p + v, :n], rt)
    fe.close()
