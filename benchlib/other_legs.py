"""Secondary legs: configs[2] quadcam on one GPU and the single-call latencies through the host-pointer C ABI."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403

def run_quadcam(args, torch, api, weights, dev, local_rank, world, rank=0):
    """BASELINE configs[2] on one GPU: quadcam FOURCORNER_FISHEYE, 4 raw 1280x800 frames -> FisheyeUndist (800x400, photometric
    gain) -> SuperPoint (100 keypoints, threshold 0.15: config/quadcam/quadcam_single.yaml:83,117) + NetVLAD on every view ->
    neighbour matching as D2FeatureTracker::matchLocalFeatures does it for quadcam (d2featuretracker.cpp:1144-1182: half-image filter on
    both views, a-side x shifted by +-move_cols, matchKNN with the search radius, index remap) + temporal matchKNN per view."""
    import torch.distributed as dist
    from d2slam_amd import netvlad as nvm, quadcam, swarm
    from d2slam_amd.synth import synth_image
    RH, RW, UH, UW, CAPQ = 800, 1280, 400, 800, 100
    Q = max(1, args.frames // 4)          # quad frames per step
    NI = 4 * Q
    prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[args.precision]
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAPQ, input_width=UW, input_height=UH, max_batch=NI, precision=prec,
                                           keypoint_threshold=0.15, device_id=local_rank))
    fe.load_superpoint(synthetic_sp_for_threshold(weights))
    fe.load_netvlad(nvm.synthetic_netvlad_weights())
    main = torch.cuda.Stream(device=dev); torch.cuda.set_stream(main)
    st = main.cuda_stream
    # raw frames camera-major: [c0: q0..q(Q-1) | c1: ... ] so that one undistort launch per camera writes a contiguous slab
    # every agent flies through the same scenes (seed 7000 + i) with its own sensor noise, so that cross-agent matches exist
    def frame(i):
        im = synth_image(RH, RW, 7000 + i)
        if world > 1:
            rng = np.random.RandomState(977 * rank + i)
            im = np.clip(im.astype(np.int16) + rng.randint(-2, 3, im.shape), 0, 255).astype(np.uint8)
        return im
    raw = torch.from_numpy(np.stack([frame(i) for i in range(NI)])).to(dev)
    maps = [tuple(torch.from_numpy(m).to(dev) for m in quadcam.synthetic_maps(c, RH, RW, UH, UW)) for c in range(4)]
    chain = quadcam.QuadcamChain(fe, torch, dev, Q, UH, UW, CAPQ, undistort_fov=200.0, knn_ratio=0.8, search_local_max_dist=0.2)
    qs = swarm.QuadSwarm(chain, torch, dev, world, rank, fe.netvlad_dim, NETVLAD_GATE, mode=os.environ.get("D2FE_QUAD_SWARM_MODE", "all2all"),
                         exchange=args.exchange) if world > 1 else None

    side = torch.cuda.Stream(device=dev) if world > 1 else None

    def step():
        chain.step(raw, RH, RW, maps, st)
        if qs:
            # configs[4]: one block per view, ONE all-gather, the quadcam NetVLAD gate, view x view cross-agent matchKNN -- on a stream of its own behind a
            # snapshot of the step's outputs, beside the next step's convolutions (the main stream waits for the 1.7 MB snapshot, never for the collective)
            qs.step_overlapped(main, side)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    fe.profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    prof = fe.profile_read(); fe.profile_enable(0)
    c1b_ms, c1b_n = prof["conv1b"]
    avg_ms = c1b_ms / max(c1b_n, 1)
    flop = 2.0 * UH * UW * 64 * 576 * NI
    peak = PEAK_TFLOPS[args.precision]
    items = NI * (UH // 8) * (UW // 16)
    executed = items * 1084 * 4096.0 if args.precision == "wino" else flop * (3.0 if args.precision == "f16x2" else 1.0)
    ach = executed / (avg_ms * 1e-3) / 1e12 if avg_ms else 0.0
    out = {"metric": "quad frames/sec undistort+SuperPoint+NetVLAD+match, 4x(1280x800->800x400)", "value": round(Q * world * args.steps / el, 2),
           "unit": "quad_frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision != "f16x2" else "f16x2(hi+lo split)/f32-acc", "data": "synthetic",
           "config": {"workload": ("configs[2]: quadcam FOURCORNER_FISHEYE 1280x800 x4 virtual cams, undistort + SuperPoint + NetVLAD + "
                                   "neighbour matching (half-image filter, +-move_cols shift, radius gate, index remap) + temporal matchKNN, 1 MI355X") if world == 1 else
                                  ("configs[4]: %d-agent quadcam swarm, one agent per GPU: the configs[2] chain per agent + one exchange block per view "
                                   "(4 per quad frame), ONE all-gather, the quadcam NetVLAD gate (getMatchedPrevKeyframe, FOURCORNER_FISHEYE branch) on the "
                                   "device and view x view cross-agent matchKNN against every remote agent (%s)" % (world, qs.mode)),
                      "quad_frames_per_step_per_gpu": Q, "max_keypoints": CAPQ, "threshold": 0.15, "undistort_fov": 200.0, "search_radius_px": 0.2 * UW,
                      "precision": args.precision},
           "avg_keypoints_per_image": round(chain.cnt[:NI].float().mean().item(), 1),
           "avg_matches_per_pair": round(chain.mn.float().mean().item(), 1),
           "roofline": {"kernel": "conv1b (executed MFMA FLOPs)", "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "frac_executed": round(ach / peak, 4), "frac_algorithmic": round(flop / (avg_ms * 1e-3) / 1e12 / peak, 4) if avg_ms else 0,
                        "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches": c1b_n,
                        "measured": "HIP events on the one stream the quadcam chain runs on (undistort, NetVLAD, SuperPoint, matching in stream order): the launch has the device to itself",
                        "traffic_note": "not collected for this leg (the d435 headline's conv1b launch is the same kernel: `roofline.traffic`)"},
           "cpu_baseline": None}
    if qs:
        dp = qs.dir_prev.cpu().numpy()
        out["cross_agent"] = {"jobs_per_step_per_gpu": qs.njobs, "view_pairs_per_step_per_gpu": qs.NP, "mode": qs.mode,
                              "avg_matches_per_view_pair": round(qs.mn.float().mean().item(), 2),
                              "wire_precision": qs.exchange, "block_bytes": qs.block_bytes, "all_gather_bytes_received_per_step": qs.block_bytes * NI * (world - 1),
                              "stream": "its own, behind a snapshot of the step's outputs (QuadSwarm.step_overlapped): the main stream never waits for the collective"}
        out["netvlad_gate"] = {"jobs": qs.njobs, "passing_netvlad_gate": int(qs.n_pass.item()), "threshold": NETVLAD_GATE,
                               "rotation_histogram_dir_prev": {str(k): int((dp == k).sum()) for k in (-1, 0, 1, 2, 3)},
                               "rule": "remote view 2 vs local views 2,3,0,1 in order, first similarity >= threshold (d2featuretracker.cpp:212-233)"}
    fe.close()
    return out


def run_latency(api, weights, nv_weights, device_id, precision, calls):
    """Single-call latency through the boundary AS THE REFERENCE CALLS IT: one image per SuperPoint::infer / MobileNetVLADONNX::inference
    call (loop_cam.cpp:609-616), one matchKNN per pair, host pointers in and out (the H2D / D2H copies and the synchronisation are inside).
    Raw ctypes calls into the C ABI with preallocated buffers; p50 / p99 over `calls` calls after 20 warm-up calls."""
    import ctypes as C
    from d2slam_amd.synth import synth_descriptor_pair, synth_image, synth_stereo
    prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[precision]
    lib = api.load_library()
    P = lambda a: a.ctypes.data_as(C.c_void_p)

    def stats(fn):
        for _ in range(20):
            fn()
        t = np.empty(calls)
        for i in range(calls):
            t0 = time.perf_counter(); fn(); t[i] = time.perf_counter() - t0
        t *= 1e3
        return {"p50_ms": round(float(np.percentile(t, 50)), 4), "p99_ms": round(float(np.percentile(t, 99)), 4), "mean_ms": round(float(t.mean()), 4)}

    out = {"calls": calls, "precision": precision, "api": "host-pointer C ABI (sync H2D + kernels + D2H per call), python ctypes with preallocated buffers",
           "reference_call_sites": "loop_cam.cpp:609-616 (infer / inference, one image per call), d2featuretracker.cpp:1134-1138 (matchKNN)"}
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=2, precision=prec, device_id=device_id))
    fe.load_superpoint(weights)
    if nv_weights is not None:
        fe.load_netvlad(nv_weights)
    h = fe._h
    l, r = synth_stereo(H, W, seed=3)
    pair = np.ascontiguousarray(np.stack([l, r]))
    kps = np.zeros((2, CAP, 2), np.float32); sc = np.zeros((2, CAP), np.float32); desc = np.zeros((2, CAP, 256), np.float32)
    cnt = np.zeros(2, np.int32)
    one = lambda: lib.d2fe_superpoint_extract(h, P(pair), W, H, W, P(kps), P(sc), P(desc), CAP, P(cnt))
    two = lambda: lib.d2fe_superpoint_extract_batch(h, P(pair), 2, W, H, W, H * W, P(kps), P(sc), P(desc), CAP, P(cnt))
    assert one() == 0 and two() == 0
    out["d2fe_superpoint_extract_1_image"] = stats(one)
    out["d2fe_superpoint_extract_batch_2_images"] = stats(two)
    if nv_weights is not None:
        g = np.zeros(fe.netvlad_dim, np.float32)
        nvc = lambda: lib.d2fe_netvlad(h, P(l), W, H, W, P(g))
        assert nvc() == 0
        out["d2fe_netvlad_1_image"] = stats(nvc)
    two()
    na, nb = int(cnt[0]), int(cnt[1])
    da, db = desc[0, :na].copy(), desc[1, :nb].copy()
    q = np.zeros(CAP, np.int32); t = np.zeros(CAP, np.int32); d = np.zeros(CAP, np.float32); nm = C.c_int(0)
    mk = lambda: lib.d2fe_match_knn(h, P(da), na, P(db), nb, 256, C.c_double(0.8), None, None, C.c_double(-1.0), P(q), P(t), P(d), CAP, C.byref(nm))
    assert mk() == 0
    out["d2fe_match_knn_%dx%dx256" % (na, nb)] = stats(mk)

    def stereo():            # what trackLocalFrames costs per stereo frame without NetVLAD: 2 images + L<->R + L<->prevL
        two(); mk(); mk()
    out["stereo_frame_2_images_2_matches_host_to_host"] = stats(stereo)
    if nv_weights is not None:
        def stereo_nv():
            two(); nvc(); mk(); mk()
        out["stereo_frame_with_netvlad_host_to_host"] = stats(stereo_nv)
        # the fused entry point: ONE upload, SuperPoint (L+R) and NetVLAD (L) side by side on two streams (loop_cam.cpp:609-616 makes the two calls
        # back to back for the same image)
        g1 = np.zeros((1, fe.netvlad_dim), np.float32)
        all1 = lambda: lib.d2fe_extract_all(h, P(pair), W, H, W, P(kps), P(sc), P(desc), CAP, P(cnt), P(g1))
        all2 = lambda: lib.d2fe_extract_all_batch(h, P(pair), 2, W, H, W, H * W, P(kps), P(sc), P(desc), CAP, P(cnt), 1, P(g1))
        assert all1() == 0 and all2() == 0
        out["d2fe_extract_all_1_image_superpoint_and_netvlad"] = stats(all1)
        out["d2fe_extract_all_batch_stereo_pair_netvlad_left"] = stats(all2)

        def stereo_all():
            all2(); mk(); mk()
        out["stereo_frame_with_netvlad_fused_host_to_host"] = stats(stereo_all)
        # the same frame through the pipe with ONE frame in flight (submit + wait): one upload, both networks, ONE matcher launch over both pairs with the
        # previous left frame's descriptors still on the device, ONE download -- the whole per-frame work of processStereoframe as a single round trip
        pipe = api.StereoPipe(fe, lanes=1, frames=1, width=W, height=H, cap=CAP, netvlad=True, ratio=0.8)
        l1, r1 = np.ascontiguousarray(l[None]), np.ascontiguousarray(r[None])
        def through_pipe():
            pipe.wait_raw(pipe.submit_ptr(l1.ctypes.data, r1.ctypes.data))
        through_pipe()
        out["stereo_frame_with_netvlad_through_the_pipe_one_in_flight"] = stats(through_pipe)
        pipe.close()
    fe.close()
    # one quadcam frame (configs[2] geometry): 4 undistorted 800x400 views through extract_batch + netvlad_batch
    UH, UW, CAPQ = 400, 800, 100
    fq = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAPQ, input_width=UW, input_height=UH, max_batch=4, precision=prec, keypoint_threshold=0.15,
                                           device_id=device_id))
    fq.load_superpoint(synthetic_sp_for_threshold(weights))
    views = np.ascontiguousarray(np.stack([synth_image(UH, UW, 7100 + i) for i in range(4)]))
    k4 = np.zeros((4, CAPQ, 2), np.float32); s4 = np.zeros((4, CAPQ), np.float32); d4 = np.zeros((4, CAPQ, 256), np.float32); c4 = np.zeros(4, np.int32)
    quad = lambda: lib.d2fe_superpoint_extract_batch(fq._h, P(views), 4, UW, UH, UW, UH * UW, P(k4), P(s4), P(d4), CAPQ, P(c4))
    assert quad() == 0
    out["quadcam_frame_4_views_800x400_extract_batch"] = stats(quad)
    if nv_weights is not None:
        fq.load_netvlad(nv_weights)
        g4 = np.zeros((4, fq.netvlad_dim), np.float32)
        qnv = lambda: lib.d2fe_netvlad_batch(fq._h, P(views), 4, UW, UH, UW, UH * UW, P(g4))
        assert qnv() == 0
        out["quadcam_frame_4_views_netvlad_batch"] = stats(qnv)
        qall = lambda: lib.d2fe_extract_all_batch(fq._h, P(views), 4, UW, UH, UW, UH * UW, P(k4), P(s4), P(d4), CAPQ, P(c4), 4, P(g4))
        assert qall() == 0
        out["quadcam_frame_4_views_extract_all_batch"] = stats(qall)
    fq.close()
    return out


def synthetic_sp_for_threshold(weights):
    """quadcam uses threshold 0.15: lower the dustbin bias so the random-init net still yields >100 candidates per view."""
    w = dict(weights)
    Wt, b = w["convPb"]
    b = b.copy(); b[64] -= np.float32(3.5)
    w["convPb"] = (Wt, b)
    return w
