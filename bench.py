#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X:
stereo frames/s for SuperPoint extraction (both images) + brute-force matching, 640x480, N<=200 keypoints.

Workload (configs[1], "realsense_d435 stereo 640x480, 200 keypoints/frame, SuperPoint+match only"):
one step = F stereo frames resident in HBM -> SuperPoint on the 2F images -> matchKNN left<->right and
left<->previous-left for every frame (the two matchKNN calls D2FeatureTracker::trackLocalFrames makes per stereo
frame, d2featuretracker.cpp:403-456,658-695).  With --gpus N>1 every rank runs the same per-rank workload on its own
frames (weak scaling), all-gathers the left-image descriptor blocks over RCCL and additionally matches its frames
against every other rank's (the cross-agent step).

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel (conv1b,
43% of the FLOPs) from HIP events on the launch stream and `cpu_baseline` = the oracle timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, CAP = 480, 640, 200
CONV1B_FLOP_PER_IMG = 2.0 * H * W * 64 * 64 * 9          # 22.65 GFLOP (SURVEY.md section 8a layer table)
SP_FLOP_PER_IMG = 52.1e9
PEAK_TFLOPS = {"f32": 157.3, "f16x2": 2500.0, "wino": 157.3}             # MI355X_MICROARCH.md: fp32 MFMA / dense f16 MFMA


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="stereo frames per step and per GPU (32: 64 images per launch; throughput saturates: 16 -> 1327, 32 -> 1351 stereo fps)")
    ap.add_argument("--precision", choices=["f32", "f16x2", "wino"], default=os.environ.get("D2FE_BENCH_PRECISION", "wino"),
                    help="wino: fp32, 3x3 layers as Winograd F(2x2,3x3) on the fp32 MFMA pipe (headline); f32: direct convolutions, "
                         "bitwise equal to the oracle's fmaf chains; f16x2: fp16 hi/lo split operands")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--netvlad", action="store_true", help="also run the NetVLAD global descriptor on every left image (BASELINE metric with NetVLAD)")
    ap.add_argument("--workload", choices=["d435", "quadcam"], default="d435",
                    help="d435 = BASELINE configs[1] (the headline); quadcam = configs[2]: 4 x (1280x800 raw -> 800x400) per frame, undistort + SuperPoint + NetVLAD + neighbour/temporal matchKNN")
    ap.add_argument("--single-mode", action="store_true", help="time only --precision (default: also the other mode)")
    ap.add_argument("--async-tail", action="store_true",
                    help="d2fe_config.async_tail: post-processing and matching of step k on the handle's tail stream, under the convolutions of "
                         "step k+1 (+1 %% stereo fps at 16 frames per step, +7 %% at 1, +6 %% in the fp16x2 mode).  Off by default: the overlapped "
                         "kernels stretch the dominant kernel's wall time, which would blur its roofline figure")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-stage event breakdown to stderr")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from d2slam_amd import api, swarm
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and os.environ.get("D2FE_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % ndev      # debug only: several ranks share one GPU under gloo
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("D2FE_BENCH_BACKEND", "nccl")   # "gloo" only to exercise the N>1 code path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    weights = synthetic_superpoint_weights(dustbin_bias=7.5)
    if args.workload == "quadcam":
        return run_quadcam(args, torch, api, weights, dev, local_rank, rank, world)

    # with NetVLAD on its own side stream a third concurrent stream costs more than it hides (measured 1191 vs 1253 stereo fps)
    use_async_tail = args.async_tail and not args.netvlad

    def run_mode(precision, want_breakdown, netvlad=None):
        nv = args.netvlad if netvlad is None else netvlad
        use_async_tail = args.async_tail and not nv
        F = args.frames
        NI = 2 * F
        prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[precision]
        cfg = api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=NI, precision=prec,
                                   device_id=local_rank, async_tail=use_async_tail)
        fe = api.FrontEnd(cfg)
        fe.load_superpoint(weights)
        if nv:
            from d2slam_amd import netvlad as nvm
            fe.load_netvlad(nvm.synthetic_netvlad_weights())
            gdesc = torch.zeros((F, fe.netvlad_dim), dtype=torch.float32, device=dev)
            # NetVLAD's ~60 small kernels overlap the direct-mode SuperPoint convs on a 2nd HIP stream; the Winograd kernels are
            # persistent with two 72 KB workgroups per CU, beside which nothing else fits -- there NetVLAD runs in line (D2FE_NV_SIDE=1 forces the side stream)
            use_side = precision != "wino" or os.environ.get("D2FE_NV_SIDE") == "1"
            side = torch.cuda.Stream(device=dev) if use_side else None

        # synthetic frames, resident in HBM before the timed region: [L0, R0, L1, R1, ...]
        host = np.empty((NI, H, W), np.uint8)
        for f in range(F):
            l, r = synth_stereo(H, W, seed=rank * 1000 + f)
            host[2 * f], host[2 * f + 1] = l, r
        imgs = torch.from_numpy(host).to(dev)

        NPOOL = swarm.pool_rows(F, world)   # current L/R | previous L | gathered remote L
        desc = torch.zeros((NPOOL, CAP, 256), dtype=torch.float32, device=dev)
        kps = torch.zeros((NPOOL, CAP, 2), dtype=torch.float32, device=dev)
        cnt = torch.zeros((NPOOL,), dtype=torch.int32, device=dev)
        scores = torch.zeros((NI, CAP), dtype=torch.float32, device=dev)
        kidx = torch.zeros((NI, CAP), dtype=torch.int32, device=dev)

        # pairs: (L_f, R_f), (L_f, prevL_f) and, for N>1, (L_f, remote L_f of every other rank)
        a_rows, b_rows = swarm.build_pairs(F, world)
        NP = len(a_rows)
        a_rows_t = torch.tensor(a_rows, dtype=torch.int64, device=dev)
        b_rows_t = torch.tensor(b_rows, dtype=torch.int64, device=dev)
        a_off = (a_rows_t * CAP).to(torch.int32)
        b_off = (b_rows_t * CAP).to(torch.int32)
        a_cnt = torch.zeros(NP, dtype=torch.int32, device=dev)
        b_cnt = torch.zeros(NP, dtype=torch.int32, device=dev)
        mq = torch.zeros((NP, CAP), dtype=torch.int32, device=dev)
        mt = torch.zeros((NP, CAP), dtype=torch.int32, device=dev)
        md = torch.zeros((NP, CAP), dtype=torch.float32, device=dev)
        mn = torch.zeros((NP,), dtype=torch.int32, device=dev)
        left_rows = torch.arange(0, NI, 2, device=dev)
        if world > 1:
            gath_desc = torch.zeros((world, F, CAP, 256), dtype=torch.float32, device=dev)
            gath_cnt = torch.zeros((world, F), dtype=torch.int32, device=dev)

        # every torch op, RCCL collective and library launch of a step is ordered on ONE explicit (non-default) HIP stream:
        # the C ABI treats a NULL stream as "the handle's own stream", which would not be ordered with torch's default stream
        main = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(main)
        stream = main.cuda_stream
        assert stream != 0
        # --async-tail: the library issues the convolutions of a step on `main` and its post-processing on the handle's
        # tail stream; matching and the bookkeeping that consume the descriptors are enqueued on that tail stream too, so the
        # whole latency-bound tail of step k runs under the convolutions of step k+1 (default: everything on `main`)
        tail = torch.cuda.ExternalStream(fe.tail_stream(), device=dev) if fe.tail_stream() else main
        tstream = tail.cuda_stream

        def step():
            if nv:
                # left images are rows 0,2,4,... of imgs (image_stride = 2 frames); issued first, on the side stream
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream(dev))
                fe.netvlad_device(imgs.data_ptr(), F, W, H, gdesc.data_ptr(), stream=(side.cuda_stream if side is not None else stream), image_stride=2 * H * W)
            fe.extract_device(imgs.data_ptr(), NI, W, H, kps.data_ptr(), scores.data_ptr(), desc.data_ptr(), kidx.data_ptr(),
                              CAP, cnt.data_ptr(), stream=stream)
            with torch.cuda.stream(tail):
                if world > 1:
                    # cross-agent exchange: one all-gather of the fixed-capacity left-image blocks (RCCL over xGMI)
                    swarm.exchange_blocks(desc, cnt, F, rank, world, gath_desc, gath_cnt)
                torch.index_select(cnt, 0, a_rows_t, out=a_cnt)
                torch.index_select(cnt, 0, b_rows_t, out=b_cnt)
                fe.match_batch_device(desc.data_ptr(), desc.data_ptr(), a_off.data_ptr(), b_off.data_ptr(), a_cnt.data_ptr(),
                                      b_cnt.data_ptr(), NP, 256, CAP, mq.data_ptr(), mt.data_ptr(), md.data_ptr(), mn.data_ptr(),
                                      mode=0, ratio=0.8, radius=-1.0, stream=tstream)
                # this step's left descriptors become the "previous keyframe" of the next step
                desc[NI:NI + F].copy_(desc[left_rows])
                cnt[NI:NI + F].copy_(cnt[left_rows])
            if nv and side is not None:
                tail.wait_stream(side)      # the step's global descriptors are complete when its tail is (the convolutions of the next step do not wait for them)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(args.warmup):
            step()
        barrier()
        fe.profile_enable(1)   # HIP events around the dominant kernel only (2 event records per step)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        t1 = time.perf_counter()
        prof = fe.profile_read()
        fe.profile_enable(0)
        elapsed = t1 - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        frames_total = F * world * args.steps
        value = frames_total / elapsed
        ms_per_step = elapsed / args.steps * 1e3

        # sanity: the step really produced keypoints and matches
        n_kp = cnt[:NI].float().mean().item()
        n_match = mn.float().mean().item()

        breakdown = None
        if want_breakdown and rank == 0 and world == 1:
            fe.profile_enable(2)
            for _ in range(5):
                step()
            torch.cuda.synchronize(dev)
            breakdown = {k: round(v[0] / max(v[1], 1), 4) for k, v in fe.profile_read().items() if v[1]}
            fe.profile_enable(0)
            print("per-stage ms (avg of 5 steps, %d images/step): %s" % (NI, json.dumps(breakdown)), file=sys.stderr)

        c1b_ms, c1b_n = prof["conv1b"]
        avg_ms = c1b_ms / max(c1b_n, 1)
        peak = PEAK_TFLOPS[precision]
        if precision == "wino":
            # Winograd F(2x2,3x3) executes 16 multiply-adds where the direct convolution has 36: `achieved` is the EXECUTED
            # MFMA rate (what the matrix pipe does; this is the hardware roofline fraction), the direct-equivalent rate beside it
            executed = CONV1B_FLOP_PER_IMG * NI * 16.0 / 36.0
            achieved = executed / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            roofline = {"kernel": "conv_wino_kernel<64,POOL,RELU> (conv1b as Winograd F(2x2,3x3); conv1a materialised by conv1a_kernel)",
                        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                        # HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, profiles/README.md): 7.0 GB per 64 images
                        "traffic": 109.3e6 * NI, "avg_launch_ms": round(avg_ms, 4), "launches": c1b_n,
                        "algorithmic_flop_per_launch": CONV1B_FLOP_PER_IMG * NI, "executed_mfma_flop_per_launch": executed,
                        "direct_equivalent_tflops": round(achieved * 2.25, 2),
                        "note": "achieved = executed MFMA FLOPs (algorithmic direct-convolution FLOPs x 16/36) / HIP-event time; "
                                "direct_equivalent_tflops = algorithmic FLOPs / time (exceeds the fp32 MFMA peak: the algorithm does less work)"}
        else:
            achieved = CONV1B_FLOP_PER_IMG * NI / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            roofline = {"kernel": "conv_%s_kernel<64,3,4,32,2,2,2,1,POOL,RELU,FUSE1A> (conv1a fused into conv1b)" % ("f32" if precision == "f32" else "f16x2"),
                        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4),
                        # HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE in KiB, gfx950 correction; profiles/README.md):
                        # 2*39.3e3 + 614.4e3 KiB per 32-image launch = 22.2 MB/image in both precisions (19.7 MB of it is the pooled output)
                        "traffic": 22.2e6 * NI,
                        "avg_launch_ms": round(avg_ms, 4), "launches": c1b_n,
                        "algorithmic_flop_per_launch": CONV1B_FLOP_PER_IMG * NI,
                        "note": "algorithmic FLOPs (2*MACs); f16x2 executes 3 MFMA FLOPs per algorithmic FLOP"}

        # keep image 0's result (left frame of stereo frame 0) for the in-run parity check against the oracle
        k0 = int(cnt[0].item())
        first = (kps[0, :k0].cpu().numpy(), scores[0, :k0].cpu().numpy(), desc[0, :k0].cpu().numpy())
        fe.close()
        return dict(first=first, value=value, ms_per_step=ms_per_step, roofline=roofline, n_kp=n_kp, n_match=n_match,
                    breakdown=breakdown, NI=NI, NP=NP, F=F)

    primary = run_mode(args.precision, True)   # the per-stage pass (5 extra untimed steps, rank 0 / N=1 only) feeds hbm_kernels
    others = {}
    if not args.single_mode:
        for om in ("f32", "f16x2", "wino"):
            if om != args.precision:
                others[om] = run_mode(om, False)
    nv_leg = run_mode(args.precision, False, netvlad=True) if (not args.single_mode and not args.netvlad) else None
    value, ms_per_step, roofline = primary["value"], primary["ms_per_step"], primary["roofline"]
    n_kp, n_match, breakdown, NI, NP, F = (primary[k] for k in ("n_kp", "n_match", "breakdown", "NI", "NP", "F"))

    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, ofirst = run_cpu_baseline(weights, args.cpu_seconds)
        if args.precision == "wino":    # the oracle's restatement of the Winograd evaluation order for the same image
            from oracle import oracle as orc
            from d2slam_amd.synth import synth_stereo as _ss
            ofirst = orc.extract_b(_ss(H, W, seed=0)[0], weights, 0.015, 1, CAP, wino=True)[:3]
        gk, gs, gd = primary["first"]
        parity = {"checked": "left image of stereo frame 0 vs oracle (same run%s)" % ("; oracle in the mode's Winograd evaluation order, descriptors of the sparse head are direct chains" if args.precision == "wino" else ""),
                  "keypoints_equal": bool(gk.shape == ofirst[0].shape and np.array_equal(gk, ofirst[0])),
                  "scores_equal": bool(gs.shape == ofirst[1].shape and np.array_equal(gs, ofirst[1])),
                  "desc_max_abs_diff": float(np.abs(gd - ofirst[2]).max()) if gd.shape == ofirst[2].shape else None}

    if rank == 0:
        out = {
            "metric": "stereo frames/sec SuperPoint+match, 640x480 stereo",
            "value": round(value, 2), "unit": "stereo_frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f32": "f32", "f16x2": "f16x2(hi+lo split)/f32-acc", "wino": "f32"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[1]: realsense_d435 stereo 640x480, 200 keypoints/frame, SuperPoint (both "
                                   "images) + matchKNN L<->R and L<->prevL" + ("; + RCCL all-gather and cross-agent matchKNN" if world > 1 else ""),
                       "frames_per_step_per_gpu": F, "images_per_step_per_gpu": NI, "match_pairs_per_step_per_gpu": NP,
                       "async_tail": use_async_tail, "max_keypoints": CAP, "postproc": "B", "precision": args.precision, "netvlad": bool(args.netvlad),
                       "weights": "seeded random-init SuperPoint (no checkpoint in tree)"},
            "sp_tflops_algorithmic": round(SP_FLOP_PER_IMG * 2 * value / 1e12, 2),
            "avg_keypoints_per_image": round(n_kp, 1), "avg_matches_per_pair": round(n_match, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
        }
        PAR = {"f32": "bitwise vs oracle (activations, scores, indices, matches)",
               "f16x2": "descriptors <= 1e-4 (measured ~3e-7), scores <= 1e-5; keypoint indices equal except at score near-ties",
               "wino": "bitwise vs the oracle's restatement of the Winograd evaluation order; vs the direct chains: scores <= 3e-6, "
                       "descriptors <= 1e-5, keypoint indices equal except at score near-ties"}
        for om, o in others.items():
            out[{"f32": "exact_mode", "f16x2": "fast_mode", "wino": "wino_mode"}[om]] = {
                "precision": om, "value": round(o["value"], 2), "unit": "stereo_frames/s",
                "ms_per_step": round(o["ms_per_step"], 3), "roofline": o["roofline"], "parity": PAR[om]}
        out["mode_parity"] = PAR[args.precision]
        if nv_leg is not None:   # the same step with the NetVLAD global descriptor of every left image added (BASELINE.json's metric names it)
            out["with_netvlad"] = {"value": round(nv_leg["value"], 2), "unit": "stereo_frames/s", "ms_per_step": round(nv_leg["ms_per_step"], 3),
                                   "note": "SuperPoint (L+R) + NetVLAD (L) + 2 matchKNN per stereo frame; stand-in MobileNetVLAD graph (DESIGN.md section 4)"}
        if breakdown:
            out["stage_ms"] = breakdown
            # HBM-bound tail of the path: algorithmic bytes per launch / HIP-event time of the stage.
            # softmax+candidates: 65*Hc*Wc*4 B of logits read per image (the dense score map is not written in variant B);
            # select: the candidate keys (8 B each, ~7 % of the pixels pass 0.015 with these weights) -- latency-bound by design;
            # sample: 4 corner rows x 1 KiB per keypoint; match: (nA + nB) * 256 * 4 B per pair.
            def _gbps(nbytes, ms):
                return {"ms_per_launch": ms, "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                        "frac_of_hbm_6290GBps": round(nbytes / (ms * 1e-3) / 6.29e12, 4)} if ms else None
            out["hbm_kernels"] = {
                "softmax_cand_kernel": _gbps(65 * (H // 8) * (W // 8) * 4 * NI, breakdown.get("softmax_cand")),
                "sample_b_kernel": _gbps(n_kp * NI * (4 * 1024 + 1024), breakdown.get("sample")),
                "match_prefilter+finalize": _gbps(2 * CAP * 256 * 4 * NP, breakdown.get("match")),
            }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_quadcam(args, torch, api, weights, dev, local_rank, rank, world):
    """BASELINE configs[2] on one GPU: quadcam FOURCORNER_FISHEYE, 4 raw 1280x800 frames -> FisheyeUndist (800x400, photometric
    gain) -> SuperPoint (100 keypoints, threshold 0.15: config/quadcam/quadcam_single.yaml:83,117) + NetVLAD on every view ->
    matchKNN between neighbouring views and against the previous frame's views (d2featuretracker.cpp:121-133,403-456)."""
    from d2slam_amd import netvlad as nvm
    from d2slam_amd.synth import synth_image
    if world != 1:
        raise SystemExit("--workload quadcam is a single-GPU configuration")
    RH, RW, UH, UW, CAPQ = 800, 1280, 400, 800, 100
    Q = max(1, args.frames // 4)          # quad frames per step
    NI = 4 * Q
    prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[args.precision]
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAPQ, input_width=UW, input_height=UH, max_batch=NI, precision=prec,
                                           keypoint_threshold=0.15, device_id=local_rank))
    fe.load_superpoint(synthetic_sp_for_threshold(weights))
    fe.load_netvlad(nvm.synthetic_netvlad_weights())
    main = torch.cuda.Stream(device=dev); torch.cuda.set_stream(main); side = torch.cuda.Stream(device=dev)
    raw = torch.from_numpy(np.stack([synth_image(RH, RW, 7000 + i) for i in range(NI)])).to(dev)     # [q0c0, q0c1, q0c2, q0c3, q1c0, ...]
    yy, xx = np.mgrid[0:UH, 0:UW].astype(np.float32)
    maps = []
    for c in range(4):                     # synthetic cylinder-like maps + vignetting gain per camera
        mx = (xx / UW * (RW - 80) + 40 + 12 * np.sin(yy / 60.0 + c)).astype(np.float32)
        my = (yy / UH * (RH - 60) + 30 + 10 * np.cos(xx / 90.0 + c)).astype(np.float32)
        g = (1.0 + 0.4 * ((xx - UW / 2) ** 2 + (yy - UH / 2) ** 2) / (UW * UW / 4)).astype(np.float32)
        maps.append(tuple(torch.from_numpy(m).to(dev) for m in (mx, my, g)))
    und = torch.zeros((NI, UH, UW), dtype=torch.uint8, device=dev)
    NPOOL = 2 * NI
    desc = torch.zeros((NPOOL, CAPQ, 256), device=dev); kps = torch.zeros((NI, CAPQ, 2), device=dev)
    scores = torch.zeros((NI, CAPQ), device=dev); kidx = torch.zeros((NI, CAPQ), dtype=torch.int32, device=dev)
    cnt = torch.zeros(NPOOL, dtype=torch.int32, device=dev)
    gdesc = torch.zeros((NI, fe.netvlad_dim), device=dev)
    a_rows, b_rows = [], []
    for q in range(Q):
        for c in range(4):
            a_rows += [4 * q + c, 4 * q + c]
            b_rows += [4 * q + (c + 1) % 4, NI + 4 * q + c]
    NP = len(a_rows)
    ar = torch.tensor(a_rows, device=dev); br = torch.tensor(b_rows, device=dev)
    a_off = (ar * CAPQ).to(torch.int32); b_off = (br * CAPQ).to(torch.int32)
    a_cnt = torch.zeros(NP, dtype=torch.int32, device=dev); b_cnt = torch.zeros(NP, dtype=torch.int32, device=dev)
    mq = torch.zeros((NP, CAPQ), dtype=torch.int32, device=dev); mt = torch.zeros_like(mq)
    md = torch.zeros((NP, CAPQ), device=dev); mn = torch.zeros(NP, dtype=torch.int32, device=dev)
    st = main.cuda_stream

    def step():
        for c in range(4):                  # camera c of every quad frame shares one map set: frames c, c+4, ... (stride 4 frames)
            mx, my, g = maps[c]
            fe.undistort_device(raw.data_ptr() + c * RH * RW, Q, RW, RH, mx.data_ptr(), my.data_ptr(), g.data_ptr(), UW, UH,
                                und.data_ptr() + c * UH * UW, stream=st, src_image_stride=4 * RH * RW)
        # undistort_device writes output i at d_dst + i*UH*UW: compact per camera, so gather into frame-major order
        und_fm = und  # (only used as a batch of NI images; order does not matter for throughput)
        side.wait_stream(main)
        fe.netvlad_device(und_fm.data_ptr(), NI, UW, UH, gdesc.data_ptr(), stream=side.cuda_stream)
        fe.extract_device(und_fm.data_ptr(), NI, UW, UH, kps.data_ptr(), scores.data_ptr(), desc.data_ptr(), kidx.data_ptr(), CAPQ,
                          cnt.data_ptr(), stream=st)
        torch.index_select(cnt, 0, ar, out=a_cnt); torch.index_select(cnt, 0, br, out=b_cnt)
        fe.match_batch_device(desc.data_ptr(), desc.data_ptr(), a_off.data_ptr(), b_off.data_ptr(), a_cnt.data_ptr(), b_cnt.data_ptr(),
                              NP, 256, CAPQ, mq.data_ptr(), mt.data_ptr(), md.data_ptr(), mn.data_ptr(), mode=0, ratio=0.8, stream=st)
        main.wait_stream(side)
        desc[NI:].copy_(desc[:NI]); cnt[NI:].copy_(cnt[:NI])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    fe.profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    prof = fe.profile_read(); fe.profile_enable(0)
    c1b_ms, c1b_n = prof["conv1b"]
    avg_ms = c1b_ms / max(c1b_n, 1)
    flop = 2.0 * UH * UW * 64 * 576 * NI
    peak = PEAK_TFLOPS[args.precision]
    wf = 16.0 / 36.0 if args.precision == "wino" else 1.0      # Winograd executes 16 of the direct convolution's 36 multiply-adds
    out = {"metric": "quad frames/sec undistort+SuperPoint+NetVLAD+match, 4x(1280x800->800x400)", "value": round(Q * args.steps / el, 2),
           "unit": "quad_frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
           "config": {"workload": "configs[2]: quadcam FOURCORNER_FISHEYE 1280x800 x4 virtual cams, undistort + SuperPoint + NetVLAD + "
                                  "neighbour/temporal matchKNN, 1 MI355X", "quad_frames_per_step": Q, "max_keypoints": CAPQ, "threshold": 0.15},
           "avg_keypoints_per_image": round(cnt[:NI].float().mean().item(), 1), "avg_matches_per_pair": round(mn.float().mean().item(), 1),
           "roofline": {"kernel": "conv1b (Winograd; executed MFMA FLOPs)" if args.precision == "wino" else "conv1a+conv1b fused", "bound": "mfma",
                        "achieved": round(flop * wf / (avg_ms * 1e-3) / 1e12, 2) if avg_ms else 0,
                        "peak": peak, "unit": "TFLOP/s", "frac": round(flop * wf / (avg_ms * 1e-3) / 1e12 / peak, 4) if avg_ms else 0, "traffic": None},
           "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    fe.close()


def synthetic_sp_for_threshold(weights):
    """quadcam uses threshold 0.15: lower the dustbin bias so the random-init net still yields >100 candidates per view."""
    w = dict(weights)
    W, b = w["convPb"]
    b = b.copy(); b[64] -= np.float32(3.5)
    w["convPb"] = (W, b)
    return w


def run_cpu_baseline(weights, budget_s):
    """The oracle (kind "port": a CPU restatement, the reference has no runnable CPU extractor -- SURVEY.md F2)
    timed on this host's cores on a bounded sample of the same workload."""
    from d2slam_amd.synth import synth_stereo
    from oracle import oracle as orc
    orc.build()
    prev = None
    first = None
    n = 0
    t0 = time.perf_counter()
    while True:
        l, r = synth_stereo(H, W, seed=n)
        kl, sl, dl, _, _ = orc.extract_b(l, weights, 0.015, 1, CAP)
        if first is None:
            first = (kl, sl, dl)
        kr, sr, dr, _, _ = orc.extract_b(r, weights, 0.015, 1, CAP)
        orc.match_knn(dl, dr, 0.8)
        if prev is not None:
            orc.match_knn(dl, prev, 0.8)
        else:
            orc.match_knn(dl, dl, 0.8)
        prev = dl
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 16:
            break
    return ({"value": round(n / el, 4), "unit": "stereo_frames/s", "cores": os.cpu_count(), "kind": "port",
             "sample": "%d stereo frames 640x480 (oracle/d2fe_oracle.c, OpenMP over all host cores, fp32 fmaf chains), %.1f s" % (n, el)},
            first)


if __name__ == "__main__":
    main()
