#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: stereo frames/s for SuperPoint + NetVLAD + match at 640x480.

Workload of `value` (the metric's configuration), EVERY --gpus N: one step = F stereo frames (u8, 640x480) through the frames-in-flight pipe of the C ABI
(include/d2fe.h, d2fe_pipe_submit / d2fe_pipe_wait) with 4 submits in flight -- pinned HOST frames in, HOST results out, inside the timed region:
  H2D of the 2F frames,
  SuperPoint on the 2F images (200 keypoints, variant-B post-processing: the live TensorRT path of the reference),
  NetVLAD global descriptor of the F left images (loop_cam.cpp:446-451: left image SP+NetVLAD, right image SP only),
  matchKNN left<->right and left<->previous-left for every frame (the two calls D2FeatureTracker::trackLocalFrames makes per
  stereo frame, d2featuretracker.cpp:403-456,658-695) as ONE matcher launch,
  ONE D2H of keypoints / scores / descriptors / counts / NetVLAD descriptors / match lists.
`configs1` beside it is BASELINE configs[1] (the same without NetVLAD).  With --gpus N>1 every rank runs that SAME per-rank path on its own frames (weak
scaling) and, per submit, on a stream of its own behind d2fe_pipe_device_view (d2slam_amd.swarm.PipeExchange): packs one exchange block per left frame
{desc, kps, scores, netvlad, n}, ships them with ONE RCCL all-gather, evaluates the reference's NetVLAD gate for every (local frame, remote frame) pair on the
device, matches its frames against every other rank's frame of the same time index and delivers those match lists to host memory too.

Prints ONE JSON line on rank 0 with `roofline` (the dominant kernel: conv1b, HIP events on the launch stream, one submit in flight), `step_roofline` (the whole
step against the matrix pipe), `roofline_netvlad` and `cpu_baseline` (torch/oneDNN network + the C oracle's post-processing and matcher on the host cores,
bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The frames-in-flight pipe gives every lane two HIP streams; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
# streams that share a queue serialise.  Must be set before the HIP runtime initialises; recorded in the JSON line (`env_overrides`).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from benchlib.common import *  # noqa: F401,F403,E402  (H, W, CAP, LANES_FOR, NV_FLOP_PER_IMG, PEAK_TFLOPS, ...)
from benchlib.stdout_contract import LINE_BUDGET, HEADLINE_KEYS, claim_stdout, silence_this_process, emit_line, slim, extras_path, headline  # noqa: E402
from benchlib.rooflines import (sp_executed_gflop_per_image, step_roofline, flag_above_peak, build_record, netvlad_roofline, index_parity_evidence,  # noqa: E402
                                profiled_traffic, live_traffic, conv1b_roofline)
from benchlib.pipe_legs import pipe_frames, exchange_on_one_gpu, stream_classes, run_pipe, mode_disagreement, self_launch, collective_evidence  # noqa: E402
from benchlib.other_legs import run_quadcam, run_latency, synthetic_sp_for_threshold  # noqa: E402


def main():
    claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=32, help="stereo frames per step and per GPU (32: 64 images per launch)")
    ap.add_argument("--precision", choices=["f32", "f16x2", "wino"], default=os.environ.get("D2FE_BENCH_PRECISION", "wino"),
                    help="wino: fp32, 3x3 layers as Winograd F(2x2,3x3) on the fp32 MFMA pipe (headline); f32: direct convolutions, "
                         "bitwise equal to the oracle's fmaf chains; f16x2: fp16 hi/lo split operands")
    ap.add_argument("--lanes", type=int, default=0, help="submits in flight of the frames-in-flight pipe (0: LANES_FOR[frames])")
    ap.add_argument("--no-batch-curve", action="store_true", help="skip the batch curve (stereo fps at 1, 2, 4, 8, 16, 32 stereo frames per submit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic (HBM bytes per conv1b launch)")
    ap.add_argument("--no-netvlad", action="store_true", help="time BASELINE configs[1] (SuperPoint + match only) as `value`")
    ap.add_argument("--no-h2d", action="store_true", help="frames resident in HBM before the timed region (no copy stream)")
    ap.add_argument("--workload", choices=["d435", "quadcam"], default="d435",
                    help="d435 = the metric's configuration; quadcam = configs[2]: 4 x (1280x800 raw -> 800x400) per frame, undistort + "
                         "SuperPoint + NetVLAD + neighbour (half-image, shifted, radius-gated) and temporal matchKNN")
    ap.add_argument("--single-mode", action="store_true", help="time only the headline leg (no configs1 / other modes / quadcam legs)")
    ap.add_argument("--async-tail", action="store_true", help="d2fe_config.async_tail without NetVLAD in between (post-processing of step k under the convolutions of step k+1)")
    ap.add_argument("--overlap", action="store_true",
                    help="d2fe_config.async_tail with NetVLAD queued on the main stream right behind the SuperPoint convolutions, beside SuperPoint's "
                         "latency-bound tail on the handle's tail stream.  Measured slower than plain stream order (2214 vs 2258 stereo fps: the two "
                         "sequences stretch each other, NetVLAD 0.94 -> 1.28 ms), hence off by default")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-stage event breakdown to stderr")
    ap.add_argument("--exchange", choices=["fp32", "int8", "int8-renorm256"], default=os.environ.get("D2FE_BENCH_EXCHANGE", "fp32"),
                    help="N>1: precision of the exchange blocks on the wire.  int8 = the reference's LCM wire format (VisualImageDesc::toLCM quantisation, "
                         "decoded exactly as its LCM constructor does: q/127 and the hard-coded 32-float renormalisation of the first n segments, which with "
                         "256-D descriptors leaves most rows un-normalised -- the reference's own cross-agent numerics); int8-renorm256 = the same bytes, every "
                         "descriptor re-normalised over its 256 floats on decode.  Either way 3.9x fewer all-gather bytes")
    ap.add_argument("--exchange-impl", choices=["capi", "torch"], default=os.environ.get("D2FE_BENCH_EXCHANGE_IMPL", "capi"),
                    help="N>1: capi = d2fe_exchange_* of the C ABI (csrc/exchange.hip): the sequence queued by the library on one stream of its own, "
                         "ncclAllGather on an RCCL communicator of the library's own (dlopen); torch = round 5's form (Python-driven, torch.distributed "
                         "collective, one stream of its own).  Two lanes either way.  capi falls back to torch when the C exchange cannot be created (recorded in `exchange.impl`)")
    ap.add_argument("--exchange-lane-streams", action="store_true", help="capi: the exchange on the producing lanes' streams instead of ONE stream of its own (A/B: measured 5-7 %% slower, "
                                                                         "profiles/r06_exchange_placement_ab.txt)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-call latency leg (host-pointer C ABI, one frame per call)")
    ap.add_argument("--latency-calls", type=int, default=300)
    ap.add_argument("--no-width-sensitivity", action="store_true", help="skip the NetVLAD trunk-width legs (`netvlad_width_sensitivity`)")
    ap.add_argument("--no-exchange-loopback", action="store_true", help="skip the one-GPU RCCL leg of the cross-agent exchange (`exchange_on_one_gpu_rccl`)")
    ap.add_argument("--no-parity-study", action="store_true", help="skip the in-run 128-image index-parity study (`index_parity_in_run`)")
    ap.add_argument("--no-solo", action="store_true", help="with --single-mode: skip the extra leg with ONE submit in flight that measures the dominant kernel alone")
    ap.add_argument("--latency-only", action="store_true", help="print only the single-call latency leg, in a process that does nothing but call the C ABI")
    ap.add_argument("--cpu-baseline-only", choices=["all_cores", "single_thread"], default=None,
                    help="print only that half of `cpu_baseline` (SURVEY.md section 8d protocol: warm-up 5, >= 50 timed iterations, median + p95); the default run starts both as "
                         "child processes beside its secondary GPU legs")
    ap.add_argument("--cpu-iterations", type=int, default=50, help="timed iterations of each cpu_baseline half (SURVEY.md section 8d: >= 50)")
    ap.add_argument("--force-dist", action="store_true", default=bool(int(os.environ.get("D2FE_BENCH_FORCE_DIST", "0") or 0)),
                    help="send --gpus 1 through the path --gpus 8 takes: self-launch under torch.distributed.run, init_process_group('nccl', device_id=...), the "
                         "collective evidence, the cross-agent exchange behind the pipe over the one-rank communicator (loopback: the rank's own blocks as the remote "
                         "agent), the N > 1 lane count, destroy_process_group")
    args = ap.parse_args()
    for k in REFUSED_ENV:
        if os.environ.get(k):
            raise SystemExit("%s is set: that switch changes what the kernels compute; bench.py refuses to time it" % k)

    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: re-launch this command line as N ranks (one process per GPU) under torch.distributed.run;
        # rank 0's JSON line passes through on (the original) stdout and the exit code is the launcher's
        rc = self_launch(args.gpus)
        silence_this_process()
        raise SystemExit(rc)

    if args.cpu_baseline_only:
        from d2slam_amd import netvlad as nvm
        from d2slam_amd.weights import synthetic_superpoint_weights
        emit_line({"cpu_baseline_half": args.cpu_baseline_only,
                   "result": run_cpu_baseline_half(args.cpu_baseline_only, synthetic_superpoint_weights(dustbin_bias=7.5), None if args.no_netvlad else nvm.synthetic_netvlad_weights(),
                                                   args.cpu_iterations)})
        return

    if args.latency_only:
        # the single-call latency leg in a process of its own that does nothing but call the C ABI (no torch, no other streams): how a D2SLAM
        # front end would use the library.  The default run spawns exactly this (below) -- inside the benchmark process, whose dozens of streams
        # share the runtime's few hardware queues, the two-stream entry points measure up to 70 % slower
        from d2slam_amd import api, netvlad as nvm
        from d2slam_amd.weights import synthetic_superpoint_weights
        emit_line({"latency": run_latency(api, synthetic_superpoint_weights(dustbin_bias=7.5), nvm.synthetic_netvlad_weights(),
                                          int(os.environ.get("LOCAL_RANK", "0")), args.precision, args.latency_calls)})
        return

    import torch
    import torch.distributed as dist
    from d2slam_amd import api, swarm
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or run `python bench.py --gpus %d` bare)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("D2FE_BENCH_BACKEND", "nccl")   # "gloo" only to exercise the N>1 code path on a 1-GPU box
    if local_rank >= ndev:
        if backend == "nccl":
            raise SystemExit("rank %d needs GPU %d but only %d are visible (RCCL does not put two ranks on one device; "
                             "D2FE_BENCH_BACKEND=gloo exercises the N>1 path on fewer GPUs)" % (rank, local_rank, ndev))
        local_rank = local_rank % ndev      # debug only: several ranks share one GPU under gloo
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # `dist_path`: the N > 1 path -- process group, collective evidence, the exchange behind the pipe, the N > 1 lane count.  --force-dist takes it with ONE rank
    # (the exchange then loops the rank's own blocks back as the remote agent), so that the first 8-GPU run can only fail for reasons that need 8 GPUs
    dist_path = world > 1 or args.force_dist
    loopback = dist_path and world == 1
    if dist_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def finish(full):
        """every rank: leave the process group; rank 0 alone owns stdout and writes the line LAST"""
        if rank != 0:
            silence_this_process()
        if dist_path:
            dist.destroy_process_group()
        if rank == 0:
            line, _ = headline(full)
            emit_line(line)

    rccl = collective_evidence(torch, dist, dev, backend, rank, world) if dist_path else None

    weights = synthetic_superpoint_weights(dustbin_bias=7.5)
    if args.workload == "quadcam":
        out = run_quadcam(args, torch, api, weights, dev, local_rank, world, rank)
        if rccl:
            out["rccl"] = rccl
        finish(out)
        return

    from d2slam_amd import netvlad as nvm
    nv_weights = nvm.synthetic_netvlad_weights()

    def run_mode(precision, want_breakdown, netvlad=True, steps=None):
        """N = 1 only: the same step through the DEVICE API on one handle (frames uploaded on a copy stream inside the timed region, results left in HBM).  Kept beside
        the pipe-timed `value` as a labelled extra (`device_resident`) and for the per-stage event breakdown (`stage_ms`); every `--gpus N` goes through run_pipe."""
        steps = steps or args.steps
        F = args.frames
        NI = 2 * F
        prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[precision]
        overlap = netvlad and args.overlap
        cfg = api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=NI, precision=prec,
                                   device_id=local_rank, async_tail=bool(args.async_tail or overlap))
        fe = api.FrontEnd(cfg)
        fe.load_superpoint(weights)
        G = 0
        if netvlad:
            fe.load_netvlad(nv_weights)
            G = fe.netvlad_dim
        # frame order within a step: [L0 .. L(F-1), R0 .. R(F-1)] -- the left images are one contiguous batch for NetVLAD
        # two alternating frame sets: even steps see the frames, odd steps the same scenes after a small camera motion (shifted by 3 x 2
        # pixels), so that L <-> previous-L is a real temporal match between DIFFERENT keypoint sets, not a frame against itself
        host = np.empty((2, NI, H, W), np.uint8)
        for f in range(F):
            l, r = synth_stereo(H, W, seed=rank * 1000 + f)
            host[0, f], host[0, F + f] = l, r
            host[1, f], host[1, F + f] = np.roll(l, (2, 3), (0, 1)), np.roll(r, (2, 3), (0, 1))
        host_pins = [torch.from_numpy(host[i]).pin_memory() for i in range(2)]
        imgs = [torch.empty((NI, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]

        # one pool of 256-float rows: [0, 2F*CAP) current L|R descriptors, [2F*CAP, 3F*CAP) previous L
        n_local_rows = 3 * F * CAP
        pool = torch.zeros((n_local_rows * 256,), dtype=torch.float32, device=dev)
        desc = pool.view(3 * F, CAP, 256)
        kps = torch.zeros((3 * F, CAP, 2), dtype=torch.float32, device=dev)
        cnt = torch.zeros((3 * F,), dtype=torch.int32, device=dev)
        scores = torch.zeros((NI, CAP), dtype=torch.float32, device=dev)
        kidx = torch.zeros((NI + F, CAP), dtype=torch.int32, device=dev)       # rows [NI, NI + F): the previous step's left images
        gdesc = torch.zeros((max(F, 1), max(G, 4)), dtype=torch.float32, device=dev)

        # pairs: (L_f, R_f), (L_f, prevL_f)
        pl = swarm.PairList(F, CAP, 1, 0, 0)
        NP = pl.npairs
        a_off = torch.tensor(pl.a_off, dtype=torch.int32, device=dev); b_off = torch.tensor(pl.b_off, dtype=torch.int32, device=dev)
        a_src = torch.tensor(pl.a_cnt_row, dtype=torch.int64, device=dev)
        b_src_local = torch.tensor(pl.b_cnt_row[:pl.n_local], dtype=torch.int64, device=dev)
        a_cnt = torch.zeros(NP, dtype=torch.int32, device=dev); b_cnt = torch.zeros(NP, dtype=torch.int32, device=dev)
        mq = torch.zeros((NP, CAP), dtype=torch.int32, device=dev); mt = torch.zeros((NP, CAP), dtype=torch.int32, device=dev)
        md = torch.zeros((NP, CAP), dtype=torch.float32, device=dev); mn = torch.zeros((NP,), dtype=torch.int32, device=dev)

        # every torch op and library launch of a step is ordered on ONE explicit (non-default) HIP stream:
        # the C ABI treats a NULL stream as "the handle's own stream", which would not be ordered with torch's default stream
        main = torch.cuda.Stream(device=dev)
        copy_s = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(main)
        stream = main.cuda_stream
        assert stream != 0
        tail = torch.cuda.ExternalStream(fe.tail_stream(), device=dev) if fe.tail_stream() else main
        tstream = tail.cuda_stream
        ev_copy = [torch.cuda.Event() for _ in range(2)]
        ev_free = [torch.cuda.Event() for _ in range(2)]
        use_h2d = not args.no_h2d
        state = {"k": 0}

        def upload(b):
            """frames of the next step: pinned host -> HBM on the copy stream (19.7 MB per 32 stereo frames)"""
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(ev_free[b])
                imgs[b].copy_(host_pins[b], non_blocking=True)
                ev_copy[b].record(copy_s)

        def step():
            k = state["k"]; b = k & 1; state["k"] = k + 1
            if use_h2d:
                main.wait_event(ev_copy[b])
                upload(b ^ 1)                      # next step's frames travel while this step computes
            im = imgs[b]
            if netvlad and not overlap:
                fe.netvlad_device(im.data_ptr(), F, W, H, gdesc.data_ptr(), stream=stream)
            fe.extract_device(im.data_ptr(), NI, W, H, kps.data_ptr(), scores.data_ptr(), desc.data_ptr(), kidx.data_ptr(),
                              CAP, cnt.data_ptr(), stream=stream)
            if netvlad and overlap:
                # behind the convolutions on `main`, beside SuperPoint's tail on the tail stream
                fe.netvlad_device(im.data_ptr(), F, W, H, gdesc.data_ptr(), stream=stream)
            if use_h2d:
                ev_free[b].record(main)            # the convolutions were the last readers of the frames (async tail: the trunk is on `main`)
            with torch.cuda.stream(tail):
                torch.index_select(cnt, 0, a_src, out=a_cnt)
                b_cnt[:pl.n_local] = cnt[b_src_local]
                fe.match_batch_device(pool.data_ptr(), pool.data_ptr(), a_off.data_ptr(), b_off.data_ptr(), a_cnt.data_ptr(),
                                      b_cnt.data_ptr(), NP, 256, CAP, mq.data_ptr(), mt.data_ptr(), md.data_ptr(), mn.data_ptr(),
                                      mode=0, ratio=0.8, radius=-1.0, stream=tstream)
                # this step's left descriptors become the "previous keyframe" of the next step
                desc[NI:NI + F].copy_(desc[:F])
                cnt[NI:NI + F].copy_(cnt[:F])
                kidx[NI:NI + F].copy_(kidx[:F])

        if use_h2d:
            for e in ev_free:
                e.record(main)
            upload(0)
        else:
            for b, im in enumerate(imgs):
                im.copy_(host_pins[b])
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(dev)
        fe.profile_enable(1)   # HIP events around the dominant kernel and the NetVLAD sequence only (4 event records per step)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        prof = fe.profile_read()
        fe.profile_enable(0)
        elapsed = t1 - t0
        value = F * steps / elapsed
        ms_per_step = elapsed / steps * 1e3
        n_kp = cnt[:NI].float().mean().item()
        breakdown = None
        if want_breakdown:
            fe.profile_enable(2)
            for _ in range(5):
                step()
            torch.cuda.synchronize(dev)
            breakdown = {k: round(v[0] / max(v[1], 1), 4) for k, v in fe.profile_read().items() if v[1]}
            fe.profile_enable(0)
            if args.breakdown:
                print("per-stage ms (avg of 5 steps, %d images/step): %s" % (NI, json.dumps(breakdown)), file=sys.stderr)
        nv_ms, nv_n = prof["netvlad"]
        roofline_nv = netvlad_roofline(nv_ms / nv_n, F, "HIP events around the whole sequence on the launch stream") if netvlad and nv_n else None
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
        fe.close()
        return dict(value=value, ms_per_step=ms_per_step, roofline_nv=roofline_nv, n_kp=n_kp, breakdown=breakdown, NI=NI, NP=NP, F=F)

    use_nv = not args.no_netvlad
    solo = None
    batch_curve = None
    device_resident = None
    noexch = None
    width_sens = None
    parity_in_run = None
    exch_1gpu = None
    # EVERY --gpus N times the frames-in-flight pipe (include/d2fe.h, d2fe_pipe_*): host frames in, host results out, `lanes` submits in flight.  N > 1 adds the
    # cross-agent exchange on a stream of its own beside it (run_pipe / swarm.PipeExchange) and keeps two submits in flight instead of four (LANES_WITH_EXCHANGE: the
    # exchange stream then has a hardware pipe it shares with a NetVLAD stream only); nothing else differs between `--gpus 1` and `--gpus 8`
    lanes = args.lanes or (LANES_FOR.get(args.frames, 2) if not dist_path else LANES_WITH_EXCHANGE.get(args.frames, LANES_FOR.get(args.frames, 2)))
    xmode = args.exchange if dist_path else None
    pk = dict(world=world, dist=dist, exchange=xmode, loopback=loopback, exchange_impl=args.exchange_impl, exchange_own_stream=not args.exchange_lane_streams)
    full_run = world == 1 and not dist_path and not args.single_mode          # the default `python bench.py`: every secondary leg
    primary = run_pipe(torch, api, weights, nv_weights, args.precision, args.frames, lanes, args.steps, args.warmup, local_rank, rank, netvlad=use_nv, **pk)
    if rank == 0 and not (primary.get("stream_placement") or {}).get("classes_told_apart"):
        # a pipe whose stream placement could not be MEASURED (device not quiet when it was created: two ranks on one GPU, another handle at work) used creation
        # order, which may put both of a lane's streams on one hardware pipe -- say so next to the number (`config.stream_placement.classes_told_apart` = 0 in the line)
        print("bench.py: WARNING: the timed pipe's stream placement was not measured (classes_told_apart = 0): streams in creation order", file=sys.stderr, flush=True)
    if world == 1 and not dist_path and lanes > 1 and not args.no_solo:
        # the step with ONE submit in flight: every kernel has the device to itself -- the dominant kernel's own roofline measurement (also the batch curve's F x 1 point)
        solo = run_pipe(torch, api, weights, nv_weights, args.precision, args.frames, 1, max(5, args.steps // 2) if args.single_mode else max(12, min(400, int(700 / args.frames))), 4,
                        local_rank, rank, netvlad=use_nv)
    latency = None
    if rank == 0 and full_run and not args.no_latency:
        # in a fresh process (see --latency-only); falls back to this process if the child fails.  Before the CPU children start: single calls are host-latency bound
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--latency-only", "--precision", args.precision, "--latency-calls",
                                str(args.latency_calls)], capture_output=True, text=True, timeout=600, env=dict(os.environ, LOCAL_RANK=str(local_rank)))
            latency = json.loads(r.stdout.strip().splitlines()[-1])["latency"]
            latency["process"] = "a process of its own that only calls the C ABI (python bench.py --latency-only)"
        except Exception as e:      # noqa: BLE001
            latency = run_latency(api, weights, nv_weights, local_rank, args.precision, args.latency_calls)
            latency["process"] = "the benchmark process (the child process failed: %s)" % str(e)[:100]
    # the batch curve BEFORE the CPU children start: its one-frame points are host-latency-bound (measured: 1338 stereo fps at 1 x 4 beside the children, 1686-1691 without)
    if full_run and not args.no_batch_curve:
        batch_curve = []
        for Fc in (1, 2, 4, 8, 16, 32):
            for Kc in sorted({1, LANES_FOR[Fc]} | ({2} if Fc <= 2 else set())):      # one or two frames per submit: also TWO in flight (the best plain configuration at F = 1)
                if Fc == args.frames and Kc == lanes:
                    r = primary
                elif Fc == args.frames and Kc == 1 and solo is not None:
                    r = solo
                else:
                    r = run_pipe(torch, api, weights, nv_weights, args.precision, Fc, Kc, max(12, min(400, int(700 / Fc))), max(2 * Kc, 4), local_rank, rank, netvlad=use_nv,
                                 light=True)
                batch_curve.append({"stereo_frames_per_submit": Fc, "submits_in_flight": Kc, "coalesce": 1, "stereo_fps": round(r["value"], 1), "ms_per_submit": round(r["ms_per_step"], 4),
                                    "host_ms_per_submit_call": round(r["host_submit_ms"], 4), "stream_classes": stream_classes(r)})
            if Fc == 1:
                # one stereo frame per submit, consecutive submits coalesced into one launch sequence when they arrive before anybody waits
                # (d2fe_pipe_config.coalesce): what a caller that receives single frames gets without batching by hand
                for cc in (2, 4):
                    r = run_pipe(torch, api, weights, nv_weights, args.precision, 1, 4, 600, 16, local_rank, rank, netvlad=use_nv, light=True, coalesce=cc)
                    batch_curve.append({"stereo_frames_per_submit": 1, "submits_in_flight": 4 * cc, "coalesce": cc, "stereo_fps": round(r["value"], 1),
                                        "ms_per_submit": round(r["ms_per_step"], 4), "host_ms_per_submit_call": round(r["host_submit_ms"], 4), "stream_classes": stream_classes(r),
                                        "note": "submit() stages the frame (its H2D starts at once); every %d-th submit launches ONE sequence over the staged frames, "
                                                "4 such passes in flight; per-ticket results are bit-identical to the single calls (tests/test_pipe.py)" % cc})
                if use_nv:
                    # four single-frame passes in flight, the NetVLAD descriptors of four consecutive submits from ONE call (d2fe_pipe_config.netvlad_group: the
                    # global descriptor feeds loop detection, not the tracker, so it may trail the keypoints by up to three submits); SuperPoint and the matches of
                    # every submit still start at once
                    r = run_pipe(torch, api, weights, nv_weights, args.precision, 1, 4, 600, 16, local_rank, rank, netvlad=True, light=True, nv_group=4)
                    batch_curve.append({"stereo_frames_per_submit": 1, "submits_in_flight": 4, "coalesce": 1, "netvlad_group": 4, "stereo_fps": round(r["value"], 1),
                                        "ms_per_submit": round(r["ms_per_step"], 4), "host_ms_per_submit_call": round(r["host_submit_ms"], 4), "stream_classes": stream_classes(r),
                                        "note": "netvlad_group = 4: one NetVLAD call per four consecutive single-frame submits; bit-identical results"})
                # ONE 4-lane pipe of plain single-frame passes (coalesce = 1) under callers that keep 1 / 2 / 3 submits outstanding (4: the K = 4 point above): the
                # pipe decides per pass which stream NetVLAD goes to (d2fe_pipe_config.netvlad_inline = auto), so the lone pass keeps the single-lane latency
                for infl in (1, 2, 3):
                    r = run_pipe(torch, api, weights, nv_weights, args.precision, 1, 4, 600, 16, local_rank, rank, netvlad=use_nv, light=True, inflight=infl)
                    batch_curve.append({"stereo_frames_per_submit": 1, "submits_in_flight": infl, "lanes": 4, "coalesce": 1, "stereo_fps": round(r["value"], 1),
                                        "ms_per_submit": round(r["ms_per_step"], 4), "host_ms_per_submit_call": round(r["host_submit_ms"], 4), "stream_classes": stream_classes(r),
                                        "note": "the 4-lane pipe of the K = 4 point with fewer submits outstanding"})
                # dynamic batching (coalesce_depth = 2): a pass is launched as soon as fewer than two are in flight, so the SAME configuration serves a caller
                # that waits for every frame (1 in flight: launched at once) and one that keeps 16 in flight (passes grow to 4 frames)
                for infl in (1, 4, 16):
                    r = run_pipe(torch, api, weights, nv_weights, args.precision, 1, 4, 600, 16, local_rank, rank, netvlad=use_nv, light=True, coalesce=4, depth=2, inflight=infl)
                    batch_curve.append({"stereo_frames_per_submit": 1, "submits_in_flight": infl, "coalesce": 4, "coalesce_depth": 2, "stereo_fps": round(r["value"], 1),
                                        "ms_per_submit": round(r["ms_per_step"], 4), "host_ms_per_submit_call": round(r["host_submit_ms"], 4), "stream_classes": stream_classes(r),
                                        "note": "dynamic batching: up to 4 consecutive submits per pass, launched early whenever fewer than 2 passes are in flight"})
    # cpu_baseline (SURVEY.md section 8d protocol: warm-up 5, >= 50 timed iterations, median + p95; one thread and all cores): two child processes started HERE, after the
    # headline, solo, latency and batch-curve legs, so that their 1-2 minutes pass beside the secondary GPU legs below (a few host threads on a 256-core box) instead of adding to the run
    cpu_children = start_cpu_baseline(args, use_nv) if (rank == 0 and world == 1 and not dist_path and not args.no_cpu_baseline) else None
    legs, legs_solo = {}, {}
    short = max(5, args.steps // 2)
    if dist_path:
        # the same ranks, the same step, WITHOUT the exchange: what the exchange costs the step (its kernels share the device with the lanes' launches)
        noexch = run_pipe(torch, api, weights, nv_weights, args.precision, args.frames, lanes, short, 2, local_rank, rank, netvlad=use_nv, light=True, world=world, dist=dist)
    if not args.single_mode:
        legs["configs1"] = run_pipe(torch, api, weights, nv_weights, args.precision, args.frames, lanes, short, 2, local_rank, rank, netvlad=False, **pk)
        for om in ("f32", "f16x2", "wino"):
            if om != args.precision:
                legs[om] = run_pipe(torch, api, weights, nv_weights, om, args.frames, lanes, short, 2, local_rank, rank, netvlad=use_nv, **pk)
                if not dist_path:
                    # the mode's OWN roofline object: the same step with ONE submit in flight (with several, the HIP-event duration of a launch includes the
                    # other lane's launches it shares the device with -- VERDICT r04 weak #5)
                    legs_solo[om] = run_pipe(torch, api, weights, nv_weights, om, args.frames, 1, short, 2, local_rank, rank, netvlad=use_nv)
    if full_run:
        dr = run_mode(args.precision, True, netvlad=use_nv, steps=short)
        device_resident = {"value": round(dr["value"], 2), "unit": "stereo_frames/s", "ms_per_step": round(dr["ms_per_step"], 3),
                           "workload": "the same step through the device API on one handle: frames uploaded on a copy stream inside the timed region, keypoints / "
                                       "descriptors / matches left in HBM (round 3's headline configuration; no D2H)", "breakdown": dr["breakdown"], "n_kp": dr["n_kp"], "NI": dr["NI"], "NP": dr["NP"], "roofline_nv": dr["roofline_nv"]}
        if use_nv and not args.no_width_sensitivity:
            # A9's architecture is an assumption (the reference's ONNX graph is not in its tree): what `value` becomes at other trunk widths
            from d2slam_amd import netvlad as nvm2
            width_sens = {"what": "`value` (same step, same pipe configuration) with the MobileNetVLAD stand-in at other MobileNetV2 depth multipliers; 0.75 (SURVEY A9's, HF-Net's) is "
                                  "the headline's, 0.35 was the width of rounds 2-5.  Both have specialised block kernels; 0.5 and 1.0 have channel counts those do not cover and run "
                                  "the generic fused-block / per-layer kernels (tests/test_gpu_parity.py::test_netvlad_other_trunk_widths holds every width to the oracle)", "points": []}
            for mult in (0.35, 0.5, 0.75, 1.0):
                if mult == NV_MULT:
                    r = primary
                else:
                    # best of two short runs: a leg of ten steps beside the CPU children's start-up has shown single outliers of 15 % (2039 / 2417 for the same build)
                    wts = nvm2.synthetic_netvlad_weights(depth_multiplier=mult)
                    r = min((run_pipe(torch, api, weights, wts, args.precision, args.frames, lanes, short, 4, local_rank, rank, netvlad=True, light=True,
                                      nv_flop_per_img=nvm2.arch_flops(mult, H, W)) for _ in range(2)), key=lambda q: q["ms_per_step"])
                width_sens["points"].append({"depth_multiplier": mult, "trunk_gflop_per_image": round(nvm2.arch_flops(mult, H, W) / 1e9, 3), "value": round(r["value"], 1),
                                             "ms_per_step": round(r["ms_per_step"], 3)})
        if not args.no_parity_study:
            # index parity of the TIMED build, collected in this run: a 128-image subset of tools/mode_disagreement.py's study
            from d2slam_amd import parity_study as ps
            t_ps = time.time()
            imgs_ps, pairs_ps, n_syn = ps.frames(48, n_real=16)
            rec = ps.study(api, imgs_ps, pairs_ps, n_syn, 0.015, CAP, 32, local_rank)
            parity_in_run = {"what": "keypoint / match index sets of both fast modes against the bitwise-exact fp32 mode, THIS build, THIS run: %d images (%d synthetic stereo "
                                     "pairs + %d frame pairs derived from the real crops of the reference's sample image), 640x480, N = %d, threshold 0.015; symmetric differences"
                                     % (len(imgs_ps), n_syn // 2, (len(imgs_ps) - n_syn) // 2, CAP),
                             "wino_vs_f32": rec["wino_vs_f32_all"], "f16x2_vs_f32": rec["f16x2_vs_f32_all"], "wino_vs_f32_real_derived": rec["wino_vs_f32_real_derived"],
                             "seconds": round(time.time() - t_ps, 1)}
    disagreement = None
    disagreement_fast = None
    if rank == 0 and world == 1 and "f32" in legs and args.precision == "wino":
        disagreement = mode_disagreement(primary["sel"], legs["f32"]["sel"], primary["F"])
        if "f16x2" in legs:
            disagreement_fast = mode_disagreement(legs["f16x2"]["sel"], legs["f32"]["sel"], primary["F"])

    quad = None
    if rank == 0 and full_run:
        qa = argparse.Namespace(**vars(args)); qa.steps = 8; qa.warmup = 2
        # best of two, as the width legs: eight steps of ~9 ms beside the CPU-baseline children have shown one outlier of -27 % (640 against 873-894 quad frames/s)
        runs = [run_quadcam(qa, torch, api, weights, dev, local_rank, world) for _ in range(2)]
        quad = max(runs, key=lambda q: q.get("value") or 0.0)
        quad["runs"] = [q.get("value") for q in runs]

    if rank == 0 and full_run and not args.no_exchange_loopback:
        # LAST leg of the run: it creates (and destroys) a one-rank RCCL communicator, whose proxy threads and streams must not sit beside any other measurement
        exch_1gpu = exchange_on_one_gpu(torch, dist, api, weights, nv_weights, args, args.lanes or LANES_WITH_EXCHANGE.get(args.frames, lanes), short, local_rank, rank, use_nv, dev)

    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not dist_path and not args.no_cpu_baseline:
        cpu_baseline = join_cpu_baseline(cpu_children, weights)
        parity = run_parity_check(primary, weights, nv_weights if use_nv else None, args.precision)

    out = None
    if rank == 0:
        value, ms_per_step = primary["value"], primary["ms_per_step"]
        NI, NP, F = primary["NI"], primary["NP"], primary["F"]
        PAR = {"f32": "bitwise vs oracle (activations, scores, indices, matches)",
               "f16x2": "descriptors <= 1e-4 (measured ~3e-7), scores <= 1e-5; keypoint indices equal except at score near-ties",
               "wino": "bitwise vs the oracle's restatement of the Winograd evaluation order; vs the direct chains: scores <= 3e-6, "
                       "descriptors <= 1e-5, keypoint indices equal except at score near-ties (with the seeded random weights near-ties are "
                       "frequent: e.g. 16 of 200 keypoints at 96x128; tests/test_wino.py)"}
        out = {
            "metric": "frames/sec SuperPoint+NetVLAD+match, 640x480 stereo" if use_nv else "frames/sec SuperPoint+match, 640x480 stereo (configs[1])",
            "value": round(value, 2), "unit": "stereo_frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"f32": "f32", "f16x2": "f16x2(hi+lo split)/f32-acc", "wino": "f32"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "BASELINE metric configuration: realsense_d435 stereo 640x480, 200 keypoints/frame: SuperPoint (both images) + "
                                   + ("NetVLAD (left image; stand-in MobileNetVLAD graph: MobileNetV2 alpha = 0.75 trunk, K = 32 -> 4096 (SURVEY A9); the reference's ONNX is not in its tree) + " if use_nv else "")
                                   + "matchKNN L<->R and L<->prevL"
                                   + ("; + one RCCL all-gather of exchange blocks, device NetVLAD gate, cross-agent matchKNN vs every remote frame" if dist_path else "")
                                   + ("; ONE rank through the N > 1 path (--force-dist): the rank's own blocks come back as the remote agent" if loopback else ""),
                       "frames_per_step_per_gpu": F, "images_per_step_per_gpu": NI, "match_pairs_per_step_per_gpu": NP,
                       "api": ("d2fe_pipe_submit / d2fe_pipe_wait (include/d2fe.h): host frames in (pinned), host results out (pinned), %d submits in flight" % primary["lanes"])
                              + ("" if not dist_path else "; cross-agent exchange per submit (d2fe_exchange_*: on the producing lane's stream) behind d2fe_pipe_device_view / _release (pack -> ONE all-gather -> gate -> "
                                                       "remote matchKNN -> D2H), enqueued one submit behind the pipe -- the SAME path as --gpus 1 plus that stream"),
                       "same_path_for_every_n_gpus": True,
                       "h2d_in_timed_region": True,
                       "d2h_in_timed_region": True,
                       "d2h_bytes_per_step": primary.get("d2h_bytes"),
                       "delivered": "keypoints, scores, descriptors, counts, NetVLAD descriptors and both match lists of every frame land in host memory inside the timed "
                                    "region (the reference's contract ends in host std::vectors, superpoint_tensorrt.cpp:172-180, loop_cam.cpp:619-645)" + ("" if not dist_path else
                                    "; N > 1: the cross-agent match lists and gate decisions too (a ring of pinned slots, one D2H per submit on the exchange stream)"),
                       "submits_in_flight": primary.get("lanes"), "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "stream_placement": dict(primary.get("stream_placement") or {}, what="d2fe_pipe_stream_placement of the timed pipe: the hardware-pipe class d2fe_pipe_create MEASURED "
                                                "for each lane's (own, second) stream; streams of one class take turns on the device; classes_told_apart 0 = the device was not quiet "
                                                "when the pipe was created (e.g. two ranks sharing one GPU) and creation order was used"),
                       "netvlad_overlaps_superpoint": "d2fe_pipe_config.netvlad_inline = auto: a pass's NetVLAD call runs on the lane's second stream beside its SuperPoint while at most one other pass is in "
                                                      "flight (always with one or two lanes), in front of its SuperPoint on the lane's own stream beyond that -- beside the OTHER lanes' work",
                       "max_keypoints": CAP, "postproc": "B",
                       "precision": args.precision, "netvlad": use_nv,
                       "weights": "seeded random-init SuperPoint / MobileNetVLAD stand-in (no checkpoints in the reference tree)"},
            "sp_tflops_algorithmic": round(SP_FLOP_PER_IMG * 2 * value / 1e12, 2),
            "step_roofline": step_roofline(args.precision, F, ms_per_step, use_nv, NP),
            "build": build_record(),
            "avg_keypoints_per_image": round(primary["n_kp"], 1), "avg_matches_per_pair": round(primary["n_match"], 1),
            "matcher_candidates_reranked_beyond_two_per_query_per_step": round(primary["fallback_rows"][0], 2),
            "matcher_exact_scan_rows_per_step": round(primary["fallback_rows"][1], 2),
            "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("D2FE_", "GPU_MAX_HW_QUEUES", "HIP_", "HSA_", "ROCR_", "AMD_"))},
            "roofline": primary["roofline"], "roofline_netvlad": primary["roofline_nv"], "cpu_baseline": cpu_baseline, "parity": parity,
            "mode_parity": PAR[args.precision],
        }
        if solo and solo.get("roofline") and primary.get("lanes", 1) > 1:
            # the kernel's own roofline fraction: measured with one submit in flight (same step, same frames, HIP events on the launch stream); with several
            # submits in flight a launch shares the device with the other lanes' launches, so its duration says nothing about the kernel
            shared = primary["roofline"]
            out["roofline"] = dict(solo["roofline"], measured="HIP events over a timed region of %d submits of the same step with ONE submit in flight (the `batch_curve` point "
                                   "%d x 1, %.1f stereo fps): the launch has the device to itself" % (solo["steps"], solo["F"], solo["value"]),
                                   in_timed_region_of_value={"avg_launch_ms": shared["avg_launch_ms"], "launches": shared["launches"], "frac_executed": shared["frac_executed"],
                                                             "note": "with %d submits in flight the launch overlaps the other lanes' NetVLAD / post-processing / copy work" % primary["lanes"]})
        if full_run and not args.no_live_traffic and args.precision == "wino" and args.frames == 32 and out["roofline"].get("kernel", "").startswith("conv_wino"):
            live = live_traffic("conv_wino_kernel<64, true, true, 0, 1, true")
            if live:
                out["roofline"]["traffic"] = live["traffic"]; out["roofline"]["traffic_note"] = live["note"]; out["roofline"]["traffic_counters"] = live["counters"]
        if device_resident and device_resident.get("roofline_nv"):
            # NetVLAD by itself on the device (the device-API leg queues it in front of SuperPoint on one stream); in the pipe it runs beside the lanes'
            # full-device launches, where its wall time is mostly waiting for compute units
            out["roofline_netvlad"] = dict(device_resident["roofline_nv"], measured="the device-API leg of this run (`device_resident`): the sequence alone on its stream",
                                           in_timed_region_of_value={"ms_per_call": (primary.get("roofline_nv") or {}).get("ms_per_call"),
                                                                     "note": "wall time of the sequence inside the running pipe (beside or in front of its lane's SuperPoint, the other lanes' "
                                                                             "full-device launches on the chip): waits for compute units included"})
        flag_above_peak(out["roofline"])
        if world == 1 and SP_FLOP_PER_IMG * 2 * value / 1e12 > PEAK_TFLOPS[args.precision] and args.precision == "wino":
            out["sp_tflops_algorithmic_note"] = ("algorithmic (direct-convolution) FLOPs of SURVEY.md section 8(a) per second: above the %.1f TF fp32-MFMA peak because the "
                                                 "Winograd layers execute 16/36 of those multiplies; the executed figure is `step_roofline`" % PEAK_TFLOPS[args.precision])
        if primary["gated"]:
            out["netvlad_gate"] = primary["gated"]
        if primary["exch"]:
            out["exchange"] = primary["exch"]
            if noexch:
                d = primary["ms_per_step"] - noexch["ms_per_step"]
                busy = (primary["exch"].get("step_timeline_ms") or {}).get("exchange_stream_busy_ms_per_submit")
                out["exchange"].update({"ms_per_step_without_exchange": round(noexch["ms_per_step"], 3), "value_without_exchange": round(noexch["value"], 2),
                                        "exchange_cost_ms_per_step": round(d, 3), "exchange_cost_frac_of_step": round(d / primary["ms_per_step"], 4),
                                        "overlapped": bool(busy is not None and d < busy),
                                        "overlapped_rule": "the step grows by less than the exchange stream's own busy time per submit: its work ran beside the lanes' launches, and "
                                                           "no lane stream ever waits for it (only a block's next writer, 2 x lanes passes later, waits for the release event)"})
        if rccl:
            out["rccl"] = rccl
        names = {"configs1": "configs1", "f32": "exact_mode", "f16x2": "fast_mode", "wino": "wino_mode"}
        for k, o in legs.items():
            e = {"value": round(o["value"], 2), "unit": "stereo_frames/s", "ms_per_step": round(o["ms_per_step"], 3)}
            if k == "configs1":
                e["workload"] = "BASELINE configs[1]: the same step without NetVLAD (SuperPoint + match only)"
                e["step_roofline"] = step_roofline(args.precision, F, o["ms_per_step"], False, o["NP"])
            else:
                e["precision"] = k; e["parity"] = PAR[k]
                e["step_roofline"] = step_roofline(k, F, o["ms_per_step"], use_nv, o["NP"])
                so = legs_solo.get(k)
                if so and so.get("roofline"):
                    # measured like the headline's: the mode's step with ONE submit in flight (the launch has the device to itself)
                    e["roofline"] = flag_above_peak(dict(so["roofline"], measured="HIP events, %d submits of this mode's step with ONE submit in flight (%.1f stereo fps)" % (so["steps"], so["value"])))
            out[names[k]] = e
        if disagreement:
            out["wino_vs_exact_on_bench_frames"] = disagreement
        if disagreement_fast:
            out["f16x2_vs_exact_on_bench_frames"] = disagreement_fast
        if parity_in_run:
            out["index_parity_in_run"] = parity_in_run
        ev = index_parity_evidence()
        if ev:
            out["index_parity_evidence"] = ev
        if width_sens:
            out["netvlad_width_sensitivity"] = width_sens
        if exch_1gpu:
            out["exchange_on_one_gpu_rccl"] = exch_1gpu
        if batch_curve:
            out["batch_curve"] = {"what": "stereo fps of the same step (H2D, SuperPoint L+R, NetVLAD L, matchKNN L<->R and L<->previous L, D2H of everything) against the stereo frames "
                                          "per submit, through d2fe_pipe_*; submits_in_flight = 1 is the synchronous single-call form (the way the reference calls the path, "
                                          "loop_cam.cpp:609-616), the other line keeps that many submits in flight on separate streams", "points": batch_curve}
        if device_resident:
            out["device_resident"] = {k: device_resident[k] for k in ("value", "unit", "ms_per_step", "workload")}
        if latency:
            out["latency"] = latency
        if quad:
            out["quadcam"] = {k: quad[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "avg_keypoints_per_image", "avg_matches_per_pair", "roofline", "runs")}
        b = primary["breakdown"] or (device_resident or {}).get("breakdown")
        if b:
            out["stage_ms"] = b
            n_kp = primary["n_kp"]
            if not primary["breakdown"]:
                NI, NP = device_resident["NI"], device_resident["NP"]

            def _gbps(nbytes, ms):
                return {"ms_per_launch": ms, "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                        "frac_of_hbm_8000GBps": round(nbytes / (ms * 1e-3) / 8.0e12, 4)} if ms else None
            out["hbm_kernels"] = {
                "softmax_cand_kernel": dict(_gbps(65 * (H // 8) * (W // 8) * 4 * NI, b.get("softmax_cand")) or {},
                                            note="bound by VALU issue, not HBM: 65 correctly rounded exponentials + divisions per cell (bitwise scores) are ~1030 VALU instructions per "
                                                 "wave, 2.0e7 per 64 images (SQ_INSTS_VALU, profiles/r04_wino_rocprofv3_summary.txt) = 32 us at the chip's full issue rate"),
                "sample_b_kernel": _gbps(n_kp * NI * (4 * 1024 + 1024), b.get("sample")),
                "match_kernel": dict(_gbps(2 * CAP * 256 * 4 * NP, b.get("match")) or {}, pairs_per_launch=NP,
                                     mfma_tflops=round(2 * 2.0 * CAP * CAP * 256 * NP / (b.get("match") * 1e-3) / 1e12, 2) if b.get("match") else None,
                                     note="two distance strips per pair (one per direction): 2 x 2 x na x nb x 256 FLOP on v_mfma_f32_16x16x4_f32"),
            }
    finish(out)


# ---- cpu_baseline (kind "port") and the in-run parity check: the ONLY places of the benchmark that touch oracle/ (the checker and the reported baseline, never the
# thing measured) -- kept in this file, run as child processes (--cpu-baseline-only) beside the secondary GPU legs ------------------------------------------------------
def _torch_superpoint(torch, F, x, w):
    """superpoint.ipynb:300-374 in plain PyTorch (oneDNN convolutions on the host): image [n,1,H,W] fp32 in [0,1] -> semi [n,H,W], raw desc [n,256,H/8,W/8]."""
    def cv(t, name, relu=True):
        Wt, b = w[name]
        y = F.conv2d(t, Wt, b, padding=Wt.shape[-1] // 2)
        return torch.relu_(y) if relu else y
    x = cv(x, "conv1a"); x = cv(x, "conv1b"); x = F.max_pool2d(x, 2, 2)
    x = cv(x, "conv2a"); x = cv(x, "conv2b"); x = F.max_pool2d(x, 2, 2)
    x = cv(x, "conv3a"); x = cv(x, "conv3b"); x = F.max_pool2d(x, 2, 2)
    x = cv(x, "conv4a"); x = cv(x, "conv4b")
    semi = cv(cv(x, "convPa"), "convPb", False)
    desc = cv(cv(x, "convDa"), "convDb", False)
    sm = torch.softmax(semi, 1)[:, :64]
    n, _, hc, wc = sm.shape
    sm = sm.permute(0, 2, 3, 1).reshape(n, hc, wc, 8, 8).permute(0, 1, 3, 2, 4).reshape(n, hc * 8, wc * 8)
    desc = desc / torch.norm(desc, p=2, dim=1, keepdim=True)
    return sm, desc


def run_cpu_baseline_half(which, weights, nv_weights, iterations):
    """One half of `cpu_baseline`: the same step on the host cores, BASELINE.md section 3 / SURVEY.md section 8(d).  The reference has no runnable CPU extractor
    (SURVEY F2), so this is kind "port": the network in PyTorch (oneDNN convolutions) + the C oracle's variant-B post-processing, NetVLAD and matchKNN.
    Inputs are generated BEFORE timing; 5 warm-up iterations, then `iterations` (>= 50 by default) timed ones of ONE stereo frame each (all_cores: TWO, so that the
    batch dimension feeds the threads), every iteration timed per stage (prep u8 -> f32/255, conv stack, post-processing, NetVLAD, matching: L<->R and
    L<->previous L); median and p95 per stage and end to end.  `single_thread`: ONE thread -- what the reference pins every host library to (d2frontend.cpp:303,
    onnx_generic.h:32, superpoint_onnx.cpp:26).  `all_cores`: the best of 16 / 32 / 64 threads (3-iteration probes; more intra-op threads collapse oneDNN's
    convolutions), the C stages threaded over frames (ctypes releases the GIL) and OpenMP inside the NetVLAD convolutions."""
    import ctypes
    import torch
    import torch.nn.functional as F
    from concurrent.futures import ThreadPoolExecutor
    from d2slam_amd.synth import synth_stereo
    from oracle import oracle as orc
    orc.build()
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    tw = {k: (torch.from_numpy(np.ascontiguousarray(v[0])), torch.from_numpy(np.ascontiguousarray(v[1]))) for k, v in weights.items()}
    ncores = os.cpu_count() or 1
    NF = 8
    frames = [synth_stereo(H, W, seed=i) for i in range(NF)]          # before any timing
    STAGES = ["prep", "conv_stack", "postproc"] + (["netvlad"] if nv_weights is not None else []) + ["match"]

    def iteration(B, first, pool, state):
        """one batch of B stereo frames; returns seconds per stage"""
        t = {}
        fr = [frames[(first + i) % NF] for i in range(B)]
        t0 = time.perf_counter()
        x = torch.from_numpy(np.stack([p[0] for p in fr] + [p[1] for p in fr]).astype(np.float32) * np.float32(1.0 / 255.0))[:, None]
        t["prep"] = time.perf_counter() - t0; t0 = time.perf_counter()
        with torch.no_grad():
            semi, desc = _torch_superpoint(torch, F, x, tw)
            semi = semi.numpy(); desc = desc.permute(0, 2, 3, 1).contiguous().numpy()
        t["conv_stack"] = time.perf_counter() - t0; t0 = time.perf_counter()

        def post(i):
            k, _, _ = orc.select_b(semi[i], 0.015, 1, CAP)
            return orc.sample_b(desc[i], k)
        d = list(pool.map(post, range(2 * B))) if pool else [post(i) for i in range(2 * B)]
        t["postproc"] = time.perf_counter() - t0; t0 = time.perf_counter()
        if nv_weights is not None:
            for i in range(B):
                orc.netvlad_forward(fr[i][0], nv_weights)        # OpenMP inside
            t["netvlad"] = time.perf_counter() - t0; t0 = time.perf_counter()
        prev = state.get("prev") or d[:B]
        jobs = [(d[i], d[B + i]) for i in range(B)] + [(d[i], prev[i % len(prev)]) for i in range(B)]
        mm = (lambda ab: orc.match_knn(ab[0], ab[1], 0.8))
        _ = list(pool.map(mm, jobs)) if pool else [mm(j) for j in jobs]
        state["prev"] = d[:B]
        t["match"] = time.perf_counter() - t0
        return t

    def measure(nthreads, B, warm, iters):
        torch.set_num_threads(nthreads)
        if gomp is not None:
            gomp.omp_set_num_threads(int(nthreads))
        pool = ThreadPoolExecutor(max_workers=min(nthreads, 2 * B)) if nthreads > 1 else None
        state = {}
        for w_ in range(warm):
            iteration(B, w_ * B, pool, state)
        rows = [iteration(B, i * B, pool, state) for i in range(iters)]
        if pool:
            pool.shutdown()
        per = {k: np.array([r[k] for r in rows]) / B * 1e3 for k in STAGES}          # ms per stereo frame
        tot = sum(per.values())
        return {"threads": nthreads, "stereo_frames_per_iteration": B, "warmup_iterations": warm, "timed_iterations": len(rows),
                "stereo_frames_per_s_median": round(1e3 / float(np.median(tot)), 4), "stereo_frames_per_s_p95_slowest": round(1e3 / float(np.percentile(tot, 95)), 4),
                "ms_per_stereo_frame": {k: {"median": round(float(np.median(v)), 3), "p95": round(float(np.percentile(v, 95)), 3)} for k, v in per.items()}}

    t0 = time.perf_counter()
    if which == "single_thread":
        r = measure(1, 1, CPU_WARMUP, iterations)
        r["note"] = "the reference pins its host libraries to one thread (d2frontend.cpp:303)"
    else:
        cands = sorted({min(ncores, c) for c in (16, 32, 64)})
        probe = {c: measure(c, 2, 1, 3)["stereo_frames_per_s_median"] for c in cands}
        r = measure(max(probe, key=probe.get), 2, CPU_WARMUP, iterations)
        r["thread_count_probe"] = {str(k): v for k, v in probe.items()}
    r["host_cores"] = ncores
    r["seconds"] = round(time.perf_counter() - t0, 1)
    return r


def start_cpu_baseline(args, use_nv):
    """both halves of `cpu_baseline` as child processes of their own (python bench.py --cpu-baseline-only ...): no GPU, no torch.cuda in them"""
    import subprocess
    kids = {}
    for which in ("all_cores", "single_thread"):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", which, "--cpu-iterations", str(args.cpu_iterations)] + ([] if use_nv else ["--no-netvlad"])
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", D2FE_BENCH_CHILD="1")
        kids[which] = (subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env), time.time())
    return kids


def join_cpu_baseline(kids, weights):
    from oracle import oracle as orc
    from d2slam_amd.synth import synth_stereo
    halves = {}
    waited = time.time()
    for which, (p, t_start) in (kids or {}).items():
        try:
            so, se = p.communicate(timeout=900)
            halves[which] = json.loads(so.strip().splitlines()[-1])["result"]
        except Exception as e:      # noqa: BLE001
            p.kill()
            halves[which] = {"error": str(e)[:200]}
    waited = time.time() - waited
    best, one = halves.get("all_cores") or {}, halves.get("single_thread") or {}
    ncores = os.cpu_count() or 1
    # the scalar fmaf-chain oracle the parity tests use (OpenMP over all cores): a labelled extra, not a tuned CPU path.  After the children: it takes every core
    orc.build()
    t0 = time.perf_counter()
    l, r = synth_stereo(H, W, seed=0)
    kl, sl, dl, _, _ = orc.extract_b(l, weights, 0.015, 1, CAP)
    kr, sr, dr, _, _ = orc.extract_b(r, weights, 0.015, 1, CAP)
    orc.match_knn(dl, dr, 0.8); orc.match_knn(dl, dl, 0.8)
    fps_orc = 1.0 / (time.perf_counter() - t0)
    return {"value": best.get("stereo_frames_per_s_median"), "unit": "stereo_frames/s", "cores": best.get("threads"), "kind": "port", "host_cores": ncores,
            "sample": "%d + %d iterations of 2 stereo frames (all cores) / 1 stereo frame (one thread), 640x480, SuperPoint (L+R) + NetVLAD (L) + 2 matchKNN per frame; 8 pre-generated "
                      "stereo frames cycled; network in PyTorch-CPU (oneDNN), post-processing / NetVLAD / matching in the C oracle" % (CPU_WARMUP, best.get("timed_iterations") or 0),
            "protocol": "SURVEY.md section 8(d): inputs generated before timing; warm-up %d; >= 50 timed iterations; per-stage wall time per iteration; median / p95.  Both halves ran as child "
                        "processes beside this run's secondary GPU legs (a few host threads); the run waited %.1f s for them at the end" % (CPU_WARMUP, waited),
            "all_cores": best, "single_thread": one,
            "fmaf_oracle": {"value": round(fps_orc, 4), "unit": "stereo_frames/s", "cores": ncores,
                            "sample": "1 stereo frame through oracle/d2fe_oracle.c (one fp32 fmaf chain per output, OpenMP): the parity checker, not a tuned CPU path"}}


def run_parity_check(primary, weights, nv_weights, precision):
    """In-run smoke of the timed configuration: the left image of stereo frame 0 against the oracle in the mode's evaluation order."""
    from d2slam_amd.synth import synth_stereo
    from oracle import oracle as orc
    l = synth_stereo(H, W, seed=0)[0]
    ok, os_, od = orc.extract_b(l, weights, 0.015, 1, CAP, wino=(precision == "wino"))[:3]
    gk, gs, gd = primary["first"]
    p = {"checked": "left image of stereo frame 0 vs the oracle (same run; the parity evidence proper is tests/ -m gpu, incl. the reference's own C++ via oracle/_ref)",
         "keypoints_equal": bool(gk.shape == ok.shape and np.array_equal(gk, ok)),
         "scores_equal": bool(gs.shape == os_.shape and np.array_equal(gs, os_)),
         "desc_max_abs_diff": float(np.abs(gd - od).max()) if gd.shape == od.shape else None}
    if nv_weights is not None and primary["gfirst"] is not None:
        ref = orc.netvlad_forward(l, nv_weights)
        p["netvlad_max_abs_diff"] = float(np.abs(primary["gfirst"] - ref).max())
    return p


if __name__ == "__main__":
    main()
