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
# lanes in flight per frames-per-submit for the batch curve (measured: tools/pipe_probe.py; more lanes than this do not pay)
LANES_FOR = {1: 4, 2: 4, 4: 4, 8: 4, 16: 4, 32: 4}
# ... and with the cross-agent exchange beside the lanes (--gpus N > 1, and the one-GPU RCCL leg): the device runs four busy streams side by side, so two lanes (SuperPoint
# and NetVLAD stream each) leave the exchange stream a hardware pipe it shares with a NetVLAD stream only: 2445-2453 stereo frames/s per rank with the exchange against
# 2416-2425 with three or four lanes, where it takes turns with a lane's SuperPoint stream (measured over one-rank RCCL)
LANES_WITH_EXCHANGE = {16: 2, 32: 2}          # round 6 A/B over RCCL loopback (profiles/r06_exchange_placement_ab.txt): d2fe_exchange_* on ONE stream of its own beside TWO lanes
                                              # 2230-2235; beside four lanes 2200-2206; on the producing lanes' streams 2128-2151 (the lane's next pass waits for the sequence)
REFUSED_ENV = ("D2FE_ABLATE", "D2FE_MATCH_NOFALLBACK")     # switches that make results wrong or parity unproven: never inside a benchmark

H, W, CAP = 480, 640, 200
CONV1B_FLOP_PER_IMG = 2.0 * H * W * 64 * 64 * 9          # 22.65 GFLOP (SURVEY.md section 8a layer table)
SP_FLOP_PER_IMG = 52.1e9
NV_MULT = 0.75                                            # MobileNetV2 alpha = 0.75 trunk (HF-Net's width) -> NetVLAD K = 32 -> 4096
NV_FLOP_PER_IMG = 2.4780096e9                             # the stand-in MobileNetVLAD trunk at that width, 640x480: 1.239 GMAC (d2slam_amd.netvlad.arch_flops(0.75))
PEAK_TFLOPS = {"f32": 157.3, "f16x2": 2500.0, "wino": 157.3}             # MI355X_MICROARCH.md: fp32 MFMA / dense f16 MFMA
NETVLAD_GATE = 0.8                                        # track_remote_netvlad_thres stand-in (the YAMLs carry 0.5..0.8)

# ---- the stdout contract: ONE JSON line, the LAST thing on stdout, on every rank, small enough to survive a tail ------------------------------------------------------
# RCCL prints its version banner through C stdio, which on a pipe or file is flushed at process exit -- behind anything Python printed (round 5's line was lost to that).
# So (i) fd 1 is pointed at stderr for the whole run (the banner, torch, rocprofv3 children, stray prints land there), the JSON line goes to the saved descriptor;
# (ii) before the line is written every C stream is flushed (a driver that merges stderr into stdout still sees the banner BEFORE the line); (iii) after the line fds 1 and 2
# of this process go to /dev/null: nothing can follow it.  Ranks other than 0 never own stdout at all.
_REAL_STDOUT = None
LINE_BUDGET = 6000                  # bytes; the driver keeps an 8 KB tail
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                 "roofline", "cpu_baseline", "step_roofline", "parity", "roofline_netvlad", "rccl", "exchange", "netvlad_gate", "cross_agent", "extras")
# dropped from the line (kept in the extras file) in this order while the line is above LINE_BUDGET
SHED_ORDER = ("cross_agent", "netvlad_gate", "exchange", "roofline_netvlad", "rccl")


def claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def silence_this_process():
    """fds 1 and 2 -> /dev/null (after flushing every C and Python stream): whatever this process still prints (exit-time banners, teardown warnings) goes nowhere"""
    import ctypes
    try:
        sys.stdout.flush(); sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    nul = os.open(os.devnull, os.O_WRONLY)
    os.dup2(nul, 1); os.dup2(nul, 2)


def emit_line(obj):
    """the JSON line: everything buffered so far is flushed first, the line is written to the process's ORIGINAL stdout in one piece, then the process goes silent"""
    global _REAL_STDOUT
    import ctypes
    data = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush(); sys.stderr.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    fd = _REAL_STDOUT if _REAL_STDOUT is not None else 1
    while data:
        n = os.write(fd, data)
        data = data[n:]
    silence_this_process()
    if _REAL_STDOUT is not None:
        os.close(_REAL_STDOUT)
        _REAL_STDOUT = None


def slim(o, maxlen=150, keep=("workload", "sample", "kernel", "api")):
    """the headline form of a (nested) record: prose longer than `maxlen` characters lives in the extras file; the strings a reader needs to identify the
    workload stay, cut to 320 characters"""
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if isinstance(v, str) and len(v) > maxlen:
                if k in keep:
                    out[k] = v if len(v) <= 320 else v[:317] + "..."
                continue
            out[k] = slim(v, maxlen, keep)
        return out
    if isinstance(o, list):
        return [slim(v, maxlen, keep) for v in o]
    return o


def extras_path():
    p = os.environ.get("D2FE_BENCH_EXTRAS")
    if p:
        return p
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, "bench_extras.json")
    except OSError:
        import tempfile
        return os.path.join(tempfile.gettempdir(), "d2fe_bench_extras.json")


def headline(full):
    """(line, path): the full record goes to the extras file (named in the line), the line keeps HEADLINE_KEYS in slim form and fits LINE_BUDGET"""
    path = extras_path()
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        rel = os.path.relpath(path, ROOT)
        named = rel if not rel.startswith("..") else path
    except OSError as e:
        named = "not written: %s" % str(e)[:80]
    always = ("vs_baseline", "cpu_baseline", "roofline")          # present (null when this run has none) in every line
    line = {k: slim(full.get(k)) for k in HEADLINE_KEYS if k in always or full.get(k) is not None}
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):          # the per-stage tables and the fmaf-chain extra stay in the file
        cb = {k: ({a: b for a, b in v.items() if a != "ms_per_stereo_frame"} if isinstance(v, dict) else v) for k, v in cb.items() if k != "fmaf_oracle"}
        line["cpu_baseline"] = dict(cb, per_stage="extras file: cpu_baseline.{all_cores,single_thread}.ms_per_stereo_frame")
    line["extras"] = {"file": named, "keys": sorted(k for k in full if k not in line)}
    if isinstance(line.get("rccl"), dict) and len(line["rccl"].get("ranks") or []) > 2:
        line["rccl"] = dict(line["rccl"], ranks="%d entries in the extras file" % len(line["rccl"]["ranks"]))
    for k in SHED_ORDER:
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        if k in line:
            del line[k]
            line["extras"]["keys"] = sorted(line["extras"]["keys"] + [k])
    if len(json.dumps(line)) > LINE_BUDGET:
        line["extras"]["keys"] = "see the file"
    return line, path


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
        quad = run_quadcam(qa, torch, api, weights, dev, local_rank, world)

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
            out["quadcam"] = {k: quad[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "avg_keypoints_per_image", "avg_matches_per_pair", "roofline")}
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


# executed matrix-pipe FLOPs of ONE 640x480 image through SuperPoint (sparse descriptor head at <= 4 corner cells per keypoint, 200 keypoints): the layer table of
# SURVEY.md section 8(a) with the arithmetic each mode runs.  GFLOP: conv1a 0.354 (as staged inside conv1b's kernel: 60 MFMAs per 8x16 item = 0.590), conv1b 22.65,
# conv2a 5.66, conv2b 5.66, conv3a 2.83, conv3b 5.66, conv4a 1.416, conv4b 1.416, convPa 2.831, convPb 0.160, descriptor head at the selected cells 0.28 (18 GFLOP per
# 64 images; the dense convDa + convDb would be 3.46)
_SP_WINO_LAYERS_GF = 22.65 + 5.66 + 5.66 + 2.83 + 5.66 + 1.416 + 1.416 + 2.831       # the eight 3x3 layers the Winograd mode runs as F(2x2,3x3): 16/36 of these
_SP_OTHER_GF = 0.160 + 0.28


def sp_executed_gflop_per_image(precision):
    if precision == "wino":
        return 0.590 + _SP_WINO_LAYERS_GF * 16.0 / 36.0 + _SP_OTHER_GF
    direct = 0.354 + _SP_WINO_LAYERS_GF + _SP_OTHER_GF
    return direct * (3.0 if precision == "f16x2" else 1.0)


def step_roofline(precision, F, ms_per_step, netvlad, npairs):
    """The WHOLE step against the matrix pipe: executed MFMA FLOPs of everything a step launches / ms_per_step / peak.  Analytic counts (the layer
    table above; NetVLAD 2.478 GFLOP per left image (MobileNetV2-0.75 trunk) on the fp32 pipe; matchKNN 2 strips x 2 na nb 256 per pair); the per-kernel SQ_INSTS_MFMA sums of the committed
    profile of the same command agree (profiles/: 3.46e8 x 4096 = 1.42 TFLOP per 64-image step in Winograd mode)."""
    peak = PEAK_TFLOPS[precision]
    sp = sp_executed_gflop_per_image(precision) * 1e9 * 2 * F
    nv = NV_FLOP_PER_IMG * F if netvlad else 0.0
    mt = 2 * 2.0 * CAP * CAP * 256 * npairs
    # NetVLAD and the matcher run on the fp32 pipe in every mode; in f16x2 mode their FLOPs are priced at the fp32 peak separately
    if precision == "f16x2":
        frac = (sp / (peak * 1e12) + (nv + mt) / (PEAK_TFLOPS["f32"] * 1e12)) / (ms_per_step * 1e-3)
    else:
        frac = (sp + nv + mt) / (peak * 1e12) / (ms_per_step * 1e-3)
    return {"bound": "mfma", "executed_mfma_tflop_per_step": round((sp + nv + mt) / 1e12, 4), "superpoint": round(sp / 1e12, 4), "netvlad": round(nv / 1e12, 4), "matcher": round(mt / 1e12, 4),
            "ms_per_step": round(ms_per_step, 3), "achieved": round((sp + nv + mt) / (ms_per_step * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(frac, 4),
            "algorithmic_tflop_per_step": round((SP_FLOP_PER_IMG * 2 * F + nv + mt / 2) / 1e12, 4),
            "note": "executed matrix-pipe FLOPs of every launch of a step / wall time of the step (H2D, D2H, post-processing and launch gaps included) / peak"
                    + ("; f16x2: SuperPoint's 3 MFMA FLOPs per algorithmic FLOP against the f16 peak, NetVLAD + matcher against the fp32 peak" if precision == "f16x2" else "")}


def flag_above_peak(r):
    """a roofline object whose ALGORITHMIC fraction exceeds 1 says why, next to the number"""
    if r and (r.get("frac_algorithmic") or 0) > 1.0:
        r["algorithmic_above_peak"] = "Winograd F(2x2,3x3): 16/36 of the direct convolution's multiplies are executed; `frac` (= frac_executed) is the matrix pipe's fraction"
    return r


def build_record():
    """what d2slam_amd.build recorded for the library this run timed (a build that fell back to untuned flags must be visible in the line)"""
    try:
        from d2slam_amd import build as hb
        bi = hb.build_info() or {}
        return {"hipcc": bi.get("hipcc"), "tuned_flags": bi.get("tuned_flags"), "compiled_without_tuned_flags": bi.get("compiled_without_tuned_flags"),
                "recorded": bool(bi)}
    except Exception:      # noqa: BLE001
        return {"recorded": False}


def pipe_frames(F, rank):
    """two alternating host frame sets for the pipe, [2 sets][L|R][F][H][W] u8: consecutive frames are pairs (scene, the scene after a small camera
    motion), so L_f <-> L_(f-1) is a real temporal match for odd f; set 1 = set 0 after a further motion (F = 1: the temporal partner is the other set)"""
    from d2slam_amd.synth import synth_stereo
    host = np.empty((2, 2, F, H, W), np.uint8)
    for f in range(F):
        l, r = synth_stereo(H, W, seed=rank * 1000 + f // 2)
        if f & 1:
            l, r = np.roll(l, (1, 2), (0, 1)), np.roll(r, (1, 2), (0, 1))
        host[0, 0, f], host[0, 1, f] = l, r
        host[1, 0, f], host[1, 1, f] = np.roll(l, (2, 3), (0, 1)), np.roll(r, (2, 3), (0, 1))
    return host


def exchange_on_one_gpu(torch, dist, api, weights, nv_weights, args, lanes, steps, local_rank, rank, use_nv, dev):
    """What the N > 1 exchange costs the step on the REAL backend, as far as one GPU can show it ("--gpus 1 through that path equals BENCH value within 1 %"):
    a one-rank RCCL communicator, the rank's own blocks as the remote agent (PipeExchange loopback: F cross-agent pairs per submit, every keypoint matches itself), the same
    pipe configuration as `value`, alternating with the same step without the exchange.  Any failure (no RCCL, rendezvous) is reported, never fatal."""
    created = False
    try:
        if not dist.is_initialized():
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = str(port)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            created = True
        runs = []
        for _ in range(2):
            w = run_pipe(torch, api, weights, nv_weights, args.precision, args.frames, lanes, steps, 2, local_rank, rank, netvlad=use_nv, light=True, world=1, dist=dist,
                         exchange=args.exchange, loopback=True, exchange_impl=args.exchange_impl)
            wo = run_pipe(torch, api, weights, nv_weights, args.precision, args.frames, lanes, steps, 2, local_rank, rank, netvlad=use_nv, light=True)
            runs.append((w, wo))
        w = min((r[0] for r in runs), key=lambda r: r["ms_per_step"]); wo = min((r[1] for r in runs), key=lambda r: r["ms_per_step"])
        return {"backend": dist.get_backend(), "impl": w["exch"].get("impl"), "what": "the step `--gpus N` runs on every rank (%d stereo frames per submit, %d submits in flight) with the cross-agent exchange over a ONE-rank "
                           "RCCL communicator (loopback: the rank's own blocks as the remote agent, %d cross-agent pairs per submit), against the same step without it; best of two "
                           "alternating runs each" % (args.frames, lanes, w["exch"]["cross_agent_pairs_per_step_per_gpu"]),
                "value_with_exchange": round(w["value"], 2), "value_without_exchange": round(wo["value"], 2), "ms_per_step_with_exchange": round(w["ms_per_step"], 3),
                "ms_per_step_without_exchange": round(wo["ms_per_step"], 3), "exchange_cost_frac_of_step": round(w["ms_per_step"] / wo["ms_per_step"] - 1.0, 4), "lanes": lanes,
                "step_timeline_ms": w["exch"]["step_timeline_ms"], "wire_precision": args.exchange}
    except Exception as e:      # noqa: BLE001
        return {"error": str(e)[:300]}
    finally:
        if created:
            try:
                dist.destroy_process_group()
            except Exception:      # noqa: BLE001
                pass


def netvlad_roofline(t_ms, F, how, flop_per_img=NV_FLOP_PER_IMG):
    ach = flop_per_img * F / (t_ms * 1e-3) / 1e12
    return {"kernel": "NetVLAD launch sequence (one launch per MobileNetV2 block: nv_fpair_kernel, nv_pblock_kernel (stride 1), nv_xblock_kernel (stride 2), nv_slab_sum_kernel, nv_tail_kernel, "
                      "nv_vlad_* x2; d2slam_amd/csrc/netvlad*.hip): MobileNetV2-%.2f trunk + NetVLAD head" % (NV_MULT if abs(flop_per_img - NV_FLOP_PER_IMG) < 1 else -1),
            "bound": "mfma", "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4), "traffic": None,
            "ms_per_call": round(t_ms, 4), "images_per_call": F, "algorithmic_flop_per_call": flop_per_img * F,
            "note": "fp32 MFMA (v_mfma_f32_16x16x4_f32) + VALU depthwise; instruction/latency-bound small layers (DESIGN.md section 4); " + how}


def stream_classes(r):
    """compact form of a run's d2fe_pipe_stream_placement for the batch curve: 'n: own/second own/second ...' (n = classes told apart, 0 = not measured)"""
    p = r.get("stream_placement") or {}
    return "%s: %s" % (p.get("classes_told_apart"), " ".join("%d/%d" % (a, b) for a, b in p.get("lanes") or []))


def run_pipe(torch, api, weights, nv_weights, precision, F, lanes, steps, warmup, local_rank, rank, netvlad=True, light=False, coalesce=1, depth=0, inflight=0,
             world=1, dist=None, exchange=None, nv_flop_per_img=NV_FLOP_PER_IMG, nv_group=1, loopback=False, exchange_impl="capi", exchange_own_stream=True):
    """`steps` submits of F stereo frames through the frames-in-flight pipe with `lanes` submits in flight: the timed region holds, per submit, the H2D of
    the 2F frames from pinned memory, SuperPoint on them, NetVLAD of the F left images, ONE matcher launch (L<->R, L<->previous L) and the D2H of every
    result into pinned memory.  EVERY --gpus N runs this function (N = 1: no process group, no barrier).  N > 1 with `exchange`: one cross-agent exchange
    (d2slam_amd.swarm.PipeExchange: pack -> ONE all-gather -> gate -> remote matches -> D2H) per submit on a stream of its own, enqueued one submit behind the
    pipe and collected with the ticket -- inside the timed region, beside the lanes' work.  Timing: barrier + device synchronisation on both sides (N > 1),
    perf_counter around exactly `steps` submits + the waits for all of them; the caller takes the MAX over ranks."""
    from d2slam_amd import swarm
    prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[precision]
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=1, precision=prec, device_id=local_rank))
    fe.load_superpoint(weights)
    if netvlad:
        fe.load_netvlad(nv_weights)
    host = torch.from_numpy(pipe_frames(F, rank)).pin_memory()
    pipe = api.StereoPipe(fe, lanes=lanes, frames=F, width=W, height=H, cap=CAP, netvlad=netvlad, ratio=0.8, pinned_input=True, coalesce=coalesce, coalesce_depth=depth, netvlad_group=nv_group)
    inflight = inflight or lanes * coalesce
    base, per_set, per_side = host.data_ptr(), 2 * F * H * W, F * H * W
    dev = torch.device("cuda", local_rank)
    NS = inflight + 3
    xch = None
    ximpl = None
    if (world > 1 or loopback) and exchange:
        G = fe.netvlad_dim if netvlad else 0
        if exchange_impl == "capi":
            try:
                xch = swarm.PipeExchange(torch, fe, pipe, dev, world, rank, F, CAP, G, exchange=exchange, gate_thres=NETVLAD_GATE, ratio=0.8, slots=NS, loopback=loopback,
                                         own_stream=exchange_own_stream)
                ximpl = "capi: d2fe_exchange_* (csrc/exchange.hip), queued on %s; collective = %s" % ("ONE stream of its own" if exchange_own_stream else "the producing lane's stream",
                    "ncclAllGather on the library's own RCCL communicator (%s)" % api.load_library().d2fe_rccl_path().decode() if xch.backend == "nccl" else "host-staged callback (%s)" % xch.backend)
            except Exception as e:      # noqa: BLE001 -- e.g. no loadable librccl: the torch.distributed form still runs (every rank decides alike: same library, same box)
                ximpl = "torch (the C exchange could not be created: %s)" % str(e)[:160]
        if xch is None:
            xch = swarm.TorchPipeExchange(torch, fe, pipe, dev, world, rank, F, CAP, G, exchange=exchange, gate_thres=NETVLAD_GATE, ratio=0.8, slots=NS, loopback=loopback)
            ximpl = ximpl or "torch: Python-driven sequence on a stream of its own, torch.distributed collective"

    def submit(i):
        o = base + (i & 1) * per_set
        return pipe.submit_ptr(o, o + per_side)

    use_dist = dist is not None and dist.is_initialized()      # N > 1, or one rank sent through the N > 1 path (--force-dist)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    last = {}

    def drive(n, start):
        tk = []
        th = 0.0
        enq = [0]

        def enq_upto(j):          # the exchange of tickets <= j is queued (one submit behind the pipe: see PipeExchange)
            while xch and enq[0] <= j:
                xch.enqueue(tk[enq[0]], enq[0] % NS); enq[0] += 1

        def finish(j):            # host results of ticket j: the pipe's block now, the cross-agent lists of ticket j - 1 (N > 1)
            pipe.wait_raw(tk[j])
            if xch:
                # one submit of slack between a frame's own results and its cross-agent results: the all-gather of step j completes when the SLOWEST rank has
                # extracted step j, and the ranks are not in lock step (on one GPU under gloo they even alternate)
                enq_upto(j)
                if j >= 1:
                    last["x"] = xch.collect((j - 1) % NS)
                if j == n - 1:
                    last["x"] = xch.collect(j % NS)
        for i in range(n):
            if i >= inflight:
                finish(i - inflight)
            ta = time.perf_counter(); tk.append(submit(start + i)); th += time.perf_counter() - ta
            enq_upto(i - 1)
        for j in range(max(0, n - inflight), n):
            finish(j)
        return tk, th
    warmup = max(warmup, 2)
    warmup += warmup & 1                       # an even number of submits: the timed region starts on frame set 0
    drive(warmup, 0)
    barrier()
    if xch:
        xch.timeline.clear()
    if not light:
        pipe.profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    tk, th = drive(steps, 0)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = pipe.profile_read() if not light else None
    if not light:
        pipe.profile_enable(0)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    fb = fe.match_fallback_rows(reset=True, full=True)
    NI, NP = 2 * F, 2 * F + (xch.NR if xch else 0)
    res = dict(steps=steps, value=F * world * steps / elapsed, ms_per_step=elapsed / steps * 1e3, host_submit_ms=th / steps * 1e3, lanes=lanes, F=F, NI=NI, NP=NP, gated=None, exch=None,
               breakdown=None, fallback_rows=(fb[0] / float(steps + warmup), fb[1] / float(steps + warmup)), roofline=None, roofline_nv=None)
    if xch:
        S = last["x"]
        res["exch"] = {"impl": ximpl, "wire_precision": exchange, "block_bytes": xch.block_bytes, "all_gather_bytes_received_per_step_per_gpu": xch.block_bytes * F * (world - 1 + (1 if loopback else 0)),
                       "cross_agent_pairs_per_step_per_gpu": xch.NR, "avg_cross_agent_matches_per_pair": round(float(S["mn"].float().mean()), 2),
                       "d2h_bytes_per_step": xch.d2h_bytes, "enqueued": "one submit behind the pipe; collected with the ticket",
                       "step_timeline_ms": dict(xch.timeline_ms() or {}, note="rank 0, medians over the timed submits, HIP events on the exchange stream (which shares the device "
                                                "with the lanes' launches: an entry is the wall time of that phase beside them); backend %s" % dist.get_backend())}
        if netvlad:
            res["gated"] = {"pairs": xch.NR, "passing_netvlad_gate": int(S["gate_n"][0]), "threshold": NETVLAD_GATE}
    if not light:
        # one more submit of frame set 0 right behind one of set 1: what this mode selected and matched (parity / mode comparison)
        tl = [submit(1), submit(0)]
        if xch:
            for j, t in enumerate(tl):
                xch.enqueue(t, j)
        pipe.wait_raw(tl[0])
        o = pipe.wait(tl[1])
        if xch:
            xch.collect(0); xch.collect(1)
        cnt = o["n_kp"].copy()
        kidx = (o["kps_xy"][:, :, 1].astype(np.int64) * W + o["kps_xy"][:, :, 0].astype(np.int64)).astype(np.int32)
        k0 = int(cnt[0])
        res["first"] = (o["kps_xy"][0, :k0].copy(), o["scores"][0, :k0].copy(), o["desc"][0, :k0].copy())
        res["gfirst"] = o["netvlad"][0].copy() if netvlad else None
        res["sel"] = {"kidx": kidx, "cnt": cnt, "mq": np.concatenate([o["lr_q"], o["prev_q"]]).copy(), "mt": np.concatenate([o["lr_t"], o["prev_t"]]).copy(),
                      "mn": np.concatenate([o["lr_n"], o["prev_n"]]).copy(), "a_row": list(range(F)) + list(range(F)),
                      "b_row": [F + f for f in range(F)] + [None] + list(range(F - 1))}
        res["n_kp"] = float(cnt.mean()); res["n_match"] = float(res["sel"]["mn"].mean())
        res["d2h_bytes"] = int(4 * (NI * CAP * 259 + F * (fe.netvlad_dim if netvlad else 0) + NI + 2 * F + 3 * 2 * F * CAP)) + (xch.d2h_bytes if xch else 0)
        c1b_ms, c1b_n = prof["conv1b"]
        res["roofline"] = conv1b_roofline(precision, c1b_ms / max(c1b_n, 1), c1b_n, NI, True)
        nv_ms, nv_n = prof["netvlad"]
        if netvlad and nv_n:
            res["roofline_nv"] = netvlad_roofline(nv_ms / nv_n, F, "HIP events around the whole sequence where the pipe queued it (netvlad_inline = auto: the lane's second stream beside that lane's SuperPoint, or the "
                                                  "lane's own stream in front of it), with the other lanes' full-device launches on the chip: the figure is the sequence's WALL time in the "
                                                  "running pipe (waits for compute units included), not its cost -- that is `roofline_netvlad` of the full line (the sequence alone)", nv_flop_per_img)
    pl, ncl = pipe.stream_placement()
    res["stream_placement"] = {"classes_told_apart": ncl, "lanes": pl, "exchange_stream_class": None}
    if xch:
        if getattr(xch, "stream", None) is not None:
            try:
                res["stream_placement"]["exchange_stream_class"] = pipe.classify_stream(xch.stream.cuda_stream)
            except Exception as e:      # not idle (should not happen here: every ticket has been waited for)
                res["stream_placement"]["exchange_stream_class"] = str(e)[:80]
        else:
            res["stream_placement"]["exchange_stream_class"] = "none: the exchange runs on the lanes' own streams"
        xch.close()
    pipe.close(); fe.close()
    return res


def mode_disagreement(a, b, F):
    """MEASURED difference between the headline mode (Winograd fp32, `a`) and the bitwise-exact direct-convolution mode (`b`) on the very
    frames the bench times: keypoints that one mode selects and the other does not (raster indices, per image), and matches
    (as pairs of raster indices, so independent of the order inside a keypoint list) that one mode reports and the other does not.
    Both modes are fp32 evaluations of the same network; they can only differ where two scores are closer than their ~1e-6 round-off."""
    NI = 2 * F
    kp_tot = kp_diff = img_diff = 0
    for i in range(NI):
        sa = set(a["kidx"][i, :a["cnt"][i]].tolist()); sb = set(b["kidx"][i, :b["cnt"][i]].tolist())
        d = len(sa ^ sb)
        kp_tot += len(sb); kp_diff += d; img_diff += d > 0
    m_tot = m_diff = lr_tot = lr_diff = 0
    for p in range(len(a["mn"])):
        ia, ib = a["a_row"][p], a["b_row"][p]        # rows of the count / raster-index arrays; [2F, 3F) = the previous step's left images
        if ib is None:                               # pipe: frame 0's temporal partner lives in the previous submit's block
            continue

        def pairs(m):
            n = int(m["mn"][p])
            return {(int(m["kidx"][ia][q]), int(m["kidx"][ib][t])) for q, t in zip(m["mq"][p, :n].tolist(), m["mt"][p, :n].tolist())}
        pa, pb = pairs(a), pairs(b)
        d = len(pa ^ pb)
        m_tot += len(pb); m_diff += d
        if ib < 2 * F:
            lr_tot += len(pb); lr_diff += d
    return {"images": NI, "keypoints_exact_mode": kp_tot, "keypoints_in_one_mode_only": kp_diff, "images_with_any_keypoint_difference": int(img_diff),
            "match_pairs": len(a["mn"]), "matches_exact_mode": m_tot, "matches_in_one_mode_only": m_diff,
            "left_right_matches_exact_mode": lr_tot, "left_right_matches_in_one_mode_only": lr_diff,
            "note": "symmetric differences over the last step's frames (current L, R and the previous step's L); seeded random-init weights compress the score distribution, so near-ties at the top-K cut are far "
                    "more frequent than with a trained network (DESIGN.md section 2)"}


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: the same command line as N ranks on this node (torch.distributed.run, rendezvous on
    127.0.0.1 and a free port).  The ranks' stdout/stderr pass through; rank 0 prints the JSON line."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks inherit this process's ORIGINAL stdout as their fd 1 (this process's own fd 1 already points at stderr, see claim_stdout)
    return subprocess.call(cmd, env=env, stdout=_REAL_STDOUT if _REAL_STDOUT is not None else None)


def collective_evidence(torch, dist, dev, backend, rank, world):
    """What the N>1 record needs to prove N ranks on N devices: the backend and world size as the process group reports them, every
    rank's device identity gathered with all_gather_object, and one all-reduce over the group on the device (sum of ranks)."""
    p = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "pid": os.getpid(), "device_index": dev.index, "name": p.name,
            "uuid": str(getattr(p, "uuid", "")), "pci_bus_id": getattr(p, "pci_bus_id", None), "pci_device_id": getattr(p, "pci_device_id", None),
            "cus": p.multi_processor_count}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    t = torch.tensor([float(rank)], device=dev)
    dist.all_reduce(t)
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    ids = {(r["uuid"], r["pci_bus_id"], r["device_index"]) for r in allr}
    return {"backend": dist.get_backend(), "is_rccl": dist.get_backend() == "nccl", "rccl_version": ver, "world_size": dist.get_world_size(),
            "allreduce_sum_of_ranks": float(t.item()), "expected_sum": float(world * (world - 1) // 2),
            "distinct_devices": len(ids), "ranks": allr}


def index_parity_evidence():
    """The committed run of tools/mode_disagreement.py (profiles/r04_mode_disagreement.json): keypoint / match index differences of the Winograd and
    fp16 hi/lo modes against the exact fp32 mode over 1056 images (992 synthetic + 64 derived from the real crops of the reference's sample image) at
    N = 100 / 150 / 200 and thresholds 0.015 / 0.15.  Summarised here as the worst rate over the six configurations; the bench frames of THIS run are
    compared live in `wino_vs_exact_on_bench_frames` / `f16x2_vs_exact_on_bench_frames`."""
    name = next((n for n in ("r05_mode_disagreement.json", "r04_mode_disagreement.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if name is None:
        return None
    j = json.load(open(os.path.join(ROOT, "profiles", name)))
    out = {"source": "profiles/" + name + " (python tools/mode_disagreement.py on MI355X, same kernels; the FULL study is not collected inside this run -- its 128-image subset is: "
                     "`index_parity_in_run`)",
           "images": j["images"], "real_derived_images": j["real_derived_images"], "pairs": j["pairs"], "configs": "N in {100, 150, 200} x threshold in {0.015, 0.15}"}
    for m in ("wino", "f16x2"):
        rows = [c["%s_vs_f32_all" % m] for c in j["configs"]]
        out[m + "_vs_f32"] = {"keypoints_compared": sum(r["keypoints"] for r in rows), "keypoints_in_one_mode_only": sum(r["keypoints_in_one_mode_only"] for r in rows),
                              "worst_per_1e4_keypoints": max(r["per_1e4_keypoints"] for r in rows),
                              "matches_compared": sum(r["matches"] for r in rows), "matches_in_one_mode_only": sum(r["matches_in_one_mode_only"] for r in rows),
                              "worst_per_1e4_matches": max(r["per_1e4_matches"] for r in rows)}
    return out


def profiled_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes of this command (profiles/r05_wino_rocprofv3_summary.txt, else round 4's:
    separate --pmc FETCH_SIZE and WRITE_SIZE passes, tools/profile.sh; KiB per dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950's wide reads).  bench.py itself does not collect counters: null when the file is absent."""
    name = next((n for n in ("r06_wino_rocprofv3_summary.txt", "r05_wino_rocprofv3_summary.txt", "r04_wino_rocprofv3_summary.txt") if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
    if name is None:
        return None, None
    path = os.path.join(ROOT, "profiles", name)
    fetch = write = None
    lines = open(path).read().split("\n")
    sect = ""
    for i, l in enumerate(lines):
        if l.startswith("== "):
            sect = l
        if kernel_tag in l and i + 1 < len(lines):
            nxt = lines[i + 1]
            if "pmc_fetch" in sect and "FETCH_SIZE=" in nxt:
                fetch = float(nxt.split("FETCH_SIZE=")[1].split()[0])
            if "pmc_write" in sect and "WRITE_SIZE=" in nxt:
                write = float(nxt.split("WRITE_SIZE=")[1].split()[0])
    if fetch is None or write is None:
        return None, None
    return int((2.0 * fetch + write) * 1024), "profiles/" + name + ": 2 x FETCH_SIZE %.4g KiB + WRITE_SIZE %.4g KiB per dispatch (rocprofv3 --pmc passes of `python bench.py`, same build; not collected inside this run)" % (fetch, write)


def live_traffic(kernel_tag):
    """roofline.traffic measured by THIS run: two child passes of this script under `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, then WRITE_SIZE --
    separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; KiB per dispatch; FETCH_SIZE doubled for gfx950's wide reads), the headline step with ONE
    submit in flight and nothing else (--single-mode), averaged over the dispatches of the dominant kernel.  None when rocprofv3 is not there or a pass fails
    (the committed profile is then quoted, see profiled_traffic)."""
    import csv, glob, shutil, subprocess, tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp) or os.environ.get("D2FE_BENCH_CHILD"):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None              # this run is itself being profiled: no nested profiler
    vals, t0 = {}, time.time()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="d2fe_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, D2FE_BENCH_CHILD="1", TMPDIR="/tmp")
            cmd = [rp, "--output-format", "csv", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "3", "--warmup", "1", "--precision", "wino", "--single-mode", "--no-cpu-baseline", "--lanes", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return None
            acc = n = 0
            for row in csv.DictReader(open(fs[0])):
                if kernel_tag in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    acc += float(row["Counter_Value"]); n += 1
            if not n:
                return None
            vals[ctr] = (acc / n, n)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    return {"traffic": int((2.0 * fetch + write) * 1024),
            "counters": {"FETCH_SIZE_KiB_per_dispatch": round(fetch, 1), "WRITE_SIZE_KiB_per_dispatch": round(write, 1), "dispatches_averaged": [vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]],
                         "seconds": round(time.time() - t0, 1)},
            "note": "measured by this run: two child passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --single-mode --lanes 1 --steps 3` (same build, same "
                    "step, one submit in flight); traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB per dispatch of the dominant kernel (the doubling: gfx950's 128-byte reads, MI355X_MICROARCH.md)"}


def conv1b_roofline(precision, avg_ms, launches, NI, fused):
    peak = PEAK_TFLOPS[precision]
    alg = CONV1B_FLOP_PER_IMG * NI
    if precision == "wino":
        # Winograd F(2x2,3x3): 16 multiply-adds per output and channel pair where the direct convolution has 36.  The launch is the
        # conv1a-fused kernel (D2FE_FUSE1A default): per 8x16-pixel work item 1024 conv1b MFMAs + 60 conv1a MFMAs (v_mfma_f32_32x32x2_f32).
        items = NI * (H // 8) * (W // 16)
        executed = items * (1024 + (60 if fused is not False else 0)) * 4096.0
        ach_e = executed / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        ach_a = alg / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, tnote = profiled_traffic("conv_wino_kernel<64, true, true, 0, 1, true") if (fused is not False and NI == 64) else (None, None)
        return {"kernel": "conv_wino_kernel<64,POOL,RELU,FUSE> (conv1a from the u8 frame fused into conv1b as Winograd F(2x2,3x3), + ReLU + 2x2 max-pool)",
                "bound": "mfma", "achieved": round(ach_e, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach_e / peak, 4),
                "frac_executed": round(ach_e / peak, 4), "frac_algorithmic": round(ach_a / peak, 4),
                "achieved_algorithmic": round(ach_a, 2),
                "traffic": traffic, "traffic_note": tnote or "HBM bytes per launch are in profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes); not collected inside this run",
                "compulsory_bytes_per_launch": int(NI * (H * W + (H // 2) * (W // 2) * 64 * 4)),
                "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                "algorithmic_flop_per_launch": alg, "executed_mfma_flop_per_launch": executed,
                "note": "frac = frac_executed = MFMA FLOPs the kernel executes / HIP-event time / 157.3 TF (the matrix pipe's roofline fraction); "
                        "frac_algorithmic = SURVEY section 8(d)'s direct-convolution FLOPs / time / peak, above 1 because Winograd executes 16/36 of them"}
    ach = alg / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    ex = 3.0 if precision == "f16x2" else 1.0
    return {"kernel": "conv_%s_kernel<64,3,4,32,2,2,2,1,POOL,RELU,FUSE1A> (conv1a fused into conv1b)" % ("f32" if precision == "f32" else "f16x2"),
            "bound": "mfma", "achieved": round(ach * ex, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach * ex / peak, 4),
            "frac_executed": round(ach * ex / peak, 4), "frac_algorithmic": round(ach / peak, 4), "achieved_algorithmic": round(ach, 2),
            "traffic": None, "traffic_note": "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): 22.2 MB per image, 19.7 MB of it the pooled output",
            "avg_launch_ms": round(avg_ms, 4), "launches": launches, "algorithmic_flop_per_launch": alg,
            "executed_mfma_flop_per_launch": alg * ex,
            "note": "f16x2 executes 3 MFMA FLOPs (hi*hi + hi*lo + lo*hi) per algorithmic FLOP" if precision == "f16x2" else "one MFMA FLOP per algorithmic FLOP"}


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


CPU_WARMUP = 5          # warm-up 5, >= 50 timed iterations, median + p95


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
