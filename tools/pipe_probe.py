#!/usr/bin/env python3
"""Throughput of the frames-in-flight pipe (d2fe_pipe_*): stereo fps for a sweep of lanes x frames-per-submit, pinned inputs, D2H of all results inside.
Usage: python tools/pipe_probe.py [--sweep "1x1,2x1,4x1,8x1,..."] [--precision wino] [--no-netvlad]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H, W, CAP = 480, 640, 200


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", default="1x1,2x1,3x1,4x1,6x1,8x1,12x1,1x2,2x2,4x2,6x2,1x4,2x4,4x4,1x8,2x8,3x8,1x16,2x16,1x32,2x32")
    ap.add_argument("--seconds", type=float, default=0.5)
    ap.add_argument("--precision", default="wino")
    ap.add_argument("--no-netvlad", action="store_true")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--nv-group", type=int, default=1, help="d2fe_pipe_config.netvlad_group")
    ap.add_argument("--dev", action="store_true", help="the development library (honours the D2FE_* schedule switches)")
    ap.add_argument("--lane-cus", type=int, default=0, help="d2fe_pipe_config.lane_cus")
    ap.add_argument("--coalesce", type=int, default=1, help="d2fe_pipe_config.coalesce (frames per submit must be 1)")
    ap.add_argument("--coalesce-depth", type=int, default=0, help="d2fe_pipe_config.coalesce_depth (dynamic batching)")
    ap.add_argument("--inflight", type=int, default=0, help="submits the caller keeps in flight (default: lanes x coalesce)")
    ap.add_argument("--nv-inline", action="store_true", help="d2fe_pipe_config.netvlad_inline = 1: NetVLAD on the lane stream, in front of SuperPoint (default: auto, decided per pass)")
    ap.add_argument("--nv-side", action="store_true", help="d2fe_pipe_config.netvlad_inline = 0: NetVLAD on a second stream per lane whatever the lane count")
    ap.add_argument("--partition", action="store_true", help="d2fe_pipe_config.cu_partition: disjoint compute units per lane")
    args = ap.parse_args()
    import torch
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights
    prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[args.precision]
    fe = (api.DevFrontEnd if args.dev else api.FrontEnd)(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=1, precision=prec))
    fe.load_superpoint(synthetic_superpoint_weights(dustbin_bias=7.5))
    if not args.no_netvlad:
        fe.load_netvlad(nvm.synthetic_netvlad_weights())
    scenes = [synth_stereo(H, W, seed=s) for s in range(8)]
    res = []
    for pt in args.sweep.split(","):
        K, F = (int(x) for x in pt.split("x"))
        # two alternating frame sets per lane slot (scene, then the scene after a small camera motion)
        NS = 2 * K
        host = torch.empty((NS, 2, F, H, W), dtype=torch.uint8).pin_memory()
        hn = host.numpy()
        for s in range(NS):
            for f in range(F):
                l, r = scenes[(s * F + f) % len(scenes)]
                sh = (s % 3, (2 * s) % 5)
                hn[s, 0, f] = np.roll(l, sh, (0, 1)); hn[s, 1, f] = np.roll(r, sh, (0, 1))
        pipe = api.StereoPipe(fe, lanes=K, frames=F, width=W, height=H, cap=CAP, netvlad=not args.no_netvlad, match_lr=not args.no_match, match_prev=not args.no_match, pinned_input=True, cu_partition=args.partition, netvlad_inline=(True if args.nv_inline else False if args.nv_side else None), coalesce=args.coalesce, lane_cus=args.lane_cus, netvlad_group=args.nv_group, coalesce_depth=args.coalesce_depth)
        base = host.data_ptr(); per = 2 * F * H * W
        def submit(i):
            s = i % NS
            return pipe.submit_ptr(base + s * per, base + s * per + F * H * W)
        KC = args.inflight or K * args.coalesce          # submits in flight = lanes x submits per pass
        tickets = [submit(i) for i in range(KC)]
        for i in range(KC, 3 * KC + 2):
            pipe.wait_raw(tickets[i - KC]); tickets.append(submit(i))
        for t in tickets[-KC:]:
            pipe.wait_raw(t)
        steps = max(2 * KC, int(args.seconds * 2200 / F))
        th = 0.0
        t0 = time.perf_counter()
        tickets = []
        for i in range(steps):
            if i >= KC:
                pipe.wait_raw(tickets[i - KC])
            ta = time.perf_counter(); tickets.append(submit(i)); th += time.perf_counter() - ta
        for t in tickets[-KC:]:
            r = pipe.wait_raw(t)
        dt = time.perf_counter() - t0
        o = pipe.wait(tickets[-1])
        rec = {"coalesce_depth": args.coalesce_depth, "inflight": KC, "nv_group": args.nv_group, "lane_cus": args.lane_cus, "coalesce": args.coalesce, "cu_partition": bool(args.partition), "nv_inline": bool(args.nv_inline), "hwq": os.environ.get("GPU_MAX_HW_QUEUES"), "lanes": K, "frames_per_submit": F, "stereo_fps": round(steps * F / dt, 1), "ms_per_submit": round(dt / steps * 1e3, 4),
               "host_submit_ms": round(th / steps * 1e3, 4), "avg_kp": float(o["n_kp"].mean()), "avg_lr": float(o["lr_n"].mean()) if o["lr_n"] is not None else None}
        pl, ncl = pipe.stream_placement()
        rec["stream_classes"] = {"n": ncl, "lanes": pl}
        print(json.dumps(rec), flush=True)
        res.append(rec)
        pipe.close()
    print(json.dumps({"pipe_probe": res, "env": {k: v for k, v in os.environ.items() if k.startswith(("D2FE_", "GPU_MAX", "HIP_"))}}))


if __name__ == "__main__":
    main()
