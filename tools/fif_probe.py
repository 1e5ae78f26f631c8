#!/usr/bin/env python3
"""Frames-in-flight probe (round 4): K lanes (one handle + one stream each), F stereo frames per lane step, round-robin
submission from ONE host thread; every lane step = H2D of 2F frames, NetVLAD(F left), SuperPoint(2F), matchKNN L<->R and L<->previous L
(the previous lane's left frames, ordered with events), ONE D2H of everything delivered.  Prints stereo fps for a sweep of (K, F).
Usage: python tools/fif_probe.py [--sweep "1x1,2x1,4x1,..."] [--steps 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W, CAP = 480, 640, 200


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", default="1x1,2x1,3x1,4x1,6x1,8x1,1x2,2x2,4x2,1x4,2x4,1x8,1x16,1x32")
    ap.add_argument("--steps", type=int, default=0, help="lane steps per point (0: ~0.4 s worth)")
    ap.add_argument("--precision", default="wino")
    ap.add_argument("--no-netvlad", action="store_true")
    ap.add_argument("--no-d2h", action="store_true")
    args = ap.parse_args()
    import torch
    from d2slam_amd import api, netvlad as nvm
    from d2slam_amd.synth import synth_stereo
    from d2slam_amd.weights import synthetic_superpoint_weights
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    weights = synthetic_superpoint_weights(dustbin_bias=7.5)
    nvw = nvm.synthetic_netvlad_weights()
    prec = {"f32": api.PREC_F32, "f16x2": api.PREC_F16X2, "wino": api.PREC_F32_WINO}[args.precision]
    frames = [synth_stereo(H, W, seed=s) for s in range(8)]
    res = []
    for pt in args.sweep.split(","):
        K, F = (int(x) for x in pt.split("x"))
        NI = 2 * F
        lanes = []
        for k in range(K):
            fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=CAP, input_width=W, input_height=H, max_batch=NI, precision=prec, device_id=0))
            fe.load_superpoint(weights)
            G = 0
            if not args.no_netvlad:
                fe.load_netvlad(nvw); G = fe.netvlad_dim
            host = np.empty((NI, H, W), np.uint8)
            for f in range(F):
                l, r = frames[(k * F + f) % len(frames)]
                host[f], host[F + f] = l, r
            L = dict(fe=fe, s=torch.cuda.Stream(device=dev), pin=torch.from_numpy(host).pin_memory(), img=torch.empty((NI, H, W), dtype=torch.uint8, device=dev))
            # one output block: desc[2F][CAP][256] | kps[2F][CAP][2] | scores[2F][CAP] | gdesc[F][G] | cnt[2F] | mq,mt,md [2F][CAP] | mn[2F]
            nw = NI * CAP * 256 + NI * CAP * 2 + NI * CAP + F * max(G, 4) + NI + 3 * NI * CAP + NI
            blk = torch.zeros(nw, dtype=torch.float32, device=dev)
            o = 0
            def take(n, dt=torch.float32):
                nonlocal o
                v = blk[o:o + n]; o += n
                return v.view(dt) if dt != torch.float32 else v
            L["desc"] = take(NI * CAP * 256); L["kps"] = take(NI * CAP * 2); L["scores"] = take(NI * CAP); L["gdesc"] = take(F * max(G, 4))
            L["cnt"] = take(NI, torch.int32); L["mq"] = take(NI * CAP, torch.int32); L["mt"] = take(NI * CAP, torch.int32); L["md"] = take(NI * CAP)
            L["mn"] = take(NI, torch.int32)
            L["blk"] = blk; L["hout"] = torch.empty(nw, dtype=torch.float32).pin_memory()
            L["kidx"] = torch.zeros((NI, CAP), dtype=torch.int32, device=dev)
            L["ev_ext"] = torch.cuda.Event(); L["ev_done"] = torch.cuda.Event(); L["ev_match"] = torch.cuda.Event()
            lanes.append(L)
        # pair arrays: addresses are absolute rows from lane's desc base is not possible across lanes (different allocations): use one pool
        # per lane for the a side and give the b side its own base pointer per launch: two launches (L<->R in-lane, L<->prevL cross-lane)
        for k, L in enumerate(lanes):
            P = lanes[(k - 1) % K]
            L["a_off"] = torch.tensor([f * CAP for f in range(F)], dtype=torch.int32, device=dev)
            L["b_off_r"] = torch.tensor([(F + f) * CAP for f in range(F)], dtype=torch.int32, device=dev)
            L["prev"] = P
        torch.cuda.synchronize()

        def submit(L):
            s = L["s"]; fe = L["fe"]; st = s.cuda_stream; P = L["prev"]
            with torch.cuda.stream(s):
                if K > 1:
                    s.wait_event(L["ev_match_next"]) if "ev_match_next" in L else None
                L["img"].copy_(L["pin"], non_blocking=True)
                if not args.no_netvlad:
                    fe.netvlad_device(L["img"].data_ptr(), F, W, H, L["gdesc"].data_ptr(), stream=st)
                fe.extract_device(L["img"].data_ptr(), NI, W, H, L["kps"].data_ptr(), L["scores"].data_ptr(), L["desc"].data_ptr(), L["kidx"].data_ptr(),
                                  CAP, L["cnt"].data_ptr(), stream=st)
                L["ev_ext"].record(s)
                fe.match_batch_device(L["desc"].data_ptr(), L["desc"].data_ptr(), L["a_off"].data_ptr(), L["b_off_r"].data_ptr(), L["cnt"].data_ptr(),
                                      L["cnt"].data_ptr() + 4 * F, F, 256, CAP, L["mq"].data_ptr(), L["mt"].data_ptr(), L["md"].data_ptr(), L["mn"].data_ptr(),
                                      stream=st)
                if P is not L:
                    s.wait_event(P["ev_ext"])
                fe.match_batch_device(L["desc"].data_ptr(), P["desc"].data_ptr(), L["a_off"].data_ptr(), L["a_off"].data_ptr(), L["cnt"].data_ptr(),
                                      P["cnt"].data_ptr(), F, 256, CAP, L["mq"].data_ptr() + 4 * F * CAP, L["mt"].data_ptr() + 4 * F * CAP,
                                      L["md"].data_ptr() + 4 * F * CAP, L["mn"].data_ptr() + 4 * F, stream=st)
                L["ev_match"].record(s)
                if not args.no_d2h:
                    L["hout"].copy_(L["blk"], non_blocking=True)
                L["ev_done"].record(s)
        for k, L in enumerate(lanes):
            if K > 1:
                L["ev_match_next"] = lanes[(k + 1) % K]["ev_match"]
        steps = args.steps or max(20, int(800 / F))
        for i in range(3 * K):
            submit(lanes[i % K])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = 0.0
        for i in range(steps):
            L = lanes[i % K]
            L["ev_done"].synchronize()
            ta = time.perf_counter()
            submit(L)
            th += time.perf_counter() - ta
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fps = steps * F / dt
        nk = float(lanes[0]["cnt"].float().mean().item()); nm = float(lanes[0]["mn"].float().mean().item())
        r = {"lanes": K, "frames_per_lane_step": F, "stereo_fps": round(fps, 1), "ms_per_lane_step": round(dt / steps * 1e3, 4), "host_submit_ms": round(th / steps * 1e3, 4), "avg_kp": nk, "avg_matches": nm}
        print(json.dumps(r), flush=True)
        res.append(r)
        for L in lanes:
            L["fe"].close()
        del lanes
        torch.cuda.empty_cache()
    print(json.dumps({"fif_probe": res, "env": {k: v for k, v in os.environ.items() if k.startswith(("D2FE_", "GPU_MAX", "HIP_"))}}))


if __name__ == "__main__":
    main()
