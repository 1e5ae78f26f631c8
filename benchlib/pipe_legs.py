"""The timed legs through the frames-in-flight pipe (every --gpus N), the N > 1 helpers and the cross-mode comparison."""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .rooflines import conv1b_roofline, netvlad_roofline
from .stdout_contract import real_stdout

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
           "--master-port", str(port), BENCH_PY] + sys.argv[1:]
    # the ranks inherit this process's ORIGINAL stdout as their fd 1 (this process's own fd 1 already points at stderr, see claim_stdout)
    return subprocess.call(cmd, env=env, stdout=real_stdout())


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
