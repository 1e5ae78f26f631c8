#!/usr/bin/env python3
"""Matcher timing (HIP events): d2fe_match_batch_device on P pairs of n x n x 256 descriptors that look like a stereo pair's (B = permuted A + noise + 20 %
outliers), and the host-pointer d2fe_match_knn.  Usage: python tools/bench_match.py [--pairs 64,2,1] [--n 200]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sets(n, seed, sigma=0.05):
    r = np.random.default_rng(seed)
    a = r.standard_normal((n, 256)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a[r.permutation(n)] + sigma * r.standard_normal((n, 256)).astype(np.float32)
    k = n // 5
    b[:k] = r.standard_normal((k, 256)).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    return a, b.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", default="64,8,2,1")
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    import torch
    from d2slam_amd import api
    fe = api.FrontEnd(api.SuperPointConfig(max_keypoints=args.n, input_width=64, input_height=64, max_batch=1))
    dev = torch.device("cuda", 0)
    n = args.n
    out = {}
    for P in [int(x) for x in args.pairs.split(",")]:
        A = np.empty((P, n, 256), np.float32); B = np.empty((P, n, 256), np.float32)
        for p in range(P):
            A[p], B[p] = sets(n, p)
        pool = torch.from_numpy(np.concatenate([A.reshape(-1, 256), B.reshape(-1, 256)])).to(dev)
        a_off = torch.arange(P, dtype=torch.int32, device=dev) * n
        b_off = a_off + P * n
        cnt = torch.full((P,), n, dtype=torch.int32, device=dev)
        mq = torch.zeros((P, n), dtype=torch.int32, device=dev); mt = torch.zeros_like(mq); md = torch.zeros((P, n), dtype=torch.float32, device=dev)
        mn = torch.zeros((P,), dtype=torch.int32, device=dev)
        s = torch.cuda.Stream(device=dev)
        def run():
            fe.match_batch_device(pool.data_ptr(), pool.data_ptr(), a_off.data_ptr(), b_off.data_ptr(), cnt.data_ptr(), cnt.data_ptr(), P, 256, n,
                                  mq.data_ptr(), mt.data_ptr(), md.data_ptr(), mn.data_ptr(), stream=s.cuda_stream)
        with torch.cuda.stream(s):
            for _ in range(5):
                run()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(args.iters):
                run()
            e1.record(s)
        s.synchronize()
        fb = fe.match_fallback_rows(reset=True, full=True)
        out["batch_%d_pairs" % P] = {"us_per_launch": round(e0.elapsed_time(e1) / args.iters * 1e3, 2), "avg_matches": round(float(mn.float().mean()), 1),
                                     "extra_candidates_per_launch": fb[0] / (args.iters + 5.0), "full_scans_per_launch": fb[1] / (args.iters + 5.0)}
    a, b = sets(n, 0)
    for _ in range(10):
        fe.match_knn(a, b)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); fe.match_knn(a, b); ts.append(time.perf_counter() - t0)
    ts.sort()
    out["host_match_knn_p50_us"] = round(ts[len(ts) // 2] * 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
