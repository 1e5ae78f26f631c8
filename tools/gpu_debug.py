"""One-shot GPU bring-up report: per-layer diffs of the HIP path against the oracle, both precisions.
Run on the GPU box:  python tools/gpu_debug.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from d2slam_amd import api
from d2slam_amd.weights import synthetic_superpoint_weights
from d2slam_amd.synth import synth_image, synth_descriptor_pair
from oracle import oracle as orc

def report(name, got, ref):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    eq = np.array_equal(got, ref)
    print("  %-10s shape %-20s bitwise=%s maxabs=%.3e refmax=%.3e nbad=%d" % (name, got.shape, eq, d.max(), np.abs(ref).max(), int((got != ref).sum())), flush=True)

def run(H, W, prec, n=2):
    print("== size %dx%d precision %d" % (H, W, prec), flush=True)
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    imgs = np.stack([synth_image(H, W, s) for s in range(n)])
    cfg = api.SuperPointConfig(max_keypoints=200, input_width=W, input_height=H, max_batch=n, precision=prec, keep_score_map=True)
    fe = api.DevFrontEnd(cfg); fe.load_superpoint(w)
    t = time.time(); res = fe.extract_batch(imgs, cap=200); print("  extract wall %.3fs" % (time.time() - t))
    refs = [orc.superpoint_forward(imgs[i], w, return_trunk=True) for i in range(n)]
    Hc, Wc = H // 8, W // 8
    # layer-by-layer (oracle recomputed per layer here)
    x = [orc.prep_u8(imgs[i])[:, :, None] for i in range(n)]
    def layer(x, name, pool=False):
        y = [orc.conv(xx, *w[name], True) for xx in x]
        return [orc.maxpool2(yy) for yy in y] if pool else y
    l1a = layer(x, "conv1a"); report("conv1a", fe.debug_read("conv1a", (n, H, W, 64)), np.stack(l1a))
    l1b = layer(l1a, "conv1b", True); report("conv1b", fe.debug_read("conv1b", (n, H//2, W//2, 64)), np.stack(l1b))
    l2a = layer(l1b, "conv2a"); report("conv2a", fe.debug_read("conv2a", (n, H//2, W//2, 64)), np.stack(l2a))
    l2b = layer(l2a, "conv2b", True); report("conv2b", fe.debug_read("conv2b", (n, H//4, W//4, 64)), np.stack(l2b))
    l3a = layer(l2b, "conv3a"); report("conv3a", fe.debug_read("conv3a", (n, H//4, W//4, 128)), np.stack(l3a))
    l3b = layer(l3a, "conv3b", True); report("conv3b", fe.debug_read("conv3b", (n, Hc, Wc, 128)), np.stack(l3b))
    l4a = layer(l3b, "conv4a"); report("conv4a", fe.debug_read("conv4a", (n, Hc, Wc, 128)), np.stack(l4a))
    l4b = layer(l4a, "conv4b"); report("conv4b", fe.debug_read("conv4b", (n, Hc, Wc, 128)), np.stack(l4b))
    report("logits", fe.debug_read("logits", (n, Hc, Wc, 65)), np.stack([r["logits"] for r in refs]))
    report("desc_raw", fe.debug_read("desc_raw", (n, Hc, Wc, 256)), np.stack([r["desc_raw"] for r in refs]))
    report("semi", fe.debug_read("semi", (n, H, W)), np.stack([r["semi"] for r in refs]))
    for i in range(n):
        kps, sc, desc = res[i]
        rk, rs, ridx = orc.select_b(refs[i]["semi"], 0.015, 1, 200)
        rd = orc.sample_b(refs[i]["desc"], rk)
        print("  img %d: n=%d ref n=%d kps_equal=%s scores_equal=%s" % (i, len(kps), len(rk), np.array_equal(kps, rk), np.array_equal(sc, rs)))
        if len(kps) == len(rk) and len(rk):
            print("     desc maxabs diff %.3e  (kps mismatches %d)" % (np.abs(desc - rd).max(), int((kps != rk).any(axis=1).sum())))
    fe.close()

def run_match():
    print("== matcher", flush=True)
    cfg = api.SuperPointConfig(max_keypoints=100, input_width=64, input_height=64, max_batch=1)
    fe = api.DevFrontEnd(cfg)
    for (na, nb, dim, ratio, radius, sigma) in [(200, 200, 256, 0.8, -1, 0.05), (150, 97, 256, 0.7, 30.0, 0.2), (33, 200, 64, 0.9, -1, 0.05),
                                                 (1, 5, 256, 0.8, -1, 0.05), (5, 1, 256, 0.8, -1, 0.05), (2, 2, 256, 0.8, -1, 0.05)]:
        a, b, pa, pb = synth_descriptor_pair(na, nb, dim, seed=na * 7 + nb, sigma=sigma)
        q, t, d = fe.match_knn(a, b, ratio, pa, pb, radius)
        rq, rt, rd = orc.match_knn(a, b, ratio, pa, pb, radius)
        print("  knn na=%d nb=%d dim=%d: n=%d ref=%d idx_equal=%s dist_equal=%s" % (na, nb, dim, len(q), len(rq), np.array_equal(q, rq) and np.array_equal(t, rt), np.array_equal(d, rd)))
        q, t, d = fe.match_crosscheck(a, b)
        rq, rt, rd = orc.match_crosscheck(a, b)
        print("  xchk na=%d nb=%d: n=%d ref=%d idx_equal=%s dist_equal=%s" % (na, nb, len(q), len(rq), np.array_equal(q, rq) and np.array_equal(t, rt), np.array_equal(d, rd)))
    fe.close()

if __name__ == "__main__":
    print(api.load_library().d2fe_version().decode())
    run_match()
    run(96, 128, 0)
    run(104, 136, 0)
    run(96, 128, 1)
    run(104, 136, 1)
