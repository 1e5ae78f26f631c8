#!/usr/bin/env python3
"""CPU study (torch, no GPU): what would Winograd F(4x4,3x3) -- 2.25 instead of 4 multiplies per output of F(2x2,3x3), 9 of the direct form -- do to the index
parity of SuperPoint?  Every 3x3 layer with Cin >= 64 is evaluated in fp32 as (a) a direct convolution, (b) F(2x2,3x3), (c) F(4x4,3x3) (transforms and the channel
contraction in fp32, filter transform in fp64 then rounded, as the product does), and in fp64 as the ground truth; then the variant-B selection (threshold, border,
top-K) on each score map.  Reported per evaluation: max |score - truth|, keypoints in one of {evaluation, truth} only, per 1e4.
Usage: python tools/wino4_study.py [n_images=32] [H=480] [W=640] [K=200] [thr=0.015]"""
import json, os, sys, time
import numpy as np
import torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d2slam_amd.synth import synth_stereo
from d2slam_amd.weights import SP_LAYERS, synthetic_superpoint_weights

BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def wino_conv(x, w, b, m):
    """x [C,H,W] fp32, w [O,C,3,3], F(m x m, 3x3), pad 1; fp32 arithmetic"""
    BT, G, AT = (BT4, G4, AT4) if m == 4 else (BT2, G2, AT2)
    a = m + 2
    C, H, W = x.shape
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = Fn.pad(x, (1, tw * m - W + 1, 1, th * m - H + 1))
    d = xp.unfold(1, a, m).unfold(2, a, m)                      # [C, th, tw, a, a]
    Bt = torch.from_numpy(BT).float()
    V = torch.einsum('ij,ctujk,lk->cilt u'.replace(' ', ''), Bt, d, Bt) if False else torch.matmul(torch.matmul(Bt, d), Bt.t())   # [C, th, tw, a, a]
    V = V.permute(3, 4, 0, 1, 2).reshape(a * a, C, th * tw)
    U = torch.from_numpy(np.einsum('ij,ocjk,lk->ocil', G, w.double().numpy(), G)).float()     # [O, C, a, a], fp64 then rounded
    U = U.permute(2, 3, 0, 1).reshape(a * a, w.shape[0], C)
    M = torch.bmm(U, V).reshape(a, a, w.shape[0], th, tw).permute(2, 3, 4, 0, 1)               # [O, th, tw, a, a]
    At = torch.from_numpy(AT).float()
    Y = torch.matmul(torch.matmul(At, M), At.t())                                               # [O, th, tw, m, m]
    Y = Y.permute(0, 1, 3, 2, 4).reshape(w.shape[0], th * m, tw * m)[:, :H, :W]
    return Y + b[:, None, None]


def forward(img_u8, wts, mode, dtype):
    x = torch.from_numpy(img_u8.astype(np.float64) / 255.0).to(dtype)[None]
    def conv(x, name, relu=True, allow_wino=True):
        w, b = wts[name]
        w = torch.from_numpy(w).to(dtype); b = torch.from_numpy(b).to(dtype)
        if mode in (2, 4) and allow_wino and w.shape[2] == 3 and w.shape[1] >= 64:
            y = wino_conv(x, w, b, mode)
        else:
            y = Fn.conv2d(x[None], w, b, padding=w.shape[2] // 2)[0]
        return torch.relu(y) if relu else y
    x = conv(x, "conv1a"); x = conv(x, "conv1b"); x = Fn.max_pool2d(x[None], 2)[0]
    x = conv(x, "conv2a"); x = conv(x, "conv2b"); x = Fn.max_pool2d(x[None], 2)[0]
    x = conv(x, "conv3a"); x = conv(x, "conv3b"); x = Fn.max_pool2d(x[None], 2)[0]
    x = conv(x, "conv4a"); x = conv(x, "conv4b")
    p = conv(x, "convPa"); p = conv(p, "convPb", relu=False)
    s = torch.softmax(p, 0)[:64]
    Hc, Wc = s.shape[1], s.shape[2]
    return s.reshape(8, 8, Hc, Wc).permute(2, 0, 3, 1).reshape(Hc * 8, Wc * 8)


def select(score, K, thr, border=4):
    s = score.clone()
    H, W = s.shape
    m = torch.zeros_like(s, dtype=torch.bool); m[border:H - border, border:W - border] = True
    idx = torch.nonzero((s > thr) & m, as_tuple=False)
    v = s[idx[:, 0], idx[:, 1]]
    if len(v) > K:
        top = torch.topk(v, K).indices
        idx = idx[top]
    return set((idx[:, 0] * W + idx[:, 1]).tolist())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    thr = float(sys.argv[5]) if len(sys.argv) > 5 else 0.015
    torch.set_num_threads(os.cpu_count())
    wts = synthetic_superpoint_weights(dustbin_bias=7.5)
    tot = {m: {"kp": 0, "diff": 0, "max_err": 0.0} for m in ("direct_f32", "wino2_f32", "wino4_f32")}
    t0 = time.time()
    for i in range(n):
        l, r = synth_stereo(H, W, seed=5000 + i // 2)
        img = (l, r)[i & 1]
        truth = forward(img, wts, 0, torch.float64)
        kt = select(truth.float(), K, thr)
        for name, mode in (("direct_f32", 0), ("wino2_f32", 2), ("wino4_f32", 4)):
            s = forward(img, wts, mode, torch.float32)
            ks = select(s, K, thr)
            tot[name]["kp"] += len(kt); tot[name]["diff"] += len(kt ^ ks)
            tot[name]["max_err"] = max(tot[name]["max_err"], float((s.double() - truth).abs().max()))
        if i % 4 == 3:
            print("  %d images, %.0f s: %s" % (i + 1, time.time() - t0, {k: (v["diff"], "%.2e" % v["max_err"]) for k, v in tot.items()}), file=sys.stderr, flush=True)
    out = {"images": n, "H": H, "W": W, "K": K, "threshold": thr, "truth": "fp64 direct convolution (torch CPU)",
           "evaluations": {k: {"keypoints_truth": v["kp"], "keypoints_in_one_only": v["diff"], "per_1e4": round(1e4 * v["diff"] / max(v["kp"], 1), 3),
                               "max_abs_score_error": v["max_err"]} for k, v in tot.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
