#!/usr/bin/env python3
"""Phase timing INSIDE match_kernel from its wall_clock64() stamps (development build: python -m d2slam_amd.build --dev;
the development library is loaded through api.DevFrontEnd).  Usage: match_stamps.py [pairs ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["D2FE_MATCH_STAMPS"] = "1"
import torch
from d2slam_amd import api
from tools.bench_match import sets

NAMES = ["start->Q frags", "tiles (MFMA)", "barrier", "scan+enumerate", "exact", "merge+emit+fallback", "ticket", "finalize (last only)"]
fe = api.DevFrontEnd(api.SuperPointConfig(max_keypoints=200, input_width=64, input_height=64, max_batch=1))
lib = api.load_library(dev=True)
lib.d2fe_debug_match_stamps.restype = C.c_long
dev = torch.device("cuda", 0); n = 200
for P in [int(x) for x in (sys.argv[1:] or ["1", "64"])]:
    A = np.empty((P, n, 256), np.float32); B = np.empty((P, n, 256), np.float32)
    for p in range(P):
        A[p], B[p] = sets(n, p)
    pool = torch.from_numpy(np.concatenate([A.reshape(-1, 256), B.reshape(-1, 256)])).to(dev)
    a_off = torch.arange(P, dtype=torch.int32, device=dev) * n; b_off = a_off + P * n
    cnt = torch.full((P,), n, dtype=torch.int32, device=dev)
    mq = torch.zeros((P, n), dtype=torch.int32, device=dev); mt = torch.zeros_like(mq); md = torch.zeros((P, n), dtype=torch.float32, device=dev); mn = torch.zeros((P,), dtype=torch.int32, device=dev)
    for _ in range(4):
        fe.match_batch_device(pool.data_ptr(), pool.data_ptr(), a_off.data_ptr(), b_off.data_ptr(), cnt.data_ptr(), cnt.data_ptr(), P, 256, n,
                              mq.data_ptr(), mt.data_ptr(), md.data_ptr(), mn.data_ptr())
    torch.cuda.synchronize()
    nwg = 14 * P
    st = np.zeros((nwg, 16), np.uint64)
    got = lib.d2fe_debug_match_stamps(fe.handle, st.ctypes.data_as(C.c_void_p), C.c_long(nwg))
    st = st.astype(np.int64)
    act = st[:, 1] > 0
    t0 = st[:, 0].min()
    print("== %d pairs, %d workgroups (%d active); wall_clock64 at 100 MHz -> us" % (P, nwg, int(act.sum())))
    print("   kernel span (first start .. last stamp): %.2f us; start spread %.2f us" % ((st.max() - t0) / 100.0, (st[:, 0].max() - t0) / 100.0))
    d = np.diff(st[act, :8], axis=1) / 100.0
    for i, nm in enumerate(NAMES[:7]):
        print("   %-22s avg %6.2f  max %6.2f us" % (nm, d[:, i].mean(), d[:, i].max()))
    last = st[:, 8] > 0
    if last.any():
        print("   %-22s avg %6.2f  max %6.2f us" % (NAMES[7], ((st[last, 8] - st[last, 7]) / 100.0).mean(), ((st[last, 8] - st[last, 7]) / 100.0).max()))
    print("   per-workgroup total avg %.2f max %.2f us" % (((st[act, 1:9].max(axis=1) - st[act, 0]) / 100.0).mean(), ((st[act, 1:9].max(axis=1) - st[act, 0]) / 100.0).max()))
