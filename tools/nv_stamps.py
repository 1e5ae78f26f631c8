"""Phase timing inside one NetVLAD block kernel (nv_xblock_kernel) from the wall_clock64() stamps its workgroups write (D2FE_NV_STAMP_STEP).
usage: D2FE_NV_FLAGS=<f> python tools/nv_stamps.py <plan step> [n_images]      (run on the GPU box)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
step = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
os.environ["D2FE_NV_STAMP_STEP"] = str(step)
import numpy as np
from d2slam_amd import api, netvlad as nvm
from d2slam_amd.synth import synth_image
H, W = 480, 640
fe = api.DevFrontEnd(api.SuperPointConfig(input_width=W, input_height=H, max_batch=n))
fe.load_netvlad(nvm.synthetic_netvlad_weights())
imgs = np.stack([synth_image(H, W, s % 4) for s in range(n)])
for _ in range(3):
    fe.netvlad(imgs)
st = fe.debug_netvlad_stamps().astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
nst = int((st > 0).sum(axis=1).max())
n_all = len(st)
st = st[(st > 0).sum(axis=1) == nst]          # the hidden-channel groups with the most chunks
rel = (st[:, :nst] - t0) * 0.01           # us (100 MHz)
print("step %d: %d workgroups (%d with all %d stamps); kernel span (first start -> last end) %.2f us" % (step, n_all, len(st), nst, rel[:, nst - 1].max()))
print("workgroup start: p50 %.2f p90 %.2f max %.2f us after the first; lifetime p50 %.2f p90 %.2f max %.2f us" % (
    np.median(rel[:, 0]), np.percentile(rel[:, 0], 90), rel[:, 0].max(), np.median(rel[:, nst - 1] - rel[:, 0]), np.percentile(rel[:, nst - 1] - rel[:, 0], 90),
    (rel[:, nst - 1] - rel[:, 0]).max()))
names = ["start", "inputs issued / w0 arrived", "w0 stored", "barrier0"]
for c in range((nst - 5) // 5):
    names += ["ch%d expand done" % c, "ch%d we stored" % c, "ch%d barrier" % c, "ch%d dw+project done" % c, "ch%d wd stored" % c]
names += ["epilogue done"]
if nst == 7:      # the first block (nv_fpair_kernel)
    names = ["start", "loads arrived and stored", "barrier", "conv done", "barrier", "dw+project done", "epilogue done"]
d = np.diff(rel, axis=1)
for i in range(1, nst):
    print("  %-28s +%6.2f us (p10 %5.2f p90 %5.2f)   at %6.2f" % (names[i] if i < len(names) else "?", np.median(d[:, i - 1]), np.percentile(d[:, i - 1], 10),
                                                                 np.percentile(d[:, i - 1], 90), np.median(rel[:, i] - rel[:, 0])))
fe.close()
