"""NetVLAD (A9) timing on the GPU: fused-block plan vs one launch per layer, HIP-event timed on the launch stream.
usage: python tools/bench_netvlad.py [n_images ...] [--mult=0.75] [--hw=480x640] [--fused-only]   (default 1 4 32)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from d2slam_amd import api, netvlad as nvm
from d2slam_amd.synth import synth_image
H, W = [int(v) for v in next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--hw=")), "480x640").split("x")]
ns = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 4, 32]
MULT = float(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--mult=")), os.environ.get("NV_MULT", "0.75")))
GF = nvm.arch_flops(MULT, H, W)
modes = ["fused"] if "--fused-only" in sys.argv else ["legacy", "fused"]
for mode in modes:
    os.environ["D2FE_NV_LEGACY"] = "1" if mode == "legacy" else "0"
    for n in ns:
        fe = api.DevFrontEnd(api.SuperPointConfig(input_width=W, input_height=H, max_batch=n))
        fe.load_netvlad(nvm.synthetic_netvlad_weights(depth_multiplier=MULT))
        dev = torch.device("cuda", 0)
        imgs = torch.from_numpy(np.stack([synth_image(H, W, s) for s in range(min(n, 4))] * ((n + 3) // 4))[:n].copy()).to(dev)
        out = torch.zeros((n, fe.netvlad_dim), device=dev)
        st = torch.cuda.Stream()
        def run():
            fe.netvlad_device(imgs.data_ptr(), n, W, H, out.data_ptr(), stream=st.cuda_stream)
        for _ in range(5): run()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(30): run()
            e1.record(st)
        torch.cuda.synchronize(); dt = e0.elapsed_time(e1) / 30
        gf = GF * n / (dt * 1e-3) / 1e12
        print("NetVLAD a=%.2f %-6s n=%2d: %.3f ms per call (%.4f ms per image, %.1f TFLOP/s)" % (MULT, mode, n, dt, dt / n, gf), flush=True)
        fe.close()
