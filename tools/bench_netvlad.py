import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from d2slam_amd import api, netvlad as nvm
from d2slam_amd.synth import synth_image
H, W = 480, 640
for n in (1, 4, 16):
    fe = api.FrontEnd(api.SuperPointConfig(input_width=W, input_height=H, max_batch=n))
    fe.load_netvlad(nvm.synthetic_netvlad_weights())
    dev = torch.device("cuda", 0)
    imgs = torch.from_numpy(np.stack([synth_image(H, W, s) for s in range(n)])).to(dev)
    out = torch.zeros((n, fe.netvlad_dim), device=dev)
    st = torch.cuda.Stream()
    def run():
        fe.netvlad_device(imgs.data_ptr(), n, W, H, out.data_ptr(), stream=st.cuda_stream)
    for _ in range(5): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30 * 1e3
    print("NetVLAD n=%d: %.3f ms per call (%.3f ms per image)" % (n, dt, dt / n))
    fe.close()
