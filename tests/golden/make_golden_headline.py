"""Reference-generated golden vectors at the BASELINE geometries (640x480 d435, 800x400 quadcam view, 512x512 TUM).

Same source of truth as make_golden_ref.py -- the two torch modules the reference defines in d2frontend/superpoint.ipynb, executed
verbatim from the notebook (nothing is copied into the repo), float64, seeded weights -- on crops of the only image the reference
ships (sample_data/fisheye.jpg).  To keep the fixture small (tests/golden/reference_headline.npz, < 1 MB) only strided sub-samples of
the dense outputs are stored: semi[::5, ::7], desc[::8 channels, ::3, ::3 cells], the full keypoint list of the notebook's
`semi > 0.2` rule and the sampled descriptors of every 32nd keypoint (the channel norms they carry depend on the WHOLE list).

Run from the repo root (needs /root/reference):  python tests/golden/make_golden_headline.py
"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import torch
from PIL import Image
from d2slam_amd.weights import SP_LAYERS, synthetic_superpoint_weights

NB = "/root/reference/d2frontend/superpoint.ipynb"
cells = ["".join(c["source"]) for c in json.load(open(NB))["cells"]]
ns = {"torch": torch}
exec(compile(cells[1], NB + ":cell1", "exec"), ns)
exec(compile(cells[5].split("# Input to the model")[0].replace("#Output to onnx.", ""), NB + ":cell5", "exec"), ns)

DUSTBIN, PB_SCALE = 0.0, 8.0        # as in make_golden_ref.py: detector 1x1 weights x 8 so that the notebook's `semi > 0.2` fires
w = synthetic_superpoint_weights(seed=1234, dustbin_bias=DUSTBIN)
w["convPb"] = (w["convPb"][0] * np.float32(PB_SCALE), w["convPb"][1])
sd = {}
for n in SP_LAYERS:
    sd[n + ".weight"] = torch.from_numpy(w[n][0]); sd[n + ".bias"] = torch.from_numpy(w[n][1])

real = np.asarray(Image.open("/root/reference/sample_data/fisheye.jpg").convert("L"))      # 800 x 1280
frames = {"d435": real[160:640, 320:960].copy(),          # 480 x 640
          "quad": real[200:600, 240:1040].copy(),         # 400 x 800
          "tum": real[144:656, 384:896].copy()}           # 512 x 512
out = {"dustbin_bias": np.float64(DUSTBIN), "pb_scale": np.float64(PB_SCALE), "seed": np.int64(1234),
       "semi_stride": np.array([5, 7]), "desc_stride": np.array([8, 3, 3]), "kdesc_stride": np.int64(32)}
torch.set_num_threads(8)
for tag, img in frames.items():
    out["img_" + tag] = img
    x64 = torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None, None].to(torch.float64)
    half = ns["SuperPointNetHalf"](); half.load_state_dict(sd); half = half.to(torch.float64).eval()
    with torch.no_grad():
        semi, desc = half(x64)
    out["semi_f64_" + tag] = semi[0].numpy()[::5, ::7].astype(np.float32)
    out["desc_f64_" + tag] = desc[0].numpy()[::8, ::3, ::3].astype(np.float32)
    full = ns["SuperPointNet"](); full.load_state_dict(sd); full = full.eval()
    with torch.no_grad():
        kps, kdesc = full(torch.from_numpy(img.astype(np.float32) / np.float32(255.0))[None, None])
    out["kps_f32_" + tag] = kps.numpy().astype(np.int16)                  # (row, col), raster order
    out["kdesc_f32_" + tag] = kdesc.numpy()[::32].astype(np.float32)      # [every 32nd keypoint, 256]
    # scores of the pixels nearest to the threshold (the only places where two fp32-accurate evaluations may list different pixels)
    s = semi[0].numpy()
    near = np.argwhere(np.abs(s - 0.2) < 5e-4)
    out["near_thr_" + tag] = near.astype(np.int16)
    print(tag, img.shape, "keypoints with semi > 0.2:", len(kps), "near-threshold pixels:", len(near))
np.savez_compressed(os.path.join(HERE, "reference_headline.npz"), **out)
print("wrote reference_headline.npz", os.path.getsize(os.path.join(HERE, "reference_headline.npz")))
