"""SuperPoint weight containers and loaders.

The reference loads `superpoint_v1_sim_int32.onnx` (config/quadcam/quadcam_single.yaml:106-115) which is
not in the tree (.MISSING_LARGE_BLOBS).  The layer set is pinned by d2frontend/superpoint.ipynb:306-321.
Weights are held as {layer: (W [cout,cin,k,k] f32, b [cout] f32)} in PyTorch state_dict layout.
"""
import numpy as np

SP_LAYERS = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b",
             "convPa", "convPb", "convDa", "convDb"]
SP_SHAPES = {
    "conv1a": (64, 1, 3), "conv1b": (64, 64, 3), "conv2a": (64, 64, 3), "conv2b": (64, 64, 3),
    "conv3a": (128, 64, 3), "conv3b": (128, 128, 3), "conv4a": (128, 128, 3), "conv4b": (128, 128, 3),
    "convPa": (256, 128, 3), "convPb": (65, 256, 1), "convDa": (256, 128, 3), "convDb": (256, 256, 1),
}


def synthetic_superpoint_weights(seed=1234, dustbin_bias=2.5):
    """Seeded random-init weights of the SuperPoint architecture (He-uniform convs, small biases).
    `dustbin_bias` raises channel 64 of convPb so only a few % of pixels pass the 0.015 threshold,
    like the trained network."""
    rng = np.random.RandomState(seed)
    w = {}
    for name in SP_LAYERS:
        cout, cin, k = SP_SHAPES[name]
        bound = np.sqrt(6.0 / (cin * k * k))
        W = rng.uniform(-bound, bound, size=(cout, cin, k, k)).astype(np.float32)
        b = rng.uniform(-0.05, 0.05, size=(cout,)).astype(np.float32)
        w[name] = (W, b)
    W, b = w["convPb"]
    b = b.copy(); b[64] += np.float32(dustbin_bias)
    w["convPb"] = (W, b)
    return w


def load_superpoint_pth(path):
    """MagicLeap superpoint_v1.pth (state_dict keys conv1a.weight, ... ; superpoint.ipynb cell 3)."""
    import torch
    sd = torch.load(path, map_location="cpu")
    return {n: (sd[n + ".weight"].float().numpy().copy(), sd[n + ".bias"].float().numpy().copy())
            for n in SP_LAYERS}


def load_superpoint_npz(path):
    z = np.load(path)
    return {n: (z[n + ".weight"].astype(np.float32), z[n + ".bias"].astype(np.float32)) for n in SP_LAYERS}


def save_superpoint_npz(path, w):
    np.savez(path, **{n + ".weight": w[n][0] for n in SP_LAYERS}, **{n + ".bias": w[n][1] for n in SP_LAYERS})
