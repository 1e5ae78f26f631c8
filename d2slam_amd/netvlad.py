"""MobileNetVLAD stand-in: architecture table, seeded weights, and the flat layer list handed to the C ABI.

The reference runs `mobilenetvlad_dyn_size.onnx` through ONNX Runtime (mobilenetvlad_onnx.h:18-74, tensor names
`image:0` -> `descriptor:0`, 4096-D, input gray float NOT scaled); that file is not in the tree and nothing in the tree
pins its graph.  The stand-in follows the HF-Net MobileNetVLAD lineage the tensor names point to:

  gray u8 -> (x-128)/128 -> MobileNetV2 (depth multiplier 0.75 -- HF-Net's width; rounds 2-5 used 0.35 -- channels rounded to multiples of 8, TF "SAME" padding,
  BatchNorm folded, ReLU6) up to the 1280-channel 1x1 conv ("layer_18", stride 32) -> 1x1 pre-projection to 128
  -> NetVLAD (K = 32 clusters, soft-assignment conv, residuals c_k - x, intra-normalisation, flatten, L2) -> 4096-D.

Layer kinds of the flat list: "conv" (3x3 full conv from the 1-channel image), "dw" (depthwise 3x3), "pw" (1x1).
`res` = index of the layer whose output is added to this layer's output (MobileNetV2 skip), or -1.
Weights layouts (PyTorch-like): conv [cout][1][3][3], dw [c][3][3], pw [cout][cin]; BatchNorm already folded.
"""
import numpy as np

DEPTH_MULTIPLIER = 0.75        # "MobileNetV2 (alpha = 0.75) trunk -> NetVLAD layer (K = 32 clusters) -> 4096" (HF-Net's MobileNetVLAD)
NETVLAD_K = 32
NETVLAD_D = 128
NETVLAD_DIM = NETVLAD_K * NETVLAD_D   # 4096, NETVLAD_DESC_RAW_SIZE (mobilenetvlad_onnx.h:5)
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2


def _div8(v, divisor=8, min_value=8):
    nv = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if nv < 0.9 * v:
        nv += divisor
    return nv


def mobilenetvlad_arch(depth_multiplier=DEPTH_MULTIPLIER):
    """Flat layer list of the trunk (MobileNetV2 table: t, c, n, s)."""
    L = []
    c_in = _div8(32 * depth_multiplier)
    L.append(dict(kind="conv", cin=1, cout=c_in, stride=2, act=ACT_RELU6, res=-1))
    blocks = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
    for t, c, n, s in blocks:
        c_out = _div8(c * depth_multiplier)
        for i in range(n):
            stride = s if i == 0 else 1
            block_in = len(L) - 1            # index of the layer producing this block's input
            hidden = c_in * t
            if t != 1:
                L.append(dict(kind="pw", cin=c_in, cout=hidden, stride=1, act=ACT_RELU6, res=-1))
            L.append(dict(kind="dw", cin=hidden, cout=hidden, stride=stride, act=ACT_RELU6, res=-1))
            L.append(dict(kind="pw", cin=hidden, cout=c_out, stride=1, act=ACT_NONE,
                          res=block_in if (stride == 1 and c_in == c_out) else -1))
            c_in = c_out
    L.append(dict(kind="pw", cin=c_in, cout=1280, stride=1, act=ACT_RELU6, res=-1))   # "layer_18"
    return L


def synthetic_netvlad_weights(seed=4321, depth_multiplier=DEPTH_MULTIPLIER):
    """Seeded random-init weights of the stand-in (He-uniform, small biases standing in for folded BatchNorm)."""
    rng = np.random.RandomState(seed)
    arch = mobilenetvlad_arch(depth_multiplier)
    layers = []
    for l in arch:
        if l["kind"] == "conv":
            fan = 9 * l["cin"]; shape = (l["cout"], l["cin"], 3, 3)
        elif l["kind"] == "dw":
            fan = 9; shape = (l["cin"], 3, 3)
        else:
            fan = l["cin"]; shape = (l["cout"], l["cin"])
        gain = 6.0 if l["act"] != ACT_NONE else 3.0
        bound = np.sqrt(gain / fan)
        w = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        b = rng.uniform(-0.05, 0.05, size=(shape[0],)).astype(np.float32)
        layers.append(dict(l, weight=w, bias=b))
    bound = np.sqrt(3.0 / 1280)
    head = dict(
        pre_w=rng.uniform(-bound, bound, size=(NETVLAD_D, 1280)).astype(np.float32),
        pre_b=rng.uniform(-0.05, 0.05, size=(NETVLAD_D,)).astype(np.float32),
        assign_w=rng.normal(0, 1.0 / np.sqrt(NETVLAD_D), size=(NETVLAD_K, NETVLAD_D)).astype(np.float32),
        assign_b=rng.uniform(-0.1, 0.1, size=(NETVLAD_K,)).astype(np.float32),
        centroids=rng.normal(0, 0.3, size=(NETVLAD_K, NETVLAD_D)).astype(np.float32),
    )
    return dict(layers=layers, head=head)


def synthetic_netvlad_pca(out_dims=1024, seed=99):
    """PCA matrices in the reference's CSV layout (row 0 = mean, rows 1.. = components, mobilenetvlad_onnx.h:35-41)."""
    rng = np.random.RandomState(seed)
    comp = rng.normal(0, 1.0 / np.sqrt(NETVLAD_DIM), size=(out_dims, NETVLAD_DIM)).astype(np.float32)
    mean = rng.normal(0, 0.002, size=(NETVLAD_DIM,)).astype(np.float32)
    return comp, mean


def arch_flops(depth_multiplier=DEPTH_MULTIPLIER, H=480, W=640, head=False):
    """Algorithmic FLOPs (2 x MACs) of one image through the stand-in's TRUNK at this width (TF "SAME" strided layers: ceil division): 2.478 GFLOP at 0.75 (0.666 at 0.35, the
    width of rounds 2-5) and 640 x 480; head=True adds the NetVLAD head's pre-projection, soft-assignment and
    residual aggregation (0.103 GFLOP)."""
    h, w = H, W
    macs = 0
    for l in mobilenetvlad_arch(depth_multiplier):
        if l["stride"] == 2:
            h, w = (h + 1) // 2, (w + 1) // 2
        if l["kind"] == "conv":
            macs += h * w * l["cout"] * 9 * l["cin"]
        elif l["kind"] == "dw":
            macs += h * w * l["cin"] * 9
        else:
            macs += h * w * l["cin"] * l["cout"]
    if head:
        macs += h * w * (1280 * NETVLAD_D + 2 * NETVLAD_K * NETVLAD_D)      # pre-projection, soft-assignment logits, residual aggregation
    return 2.0 * macs
