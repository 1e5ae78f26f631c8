"""ONNX graph -> the flat layer list of d2fe_load_netvlad (and back), dependency-free.

The reference constructs `MobileNetVLADONNX(model_path, ...)` from an ONNX FILE (d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:18-47,
tensor names `image:0` -> `descriptor:0`, input NHWC [1,H,W,1] float gray, NOT scaled by the caller, :60).  That file
(models/netvlad_series/mobilenetvlad_dyn_size.onnx) is a Dropbox download that is not in the reference tree, so its exact operator
sequence cannot be pinned here.  What this module does:

* `load_netvlad_onnx(path)`: walks the GRAPH (not just the initializers) and maps the operators a MobileNetV2 trunk + NetVLAD head is
  made of onto `dict(layers=[...], head={...})` -- the structure `FrontEnd.load_netvlad` hands to the C ABI as `d2fe_nv_layer[]`:
      trunk:  [Transpose NHWC->NCHW] [Sub 128, Div 128 | Mul 1/128]  the input scaling the kernels hard-code, verified
              Conv k3 group 1 from 1 channel          -> kind "conv"
              Conv k3 group == channels               -> kind "dw"
              Conv k1 group 1                         -> kind "pw"
              BatchNormalization after a Conv         -> folded into that Conv's weight and bias
              Relu | Clip(0, 6)                       -> act 1 | 2 of the producing layer
              Add(conv output, earlier tensor)        -> res = index of the layer that produced the earlier tensor
              pads: auto_pad SAME_UPPER, or explicit pads equal to TF "SAME" for the node's stride (checked)
      head:   Conv k1 (pre-projection, linear) ; Conv k1 -> Softmax(axis 1) (soft assignment) ; Reshape/Transpose/MatMul/ReduceSum/
              Mul(centroids)/Sub/LpNormalization(axis 1)/Reshape/LpNormalization -- the aggregation V[k] = sum_p a[p,k] (c_k - x_p),
              intra-normalisation, flatten, L2; the centroids are the Mul node's initializer.
  Anything else raises `UnsupportedOnnx` naming the node, so that a user who supplies the real file learns exactly which operator is new.
* `export_netvlad_onnx(nv, path, H, W)`: writes such a graph for a layer list (opset 13) -- used by the round-trip test and as a
  worked example of the operator subset.

Only protobuf wire-format primitives are used (no `onnx` package in the image); field numbers follow onnx.proto3.
"""
import struct

import numpy as np

from .weights import _pb_fields, _pb_varint

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2


class UnsupportedOnnx(ValueError):
    pass


# ---- reading --------------------------------------------------------------------------------------------------------------------------
def _packed_ints(v, wt):
    if wt != 2:
        return [v]
    out, p = [], 0
    while p < len(v):
        d, p = _pb_varint(v, p)
        out.append(d if d < (1 << 63) else d - (1 << 64))
    return out


def _tensor(buf):
    dims, dtype, name, raw, floats, int64s = [], 0, "", None, [], []
    for no, wt, v in _pb_fields(buf):
        if no == 1:
            dims += _packed_ints(v, wt)
        elif no == 2:
            dtype = v
        elif no == 8:
            name = bytes(v).decode()
        elif no == 9:
            raw = bytes(v)
        elif no == 4:
            floats.append(np.frombuffer(bytes(v), "<f4") if wt == 2 else np.array([v], "<u4").view("<f4"))
        elif no == 7:
            int64s += _packed_ints(v, wt)
    if dtype == 1:
        arr = np.frombuffer(raw, "<f4") if raw is not None else (np.concatenate(floats) if floats else np.zeros(0, "<f4"))
        arr = arr.astype(np.float32)
    elif dtype == 7:
        arr = np.frombuffer(raw, "<i8").astype(np.int64) if raw is not None else np.array(int64s, np.int64)
    else:
        return name, None
    return name, arr.reshape(dims if dims else arr.shape).copy()


def _attr(buf):
    name, val = "", None
    ints, floats = [], []
    for no, wt, v in _pb_fields(buf):
        if no == 1:
            name = bytes(v).decode()
        elif no == 2:
            val = struct.unpack("<f", struct.pack("<I", v))[0]
        elif no == 3:
            val = v if v < (1 << 63) else v - (1 << 64)
        elif no == 4:
            val = bytes(v).decode()
        elif no == 5:
            val = _tensor(v)[1]
        elif no == 7:
            floats += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif no == 8:
            ints += _packed_ints(v, wt)
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def read_onnx_graph(path):
    """-> (nodes, initializers, graph_inputs, graph_outputs); a node is dict(op, name, inputs, outputs, attrs)."""
    data = memoryview(open(path, "rb").read())
    nodes, init, gin, gout = [], {}, [], []
    for fno, wt, graph in _pb_fields(data):
        if fno != 7 or wt != 2:
            continue
        for gno, gwt, v in _pb_fields(graph):
            if gno == 1:
                n = dict(op="", name="", inputs=[], outputs=[], attrs={})
                for no, nwt, x in _pb_fields(v):
                    if no == 1:
                        n["inputs"].append(bytes(x).decode())
                    elif no == 2:
                        n["outputs"].append(bytes(x).decode())
                    elif no == 3:
                        n["name"] = bytes(x).decode()
                    elif no == 4:
                        n["op"] = bytes(x).decode()
                    elif no == 5:
                        k, a = _attr(x)
                        n["attrs"][k] = a
                nodes.append(n)
            elif gno == 5:
                name, arr = _tensor(v)
                if arr is not None:
                    init[name] = arr
            elif gno in (11, 12):
                for no, _, x in _pb_fields(v):
                    if no == 1:
                        (gin if gno == 11 else gout).append(bytes(x).decode())
    return nodes, init, gin, gout


def _same_pads_ok(node, stride):
    ap = node["attrs"].get("auto_pad", "NOTSET")
    if ap in ("SAME_UPPER",):
        return True
    pads = node["attrs"].get("pads")
    k = node["attrs"].get("kernel_shape", [1, 1])[0]
    if k == 1:
        return pads in (None, [0, 0, 0, 0])
    # TF "SAME" for k = 3: stride 1 -> (1,1,1,1); stride 2 on an even size -> begin 0, end 1 (tf2onnx writes exactly that)
    return pads == ([1, 1, 1, 1] if stride == 1 else [0, 0, 1, 1])


def load_netvlad_onnx(path):
    nodes, init, gin, gout = read_onnx_graph(path)
    inputs = [g for g in gin if g not in init]
    if len(inputs) != 1:
        raise UnsupportedOnnx("expected one graph input, found %s" % inputs)
    cur = inputs[0]
    consumers = {}
    for n in nodes:
        for i in n["inputs"]:
            consumers.setdefault(i, []).append(n)

    def const(name):
        if name in init:
            return init[name]
        for n in nodes:                    # Constant nodes
            if n["op"] == "Constant" and n["outputs"] == [name]:
                return n["attrs"].get("value")
        return None

    layers, produced = [], {}             # produced: tensor name -> index of the layer whose (activated / residual-added) output it is
    scaled = {"sub": False, "div": False}
    i = 0
    order = list(nodes)
    pos = {id(n): k for k, n in enumerate(order)}

    def single_consumer(t):
        c = [n for n in consumers.get(t, []) if n["op"] != "Constant"]
        return c

    feat_tensor = None
    # ---- trunk -----------------------------------------------------------------------------------------------------------------------
    while True:
        cs = single_consumer(cur)
        if not cs:
            raise UnsupportedOnnx("tensor %r has no consumer before the NetVLAD head was found" % cur)
        # the head starts where a tensor feeds BOTH a soft-assignment Conv->Softmax and the aggregation
        if len(cs) > 1 and any(c["op"] == "Conv" for c in cs) and layers and not any(c["op"] == "Add" for c in cs):
            feat_tensor = cur
            break
        # a residual source is consumed by the next block's first Conv AND by an Add: follow the Conv
        n = [c for c in cs if c["op"] != "Add"][0] if len(cs) > 1 else cs[0]
        op = n["op"]
        if op == "Transpose":
            if n["attrs"].get("perm") != [0, 3, 1, 2] or layers:
                raise UnsupportedOnnx("Transpose %r: only the leading NHWC->NCHW transpose is supported" % n["name"])
        elif op in ("Sub", "Div", "Mul") and not layers:
            c = const(n["inputs"][1])
            v = float(np.asarray(c).reshape(-1)[0]) if c is not None and np.asarray(c).size == 1 else None
            if op == "Sub" and v == 128.0:
                scaled["sub"] = True
            elif (op == "Div" and v == 128.0) or (op == "Mul" and v == 1.0 / 128.0):
                scaled["div"] = True
            else:
                raise UnsupportedOnnx("input scaling %s %r: the kernels implement (x - 128) / 128 only" % (op, v))
        elif op == "Conv":
            if not (scaled["sub"] and scaled["div"]):
                raise UnsupportedOnnx("the graph does not scale its input by (x - 128) / 128 before the first convolution")
            w = init.get(n["inputs"][1])
            if w is None:
                raise UnsupportedOnnx("Conv %r: weight is not an initializer" % n["name"])
            b = init.get(n["inputs"][2]) if len(n["inputs"]) > 2 else np.zeros(w.shape[0], np.float32)
            k = w.shape[2]
            group = n["attrs"].get("group", 1)
            stride = (n["attrs"].get("strides") or [1, 1])[0]
            if n["attrs"].get("dilations", [1, 1]) != [1, 1] or w.shape[2] != w.shape[3] or k not in (1, 3) or not _same_pads_ok(n, stride):
                raise UnsupportedOnnx("Conv %r: kernel %s / pads %s / dilations not supported" % (n["name"], w.shape, n["attrs"].get("pads")))
            if k == 3 and group == 1 and w.shape[1] == 1:
                kind, cin, cout, wt = "conv", 1, w.shape[0], w.copy()
            elif k == 3 and group == w.shape[0] and w.shape[1] == 1:
                kind, cin, cout, wt = "dw", w.shape[0], w.shape[0], w[:, 0].copy()
            elif k == 1 and group == 1:
                kind, cin, cout, wt = "pw", w.shape[1], w.shape[0], w[:, :, 0, 0].copy()
            else:
                raise UnsupportedOnnx("Conv %r: shape %s group %d is neither the stem, a depthwise 3x3 nor a 1x1" % (n["name"], w.shape, group))
            layers.append(dict(kind=kind, cin=int(cin), cout=int(cout), stride=int(stride), act=ACT_NONE, res=-1,
                               weight=wt.astype(np.float32), bias=np.asarray(b, np.float32).copy()))
        elif op == "BatchNormalization":
            if not layers:
                raise UnsupportedOnnx("BatchNormalization before any convolution")
            s, bb, m, v = (init[x] for x in n["inputs"][1:5])
            g = s / np.sqrt(v + n["attrs"].get("epsilon", 1e-5))
            L = layers[-1]
            L["weight"] = (L["weight"] * g.reshape((-1,) + (1,) * (L["weight"].ndim - 1))).astype(np.float32)
            L["bias"] = ((L["bias"] - m) * g + bb).astype(np.float32)
        elif op == "Relu":
            layers[-1]["act"] = ACT_RELU
        elif op == "Clip":
            lo = n["attrs"].get("min"); hi = n["attrs"].get("max")
            if lo is None and len(n["inputs"]) > 1:
                lo = float(np.asarray(const(n["inputs"][1])).reshape(-1)[0]); hi = float(np.asarray(const(n["inputs"][2])).reshape(-1)[0])
            if (lo, hi) != (0.0, 6.0):
                raise UnsupportedOnnx("Clip %r: only Clip(0, 6) = ReLU6 is supported" % n["name"])
            layers[-1]["act"] = ACT_RELU6
        elif op == "Add":
            other = [x for x in n["inputs"] if x != cur]
            if len(other) != 1 or other[0] not in produced:
                raise UnsupportedOnnx("Add %r: the second operand is not the output of an earlier layer" % n["name"])
            if layers[-1]["act"] != ACT_NONE:
                raise UnsupportedOnnx("Add %r after an activation: MobileNetV2 adds to the linear bottleneck output" % n["name"])
            layers[-1]["res"] = produced[other[0]]
        else:
            raise UnsupportedOnnx("operator %s (%r) is not part of the supported MobileNetV2 + NetVLAD subset" % (op, n["name"]))
        cur = n["outputs"][0]
        if layers:
            produced[cur] = len(layers) - 1
    # ---- head: the last trunk layer is the linear pre-projection -------------------------------------------------------------------------
    pre = layers.pop()
    if pre["kind"] != "pw" or pre["act"] != ACT_NONE or pre["res"] >= 0:
        raise UnsupportedOnnx("the layer feeding the NetVLAD head must be a linear 1x1 convolution (the pre-projection)")
    cs = consumers[feat_tensor]
    assign = [c for c in cs if c["op"] == "Conv"]
    if len(assign) != 1:
        raise UnsupportedOnnx("NetVLAD head: expected one soft-assignment Conv on the feature tensor")
    aw = init[assign[0]["inputs"][1]]; ab = init[assign[0]["inputs"][2]]
    sm = consumers.get(assign[0]["outputs"][0], [])
    if len(sm) != 1 or sm[0]["op"] != "Softmax" or sm[0]["attrs"].get("axis", -1) != 1:
        raise UnsupportedOnnx("NetVLAD head: the assignment Conv must feed Softmax(axis = 1)")
    K, D = aw.shape[0], aw.shape[1]
    # the rest of the head must be exactly the aggregation; find the centroids as the [K, D] initializer of a Mul, and check the op multiset
    head_ops = [n["op"] for n in order[pos[id(sm[0])] + 1:] if n["op"] != "Constant"]
    allowed = {"Reshape", "Transpose", "MatMul", "ReduceSum", "Mul", "Sub", "LpNormalization"}
    if not set(head_ops) <= allowed or head_ops.count("LpNormalization") != 2 or head_ops.count("MatMul") != 1 or head_ops.count("Sub") != 1:
        raise UnsupportedOnnx("NetVLAD head: unexpected operators after the Softmax: %s" % head_ops)
    cen = None
    for n in order[pos[id(sm[0])] + 1:]:
        if n["op"] == "Mul":
            for x in n["inputs"]:
                if x in init and init[x].shape == (K, D):
                    cen = init[x]
        if n["op"] == "Sub":
            # V = asum * c - sum_p a x  (residuals c - x): the Mul output must be the FIRST operand
            first = [m for m in order if m["outputs"] and m["outputs"][0] == n["inputs"][0]]
            if not first or first[0]["op"] != "Mul":
                raise UnsupportedOnnx("NetVLAD head: expected Sub(Mul(sum_p a, centroids), MatMul(a, x)) (residuals c - x)")
    if cen is None:
        raise UnsupportedOnnx("NetVLAD head: no [K, D] centroid initializer found")
    head = dict(pre_w=pre["weight"], pre_b=pre["bias"], assign_w=aw[:, :, 0, 0].astype(np.float32).copy(), assign_b=ab.astype(np.float32).copy(),
                centroids=cen.astype(np.float32).copy())
    return dict(layers=layers, head=head)


# ---- writing (opset 13) ------------------------------------------------------------------------------------------------------------------
def _vi(x):
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _f(no, wt, payload):
    key = _vi((no << 3) | wt)
    if wt == 2:
        return key + _vi(len(payload)) + payload
    return key + payload


def _s(no, text):
    return _f(no, 2, text.encode())


def _tensor_pb(name, arr):
    arr = np.ascontiguousarray(arr)
    dt = 1 if arr.dtype == np.float32 else 7
    out = b"".join(_f(1, 0, _vi(int(d))) for d in arr.shape) + _f(2, 0, _vi(dt)) + _s(8, name)
    out += _f(9, 2, arr.astype("<f4" if dt == 1 else "<i8").tobytes())
    return out


def _attr_pb(name, val):
    out = _s(1, name)
    if isinstance(val, str):
        out += _f(4, 2, val.encode()) + _f(20, 0, _vi(3))
    elif isinstance(val, float):
        out += _f(2, 5, struct.pack("<f", val)) + _f(20, 0, _vi(1))
    elif isinstance(val, int):
        out += _f(3, 0, _vi(val & ((1 << 64) - 1))) + _f(20, 0, _vi(2))
    else:
        out += b"".join(_f(8, 0, _vi(int(v) & ((1 << 64) - 1))) for v in val) + _f(20, 0, _vi(7))
    return out


def _node_pb(op, inputs, outputs, name, **attrs):
    out = b"".join(_s(1, i) for i in inputs) + b"".join(_s(2, o) for o in outputs) + _s(3, name) + _s(4, op)
    out += b"".join(_f(5, 2, _attr_pb(k, v)) for k, v in attrs.items())
    return out


def _value_info(name, shape):
    dims = b"".join(_f(1, 2, _f(1, 0, _vi(int(d)))) for d in shape)
    ttype = _f(1, 0, _vi(1)) + _f(2, 2, dims)
    return _s(1, name) + _f(2, 2, _f(1, 2, ttype))


def export_netvlad_onnx(nv, path, H, W):
    """Writes the layer list as an ONNX model with the reference's tensor names (`image:0` NHWC -> `descriptor:0`)."""
    nodes, inits = [], []

    def init(name, arr):
        inits.append(_tensor_pb(name, arr))
        return name

    t = "image:0"
    nodes.append(_node_pb("Transpose", [t], ["x_nchw"], "to_nchw", perm=[0, 3, 1, 2])); t = "x_nchw"
    nodes.append(_node_pb("Sub", [t, init("c128", np.array([128.0], np.float32))], ["x_sub"], "sub128")); t = "x_sub"
    nodes.append(_node_pb("Div", [t, init("c128d", np.array([128.0], np.float32))], ["x_scaled"], "div128")); t = "x_scaled"
    outs = []
    layers = list(nv["layers"]) + [dict(kind="pw", cin=nv["head"]["pre_w"].shape[1], cout=nv["head"]["pre_w"].shape[0], stride=1, act=ACT_NONE, res=-1,
                                        weight=nv["head"]["pre_w"], bias=nv["head"]["pre_b"])]
    hh, ww = H, W
    for li, L in enumerate(layers):
        w = np.asarray(L["weight"], np.float32)
        if L["kind"] == "conv":
            wt, group, k = w.reshape(L["cout"], 1, 3, 3), 1, 3
        elif L["kind"] == "dw":
            wt, group, k = w.reshape(L["cin"], 1, 3, 3), L["cin"], 3
        else:
            wt, group, k = w.reshape(L["cout"], L["cin"], 1, 1), 1, 1
        s = L["stride"]
        pads = [0, 0, 0, 0] if k == 1 else ([1, 1, 1, 1] if s == 1 else [0, 0, 1, 1])
        if k == 3 and s == 2 and (hh % 2 or ww % 2):
            raise ValueError("export: stride-2 layer on an odd size needs per-axis pads; use sizes divisible by 32")
        o = "l%d" % li
        nodes.append(_node_pb("Conv", [t, init(o + "_w", wt), init(o + "_b", np.asarray(L["bias"], np.float32))], [o], o + "_conv",
                              kernel_shape=[k, k], strides=[s, s], pads=pads, group=group, dilations=[1, 1]))
        t = o
        if L["res"] >= 0:
            nodes.append(_node_pb("Add", [t, outs[L["res"]]], [o + "_add"], o + "_add")); t = o + "_add"
        if L["act"] == ACT_RELU:
            nodes.append(_node_pb("Relu", [t], [o + "_relu"], o + "_relu")); t = o + "_relu"
        elif L["act"] == ACT_RELU6:
            nodes.append(_node_pb("Clip", [t, init(o + "_lo", np.array(0.0, np.float32)), init(o + "_hi", np.array(6.0, np.float32))], [o + "_clip"], o + "_clip"))
            t = o + "_clip"
        outs.append(t)
        hh, ww = (hh + s - 1) // s, (ww + s - 1) // s
    hd = nv["head"]
    K, D = hd["assign_w"].shape
    P = hh * ww
    feat = t
    nodes.append(_node_pb("Conv", [feat, init("assign_w", hd["assign_w"].reshape(K, D, 1, 1)), init("assign_b", hd["assign_b"])], ["assign"], "assign_conv",
                          kernel_shape=[1, 1], strides=[1, 1], pads=[0, 0, 0, 0], group=1, dilations=[1, 1]))
    nodes.append(_node_pb("Softmax", ["assign"], ["a"], "assign_softmax", axis=1))
    nodes.append(_node_pb("Reshape", ["a", init("shape_kp", np.array([K, P], np.int64))], ["a_kp"], "a_reshape"))
    nodes.append(_node_pb("Reshape", [feat, init("shape_dp", np.array([D, P], np.int64))], ["x_dp"], "x_reshape"))
    nodes.append(_node_pb("Transpose", ["x_dp"], ["x_pd"], "x_transpose", perm=[1, 0]))
    nodes.append(_node_pb("MatMul", ["a_kp", "x_pd"], ["ax"], "ax_matmul"))
    nodes.append(_node_pb("ReduceSum", ["a_kp", init("axes1", np.array([1], np.int64))], ["asum"], "asum", keepdims=1))
    nodes.append(_node_pb("Mul", ["asum", init("centroids", hd["centroids"])], ["ac"], "ac_mul"))
    nodes.append(_node_pb("Sub", ["ac", "ax"], ["v"], "residual_sub"))
    nodes.append(_node_pb("LpNormalization", ["v"], ["v_intra"], "intra_norm", axis=1, p=2))
    nodes.append(_node_pb("Reshape", ["v_intra", init("shape_flat", np.array([1, K * D], np.int64))], ["v_flat"], "flatten"))
    nodes.append(_node_pb("LpNormalization", ["v_flat"], ["descriptor:0"], "l2_norm", axis=1, p=2))
    graph = b"".join(_f(1, 2, n) for n in nodes) + _s(2, "mobilenetvlad_standin") + b"".join(_f(5, 2, i) for i in inits)
    graph += _f(11, 2, _value_info("image:0", [1, H, W, 1])) + _f(12, 2, _value_info("descriptor:0", [1, K * D]))
    model = _f(1, 0, _vi(7)) + _s(2, "d2slam_amd.onnx_graph") + _f(8, 2, _s(1, "") + _f(2, 0, _vi(13))) + _f(7, 2, graph)
    with open(path, "wb") as f:
        f.write(model)
