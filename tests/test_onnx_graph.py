"""ONNX GRAPH -> d2fe_nv_layer[] (d2slam_amd/onnx_graph.py): the loader the reference's `MobileNetVLADONNX(model_path, ...)` constructor
implies (mobilenetvlad_onnx.h:18-47).  The reference's own .onnx is not in its tree, so the loader is exercised on a synthesised ONNX of
the stand-in graph (exporter in the same module), on a hand-built Conv + BatchNormalization + Relu graph, and on unsupported operators."""
import os

import numpy as np
import pytest

from d2slam_amd import netvlad as nvm, onnx_graph as og


def test_round_trip_of_the_standin(tmp_path):
    nv = nvm.synthetic_netvlad_weights()
    p = str(tmp_path / "standin.onnx")
    og.export_netvlad_onnx(nv, p, 96, 128)
    nodes, init, gin, gout = og.read_onnx_graph(p)
    assert "image:0" in gin and gout == ["descriptor:0"]                   # the reference's tensor names (mobilenetvlad_onnx.h:20)
    assert [n["op"] for n in nodes[:3]] == ["Transpose", "Sub", "Div"]
    got = og.load_netvlad_onnx(p)
    assert len(got["layers"]) == len(nv["layers"])
    for a, b in zip(got["layers"], nv["layers"]):
        for k in ("kind", "cin", "cout", "stride", "act", "res"):
            assert a[k] == b[k], (k, a[k], b[k])
        assert np.array_equal(a["weight"].reshape(-1), np.asarray(b["weight"], np.float32).reshape(-1)) and np.array_equal(a["bias"], b["bias"])
    for k in ("pre_w", "pre_b", "assign_w", "assign_b", "centroids"):
        assert np.array_equal(got["head"][k], nv["head"][k]), k


def test_oracle_runs_the_loaded_graph(tmp_path, orc):
    """the loaded layer list is a valid network: the oracle's forward on it equals the forward on the original dict"""
    from d2slam_amd.synth import synth_image
    nv = nvm.synthetic_netvlad_weights()
    p = str(tmp_path / "standin.onnx")
    og.export_netvlad_onnx(nv, p, 64, 96)
    img = synth_image(64, 96, 3)
    assert np.array_equal(orc.netvlad_forward(img, og.load_netvlad_onnx(p)), orc.netvlad_forward(img, nv))


def _tiny_graph(path, ops):
    """Transpose, (x-128)/128, Conv 1->8 s2, [extra ops], 1x1 pre-projection, NetVLAD head with K=2."""
    nodes, inits = [], []
    init = lambda n, a: (inits.append(og._tensor_pb(n, a)), n)[1]
    rng = np.random.RandomState(0)
    nodes.append(og._node_pb("Transpose", ["image:0"], ["t0"], "tr", perm=[0, 3, 1, 2]))
    nodes.append(og._node_pb("Sub", ["t0", init("c1", np.array([128.0], np.float32))], ["t1"], "sub"))
    nodes.append(og._node_pb("Mul", ["t1", init("c2", np.array([1.0 / 128.0], np.float32))], ["t2"], "mul"))
    w0 = rng.randn(8, 1, 3, 3).astype(np.float32); b0 = rng.randn(8).astype(np.float32)
    nodes.append(og._node_pb("Conv", ["t2", init("w0", w0), init("b0", b0)], ["c0"], "conv0", kernel_shape=[3, 3], strides=[2, 2], auto_pad="SAME_UPPER", group=1))
    t = "c0"
    extra = {}
    for op in ops:
        if op == "bn":
            extra = dict(s=rng.rand(8).astype(np.float32) + 0.5, b=rng.randn(8).astype(np.float32), m=rng.randn(8).astype(np.float32), v=rng.rand(8).astype(np.float32) + 0.1)
            nodes.append(og._node_pb("BatchNormalization", [t, init("bn_s", extra["s"]), init("bn_b", extra["b"]), init("bn_m", extra["m"]), init("bn_v", extra["v"])], ["bn"], "bn", epsilon=1e-3))
            t = "bn"
        elif op == "relu":
            nodes.append(og._node_pb("Relu", [t], ["r"], "relu")); t = "r"
        else:
            nodes.append(og._node_pb(op, [t], ["u"], "unsupported")); t = "u"
    wp = rng.randn(4, 8, 1, 1).astype(np.float32); bp = rng.randn(4).astype(np.float32)
    nodes.append(og._node_pb("Conv", [t, init("wp", wp), init("bp", bp)], ["feat"], "pre", kernel_shape=[1, 1], strides=[1, 1], pads=[0, 0, 0, 0], group=1))
    K, D, P = 2, 4, 4 * 6
    nodes.append(og._node_pb("Conv", ["feat", init("aw", rng.randn(K, D, 1, 1).astype(np.float32)), init("ab", rng.randn(K).astype(np.float32))], ["as"], "assign", kernel_shape=[1, 1]))
    nodes.append(og._node_pb("Softmax", ["as"], ["a"], "sm", axis=1))
    nodes.append(og._node_pb("Reshape", ["a", init("s1", np.array([K, P], np.int64))], ["a2"], "r1"))
    nodes.append(og._node_pb("Reshape", ["feat", init("s2", np.array([D, P], np.int64))], ["x2"], "r2"))
    nodes.append(og._node_pb("Transpose", ["x2"], ["x3"], "tr2", perm=[1, 0]))
    nodes.append(og._node_pb("MatMul", ["a2", "x3"], ["ax"], "mm"))
    nodes.append(og._node_pb("ReduceSum", ["a2", init("ax1", np.array([1], np.int64))], ["asum"], "rs", keepdims=1))
    nodes.append(og._node_pb("Mul", ["asum", init("cen", rng.randn(K, D).astype(np.float32))], ["ac"], "mulc"))
    nodes.append(og._node_pb("Sub", ["ac", "ax"], ["v"], "sub2"))
    nodes.append(og._node_pb("LpNormalization", ["v"], ["vi"], "n1", axis=1, p=2))
    nodes.append(og._node_pb("Reshape", ["vi", init("s3", np.array([1, K * D], np.int64))], ["vf"], "r3"))
    nodes.append(og._node_pb("LpNormalization", ["vf"], ["descriptor:0"], "n2", axis=1, p=2))
    graph = b"".join(og._f(1, 2, n) for n in nodes) + og._s(2, "tiny") + b"".join(og._f(5, 2, i) for i in inits)
    graph += og._f(11, 2, og._value_info("image:0", [1, 8, 12, 1])) + og._f(12, 2, og._value_info("descriptor:0", [1, K * D]))
    open(path, "wb").write(og._f(1, 0, og._vi(7)) + og._f(8, 2, og._s(1, "") + og._f(2, 0, og._vi(13))) + og._f(7, 2, graph))
    return w0, b0, extra


def test_batchnorm_is_folded_and_relu_mapped(tmp_path):
    p = str(tmp_path / "bn.onnx")
    w0, b0, e = _tiny_graph(p, ["bn", "relu"])
    nv = og.load_netvlad_onnx(p)
    assert len(nv["layers"]) == 1 and nv["layers"][0]["kind"] == "conv" and nv["layers"][0]["stride"] == 2 and nv["layers"][0]["act"] == og.ACT_RELU
    g = e["s"] / np.sqrt(e["v"] + np.float32(1e-3))
    assert np.allclose(nv["layers"][0]["weight"], w0 * g.reshape(-1, 1, 1, 1), rtol=1e-6)
    assert np.allclose(nv["layers"][0]["bias"], (b0 - e["m"]) * g + e["b"], rtol=1e-6, atol=1e-7)
    assert nv["head"]["pre_w"].shape == (4, 8) and nv["head"]["centroids"].shape == (2, 4)


def test_unsupported_operator_is_named(tmp_path):
    p = str(tmp_path / "bad.onnx")
    _tiny_graph(p, ["Sigmoid"])
    with pytest.raises(og.UnsupportedOnnx, match="Sigmoid"):
        og.load_netvlad_onnx(p)


@pytest.mark.gpu
def test_hip_loads_the_onnx_graph(tmp_path, orc):
    """FrontEnd.load_netvlad_onnx: the file route of MobileNetVLADONNX's constructor, end to end on the GPU."""
    from d2slam_amd import api
    from d2slam_amd.synth import synth_image
    nv = nvm.synthetic_netvlad_weights()
    p = str(tmp_path / "standin.onnx")
    og.export_netvlad_onnx(nv, p, 96, 128)
    img = synth_image(96, 128, 4)
    fe = api.FrontEnd(api.SuperPointConfig(input_width=128, input_height=96, max_batch=1))
    fe.load_netvlad_onnx(p)
    got = fe.netvlad(img[None])[0]
    assert np.abs(got - orc.netvlad_forward(img, nv)).max() <= 1e-4
    fe.close()
