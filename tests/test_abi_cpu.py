"""CPU checks of the drop-in boundary: libd2fe_hip.so loads without a GPU and exports exactly what
include/d2fe.h declares; host-side entry points behave; device entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "d2fe.h")).read()
    return sorted(set(re.findall(r"D2FE_API\s+[\w\s\*]+?\b(d2fe_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from d2slam_amd import build
    return C.CDLL(build.build())


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    from d2slam_amd import api
    assert sorted(api.EXPORTS) == names, "api.EXPORTS out of sync with include/d2fe.h"


def test_product_library_has_no_debug_exports_and_reads_no_environment(lib):
    """test hooks, ablation switches and trace code live in the development library only."""
    import subprocess
    from d2slam_amd import api, build
    syms = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True).stdout
    assert "d2fe_debug" not in syms and "g_wino_trace" not in syms and "g_pc_trace" not in syms
    assert not any(n.startswith("d2fe_debug") for n in api.EXPORTS)
    und = subprocess.run(["nm", "-D", "--undefined-only", build.LIB], capture_output=True, text=True).stdout
    assert " getenv" not in und and "secure_getenv" not in und, "the product library must not read the environment"
    csrc = os.path.join(ROOT, "d2slam_amd", "csrc")
    hits = [f for f in os.listdir(csrc) for l in open(os.path.join(csrc, f), errors="ignore") if "getenv(" in l]
    assert len(hits) <= 1, hits          # the one in d2fe_dev_env (kernels.h), compiled only with -DD2FE_DEVTOOLS


def test_development_library_exports_the_hooks():
    from d2slam_amd import api, build
    dev = C.CDLL(build.build(dev=True))
    src = open(os.path.join(ROOT, "include", "d2fe_debug.h")).read()
    names = sorted(set(re.findall(r"D2FE_API\s+[\w\s\*]+?\b(d2fe_\w+)\s*\(", src)))
    assert names == sorted(api.DEBUG_EXPORTS)
    for n in names + list(api.EXPORTS):
        assert hasattr(dev, n), "the development library misses %s" % n


def test_version_and_default_config(lib):
    from d2slam_amd.api import _Config
    lib.d2fe_version.restype = C.c_char_p
    assert b"gfx950" in lib.d2fe_version()
    c = _Config()
    lib.d2fe_default_config(C.byref(c))
    assert c.struct_size == C.sizeof(_Config)
    # SuperPointConfig defaults, superpoint_tensorrt.h:18-24
    assert (c.max_keypoints, c.remove_borders) == (100, 1) and abs(c.keypoint_threshold - 0.015) < 1e-9


def test_half_image_filter_host_logic(lib, orc):
    rng = np.random.RandomState(0)
    pts = rng.uniform(0, 800, size=(300, 2)).astype(np.float32)
    from d2slam_amd.api import get_feature_half_img
    desc = rng.randn(300, 256).astype(np.float32)
    for left in (True, False):
        d, p, idx = get_feature_half_img(pts, desc, left, 800, 200.0)
        ref = orc.half_img(pts, left, 800, 200.0)
        assert np.array_equal(idx, ref) and np.array_equal(d, desc[ref]) and np.array_equal(p, pts[ref])


def test_invalid_arguments_do_not_crash(lib):
    lib.d2fe_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert lib.d2fe_create(None, C.byref(h)) == -1
    from d2slam_amd.api import _Config
    c = _Config()
    lib.d2fe_default_config(C.byref(c))
    c.max_width = 8
    assert lib.d2fe_create(C.byref(c), C.byref(h)) == -1 and b"at least 16" in lib.d2fe_last_error()
    c.max_width = 640; c.max_keypoints = 0
    assert lib.d2fe_create(C.byref(c), C.byref(h)) == -1 and b"max_keypoints" in lib.d2fe_last_error()
    n = C.c_int(5)
    assert lib.d2fe_match_knn(None, None, 3, None, 3, 256, C.c_double(0.8), None, None, C.c_double(-1.0), None, None, None,
                              3, C.byref(n)) == -1 and n.value == 0


def test_no_cpu_fallback():
    """Without a GPU the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from d2slam_amd import api
    with pytest.raises(api.D2FEError):
        api.FrontEnd(api.SuperPointConfig())
    sp = api.SuperPoint(api.SuperPointConfig(), weights=None)
    assert sp.build() is False
    ok, k, d, s = sp.infer(np.zeros((480, 640), np.uint8))
    assert ok is False and len(k) == 0 and len(d) == 0 and len(s) == 0      # reference clears outputs on failure


def test_product_does_not_import_oracle():
    pk = os.path.join(ROOT, "d2slam_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                s = open(os.path.join(dp, f)).read()
                assert "from oracle" not in s and "import oracle" not in s and "d2fe_oracle" not in s.replace("oracle/d2fe_oracle.c", "").replace("oracle/d2fe_oracle_lk.c", ""), f   # comments may cite the files


def _pb_key(fno, wt):
    return _pb_varint_enc((fno << 3) | wt)


def _pb_varint_enc(v):
    out = bytearray()
    while True:
        b = v & 0x7F; v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_ld(fno, payload):
    return _pb_key(fno, 2) + _pb_varint_enc(len(payload)) + payload


def test_onnx_initializer_reader_roundtrip(tmp_path):
    """The reference ships its SuperPoint weights as ONNX; d2slam_amd.weights reads the initializers without the onnx package.
    A minimal model file is serialised here by hand (ModelProto.graph.initializer[], raw_data and float_data encodings, packed
    and unpacked dims) and must come back bit for bit."""
    from d2slam_amd.weights import SP_LAYERS, load_superpoint_onnx, read_onnx_initializers, synthetic_superpoint_weights
    w = synthetic_superpoint_weights(seed=7)
    tensors = b""
    for i, n in enumerate(SP_LAYERS):
        for suffix, arr in ((".weight", w[n][0]), (".bias", w[n][1])):
            dims = b"".join(_pb_key(1, 0) + _pb_varint_enc(d) for d in arr.shape) if i % 2 else _pb_ld(1, b"".join(_pb_varint_enc(d) for d in arr.shape))
            data = _pb_ld(9, arr.astype("<f4").tobytes()) if suffix == ".weight" else _pb_ld(4, arr.astype("<f4").tobytes())
            t = dims + _pb_key(2, 0) + _pb_varint_enc(1) + _pb_ld(8, (n + suffix).encode()) + data
            tensors += _pb_ld(5, t)
    int_tensor = _pb_key(1, 0) + _pb_varint_enc(2) + _pb_key(2, 0) + _pb_varint_enc(7) + _pb_ld(8, b"shape_const") + _pb_ld(9, np.array([1, 2], "<i8").tobytes())
    graph = _pb_ld(1, _pb_ld(1, b"image")) + _pb_ld(2, b"torch_jit") + tensors + _pb_ld(5, int_tensor)
    model = _pb_key(1, 0) + _pb_varint_enc(8) + _pb_ld(2, b"pytorch") + _pb_ld(7, graph) + _pb_ld(8, _pb_ld(1, b"") + _pb_key(2, 0) + _pb_varint_enc(16))
    path = tmp_path / "sp.onnx"
    path.write_bytes(model)
    init = read_onnx_initializers(str(path))
    assert "shape_const" not in init and len(init) == 24
    back = load_superpoint_onnx(str(path))
    for n in SP_LAYERS:
        assert np.array_equal(back[n][0], w[n][0]) and np.array_equal(back[n][1], w[n][1])
