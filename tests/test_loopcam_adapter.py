"""The adapter under the reference's OWN caller (north_star: "so d2vins/d2pgo consume identical VisualImageDesc structs").

LoopCam::extractorImgDescDeepnet (d2frontend/src/loop_cam.cpp:589-648) is compiled unchanged, where it lies under /root/reference, by oracle/build_ref.py
(oracle/ref_shim/spref_loopcam.cpp) together with the structs it fills (VisualImageDesc d2frontend_types.h:85-110, LandmarkPerFrame d2landmarks.h:28-70),
extractColor (loop_utils.cpp:54-63) and camodocal's CataCamera::liftProjective -- twice:

  side "hip": superpoint_ptr / netvlad_onnx = include/d2fe_adapter.cpp (the file a D2SLAM maintainer adds) over libd2fe_hip.so,
  side "ref": superpoint_ptr = the reference's own SuperPoint::infer + processOutput (superpoint_tensorrt.cpp:161-183,200-350) on the oracle's network outputs.

The -m gpu tests compare EVERY field the function sets in the two VisualImageDesc; the CPU tests hold the "ref" side to an independent numpy statement of the
struct fill (so a mistake in the test library cannot cancel out between the two sides), including the NaN-skip misalignment of loop_cam.cpp:626-630 (a keypoint
whose ray is NaN is dropped from `landmarks` but stays in `landmark_descriptor` / `landmark_scores`)."""
import os

import numpy as np
import pytest

from d2slam_amd.synth import synth_image
from d2slam_amd.weights import synthetic_superpoint_weights, save_superpoint_d2fw, save_netvlad_d2fw

ref = pytest.importorskip("oracle.ref")
from oracle import build_ref  # noqa: E402

pytestmark = pytest.mark.skipif(not (build_ref.available() or all(os.path.exists(p) for p in build_ref.LOOPCAM_LIBS.values())),
                                reason="no reference tree and no prebuilt oracle/_ref/libspref_loopcam_*.so")

PINHOLE = (0, (385.7, 386.1, 322.3, 238.9))                                   # realsense_d435-like
# MEI cameras (xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0): a fisheye with distortion, and one whose xi > 1 and short focal length send the image corners
# past the model's domain -- 1 + (1 - xi^2) rho^2 < 0 -> sqrt of a negative number -> NaN rays (CataCamera.cc:466-480)
MEI = (1, (1.9, -0.25, 0.08, 0.0007, -0.0004, 780.0, 779.0, 321.0, 243.0))
MEI_NAN = (1, (1.5, 0.0, 0.0, 0.0, 0.0, 330.0, 330.0, 320.0, 240.0))          # NaN beyond 295 px from the centre


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    o.build()
    return o


def _lift_np(cam, pts):
    """numpy statement of cams[i]->liftProjective + normalize() + hasNaN() for the two camera kinds of the test library (pinhole without distortion; MEI WITHOUT
    distortion -- the distorted MEI model is held to camodocal itself in tests/test_ref_pin.py)"""
    kind, p = cam
    out = np.zeros((len(pts), 3))
    for i, (x, y) in enumerate(pts.astype(np.float64)):
        if kind == 0:
            fx, fy, cx, cy = p[:4]
            v = np.array([(1.0 / fx) * x + (-cx / fx), (1.0 / fy) * y + (-cy / fy), 1.0])
        else:
            xi, g1, g2, u0, v0 = p[0], p[5], p[6], p[7], p[8]
            mx, my = (1.0 / g1) * x + (-u0 / g1), (1.0 / g2) * y + (-v0 / g2)
            with np.errstate(invalid="ignore"):      # CataCamera.cc:476-486, the projective ray of the MEI model
                if xi == 1.0:
                    v = np.array([mx, my, (1.0 - mx * mx - my * my) / 2.0])
                else:
                    rho2 = mx * mx + my * my
                    v = np.array([mx, my, 1.0 - xi * (rho2 + 1.0) / (xi + np.sqrt(1.0 + (1.0 - xi * xi) * rho2))])
        z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]
        if z > 0:
            v = v / np.sqrt(z)
        out[i] = v
    return out


def _expected_from_post(img, kps, sc, desc, cam, stamp, camera_index, camera_id):
    """what loop_cam.cpp:619-645 makes of infer()'s outputs"""
    rays = _lift_np(cam, kps)
    keep = ~np.isnan(rays).any(axis=1)
    xi = np.rint(kps[keep]).astype(int)
    col = img[xi[:, 1], xi[:, 0]]
    return dict(n=int(keep.sum()), pt2d=kps[keep], pt3d=rays[keep], color=np.stack([col] * 3, 1), nan=int((~keep).sum()), desc=desc.reshape(-1), scores=sc)


@pytest.mark.parametrize("cam,cfg", [(PINHOLE, 0), (MEI_NAN, 0), (PINHOLE, 1)])
def test_reference_caller_over_reference_infer_vs_numpy(orc, cam, cfg):
    """side "ref" on CPU: landmark count and order, pt2d, pt3d_norm (bitwise: the same double arithmetic), colour, stamps / ids, the full descriptor and score vectors,
    NaN skips; camera_configuration 1 = STEREO_FISHEYE masks the bottom quarter of the image IN PLACE before inference (loop_cam.cpp:601-604)."""
    H, W, N = 96, 128, 60
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    img = synth_image(H, W, 11)
    masked = img.copy()
    if cfg == 1:
        masked[H * 3 // 4:] = 0
    camv = cam if cam is not MEI_NAN else (1, (1.8, 0.0, 0.0, 0.0, 0.0, 66.0, 66.0, 64.0, 48.0))       # the small image's version of the NaN camera
    f = orc.superpoint_forward(masked, w)
    g = np.linspace(-1, 1, 64).astype(np.float32)
    lc = ref.LoopCam("ref", W, H, N, self_id=3, camera_configuration=cfg, cams=(PINHOLE, camv))
    lc.set_network_outputs(f["semi"], f["desc"], g)
    work = img.copy()
    out = lc.extract(work, stamp=99.25, camera_index=1, camera_id=3001)
    assert np.array_equal(work, masked)                                     # the caller's image was masked in place (or left alone)
    rk, rs, rd = ref.superpoint_post(f["semi"], f["desc"], 0.015, 1, N)
    exp = _expected_from_post(masked, rk, rs, rd, camv, 99.25, 1, 3001)
    assert out["stamp"] == 99.25 and out["camera_index"] == 1 and out["camera_id"] == 3001 and out["drone_id"] == 3
    assert out["n_landmarks"] == exp["n"] and out["nan_warnings"] == exp["nan"]
    if cam is MEI_NAN:
        assert 0 < exp["nan"] < len(rk), "the NaN camera must drop some keypoints and keep others (%d of %d)" % (exp["nan"], len(rk))
    assert np.array_equal(out["pt2d"], exp["pt2d"]) and np.array_equal(out["color"], exp["color"])
    assert np.array_equal(out["pt3d_norm"], exp["pt3d"])
    assert np.all(out["lm_camera_index"] == 1) and np.all(out["lm_camera_id"] == 3001) and np.all(out["lm_stamp"] == 99.25) and np.all(out["lm_stamp_discover"] == 99.25)
    # the misalignment the reference warns about: descriptors and scores of the skipped keypoints stay in the flat vectors
    assert np.array_equal(out["landmark_descriptor"], exp["desc"]) and np.array_equal(out["landmark_scores"], exp["scores"])
    assert len(out["landmark_descriptor"]) == 256 * len(rk) >= 256 * out["n_landmarks"]
    assert np.array_equal(out["image_desc"], g)
    # superpoint_mode = true: no NetVLAD call (loop_cam.cpp:612-616)
    out2 = lc.extract(img.copy(), superpoint_mode=True, camera_index=1)
    assert len(out2["image_desc"]) == 0 and out2["n_landmarks"] == exp["n"]
    lc.close()


def test_d2fw_container_round_trip(tmp_path):
    """the weight container the adapter reads (include/d2fe_weights_file.hpp) as d2slam_amd/weights.py writes it: header and one tensor checked byte for byte"""
    import struct
    from d2slam_amd import netvlad as nvm
    w = synthetic_superpoint_weights()
    p = str(tmp_path / "sp.d2fw")
    save_superpoint_d2fw(p, w)
    b = open(p, "rb").read()
    assert b[:4] == b"D2FW" and struct.unpack("<II", b[4:12]) == (1, 24)
    nl, = struct.unpack("<I", b[12:16]); assert b[16:16 + nl] == b"conv1a.weight"
    nd, = struct.unpack("<I", b[16 + nl:20 + nl]); dims = struct.unpack("<4q", b[20 + nl:52 + nl])
    assert nd == 4 and dims == (64, 1, 3, 3)
    assert np.array_equal(np.frombuffer(b[52 + nl:52 + nl + 4 * 576], np.float32), w["conv1a"][0].reshape(-1))
    q = str(tmp_path / "nv.d2fw")
    save_netvlad_d2fw(q, nvm.synthetic_netvlad_weights())
    assert os.path.getsize(q) > 4 * 1280 * 128


CASES = [
    # name, H, W, camera, camera_configuration, max_keypoints, with NetVLAD
    ("d435_pinhole_640x480", 480, 640, PINHOLE, 0, 200, True),
    ("fisheye_crop_mei_640x480", 480, 640, MEI, 0, 150, True),
    ("mei_past_its_domain_nan_skip", 480, 640, MEI_NAN, 0, 200, False),
    ("stereo_fisheye_mask", 480, 640, PINHOLE, 1, 100, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,H,W,cam,cfg,N,with_nv", CASES, ids=[c[0] for c in CASES])
def test_adapter_under_the_reference_caller_equals_reference_infer(orc, tmp_path, name, H, W, cam, cfg, N, with_nv):
    """Every field of the two VisualImageDesc: the reference's caller over include/d2fe_adapter.cpp + libd2fe_hip.so (exact fp32 mode) against the same caller over
    the reference's own SuperPoint::infer on the oracle's network outputs.  Landmark count and order, pt2d, pt3d_norm bitwise; landmark_descriptor <= 1e-6;
    landmark_scores bitwise; image_desc <= 1e-4; the NaN-skip misalignment included."""
    from d2slam_amd import api, netvlad as nvm
    w = synthetic_superpoint_weights(dustbin_bias=7.5)
    nv = nvm.synthetic_netvlad_weights()
    sp_path, nv_path = str(tmp_path / "sp.d2fw"), str(tmp_path / "nv.d2fw")
    save_superpoint_d2fw(sp_path, w)
    save_netvlad_d2fw(nv_path, nv)
    if name.startswith("fisheye"):
        # a crop of the reference's own sample image where the golden file carries it, else a synthetic frame
        gp = os.path.join(os.path.dirname(__file__), "golden", "reference_headline.npz")
        img = None
        if os.path.exists(gp):
            z = np.load(gp)
            k = next((k for k in z.files if z[k].dtype == np.uint8 and z[k].ndim >= 2 and z[k].shape[-2:] == (H, W)), None)
            if k is not None:
                img = np.ascontiguousarray(z[k].reshape(-1, H, W)[0])
        if img is None:
            img = synth_image(H, W, 23)
    else:
        img = synth_image(H, W, 31)
    masked = img.copy()
    if cfg == 1:
        masked[H * 3 // 4:] = 0
    cams = (PINHOLE, cam)
    hip = ref.LoopCam("hip", W, H, N, self_id=2, camera_configuration=cfg, cams=cams, sp_path=sp_path, nv_path=nv_path if with_nv else None, precision=api.PREC_F32)
    rf = ref.LoopCam("ref", W, H, N, self_id=2, camera_configuration=cfg, cams=cams)
    f = orc.superpoint_forward(masked, w)
    g = orc.netvlad_forward(masked, nv) if with_nv else None
    rf.set_network_outputs(f["semi"], f["desc"], g)
    wa, wb = img.copy(), img.copy()
    a = hip.extract(wa, stamp=1234.5, camera_index=1, camera_id=2001)
    b = rf.extract(wb, stamp=1234.5, camera_index=1, camera_id=2001)
    assert np.array_equal(wa, masked) and np.array_equal(wb, masked)
    for k in ("stamp", "camera_index", "camera_id", "drone_id", "n_landmarks", "nan_warnings"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["n_landmarks"] > 20
    if name.startswith("mei_past"):
        assert 0 < a["nan_warnings"] and a["n_landmarks"] + a["nan_warnings"] == len(a["landmark_scores"])
    else:
        assert a["nan_warnings"] == 0 and a["n_landmarks"] == N
    for k in ("pt2d", "pt3d_norm", "color", "lm_camera_index", "lm_camera_id", "lm_stamp", "lm_stamp_discover", "landmark_scores"):
        assert np.array_equal(a[k], b[k]), k
    assert a["landmark_descriptor"].shape == b["landmark_descriptor"].shape == (256 * len(a["landmark_scores"]),)
    assert np.abs(a["landmark_descriptor"] - b["landmark_descriptor"]).max() <= 1e-6
    if with_nv:
        assert a["image_desc"].shape == b["image_desc"].shape == (4096,) and np.abs(a["image_desc"] - b["image_desc"]).max() <= 1e-4
    else:
        assert len(a["image_desc"]) == 0                              # no NetVLAD object: cnn_use_onnx = false
    # superpoint_mode = true skips NetVLAD on both sides (loop_cam.cpp:612-616); a second call APPENDS nothing from the first (fresh vframe)
    a2 = hip.extract(img.copy(), stamp=1.0, camera_index=1, camera_id=2001, superpoint_mode=True)
    assert len(a2["image_desc"]) == 0 and a2["n_landmarks"] == a["n_landmarks"] and np.array_equal(a2["pt2d"], a["pt2d"])
    hip.close(); rf.close()
