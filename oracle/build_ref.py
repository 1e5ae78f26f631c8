"""oracle/build_ref.py -- builds oracle/_ref/libspref.so from the reference's OWN C++ (test infrastructure).

The hot path's post-processing and matching glue in the reference depend on nothing but a handful of Eigen / OpenCV
types, so the line ranges below are compiled exactly where they lie under /root/reference against the stand-in
headers of oracle/ref_shim/ (what is and is not pinned by this: oracle/ref_shim/README.md).  Nothing is copied into
the repository: the extracted text exists only in a temporary directory for the duration of the g++ call, and the
only output is oracle/_ref/libspref.so (git-ignored, NOT gpurun-ignored: it travels to the GPU box like the product's
own .so).  /root/reference does not exist on the GPU box; there the prebuilt library is used as is.

Each range carries anchors (text that must appear on its first and last line) so that a reference tree that differs
from the surveyed one (HKUST-Aerial-Robotics/D2SLAM @ 2024-12-18) fails the build instead of compiling something else.
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("D2FE_REFERENCE", "/root/reference")
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libspref.so")
SHIM = os.path.join(HERE, "ref_shim")

# macro -> (file under REF, first line, last line, anchor on first line, anchor on last line)
RANGES = {
    "SPREF_GEN_TENSORRT_INFER": ("d2frontend/src/CNN/superpoint_tensorrt.cpp", 161, 183, "bool SuperPoint::infer(const cv::Mat & input", "}"),
    "SPREF_GEN_TENSORRT_POST": ("d2frontend/src/CNN/superpoint_tensorrt.cpp", 200, 350, "//replace to NMS", "}"),
    "SPREF_GEN_COMMON_KPS": ("d2frontend/src/CNN/superpoint_common.cpp", 8, 40, "namespace D2FrontEnd {", "}"),
    "SPREF_GEN_COMMON_NMS": ("d2frontend/src/CNN/superpoint_common.cpp", 101, 178, "bool pt_conf_comp(", "}  // namespace D2FrontEnd"),
    "SPREF_GEN_MATCHER": ("d2frontend/src/feature_matcher.cpp", 3, 43, "namespace D2FrontEnd {", "}"),
    "SPREF_GEN_HALFIMG": ("d2frontend/src/d2featuretracker.cpp", 1051, 1075, "getFeatureHalfImg(", "}"),
    "SPREF_GEN_NEIGHBOUR": ("d2frontend/src/d2featuretracker.cpp", 1146, 1181, "std::map<int, int> tmp_to_idx_a, tmp_to_idx_b;", "}"),
    # round 3 (ref_shim/spref_api2.cpp): the reference-owned code either side of the hot path
    "SPREF_GEN_CODEC_Q_LM": ("d2common/include/d2common/d2frontend_types.h", 230, 237, "Eigen::Map<const VectorXf> desc0(landmark_descriptor.data()", "}"),
    "SPREF_GEN_CODEC_Q_NV": ("d2common/include/d2common/d2frontend_types.h", 262, 268, "img_desc.header.image_desc_size_int8 = image_desc.size();", "}"),
    "SPREF_GEN_CODEC_DEQ": ("d2common/include/d2common/d2frontend_types.h", 319, 341, "if (desc.landmark_descriptor_int8.size() > 0) {", "}"),
    "SPREF_GEN_DB_QUERY": ("d2frontend/src/loop_detector.cpp", 300, 350, "int LoopDetector::queryIndexFromDatabase(", "}"),
    "SPREF_GEN_TRACKER_GATE": ("d2frontend/src/d2featuretracker.cpp", 166, 235, "bool D2FeatureTracker::getMatchedPrevKeyframe(", "}"),
    "SPREF_GEN_TRACKER_DIRS": ("d2frontend/src/d2featuretracker.cpp", 270, 284, "int max_dirs = 4;", "}"),
    "SPREF_GEN_LOOPCAM_MATCH": ("d2frontend/src/loop_cam.cpp", 156, 191, "void matchLocalFeatures(", "}"),
    # round 3 (ref_shim/spref_api3.cpp): undistortion map generation = camodocal (vendored camera_models/) + FisheyeUndist::genOneUndistMap
    "SPREF_GEN_CATA_INVK": ("camera_models/src/camera_models/CataCamera.cc", 221, 224, "m_inv_K11 = 1.0 / mParameters.gamma1();", "m_inv_K23 = -mParameters.v0() / mParameters.gamma2();"),
    "SPREF_GEN_CATA_LIFT": ("camera_models/src/camera_models/CataCamera.cc", 425, 487, "void CataCamera::liftProjective(const Eigen::Vector2d& p,", "}"),
    "SPREF_GEN_CATA_SPACE": ("camera_models/src/camera_models/CataCamera.cc", 495, 515, "void CataCamera::spaceToPlane(const Eigen::Vector3d& P,", "}"),
    "SPREF_GEN_CATA_DIST": ("camera_models/src/camera_models/CataCamera.cc", 617, 633, "void CataCamera::distortion(const Eigen::Vector2d& p_u,", "}"),
    "SPREF_GEN_CYL_INVK": ("camera_models/src/camera_models/CylindricalCamera.cc", 144, 147, "m_inv_K11 = 1.0 / mParameters.fx();", "m_inv_K23 = -mParameters.cy() / mParameters.fy();"),
    "SPREF_GEN_CYL_LIFT": ("camera_models/src/camera_models/CylindricalCamera.cc", 207, 220, "void CylindricalCamera::liftProjective(const Eigen::Vector2d& p,", "}"),
    "SPREF_GEN_MAP_LOOP_VCAM": ("d2common/include/d2common/fisheye_undistort.h", 571, 579, "for (unsigned int x = 0; x < imgWidth; x++)", "cv::Vec2f(imgPoint.x(), imgPoint.y());"),
    # round 3 (ref_shim/spref_api4.cpp): the LK tracker's glue around the (absent) OpenCV-CUDA optical flow
    "SPREF_GEN_LK_INFO": ("d2frontend/include/d2frontend/opticaltrack_utils.h", 16, 26, "template <typename T> struct LKImageInfo {", "using LKImageInfoGPU = LKImageInfo<cv::cuda::GpuMat>;"),
    "SPREF_GEN_REDUCE_VECTOR": ("d2frontend/include/d2frontend/utils.h", 21, 28, "template <typename T, typename B>", "}"),
    "SPREF_GEN_LK_INBORDER": ("d2frontend/src/opticaltrack_utils.cpp", 35, 41, "bool inBorder(const cv::Point2f &pt, cv::Size shape) {", "}"),
    "SPREF_GEN_LK_TRACKPYR": ("d2frontend/src/opticaltrack_utils.cpp", 173, 278, "LKImageInfoGPU opticalflowTrackPyr(const cv::Mat &cur_img,", "}"),
    "SPREF_GEN_MAP_LOOP_PINHOLE": ("d2common/include/d2common/fisheye_undistort.h", 627, 638, "for (unsigned int x = 0; x < imgWidth; x++)", "cv::Vec2f(imgPoint.x(), imgPoint.y());"),
}


# round 6 (ref_shim/spref_loopcam.cpp): the reference's own CALLER of the extractor, LoopCam::extractorImgDescDeepnet, with the structs it fills -- compiled
# twice: over include/d2fe_adapter.cpp + libd2fe_hip.so, and over the reference's own SuperPoint::infer post-processing (ranges above)
LOOPCAM_RANGES = {
    "SPREF_GEN_LOOPCAM_EXTRACT": ("d2frontend/src/loop_cam.cpp", 589, 648, "VisualImageDesc LoopCam::extractorImgDescDeepnet(ros::Time stamp, cv::Mat img,", "}"),
    "SPREF_GEN_VISUAL_IMAGE_DESC": ("d2common/include/d2common/d2frontend_types.h", 85, 110, "struct VisualImageDesc {", "double cur_td = 0;"),
    "SPREF_GEN_LANDMARK_PER_FRAME": ("d2common/include/d2common/d2landmarks.h", 28, 70, "struct LandmarkPerFrame {", "{}"),
    "SPREF_GEN_BASETYPES_IDS": ("d2common/include/d2common/d2basetypes.h", 18, 20, "typedef int64_t FrameIdType;", "typedef int32_t CamIdType;"),
    "SPREF_GEN_BASETYPES_CAMCFG": ("d2common/include/d2common/d2basetypes.h", 40, 46, "enum CameraConfig{", "};"),
    "SPREF_GEN_EXTRACT_COLOR": ("d2frontend/src/loop_utils.cpp", 54, 63, "cv::Vec3b extractColor(const cv::Mat &img, cv::Point2f p) {", "}"),
}
LOOPCAM_SHARED = ("SPREF_GEN_TENSORRT_INFER", "SPREF_GEN_TENSORRT_POST", "SPREF_GEN_CATA_INVK", "SPREF_GEN_CATA_LIFT", "SPREF_GEN_CATA_DIST")
LOOPCAM_LIBS = {"hip": os.path.join(OUT_DIR, "libspref_loopcam_hip.so"), "ref": os.path.join(OUT_DIR, "libspref_loopcam_ref.so")}
PRODUCT_LIBDIR = os.path.join(os.path.dirname(HERE), "d2slam_amd", "lib")


def _write_ranges(tmp, ranges):
    defs = []
    for macro, (rel, a, b, anchor_a, anchor_b) in ranges.items():
        path = os.path.join(REF, rel)
        with open(path, encoding="utf-8") as f:
            lines = f.read().split("\n")
        first, last = lines[a - 1], lines[b - 1]
        if anchor_a not in first or anchor_b not in last:
            raise RuntimeError("%s:%d-%d does not look like the surveyed reference (anchors %r / %r not found in %r / %r)" % (rel, a, b, anchor_a, anchor_b, first, last))
        inc = os.path.join(tmp, macro.lower() + ".inc")
        with open(inc, "w", encoding="utf-8") as f:
            f.write('#line %d "%s"\n' % (a, path))
            f.write("\n".join(lines[a - 1:b]) + "\n")
        defs.append('-D%s="%s"' % (macro, inc))
    return defs


def build_loopcam(force=False, verbose=False):
    """oracle/_ref/libspref_loopcam_{hip,ref}.so: LoopCam::extractorImgDescDeepnet (loop_cam.cpp:589-648) over include/d2fe_adapter.cpp + libd2fe_hip.so and over the
    reference's own SuperPoint::infer.  Returns {side: path}; sides that cannot be built here (no reference tree and no prebuilt library; the HIP side also
    needs d2slam_amd/lib/libd2fe_hip.so to link against) are absent."""
    out = {}
    if not available():
        return {k: v for k, v in LOOPCAM_LIBS.items() if os.path.exists(v)}
    src = os.path.join(SHIM, "spref_loopcam.cpp")
    adapter = [os.path.join(os.path.dirname(HERE), "include", f) for f in ("d2fe_adapter.cpp", "d2fe_weights_file.hpp", "d2fe.h")]
    ranges = dict(LOOPCAM_RANGES, **{k: RANGES[k] for k in LOOPCAM_SHARED})
    deps = [src, os.path.abspath(__file__)] + adapter + [os.path.join(REF, r[0]) for r in ranges.values()]
    for root, _, files in os.walk(SHIM):
        deps += [os.path.join(root, f) for f in files if f.endswith((".h", ".hpp"))]
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="spref_lc_") as tmp:
        defs = None
        for side, lib in LOOPCAM_LIBS.items():
            if side == "hip" and not os.path.exists(os.path.join(PRODUCT_LIBDIR, "libd2fe_hip.so")):
                continue
            if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps if os.path.exists(d)):
                out[side] = lib
                continue
            defs = defs or _write_ranges(tmp, ranges)
            cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-w", "-DLOOPCAM_SIDE_%s" % side.upper(),
                   "-I" + SHIM, "-I" + os.path.join(REF, "d2frontend", "include")] + defs + [src, "-o", lib]
            if side == "hip":
                cmd += ["-L" + PRODUCT_LIBDIR, "-ld2fe_hip", "-Wl,-rpath,$ORIGIN/../../d2slam_amd/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("building %s failed:\n%s" % (lib, (r.stdout + r.stderr)[-6000:]))
            out[side] = lib
    return out


TORCH_LIB = os.path.join(OUT_DIR, "libspref_torch.so")
TORCH_RANGE = ("SPREF_GEN_COMPUTE_DESC", "d2frontend/src/CNN/superpoint_common.cpp", 42, 99, "void computeDescriptors(const torch::Tensor& mProb, const torch::Tensor& mDesc,", "}")


def build_torch(force=False, verbose=False):
    """oracle/_ref/libspref_torch.so: computeDescriptors (superpoint_common.cpp:42-99) against the image's libtorch.  Returns the path or None
    (no reference tree and no prebuilt library, or no torch C++ headers)."""
    if not available():
        return TORCH_LIB if os.path.exists(TORCH_LIB) else None
    src = os.path.join(SHIM, "spref_torch.cpp")
    deps = [src, os.path.join(SHIM, "torch_eigen", "Eigen", "Eigen"), os.path.join(REF, TORCH_RANGE[1]), os.path.abspath(__file__)]
    if not force and os.path.exists(TORCH_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(TORCH_LIB) for d in deps):
        return TORCH_LIB
    try:
        import torch
        from torch.utils import cpp_extension as ce
    except Exception:
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    macro, rel, a, b, anchor_a, anchor_b = TORCH_RANGE
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        lines = f.read().split("\n")
    if anchor_a not in lines[a - 1] or anchor_b not in lines[b - 1]:
        raise RuntimeError("%s:%d-%d does not look like the surveyed reference" % (rel, a, b))
    with tempfile.TemporaryDirectory(prefix="spref_t_") as tmp:
        inc = os.path.join(tmp, "compute_desc.inc")
        with open(inc, "w", encoding="utf-8") as f:
            f.write('#line %d "%s"\n' % (a, os.path.join(REF, rel)))
            f.write("\n".join(lines[a - 1:b]) + "\n")
        libdir = ce.library_paths()[0]
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-w",
               "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), '-D%s="%s"' % (macro, inc),
               "-I" + os.path.join(SHIM, "torch_eigen"), "-I" + SHIM] + ["-I" + p for p in ce.include_paths()] + \
              [src, "-L" + libdir, "-ltorch_cpu", "-lc10", "-ltorch", "-Wl,-rpath," + libdir, "-o", TORCH_LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building oracle/_ref/libspref_torch.so failed:\n" + r.stdout[-3000:] + r.stderr[-6000:])
    return TORCH_LIB


def available():
    return os.path.isdir(os.path.join(REF, "d2frontend", "src"))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.abspath(__file__)]
    for root, _, files in os.walk(SHIM):
        deps += [os.path.join(root, f) for f in files]
    deps += [os.path.join(REF, r[0]) for r in RANGES.values()]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Returns the library path, or None when /root/reference is absent and no prebuilt library exists."""
    if not available():
        return LIB if os.path.exists(LIB) else None
    if not force and not needs_build():
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="spref_") as tmp:
        defs = []
        for macro, (rel, a, b, anchor_a, anchor_b) in RANGES.items():
            path = os.path.join(REF, rel)
            with open(path, encoding="utf-8") as f:
                lines = f.read().split("\n")
            first, last = lines[a - 1], lines[b - 1]
            if anchor_a not in first or anchor_b not in last:
                raise RuntimeError("%s:%d-%d does not look like the surveyed reference (anchors %r / %r not found in %r / %r)"
                                   % (rel, a, b, anchor_a, anchor_b, first, last))
            inc = os.path.join(tmp, macro.lower() + ".inc")
            with open(inc, "w", encoding="utf-8") as f:
                f.write('#line %d "%s"\n' % (a, path))
                f.write("\n".join(lines[a - 1:b]) + "\n")
            defs.append('-D%s="%s"' % (macro, inc))
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-w",
               "-I" + SHIM, "-I" + os.path.join(REF, "d2frontend", "include")] + defs + \
              [os.path.join(SHIM, "spref_api.cpp"), os.path.join(SHIM, "spref_api2.cpp"), os.path.join(SHIM, "spref_api3.cpp"), os.path.join(SHIM, "spref_api4.cpp"), "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building oracle/_ref/libspref.so failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
    if "--torch" in sys.argv:
        print(build_torch(force="-f" in sys.argv, verbose=True))
    if "--loopcam" in sys.argv:
        print(build_loopcam(force="-f" in sys.argv, verbose=True))
