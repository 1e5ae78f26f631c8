"""Host-side mirror of the reference's operator interface for the feature hot path, on top of the C ABI
(include/d2fe.h, libd2fe_hip.so).

Mirrors (names / argument meaning / error behaviour):
  * SuperPointConfig + SuperPoint.build()/infer()   d2frontend/include/d2frontend/CNN/superpoint_tensorrt.h:17-48
  * matchKNN(desc_a, desc_b, knn_match_ratio, pts_a, pts_b, search_local_dist)   d2frontend/include/d2frontend/feature_matcher.h:6-11
  * cv::BFMatcher(NORM_L2, crossCheck=True).match    loop_cam.cpp:167-170
  * getFeatureHalfImg                                d2frontend/src/d2featuretracker.cpp:1051-1075

There is NO CPU fallback: if libd2fe_hip.so is missing or no GPU is visible these calls raise.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .weights import SP_LAYERS

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libd2fe_hip.so")
DEV_LIB_PATH = os.path.join(_HERE, "lib", "libd2fe_hip_dev.so")      # -DD2FE_DEVTOOLS: d2fe_debug_* test hooks, phase stamps, D2FE_* schedule switches
if os.environ.get("D2FE_LIB"):      # developer knob: same-box A/B of two builds of the library (tools/gpu_run.sh ab)
    LIB_PATH = os.path.abspath(os.environ["D2FE_LIB"])

POSTPROC_B, POSTPROC_A = 0, 1
PREC_F32, PREC_F16X2, PREC_F32_WINO = 0, 1, 2
ERR_TRUNCATED = -4            # d2fe_status: output capacity too small, n_out holds what was written
KEEP_ALL_CAP = 1024           # host-pointer staging capacity of a keep-all handle (max_keypoints = -1)
PROF_STAGES = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPaDa",
               "convPb", "convDb", "softmax_cand", "select", "sample", "match", "netvlad"]


class D2FEError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("d2fe error %d: %s" % (code, msg))
        self.code = code


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device_id", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_batch", C.c_int32), ("max_keypoints", C.c_int32),
                ("remove_borders", C.c_int32), ("keypoint_threshold", C.c_float), ("postproc", C.c_int32),
                ("nms_dist", C.c_int32), ("precision", C.c_int32), ("keep_score_map", C.c_int32),
                ("dense_descriptors", C.c_int32), ("async_tail", C.c_int32), ("reserved", C.c_int32 * 5)]


class _ConvParams(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32),
                ("ksize", C.c_int32)]


class _SPWeights(C.Structure):
    _fields_ = [("layer", _ConvParams * 12)]


class _NvLayer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
                ("res", C.c_int32), ("weight", C.c_void_p), ("bias", C.c_void_p)]


class _NvWeights(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("layers", C.c_void_p), ("feat_dim", C.c_int32), ("proj_dim", C.c_int32),
                ("n_clusters", C.c_int32), ("pre_w", C.c_void_p), ("pre_b", C.c_void_p), ("assign_w", C.c_void_p),
                ("assign_b", C.c_void_p), ("centroids", C.c_void_p)]


class _MatchBatch(C.Structure):
    _fields_ = [("d_a", C.c_void_p), ("d_b", C.c_void_p), ("d_pts_a", C.c_void_p), ("d_pts_b", C.c_void_p),
                ("d_a_off", C.c_void_p), ("d_b_off", C.c_void_p), ("d_a_cnt", C.c_void_p), ("d_b_cnt", C.c_void_p),
                ("npairs", C.c_int32), ("dim", C.c_int32), ("max_n", C.c_int32), ("mode", C.c_int32),
                ("ratio", C.c_double), ("radius", C.c_double),
                ("d_q_idx", C.c_void_p), ("d_t_idx", C.c_void_p), ("d_dist", C.c_void_p), ("d_n_out", C.c_void_p)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)      # d2fe_all_gather_fn


class _ExchangeConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("world", C.c_int32), ("rank", C.c_int32), ("wire", C.c_int32), ("loopback", C.c_int32), ("slots", C.c_int32),
                ("own_stream", C.c_int32), ("timing", C.c_int32), ("gate_thres", C.c_double), ("ratio", C.c_double), ("all_gather", ALL_GATHER_FN),
                ("all_gather_user", C.c_void_p), ("reserved", C.c_int32 * 6)]


class _ExchangeResult(C.Structure):
    _fields_ = [("ticket", C.c_int64), ("npairs", C.c_int32), ("cap", C.c_int32), ("q_idx", C.c_void_p), ("t_idx", C.c_void_p), ("dist", C.c_void_p),
                ("n_match", C.c_void_p), ("gate_pass", C.c_void_p), ("gate_sims", C.c_void_p), ("gate_n", C.c_int32), ("phase_ms", C.c_float * 5)]


class _PipeConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("lanes", C.c_int32), ("frames", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("cap", C.c_int32), ("netvlad", C.c_int32), ("match_lr", C.c_int32), ("match_prev", C.c_int32), ("pinned_input", C.c_int32),
                ("ratio", C.c_double), ("radius_lr", C.c_double), ("radius_prev", C.c_double), ("cu_partition", C.c_int32), ("netvlad_inline", C.c_int32), ("coalesce", C.c_int32), ("lane_cus", C.c_int32), ("netvlad_group", C.c_int32), ("coalesce_depth", C.c_int32), ("reserved", C.c_int32 * 2)]


class _PipeResult(C.Structure):
    _fields_ = [("frames", C.c_int32), ("cap", C.c_int32), ("desc_dim", C.c_int32), ("netvlad_dim", C.c_int32),
                ("kps_xy", C.c_void_p), ("scores", C.c_void_p), ("desc", C.c_void_p), ("n_kp", C.c_void_p), ("netvlad", C.c_void_p),
                ("lr_q", C.c_void_p), ("lr_t", C.c_void_p), ("lr_dist", C.c_void_p), ("lr_n", C.c_void_p),
                ("prev_q", C.c_void_p), ("prev_t", C.c_void_p), ("prev_dist", C.c_void_p), ("prev_n", C.c_void_p)]


class _PipeDeviceResult(C.Structure):
    _fields_ = [("frames", C.c_int32), ("cap", C.c_int32), ("desc_dim", C.c_int32), ("netvlad_dim", C.c_int32),
                ("d_kps_xy", C.c_void_p), ("d_scores", C.c_void_p), ("d_desc", C.c_void_p), ("d_n_kp", C.c_void_p), ("d_netvlad", C.c_void_p)]


_lib = None
_dev_lib = None

EXPORTS = [
    "d2fe_last_error", "d2fe_version", "d2fe_default_config", "d2fe_create", "d2fe_destroy", "d2fe_load_superpoint", "d2fe_set_superpoint_pca",
    "d2fe_desc_dim", "d2fe_superpoint_extract", "d2fe_superpoint_extract_batch", "d2fe_superpoint_extract_device", "d2fe_extract_all",
    "d2fe_extract_all_batch", "d2fe_tail_stream", "d2fe_superpoint_wait_tail", "d2fe_load_netvlad", "d2fe_set_netvlad_pca", "d2fe_netvlad_dim",
    "d2fe_netvlad", "d2fe_netvlad_batch", "d2fe_netvlad_device", "d2fe_match_knn", "d2fe_match_crosscheck", "d2fe_match_batch_device",
    "d2fe_match_fallback_rows", "d2fe_block_words", "d2fe_block_field_offset", "d2fe_pack_blocks_device", "d2fe_gate_pairs_device",
    "d2fe_quad_gate_device", "d2fe_block_bytes_int8", "d2fe_pack_blocks_int8_device", "d2fe_unpack_blocks_int8_device", "d2fe_half_move_cols",
    "d2fe_half_image_compact_device", "d2fe_remap_matches_device", "d2fe_half_image_filter", "d2fe_undistort", "d2fe_undistort_device",
    "d2fe_db_create", "d2fe_db_destroy", "d2fe_db_ntotal", "d2fe_db_add", "d2fe_db_search", "d2fe_db_query_gated", "d2fe_quantize_int8",
    "d2fe_dequantize_int8", "d2fe_sync", "d2fe_profile_enable", "d2fe_profile_read", "d2fe_prepare_gray", "d2fe_prepare_gray_device",
    "d2fe_gen_cylinder_map", "d2fe_gen_cylinder_map_device", "d2fe_gen_pinhole_map", "d2fe_gen_pinhole_map_device", "d2fe_lk_frame_create",
    "d2fe_lk_frame_create_device", "d2fe_lk_frame_destroy", "d2fe_lk_frame_read_level", "d2fe_lk_track", "d2fe_lk_track_batch",
    "d2fe_detect_fast_by_region", "d2fe_good_features_to_track", "d2fe_pipe_default_config", "d2fe_pipe_create", "d2fe_pipe_destroy",
    "d2fe_pipe_lanes", "d2fe_pipe_stream_placement", "d2fe_pipe_classify_stream", "d2fe_pipe_submit", "d2fe_pipe_wait", "d2fe_pipe_profile_enable", "d2fe_pipe_profile_read",
    "d2fe_pipe_device_view", "d2fe_pipe_device_release", "d2fe_pipe_lane_stream", "d2fe_pipe_geometry", "d2fe_pipe_handle",
    "d2fe_exchange_default_config", "d2fe_exchange_create", "d2fe_exchange_destroy", "d2fe_exchange_enqueue", "d2fe_exchange_collect", "d2fe_exchange_pairs",
    "d2fe_exchange_block_bytes", "d2fe_exchange_stream", "d2fe_rccl_load", "d2fe_rccl_path", "d2fe_rccl_unique_id", "d2fe_rccl_comm_init_rank", "d2fe_rccl_comm_destroy"]
# the development library (lib/libd2fe_hip_dev.so, include/d2fe_debug.h) exports these on top: test hooks and kernel diagnostics
DEBUG_EXPORTS = [
    "d2fe_debug_graph_count", "d2fe_debug_read", "d2fe_debug_netvlad_layer", "d2fe_debug_netvlad_stamps", "d2fe_debug_pack_wino",
    "d2fe_debug_pack_netvlad", "d2fe_debug_netvlad_tile", "d2fe_debug_conv3x3_wino", "d2fe_debug_match_stamps"]


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's): whichever copy
    is loaded first serves both, and torch cannot see the GPU through the system copy.  If torch is installed but not yet
    imported, load ITS runtime first (located without importing torch) so that a later `import torch` stays consistent."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load_library(dev=False):
    """dlopen the C-ABI library (dev=True: the development library with the d2fe_debug_* hooks, include/d2fe_debug.h).  Raises if it has not
    been built (python -m d2slam_amd.build [--dev])."""
    global _lib, _dev_lib
    if dev:
        if _dev_lib is None:
            _dev_lib = _open_library(DEV_LIB_PATH, True)
        return _dev_lib
    if _lib is None:
        _lib = _open_library(LIB_PATH, False)
    return _lib


def _open_library(path, dev):
    if True:
        _preload_hip_runtime()
        if not os.path.exists(path):
            raise D2FEError(-100, "%s not built: run `python __graft_entry__.py` or `python -m d2slam_amd.build%s` "
                                  "(no CPU fallback exists)" % (os.path.basename(path), " --dev" if dev else ""))
        lib = C.CDLL(path)
        lib.d2fe_last_error.restype = C.c_char_p
        lib.d2fe_version.restype = C.c_char_p
        if dev:
            lib.d2fe_debug_read.restype = C.c_long
            lib.d2fe_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
            for nm in ("d2fe_debug_netvlad_layer", "d2fe_debug_netvlad_stamps", "d2fe_debug_pack_wino", "d2fe_debug_pack_netvlad", "d2fe_debug_match_stamps"):
                getattr(lib, nm).restype = C.c_long
        lib.d2fe_destroy.restype = None
        lib.d2fe_half_move_cols.restype = C.c_float
        lib.d2fe_half_move_cols.argtypes = [C.c_int, C.c_double]
        lib.d2fe_default_config.restype = None
        lib.d2fe_superpoint_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                      C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.d2fe_superpoint_extract_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                       C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_int, C.c_void_p, C.c_void_p]
        lib.d2fe_extract_all_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.d2fe_extract_all.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p]
        lib.d2fe_superpoint_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.d2fe_match_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p]
        lib.d2fe_match_crosscheck.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.d2fe_match_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.d2fe_half_image_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        lib.d2fe_create.argtypes = [C.c_void_p, C.c_void_p]
        lib.d2fe_destroy.argtypes = [C.c_void_p]
        lib.d2fe_load_superpoint.argtypes = [C.c_void_p, C.c_void_p]
        lib.d2fe_sync.argtypes = [C.c_void_p]
        lib.d2fe_match_fallback_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.d2fe_match_fallback_rows.restype = C.c_long
        lib.d2fe_tail_stream.argtypes = [C.c_void_p]
        lib.d2fe_tail_stream.restype = C.c_void_p
        lib.d2fe_superpoint_wait_tail.argtypes = [C.c_void_p, C.c_void_p]
        lib.d2fe_undistort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_undistort_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.d2fe_db_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_db_destroy.argtypes = [C.c_void_p]
        lib.d2fe_db_destroy.restype = None
        lib.d2fe_db_ntotal.argtypes = [C.c_void_p]
        lib.d2fe_db_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.d2fe_db_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.d2fe_db_query_gated.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        lib.d2fe_quantize_int8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_dequantize_int8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_load_netvlad.argtypes = [C.c_void_p, C.c_void_p]
        lib.d2fe_set_netvlad_pca.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.d2fe_netvlad_dim.argtypes = [C.c_void_p]
        lib.d2fe_netvlad_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        lib.d2fe_netvlad.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_netvlad_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.d2fe_set_superpoint_pca.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.d2fe_desc_dim.argtypes = [C.c_void_p]
        lib.d2fe_profile_enable.argtypes = [C.c_void_p, C.c_int]
        lib.d2fe_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.d2fe_prepare_gray.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_prepare_gray_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int,
                                                 C.c_int, C.c_void_p, C.c_void_p]
        lib.d2fe_gen_cylinder_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        lib.d2fe_gen_cylinder_map_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.d2fe_gen_pinhole_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        lib.d2fe_gen_pinhole_map_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                                    C.c_void_p, C.c_void_p]
        lib.d2fe_lk_frame_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_lk_frame_create_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.d2fe_lk_frame_destroy.argtypes = [C.c_void_p]
        lib.d2fe_lk_frame_destroy.restype = None
        lib.d2fe_lk_frame_read_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.d2fe_lk_frame_read_level.restype = C.c_long
        lib.d2fe_lk_track.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.d2fe_lk_track_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p]
        lib.d2fe_detect_fast_by_region.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                   C.c_void_p, C.c_int, C.c_void_p]
        lib.d2fe_good_features_to_track.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p,
                                                    C.c_int, C.c_void_p]
        lib.d2fe_pipe_default_config.argtypes = [C.c_void_p]
        lib.d2fe_pipe_default_config.restype = None
        lib.d2fe_pipe_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.d2fe_pipe_destroy.argtypes = [C.c_void_p]
        lib.d2fe_pipe_destroy.restype = None
        lib.d2fe_pipe_lanes.argtypes = [C.c_void_p]
        lib.d2fe_pipe_stream_placement.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.d2fe_pipe_classify_stream.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        lib.d2fe_pipe_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        lib.d2fe_pipe_wait.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.d2fe_pipe_device_view.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        lib.d2fe_pipe_device_release.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.d2fe_pipe_profile_enable.argtypes = [C.c_void_p, C.c_int]
        lib.d2fe_pipe_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.d2fe_pipe_lane_stream.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.d2fe_pipe_geometry.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.d2fe_pipe_handle.argtypes = [C.c_void_p]; lib.d2fe_pipe_handle.restype = C.c_void_p
        lib.d2fe_exchange_default_config.argtypes = [C.c_void_p]; lib.d2fe_exchange_default_config.restype = None
        lib.d2fe_exchange_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.d2fe_exchange_destroy.argtypes = [C.c_void_p]; lib.d2fe_exchange_destroy.restype = None
        lib.d2fe_exchange_enqueue.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        lib.d2fe_exchange_collect.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.d2fe_exchange_pairs.argtypes = [C.c_void_p]
        lib.d2fe_exchange_block_bytes.argtypes = [C.c_void_p]
        lib.d2fe_exchange_stream.argtypes = [C.c_void_p]; lib.d2fe_exchange_stream.restype = C.c_void_p
        lib.d2fe_rccl_load.argtypes = [C.c_char_p]
        lib.d2fe_rccl_path.restype = C.c_char_p
        lib.d2fe_rccl_unique_id.argtypes = [C.c_void_p]
        lib.d2fe_rccl_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.d2fe_rccl_comm_destroy.argtypes = [C.c_void_p]
    return lib


def _check(rc):
    if rc != 0:
        raise D2FEError(rc, load_library().d2fe_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class SuperPointConfig:
    """Field-for-field SuperPointConfig (superpoint_tensorrt.h:17-33); TensorRT-only fields are kept for
    source compatibility and ignored (onnx_path/engine_path/tensor names/dla_core/fp_16)."""
    max_keypoints: int = 100
    remove_borders: int = 1
    dla_core: int = 0
    fp_16: int = 0
    input_width: int = 640
    input_height: int = 480
    superpoint_pca_dims: int = -1
    keypoint_threshold: float = 0.015
    input_tensor_names: List[str] = field(default_factory=lambda: ["input"])
    output_tensor_names: List[str] = field(default_factory=lambda: ["scores", "descriptors"])
    onnx_path: str = ""
    engine_path: str = ""
    pca_mean_path: str = ""
    pca_comp_path: str = ""
    enable_pca: bool = False
    # device-side additions
    device_id: int = 0
    max_batch: int = 2
    postproc: int = POSTPROC_B
    nms_dist: int = 10
    precision: int = PREC_F32
    keep_score_map: bool = False   # debug: also write the dense score map
    dense_descriptors: bool = False  # debug: dense descriptor map instead of the sparse descriptor head (variant B)
    async_tail: bool = False         # extract_device: post-processing on the handle's tail stream, under the next call's convolutions


class FrontEnd:
    """One device context (== one LoopCam's networks, loop_cam.cpp:24-70)."""

    def __init__(self, cfg: SuperPointConfig, dev: bool = False):
        """dev=True: a handle of the development library (d2fe_debug_* hooks, D2FE_* schedule switches); the product library otherwise."""
        lib = load_library(dev)
        self.dev = bool(dev)
        c = _Config()
        lib.d2fe_default_config(C.byref(c))
        c.device_id = cfg.device_id
        c.max_width = cfg.input_width
        c.max_height = cfg.input_height
        c.max_batch = cfg.max_batch
        c.max_keypoints = cfg.max_keypoints
        c.remove_borders = cfg.remove_borders
        c.keypoint_threshold = cfg.keypoint_threshold
        c.postproc = cfg.postproc
        c.nms_dist = cfg.nms_dist
        c.precision = cfg.precision
        c.keep_score_map = int(cfg.keep_score_map)
        c.dense_descriptors = int(cfg.dense_descriptors)
        c.async_tail = int(cfg.async_tail)
        self.cfg = cfg
        self._h = C.c_void_p()
        _check(lib.d2fe_create(C.byref(c), C.byref(self._h)))
        self._lib = lib
        self._keep = None

    def _need_dev(self, what):
        if not self.dev:
            raise D2FEError(-5, "%s is a test hook of the development library: create the handle with FrontEnd(cfg, dev=True) / DevFrontEnd(cfg) "
                                "(include/d2fe_debug.h; the product library exports no d2fe_debug_* symbol)" % what)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            # pipes run on this handle's packed weights; the library refuses to destroy a handle that still has live pipes (d2fe_destroy), so they go first
            for p in list(getattr(self, "_pipes", ())):
                p.close()
            self._lib.d2fe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def load_superpoint(self, weights):
        """weights: {layer: (W [cout,cin,k,k], b [cout])} for the 12 layers of superpoint.ipynb:306-321."""
        sw = _SPWeights()
        keep = []
        for i, name in enumerate(SP_LAYERS):
            W, b = weights[name]
            W = np.ascontiguousarray(W, np.float32); b = np.ascontiguousarray(b, np.float32)
            keep += [W, b]
            sw.layer[i].weight = W.ctypes.data
            sw.layer[i].bias = b.ctypes.data
            sw.layer[i].cout, sw.layer[i].cin, sw.layer[i].ksize = W.shape[0], W.shape[1], W.shape[2]
        _check(self._lib.d2fe_load_superpoint(self._h, C.byref(sw)))

    def set_pca(self, comp, mean):
        """comp [pca_dims,256] (CSV layout of superpoint_onnx.cpp:47-53), mean [256]; variant A only."""
        if comp is None:
            _check(self._lib.d2fe_set_superpoint_pca(self._h, None, None, 0))
            return
        comp = np.ascontiguousarray(comp, np.float32); mean = np.ascontiguousarray(mean, np.float32)
        _check(self._lib.d2fe_set_superpoint_pca(self._h, _ptr(comp), _ptr(mean), comp.shape[0]))

    @property
    def desc_dim(self):
        return int(self._lib.d2fe_desc_dim(self._h))

    # ---- extractor ---------------------------------------------------------------------------------------------
    def extract_batch(self, images, cap=None):
        """images: u8 [n,H,W].  Returns list of (kps [k,2], scores [k], desc [k,256])."""
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim == 2:
            images = images[None]
        n, H, W = images.shape
        if not cap:      # keep-all handles (max_keypoints = -1) have no configured cap: the library's staging capacity applies
            cap = self.cfg.max_keypoints if self.cfg.max_keypoints > 0 else KEEP_ALL_CAP
        kps = np.zeros((n, cap, 2), np.float32); sc = np.zeros((n, cap), np.float32)
        desc = np.zeros((n, cap, self.desc_dim), np.float32); cnt = np.zeros(n, np.int32)
        rc = self._lib.d2fe_superpoint_extract_batch(self._h, _ptr(images), n, W, H, W, H * W, _ptr(kps), _ptr(sc),
                                                     _ptr(desc), cap, _ptr(cnt))
        # D2FE_ERR_TRUNCATED: the strongest `cap` keypoints were written and n_out is valid -- report it, keep the results
        self.last_truncated = rc == ERR_TRUNCATED
        if rc != 0 and rc != ERR_TRUNCATED:
            _check(rc)
        return [(kps[i, :cnt[i]].copy(), sc[i, :cnt[i]].copy(), desc[i, :cnt[i]].copy()) for i in range(n)]

    def extract_all_batch(self, images, n_netvlad, cap=None):
        """d2fe_extract_all_batch: SuperPoint on all images + NetVLAD on the first n_netvlad, one upload, two streams.
        Returns (list of (kps, scores, desc), netvlad [n_netvlad, G])."""
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim == 2:
            images = images[None]
        n, H, W = images.shape
        if not cap:
            cap = self.cfg.max_keypoints if self.cfg.max_keypoints > 0 else KEEP_ALL_CAP
        kps = np.zeros((n, cap, 2), np.float32); sc = np.zeros((n, cap), np.float32)
        desc = np.zeros((n, cap, self.desc_dim), np.float32); cnt = np.zeros(n, np.int32)
        g = np.zeros((max(n_netvlad, 1), self.netvlad_dim), np.float32)
        rc = self._lib.d2fe_extract_all_batch(self._h, _ptr(images), n, W, H, W, H * W, _ptr(kps), _ptr(sc), _ptr(desc), cap, _ptr(cnt),
                                              int(n_netvlad), _ptr(g))
        self.last_truncated = rc == ERR_TRUNCATED
        if rc != 0 and rc != ERR_TRUNCATED:
            _check(rc)
        return [(kps[i, :cnt[i]].copy(), sc[i, :cnt[i]].copy(), desc[i, :cnt[i]].copy()) for i in range(n)], g[:n_netvlad].copy()

    def graph_count(self):
        """(cached hipGraphs of host-pointer launch sequences, geometries whose capture was rejected) -- include/d2fe.h."""
        self._need_dev("graph_count")
        bad = C.c_int(0)
        self._lib.d2fe_debug_graph_count.argtypes = [C.c_void_p, C.c_void_p]
        n = int(self._lib.d2fe_debug_graph_count(self._h, C.byref(bad)))
        return n, int(bad.value)

    def match_fallback_rows(self, reset=True, full=False):
        """Queries that needed more than the matcher's first four candidates since the last reset (include/d2fe.h); full=True returns
        (that count, the number that took the exact scan of all train rows)."""
        fs = C.c_long(0)
        r = int(self._lib.d2fe_match_fallback_rows(self._h, int(bool(reset)), C.byref(fs)))
        if r < 0:
            _check(r)
        return (r, int(fs.value)) if full else r

    def extract_device(self, d_gray, n, W, H, d_kps, d_scores, d_desc, d_idx, cap, d_n, stream=None, stride=None,
                       image_stride=None):
        """Device-resident form; arguments are raw device addresses (ints)."""
        _check(self._lib.d2fe_superpoint_extract_device(self._h, d_gray, n, W, H, stride or W, image_stride or H * W,
                                                        d_kps, d_scores, d_desc, d_idx, cap, d_n, stream))

    def tail_stream(self):
        """async_tail mode: raw hipStream_t on which the outputs of extract_device become valid (0 when the mode is off)."""
        return int(self._lib.d2fe_tail_stream(self._h) or 0)

    def wait_tail(self, stream=None):
        _check(self._lib.d2fe_superpoint_wait_tail(self._h, C.c_void_p(stream or 0)))

    def debug_read(self, name, shape):
        self._need_dev("debug_read")
        out = np.empty(shape, np.float32)
        r = self._lib.d2fe_debug_read(self._h, name.encode(), _ptr(out), out.nbytes)
        if r < 0:
            _check(int(r))
        if r != out.nbytes:
            raise D2FEError(-1, "debug_read size mismatch %d vs %d" % (r, out.nbytes))
        return out

    def debug_conv3x3_wino(self, x, weight, bias, pool=False, relu=True, iters=0):
        """One 3x3 layer through the Winograd kernels: x [n,H,W,Cin] NHWC, weight [Cout,Cin,3,3].  Returns (out, ms_per_launch)."""
        self._need_dev("debug_conv3x3_wino")
        x = np.ascontiguousarray(x, np.float32); weight = np.ascontiguousarray(weight, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        n, H, W, cin = x.shape
        cout = weight.shape[0]
        out = np.empty((n, H // 2, W // 2, cout) if pool else (n, H, W, cout), np.float32)
        ms = C.c_float(0.0)
        _check(self._lib.d2fe_debug_conv3x3_wino(self._h, _ptr(x), n, H, W, cin, _ptr(weight), _ptr(bias), cout, int(pool),
                                                 int(relu), _ptr(out), int(iters), C.byref(ms)))
        return out, float(ms.value)

    def sync(self):
        _check(self._lib.d2fe_sync(self._h))

    def profile_enable(self, mode):
        """0 off, 1 dominant kernel (conv1b) only, 2 every stage (HIP events on the launch stream)."""
        _check(self._lib.d2fe_profile_enable(self._h, int(mode)))

    def profile_read(self):
        ms = np.zeros(len(PROF_STAGES), np.float32); cnt = np.zeros(len(PROF_STAGES), np.int32)
        _check(self._lib.d2fe_profile_read(self._h, _ptr(ms), _ptr(cnt)))
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(PROF_STAGES)}

    # ---- NetVLAD (MobileNetVLADONNX, mobilenetvlad_onnx.h:18-74) ---------------------------------------------------------
    def load_netvlad(self, nv):
        """nv: dict(layers=[dict(kind, cin, cout, stride, act, res, weight, bias)...], head=dict(pre_w, pre_b, assign_w,
        assign_b, centroids)) as produced by d2slam_amd.netvlad.synthetic_netvlad_weights()."""
        kinds = {"conv": 0, "pw": 1, "dw": 2}
        n = len(nv["layers"])
        arr = (_NvLayer * n)()
        keep = []
        for i, l in enumerate(nv["layers"]):
            w = np.ascontiguousarray(l["weight"], np.float32); b = np.ascontiguousarray(l["bias"], np.float32)
            keep += [w, b]
            arr[i] = _NvLayer(kinds[l["kind"]], l["cin"], l["cout"], l["stride"], l["act"], l["res"], w.ctypes.data, b.ctypes.data)
        hd = {k: np.ascontiguousarray(v, np.float32) for k, v in nv["head"].items()}
        ws = _NvWeights(n, C.cast(arr, C.c_void_p), hd["pre_w"].shape[1], hd["pre_w"].shape[0], hd["assign_w"].shape[0],
                        hd["pre_w"].ctypes.data, hd["pre_b"].ctypes.data, hd["assign_w"].ctypes.data,
                        hd["assign_b"].ctypes.data, hd["centroids"].ctypes.data)
        _check(self._lib.d2fe_load_netvlad(self._h, C.byref(ws)))

    def load_netvlad_onnx(self, path):
        """MobileNetVLADONNX's constructor takes an ONNX file (mobilenetvlad_onnx.h:18-47): graph -> layer list -> d2fe_load_netvlad."""
        from . import onnx_graph
        self.load_netvlad(onnx_graph.load_netvlad_onnx(path))

    def set_netvlad_pca(self, comp, mean):
        if comp is None:
            _check(self._lib.d2fe_set_netvlad_pca(self._h, None, None, 0))
            return
        comp = np.ascontiguousarray(comp, np.float32); mean = np.ascontiguousarray(mean, np.float32)
        _check(self._lib.d2fe_set_netvlad_pca(self._h, _ptr(comp), _ptr(mean), comp.shape[0]))

    @property
    def netvlad_dim(self):
        r = int(self._lib.d2fe_netvlad_dim(self._h))
        if r < 0:
            _check(r)
        return r

    def debug_netvlad_layer(self, layer, shape):
        """Output of one layer of the last netvlad call ([n,h,w,c] fp32) or None when it only exists inside a fused block."""
        self._need_dev("debug_netvlad_layer")
        out = np.empty(shape, np.float32)
        self._lib.d2fe_debug_netvlad_layer.restype = C.c_long
        r = self._lib.d2fe_debug_netvlad_layer(self._h, int(layer), int(shape[0]), _ptr(out), C.c_size_t(out.nbytes))
        if r == -3:      # D2FE_ERR_NOT_READY
            return None
        if r < 0:
            _check(int(r))
        assert r == out.nbytes, (r, out.nbytes)
        return out

    def debug_netvlad_stamps(self, max_wgs=32768):
        """[workgroups][32] uint64 phase stamps of the plan step named by D2FE_NV_STAMP_STEP (see include/d2fe.h)."""
        self._need_dev("debug_netvlad_stamps")
        out = np.zeros((max_wgs, 32), np.uint64)
        self._lib.d2fe_debug_netvlad_stamps.restype = C.c_long
        r = int(self._lib.d2fe_debug_netvlad_stamps(self._h, _ptr(out), C.c_long(max_wgs)))
        if r < 0:
            _check(r)
        return out[:r]

    def netvlad(self, images):
        """MobileNetVLADONNX::inference for a batch of u8 images [n,H,W] -> [n, netvlad_dim]."""
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim == 2:
            images = images[None]
        n, H, W = images.shape
        out = np.zeros((n, self.netvlad_dim), np.float32)
        _check(self._lib.d2fe_netvlad_batch(self._h, _ptr(images), n, W, H, W, H * W, _ptr(out)))
        return out

    def netvlad_device(self, d_gray, n, W, H, d_out, stream=None, stride=None, image_stride=None):
        _check(self._lib.d2fe_netvlad_device(self._h, d_gray, n, W, H, stride or W, image_stride or H * W, d_out, stream))

    # ---- SURVEY 8(f) next rows ------------------------------------------------------------------------------------------------
    def undistort(self, src, mapx, mapy, gain=None):
        """FisheyeUndist::undist_id_cuda (fisheye_undistort.h:152-176): remap(INTER_LINEAR) [+ photometric gain] -> u8."""
        src = np.ascontiguousarray(src, np.uint8); mapx = np.ascontiguousarray(mapx, np.float32); mapy = np.ascontiguousarray(mapy, np.float32)
        g = np.ascontiguousarray(gain, np.float32) if gain is not None else None
        dst = np.empty(mapx.shape, np.uint8)
        _check(self._lib.d2fe_undistort(self._h, _ptr(src), src.shape[1], src.shape[0], src.shape[1], _ptr(mapx), _ptr(mapy),
                                        _ptr(g), mapx.shape[1], mapx.shape[0], _ptr(dst)))
        return dst

    def undistort_device(self, d_src, n, sw, sh, d_mapx, d_mapy, d_gain, dw, dh, d_dst, stream=None, sstride=None,
                         src_image_stride=None):
        """Device-resident undistort of n frames sharing one map set (raw addresses)."""
        _check(self._lib.d2fe_undistort_device(self._h, d_src, n, sw, sh, sstride or sw, src_image_stride if src_image_stride is not None else sw * sh,
                                               d_mapx, d_mapy, d_gain, dw, dh, d_dst, stream))

    def prepare_gray(self, img, width, height):
        """cv::cvtColor(BGR2GRAY) if 3 channels + cv::resize to (width, height) if the size differs (superpoint_onnx.cpp:76-83)."""
        img = np.ascontiguousarray(img, np.uint8)
        ch = 1 if img.ndim == 2 else img.shape[2]
        sh, sw = img.shape[:2]
        out = np.zeros((height, width), np.uint8)
        _check(self._lib.d2fe_prepare_gray(self._h, _ptr(img), ch, sw, sh, sw * ch, int(width), int(height), _ptr(out)))
        return out

    @staticmethod
    def _mei(cam):
        """cam: dict(xi,k1,k2,p1,p2,gamma1,gamma2,u0,v0) or the 9 values in that order (kalibr 'omni' + 'radtan')."""
        keys = ("xi", "k1", "k2", "p1", "p2", "gamma1", "gamma2", "u0", "v0")
        vals = [cam[k] for k in keys] if isinstance(cam, dict) else list(cam)
        return (C.c_double * 9)(*[float(v) for v in vals])

    def gen_cylinder_map(self, cam, width, height, fov_deg):
        """FisheyeUndist::generateCylinderMap (fisheye_undistort.h:458-500) -> (mapx, mapy) float32 [height, width]."""
        mx = np.zeros((height, width), np.float32); my = np.zeros((height, width), np.float32)
        _check(self._lib.d2fe_gen_cylinder_map(self._h, self._mei(cam), int(width), int(height), float(fov_deg), _ptr(mx), _ptr(my)))
        return mx, my

    def gen_cylinder_map_device(self, cam, width, height, fov_deg, d_mapx, d_mapy, stream=None):
        _check(self._lib.d2fe_gen_cylinder_map_device(self._h, self._mei(cam), int(width), int(height), float(fov_deg),
                                                      C.c_void_p(d_mapx), C.c_void_p(d_mapy), C.c_void_p(stream or 0)))

    def gen_pinhole_map(self, cam, q_wxyz, width, height, f):
        """genOneUndistMap(id, cam, rotation, w, h, f) (fisheye_undistort.h:615-660) -> (mapx, mapy)."""
        mx = np.zeros((height, width), np.float32); my = np.zeros((height, width), np.float32)
        q = (C.c_double * 4)(*[float(v) for v in q_wxyz])
        _check(self._lib.d2fe_gen_pinhole_map(self._h, self._mei(cam), q, int(width), int(height), float(f), _ptr(mx), _ptr(my)))
        return mx, my

    def quantize_int8(self, x, double_max=False):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        out = np.empty(x.shape[0], np.int8)
        _check(self._lib.d2fe_quantize_int8(self._h, _ptr(x), x.shape[0], int(double_max), _ptr(out)))
        return out

    def dequantize_int8(self, q, landmark_num=-1):
        q = np.ascontiguousarray(q, np.int8).reshape(-1)
        out = np.empty(q.shape[0], np.float32)
        _check(self._lib.d2fe_dequantize_int8(self._h, _ptr(q), q.shape[0], int(landmark_num), _ptr(out)))
        return out

    # ---- matcher -------------------------------------------------------------------------------------------------
    def match_knn(self, desc_a, desc_b, knn_match_ratio=0.8, pts_a=None, pts_b=None, search_local_dist=-1.0):
        a = np.ascontiguousarray(desc_a, np.float32); b = np.ascontiguousarray(desc_b, np.float32)
        na = a.shape[0] if a.ndim == 2 else 0
        nb = b.shape[0] if b.ndim == 2 else 0
        dim = a.shape[1] if na else (b.shape[1] if nb else 256)
        cap = max(na, 1)
        q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32); n = C.c_int(0)
        pa = np.ascontiguousarray(pts_a, np.float32) if pts_a is not None and len(pts_a) else None
        pb = np.ascontiguousarray(pts_b, np.float32) if pts_b is not None and len(pts_b) else None
        _check(self._lib.d2fe_match_knn(self._h, _ptr(a), na, _ptr(b), nb, dim, float(knn_match_ratio), _ptr(pa), _ptr(pb),
                                        float(search_local_dist), _ptr(q), _ptr(t), _ptr(d), cap, C.byref(n)))
        return q[:n.value].copy(), t[:n.value].copy(), d[:n.value].copy()

    def match_crosscheck(self, desc_a, desc_b):
        a = np.ascontiguousarray(desc_a, np.float32); b = np.ascontiguousarray(desc_b, np.float32)
        na, nb = a.shape[0], b.shape[0]
        dim = a.shape[1]
        cap = max(na, 1)
        q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32); n = C.c_int(0)
        _check(self._lib.d2fe_match_crosscheck(self._h, _ptr(a), na, _ptr(b), nb, dim, _ptr(q), _ptr(t), _ptr(d), cap,
                                               C.byref(n)))
        return q[:n.value].copy(), t[:n.value].copy(), d[:n.value].copy()

    def pack_blocks_device(self, d_desc, d_kps, d_scores, d_n, d_netvlad, row0, row_step, nframes, cap, netvlad_dim, d_blocks, stream=None):
        """One exchange block per frame (desc | kps | scores | netvlad | n), raw device addresses; see include/d2fe.h."""
        _check(self._lib.d2fe_pack_blocks_device(self._h, C.c_void_p(d_desc), C.c_void_p(d_kps), C.c_void_p(d_scores), C.c_void_p(d_n),
                                                 C.c_void_p(d_netvlad or 0), int(row0), int(row_step), int(nframes), int(cap), int(netvlad_dim),
                                                 C.c_void_p(d_blocks), C.c_void_p(stream or 0)))

    def pack_blocks_int8_device(self, d_desc, d_kps, d_n, d_netvlad, row0, row_step, nframes, cap, netvlad_dim, d_blocks, stream=None):
        """Exchange blocks in the reference's int8 wire precision (VisualImageDesc::toLCM); raw device addresses."""
        V = C.c_void_p
        _check(self._lib.d2fe_pack_blocks_int8_device(self._h, V(d_desc), V(d_kps), V(d_n), V(d_netvlad or 0), int(row0), int(row_step), int(nframes),
                                                      int(cap), int(netvlad_dim), V(d_blocks), V(stream or 0)))

    def unpack_blocks_int8_device(self, d_blocks_int8, nblocks, cap, netvlad_dim, d_blocks, renorm=0, stream=None):
        """Gathered int8 blocks -> fp32 block layout with the reference's decode (renorm 0) or per-descriptor re-normalisation (1)."""
        V = C.c_void_p
        _check(self._lib.d2fe_unpack_blocks_int8_device(self._h, V(d_blocks_int8), int(nblocks), int(cap), int(netvlad_dim), int(renorm), V(d_blocks),
                                                        V(stream or 0)))

    def gate_pairs_device(self, d_q, q_stride, d_db, db_stride, dim, d_pair_q, d_pair_db, npairs, thres, d_cnt_inout=None, d_pass=None,
                          d_sims=None, d_n_pass=None, stream=None):
        """NetVLAD gate of a pair list (getMatchedPrevKeyframe's similarity test), raw device addresses."""
        _check(self._lib.d2fe_gate_pairs_device(self._h, C.c_void_p(d_q), C.c_size_t(q_stride), C.c_void_p(d_db), C.c_size_t(db_stride), int(dim),
                                                C.c_void_p(d_pair_q), C.c_void_p(d_pair_db), int(npairs), C.c_double(thres),
                                                C.c_void_p(d_cnt_inout or 0), C.c_void_p(d_pass or 0), C.c_void_p(d_sims or 0),
                                                C.c_void_p(d_n_pass or 0), C.c_void_p(stream or 0)))

    def quad_gate_device(self, d_local, local_stride, d_remote, remote_stride, dim, d_job_local_row0, d_job_remote_row0, local_view_step,
                         remote_view_step, njobs, thres, d_dir_prev=None, d_sims=None, d_cnt_inout=None, d_n_pass=None, stream=None):
        """The FOURCORNER_FISHEYE gate of getMatchedPrevKeyframe + the view pairing of trackRemoteFrames (include/d2fe.h)."""
        V = C.c_void_p
        _check(self._lib.d2fe_quad_gate_device(self._h, V(d_local), C.c_size_t(local_stride), V(d_remote), C.c_size_t(remote_stride), int(dim),
                                               V(d_job_local_row0), V(d_job_remote_row0), int(local_view_step), int(remote_view_step),
                                               int(njobs), C.c_double(thres), V(d_dir_prev or 0), V(d_sims or 0), V(d_cnt_inout or 0),
                                               V(d_n_pass or 0), V(stream or 0)))

    def half_image_compact_device(self, d_desc, d_pts, d_n, d_job_row, d_job_left, d_job_shift, njobs, cap, dim, width_undistort, undistort_fov,
                                  d_out_desc, d_out_pts, d_out_map, d_out_n, stream=None):
        """getFeatureHalfImg for a batch of jobs + the a-side x shift (d2featuretracker.cpp:1051-1075,1161-1170); raw device addresses."""
        V = C.c_void_p
        _check(self._lib.d2fe_half_image_compact_device(self._h, V(d_desc), V(d_pts), V(d_n), V(d_job_row), V(d_job_left), V(d_job_shift), int(njobs),
                                                        int(cap), int(dim), int(width_undistort), C.c_double(undistort_fov), V(d_out_desc), V(d_out_pts),
                                                        V(d_out_map), V(d_out_n), V(stream or 0)))

    def half_move_cols(self, width_undistort, undistort_fov):
        """move_cols of getFeatureHalfImg: (float)(width_undistort * 90.0 / undistort_fov)."""
        return float(self._lib.d2fe_half_move_cols(int(width_undistort), float(undistort_fov)))

    def remap_matches_device(self, d_q, d_t, d_n_match, d_map_a_job, d_map_b_job, d_maps, npairs, cap_match, cap_map, stream=None):
        """Index remap of matchLocalFeatures (d2featuretracker.cpp:1178-1181) for a batch of pairs; raw device addresses."""
        V = C.c_void_p
        _check(self._lib.d2fe_remap_matches_device(self._h, V(d_q), V(d_t), V(d_n_match), V(d_map_a_job), V(d_map_b_job), V(d_maps), int(npairs),
                                                   int(cap_match), int(cap_map), V(stream or 0)))

    def match_batch_device(self, d_a, d_b, d_a_off, d_b_off, d_a_cnt, d_b_cnt, npairs, dim, max_n, d_q, d_t, d_dist,
                           d_n, mode=0, ratio=0.8, radius=-1.0, d_pts_a=None, d_pts_b=None, stream=None):
        mb = _MatchBatch(d_a, d_b, d_pts_a, d_pts_b, d_a_off, d_b_off, d_a_cnt, d_b_cnt, npairs, dim, max_n, mode,
                         ratio, radius, d_q, d_t, d_dist, d_n)
        _check(self._lib.d2fe_match_batch_device(self._h, C.byref(mb), stream))


class DevFrontEnd(FrontEnd):
    """FrontEnd on the development library (lib/libd2fe_hip_dev.so): the d2fe_debug_* test hooks and the D2FE_* schedule switches exist only there."""

    def __init__(self, cfg: SuperPointConfig):
        super().__init__(cfg, dev=True)


class StereoPipe:
    """Frames in flight (include/d2fe.h, d2fe_pipe_*): the per-frame work of D2Frontend::processStereoframe (SuperPoint on both images, NetVLAD on
    the left one, matchKNN L<->R and L<->previous L) for `frames` stereo frames per submit with up to `lanes` submits in flight.
    submit() enqueues and returns a ticket; wait() returns views into the lane's pinned result block (copy what must outlive 2 * lanes submits)."""

    def __init__(self, fe: FrontEnd, lanes=4, frames=1, width=640, height=480, cap=None, netvlad=True, match_lr=True, match_prev=True,
                 ratio=0.8, radius_lr=-1.0, radius_prev=-1.0, pinned_input=False, cu_partition=False, netvlad_inline=None, coalesce=1, lane_cus=0, netvlad_group=1, coalesce_depth=0):
        self._lib = fe._lib
        self._fe = fe           # the pipe borrows the handle's weights
        c = _PipeConfig()
        self._lib.d2fe_pipe_default_config(C.byref(c))
        c.lanes, c.frames, c.width, c.height = int(lanes), int(frames), int(width), int(height)
        c.cap = int(cap or fe.cfg.max_keypoints)
        c.netvlad, c.match_lr, c.match_prev, c.pinned_input = int(bool(netvlad)), int(bool(match_lr)), int(bool(match_prev)), int(bool(pinned_input))
        c.ratio, c.radius_lr, c.radius_prev = float(ratio), float(radius_lr), float(radius_prev)
        c.cu_partition = int(bool(cu_partition)); c.coalesce = int(coalesce); c.lane_cus = int(lane_cus); c.netvlad_group = int(netvlad_group); c.coalesce_depth = int(coalesce_depth)
        c.netvlad_inline = 2 if netvlad_inline is None else int(bool(netvlad_inline))      # None: auto (inline when lanes > 2)
        self._p = C.c_void_p()
        _check(self._lib.d2fe_pipe_create(fe.handle, C.byref(c), C.byref(self._p)))
        if not hasattr(fe, "_pipes"):
            import weakref
            fe._pipes = weakref.WeakSet()
        fe._pipes.add(self)
        self.lanes, self.frames, self.width, self.height = int(lanes), int(frames), int(width), int(height)
        self._pinned_input = bool(pinned_input)
        self._res = _PipeResult()

    def close(self):
        if getattr(self, "_p", None) and self._p.value:
            self._lib.d2fe_pipe_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def profile_enable(self, mode):
        _check(self._lib.d2fe_pipe_profile_enable(self._p, int(mode)))

    def profile_read(self):
        ms = (C.c_float * len(PROF_STAGES))(); n = (C.c_int32 * len(PROF_STAGES))()
        _check(self._lib.d2fe_pipe_profile_read(self._p, ms, n))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(PROF_STAGES)}

    def submit_ptr(self, left_ptr, right_ptr, stride=None, image_stride=None):
        """raw host addresses (e.g. of pinned torch tensors)"""
        t = C.c_int64()
        _check(self._lib.d2fe_pipe_submit(self._p, C.c_void_p(left_ptr), C.c_void_p(right_ptr), int(stride or self.width),
                                          int(image_stride or self.width * self.height), C.byref(t)))
        return int(t.value)

    def submit(self, left, right):
        """left, right: u8 [frames, H, W] (or [H, W] when frames == 1).  The library copies the frames into its own pinned staging before this returns
        (pinned_input = 0), so the arrays -- and any contiguous temporaries made here -- are not referenced afterwards.  A pipe created with
        pinned_input=True DMAs straight from the caller's memory until the ticket has been waited for: that contract (page-locked memory that
        stays alive and unchanged) cannot be kept for numpy arrays, so it is refused here -- use submit_ptr with memory you own."""
        if self._pinned_input:
            raise ValueError("StereoPipe(pinned_input=True): submit() takes numpy arrays, which are neither page-locked nor kept alive until wait(); "
                             "use submit_ptr() with page-locked memory that outlives the ticket")
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        return self.submit_ptr(left.ctypes.data, right.ctypes.data)

    def stream_placement(self):
        """(classes, n_classes) of d2fe_pipe_stream_placement: classes[k] = (hardware-pipe class of lane k's own stream, of its second stream)."""
        K = self.lanes
        arr = (C.c_int32 * (2 * K))(); n = C.c_int32(0)
        _check(self._lib.d2fe_pipe_stream_placement(self._p, arr, C.byref(n)))
        return [(arr[2 * k], arr[2 * k + 1]) for k in range(K)], n.value

    def classify_stream(self, stream):
        """d2fe_pipe_classify_stream: the class of the pipe's streams `stream` (a hipStream_t as int) takes turns with on the device, -1: none.  Idle pipe only."""
        c = C.c_int32(-1)
        _check(self._lib.d2fe_pipe_classify_stream(self._p, C.c_void_p(int(stream)), C.byref(c)))
        return int(c.value)

    def pick_consumer_stream(self, make, handle=lambda s: s, tries=4):
        """A stream for a device-side consumer of this pipe's results that disturbs the lanes least: `make()` creates candidate streams one at a time (`handle(stream)` = its
        hipStream_t as int) until one takes turns with none of the lanes' streams, or only with second (NetVLAD) streams; after `tries` candidates the best seen.  One at a
        time because every stream a process creates costs it a hardware queue for good."""
        placement, n = self.stream_placement()
        own = {a for a, _ in placement}
        best = None
        for _ in range(tries if n >= 2 else 1):
            st = make()
            if n < 2:
                return st
            c = self.classify_stream(handle(st))
            rank = 0 if c < 0 else 1 if c not in own else 2
            if best is None or rank < best[0]:
                best = (rank, st)
            if rank < 2:
                break
        return best[1]

    def device_view(self, ticket, stream):
        """DEVICE pointers into the ticket's result block for a consumer on `stream` (a raw hipStream_t, not 0): the stream is made to wait for the ticket's SuperPoint
        and NetVLAD results; release with device_release(ticket, stream) once the consumer's work is queued (include/d2fe.h, d2fe_pipe_device_view)."""
        r = _PipeDeviceResult()
        _check(self._lib.d2fe_pipe_device_view(self._p, C.c_int64(ticket), C.c_void_p(stream), C.byref(r)))
        return r

    def device_release(self, ticket, stream):
        _check(self._lib.d2fe_pipe_device_release(self._p, C.c_int64(ticket), C.c_void_p(stream)))

    def wait_raw(self, ticket):
        _check(self._lib.d2fe_pipe_wait(self._p, C.c_int64(ticket), C.byref(self._res)))
        return self._res

    def wait(self, ticket):
        """dict of numpy VIEWS into the pinned result block"""
        r = self.wait_raw(ticket)
        F, cap, D, G = r.frames, r.cap, r.desc_dim, r.netvlad_dim

        def view(ptr, shape, dt):
            if not ptr:
                return None
            n = int(np.prod(shape))
            buf = (C.c_float * n).from_address(ptr) if dt == np.float32 else (C.c_int32 * n).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).reshape(shape)
        out = {"kps_xy": view(r.kps_xy, (2 * F, cap, 2), np.float32), "scores": view(r.scores, (2 * F, cap), np.float32),
               "desc": view(r.desc, (2 * F, cap, D), np.float32), "n_kp": view(r.n_kp, (2 * F,), np.int32),
               "netvlad": view(r.netvlad, (F, G), np.float32) if G else None}
        for k in ("lr", "prev"):
            out[k + "_q"] = view(getattr(r, k + "_q"), (F, cap), np.int32); out[k + "_t"] = view(getattr(r, k + "_t"), (F, cap), np.int32)
            out[k + "_dist"] = view(getattr(r, k + "_dist"), (F, cap), np.float32); out[k + "_n"] = view(getattr(r, k + "_n"), (F,), np.int32)
        return out


def block_words(cap, netvlad_dim):
    """Float words of one exchange block (include/d2fe.h, d2fe_block_words); callable without a GPU."""
    return int(load_library().d2fe_block_words(int(cap), int(netvlad_dim)))


def block_bytes_int8(cap, netvlad_dim):
    """Bytes of one int8 exchange block (include/d2fe.h, d2fe_block_bytes_int8)."""
    return int(load_library().d2fe_block_bytes_int8(int(cap), int(netvlad_dim)))


def block_field_offset(cap, netvlad_dim, field):
    """Word offset of a block field: 'desc', 'kps', 'scores', 'netvlad', 'n'."""
    return int(load_library().d2fe_block_field_offset(int(cap), int(netvlad_dim), ["desc", "kps", "scores", "netvlad", "n"].index(field)))


class FlatIPDatabase:
    """faiss::IndexFlatIP stand-in for the NetVLAD keyframe database (loop_detector.h:71-72, loop_detector.cpp:254-263,300-350)."""

    def __init__(self, fe: FrontEnd, dim: int, capacity: int = 65536):
        self._lib = fe._lib
        self._db = C.c_void_p()
        self._fe = fe
        _check(self._lib.d2fe_db_create(fe.handle, dim, capacity, C.byref(self._db)))
        self.dim = dim

    def close(self):
        if self._db.value:
            self._lib.d2fe_db_destroy(self._db)
            self._db = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ntotal(self):
        return int(self._lib.d2fe_db_ntotal(self._db))

    def add(self, vecs):
        v = np.ascontiguousarray(vecs, np.float32).reshape(-1, self.dim)
        r = int(self._lib.d2fe_db_add(self._db, _ptr(v), v.shape[0]))
        if r < 0:
            _check(r)
        return r

    def search(self, q, k):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, self.dim)
        sims = np.zeros((q.shape[0], k), np.float32); labels = np.zeros((q.shape[0], k), np.int32)
        _check(self._lib.d2fe_db_search(self._db, _ptr(q), q.shape[0], k, _ptr(sims), _ptr(labels)))
        return sims, labels

    def query_gated(self, q, max_index, thres):
        """LoopDetector::queryIndexFromDatabase (loop_detector.cpp:300-350) -> (label or -1, similarity)."""
        q = np.ascontiguousarray(q, np.float32).reshape(self.dim)
        label = C.c_int32(-1); sim = C.c_float(0)
        _check(self._lib.d2fe_db_query_gated(self._db, _ptr(q), int(max_index), float(thres), C.byref(label), C.byref(sim)))
        return int(label.value), float(sim.value)


# ---- LK optical-flow tracker (d2frontend/src/opticaltrack_utils.cpp) ---------------------------------
PYR_LEVEL = 2            # opticaltrack_utils.h:10
WIN_SIZE = 21            # opticaltrack_utils.cpp:25
LK_ITERS = 30            # SparsePyrLKOpticalFlow::create(WIN_SIZE, PYR_LEVEL, 30, true), :239
WHOLE_IMG_MATCH, LEFT_RIGHT_IMG_MATCH, RIGHT_LEFT_IMG_MATCH = 0, 1, 2   # TrackLRType


class LKFrame:
    """Device-resident image pyramid = LKImageInfoGPU::pyr (opticaltrack_utils.h:16-23), built by buildImagePyramid
    (opticaltrack_utils.cpp:526-542)."""

    def __init__(self, fe: FrontEnd, gray=None, levels=PYR_LEVEL, d_gray=None, width=None, height=None, stride=None, stream=None):
        self._lib = fe._lib
        self._fe = fe
        self._f = C.c_void_p()
        self.levels = levels
        if gray is not None:
            img = np.ascontiguousarray(gray, np.uint8)
            self.height, self.width = img.shape
            _check(self._lib.d2fe_lk_frame_create(fe.handle, _ptr(img), self.width, self.height, self.width, levels, C.byref(self._f)))
        else:
            self.width, self.height = int(width), int(height)
            _check(self._lib.d2fe_lk_frame_create_device(fe.handle, C.c_void_p(d_gray), self.width, self.height,
                                                         int(stride or width), levels, C.c_void_p(stream or 0), C.byref(self._f)))

    def close(self):
        if self._f.value:
            self._lib.d2fe_lk_frame_destroy(self._f)
            self._f = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level(self, l):
        w, h = self.width, self.height
        for _ in range(l):
            w, h = (w + 1) // 2, (h + 1) // 2
        out = np.zeros((h, w), np.uint8)
        r = self._lib.d2fe_lk_frame_read_level(self._f, int(l), _ptr(out), out.nbytes, None, None)
        if r < 0:
            _check(int(r))
        return out


def buildImagePyramid(fe: FrontEnd, gray, maxLevel=PYR_LEVEL) -> LKFrame:
    """buildImagePyramid(GpuMat, maxLevel) (opticaltrack_utils.cpp:526-542)."""
    return LKFrame(fe, gray, maxLevel)


def lk_track(fe: FrontEnd, prev: LKFrame, cur: LKFrame, prev_pts, cur_init, track_type=WHOLE_IMG_MATCH, move_cols=0.0,
             win=WIN_SIZE, iters=LK_ITERS):
    """Forward + reverse SparsePyrLK with the 0.5 px and inBorder tests (opticaltrack_utils.cpp:236-272) -> (cur_pts, status)."""
    pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    ci = np.ascontiguousarray(cur_init, np.float32).reshape(-1, 2)
    n = pp.shape[0]
    out = np.zeros((max(n, 1), 2), np.float32); st = np.zeros(max(n, 1), np.uint8)
    _check(fe._lib.d2fe_lk_track(fe.handle, prev._f, cur._f, _ptr(pp), _ptr(ci), n, int(track_type), float(move_cols), int(win),
                                 int(iters), _ptr(out), _ptr(st)))
    return out[:n], st[:n]


class _LKPair(C.Structure):
    _fields_ = [("prev", C.c_void_p), ("cur", C.c_void_p), ("first", C.c_int32), ("count", C.c_int32), ("type", C.c_int32),
                ("move_cols", C.c_float)]


def lk_track_batch(fe: FrontEnd, jobs, win=WIN_SIZE, iters=LK_ITERS):
    """jobs: list of (prev LKFrame, cur LKFrame, prev_pts, cur_init, track_type, move_cols) -> list of (cur_pts, status);
    all tracks run in one kernel launch (d2fe_lk_track_batch)."""
    pairs = (_LKPair * max(len(jobs), 1))()
    pp, ci, first = [], [], 0
    for k, (prev, cur, p, c, t, mv) in enumerate(jobs):
        p = np.ascontiguousarray(p, np.float32).reshape(-1, 2); c = np.ascontiguousarray(c, np.float32).reshape(-1, 2)
        pairs[k] = _LKPair(prev._f, cur._f, first, len(p), int(t), float(mv))
        pp.append(p); ci.append(c); first += len(p)
    n = first
    allp = np.concatenate(pp) if n else np.zeros((0, 2), np.float32)
    alli = np.concatenate(ci) if n else np.zeros((0, 2), np.float32)
    out = np.zeros((max(n, 1), 2), np.float32); st = np.zeros(max(n, 1), np.uint8)
    _check(fe._lib.d2fe_lk_track_batch(fe.handle, pairs, len(jobs), _ptr(np.ascontiguousarray(allp)), _ptr(np.ascontiguousarray(alli)),
                                       n, int(win), int(iters), _ptr(out), _ptr(st)))
    res = []
    for k in range(len(jobs)):
        a, cnt = pairs[k].first, pairs[k].count
        res.append((out[a:a + cnt].copy(), st[a:a + cnt].copy()))
    return res


def opticalflowTrackPyr(fe: FrontEnd, cur_img, prev_lk: dict, track_type=WHOLE_IMG_MATCH, undistort_fov=200.0):
    """opticalflowTrackPyr(cur_img, prev_lk, type) (opticaltrack_utils.cpp:173-279).  prev_lk / the return value are dicts with
    the LKImageInfoGPU fields: lk_pts [n,2], lk_ids, lk_local_index, lk_types, pyr (LKFrame)."""
    cur_img = np.ascontiguousarray(cur_img, np.uint8)
    cur_pyr = buildImagePyramid(fe, cur_img, PYR_LEVEL)
    prev_pts = np.asarray(prev_lk.get("lk_pts", np.zeros((0, 2))), np.float32).reshape(-1, 2)
    ids = np.asarray(prev_lk.get("lk_ids", np.arange(len(prev_pts))))
    types = np.asarray(prev_lk.get("lk_types", np.zeros(len(prev_pts), np.int32)))
    local = np.asarray(prev_lk.get("lk_local_index", np.arange(len(prev_pts))))
    empty = {"lk_pts": np.zeros((0, 2), np.float32), "lk_ids": ids[:0], "lk_local_index": local[:0], "lk_types": types[:0], "pyr": cur_pyr}
    if len(prev_pts) == 0:
        return empty
    move_cols = np.float32(cur_img.shape[1] * 90.0 / undistort_fov)
    if track_type == WHOLE_IMG_MATCH:
        cur_pts = prev_pts.copy()
    else:
        if track_type == LEFT_RIGHT_IMG_MATCH:
            keep = prev_pts[:, 0] < np.float32(cur_img.shape[1]) - move_cols
            cur_pts = prev_pts[keep].copy(); cur_pts[:, 0] += move_cols
        else:
            keep = prev_pts[:, 0] >= move_cols
            cur_pts = prev_pts[keep].copy(); cur_pts[:, 0] -= move_cols
        prev_pts, ids, types, local = prev_pts[keep], ids[keep], types[keep], local[keep]
    if len(cur_pts) == 0:
        return empty
    out, st = lk_track(fe, prev_lk["pyr"], cur_pyr, prev_pts, cur_pts, track_type, float(move_cols))
    k = st.astype(bool)
    return {"lk_pts": out[k], "lk_ids": ids[k], "lk_local_index": local[k], "lk_types": types[k], "pyr": cur_pyr}


def detectFastByRegion(fe: FrontEnd, frame: LKFrame, features, cols, rows, threshold=10, with_response=False):
    """detectFastByRegion(img, mask, features, cols, rows) (opticaltrack_utils.cpp:444-493)."""
    cap = max(int(features), 1)
    xy = np.zeros((cap, 2), np.float32); resp = np.zeros(cap, np.int32); n = C.c_int(0)
    _check(fe._lib.d2fe_detect_fast_by_region(fe.handle, frame._f, int(features), int(cols), int(rows), int(threshold), _ptr(xy),
                                              _ptr(resp), cap, C.byref(n)))
    return (xy[:n.value].copy(), resp[:n.value].copy()) if with_response else xy[:n.value].copy()


def goodFeaturesToTrack(fe: FrontEnd, frame: LKFrame, max_corners, quality=0.01, min_dist=20.0):
    """cv::cuda::createGoodFeaturesToTrackDetector(type, max_corners, quality, min_dist)->detect (opticaltrack_utils.cpp:404-412)."""
    cap = max(int(max_corners), 1) if max_corners > 0 else frame.width * frame.height // 4
    xy = np.zeros((cap, 2), np.float32); n = C.c_int(0)
    _check(fe._lib.d2fe_good_features_to_track(fe.handle, frame._f, int(max_corners), float(quality), float(min_dist), _ptr(xy), cap,
                                               C.byref(n)))
    return xy[:n.value].copy()


def detectPoints(fe: FrontEnd, frame: LKFrame, cur_pts, require_pts, use_fast=False, fast_rows=3, fast_cols=4,
                 feature_min_dist=20.0):
    """detectPoints (opticaltrack_utils.cpp:375-442): detect only when more than a quarter of the points are missing, ask for
    twice the shortfall when some points exist, drop candidates closer than feature_min_dist to an accepted point."""
    cur_pts = np.asarray(cur_pts, np.float32).reshape(-1, 2)
    lack = int(require_pts) - len(cur_pts)
    if not lack > int(require_pts) // 4:
        return np.zeros((0, 2), np.float32)
    num = lack * 2 if len(cur_pts) > 0 else lack
    if use_fast:
        cand = detectFastByRegion(fe, frame, num, fast_rows, fast_cols)   # (sic) the reference passes rows as `cols`, :391-392
    else:
        cand = goodFeaturesToTrack(fe, frame, num, 0.01, feature_min_dist)
    all_pts = [p for p in cur_pts]
    out = []
    for p in cand:
        near = False
        for q in all_pts:
            d = p - q
            if np.sqrt(np.float64(d[0]) * d[0] + np.float64(d[1]) * d[1]) < feature_min_dist:
                near = True
                break
        if not near:
            out.append(p); all_pts.append(p)
        if len(out) >= lack:
            break
    return np.asarray(out, np.float32).reshape(-1, 2)


class SuperPoint:
    """Mirror of class SuperPoint (superpoint_tensorrt.h:36-93): build() then infer()."""

    def __init__(self, super_point_config: SuperPointConfig, weights=None):
        self.super_point_config_ = super_point_config
        self._weights = weights
        self._fe: Optional[FrontEnd] = None

    def build(self) -> bool:
        """SuperPoint::build (superpoint_tensorrt.cpp:22-107): returns False instead of raising, like the reference."""
        try:
            self._fe = FrontEnd(self.super_point_config_)
            if self._weights is None:
                raise D2FEError(-3, "no weights given (the reference would fail to parse onnx_path)")
            self._fe.load_superpoint(self._weights)
            return True
        except D2FEError as e:
            self.last_error = str(e)
            self._fe = None
            return False

    def infer(self, image):
        """bool SuperPoint::infer(const cv::Mat&, vector<Point2f>& keypoints, vector<float>& descriptors,
        vector<float>& scores) (superpoint_tensorrt.cpp:161-183).  Returns (ok, keypoints [k,2], descriptors
        flat [k*256], scores [k]); on failure all three are empty, as the reference clears them."""
        empty = (False, np.zeros((0, 2), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32))
        if self._fe is None:
            return empty
        try:
            (kps, sc, desc), = self._fe.extract_batch(np.asarray(image)[None])
            return True, kps, desc.reshape(-1), sc
        except D2FEError as e:
            self.last_error = str(e)
            return empty

    @property
    def frontend(self):
        return self._fe


def matchKNN(fe: FrontEnd, desc_a, desc_b, knn_match_ratio=0.8, pts_a=None, pts_b=None, search_local_dist=-1.0):
    """D2FrontEnd::matchKNN (feature_matcher.cpp:4-42) -> list of (queryIdx, trainIdx, distance)."""
    q, t, d = fe.match_knn(desc_a, desc_b, knn_match_ratio, pts_a, pts_b, search_local_dist)
    return list(zip(q.tolist(), t.tolist(), d.tolist()))


def get_feature_half_img(pts, desc, require_left, width_undistort, undistort_fov, dims=256):
    """getFeatureHalfImg (d2featuretracker.cpp:1051-1075): returns (desc_half, pts_new, tmp_to_idx)."""
    lib = load_library()
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = pts.shape[0]
    m = np.zeros(max(n, 1), np.int32); cnt = C.c_int(0)
    _check(lib.d2fe_half_image_filter(_ptr(pts), n, int(require_left), int(width_undistort), float(undistort_fov),
                                      _ptr(m), C.byref(cnt)))
    idx = m[:cnt.value].copy()
    desc = np.asarray(desc, np.float32).reshape(-1, dims)
    return desc[idx].copy(), pts[idx].copy(), idx


# ---- cross-agent exchange behind a pipe (include/d2fe.h, d2fe_exchange_* / d2fe_rccl_*; csrc/exchange.hip) ---------------------------------------------
WIRE = {"fp32": 0, "int8": 1, "int8-renorm256": 2}
EXCHANGE_PHASES = ("pack_blocks", "all_gather", "decode_counts_gate", "match_remote", "release_and_d2h")


def rccl_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes it; every rank needs the same bytes)"""
    buf = (C.c_char * 128)()
    _check(load_library().d2fe_rccl_unique_id(buf))
    return bytes(buf.raw)


def rccl_comm_init_rank(uid, world, rank, device=0):
    """ncclCommInitRank through the library's own dlopen of librccl -> an opaque ncclComm_t (int address)"""
    assert len(uid) == 128
    comm = C.c_void_p()
    _check(load_library().d2fe_rccl_comm_init_rank(C.create_string_buffer(uid, 128), int(world), int(rank), int(device), C.byref(comm)))
    return comm.value


def rccl_comm_destroy(comm):
    if comm:
        _check(load_library().d2fe_rccl_comm_destroy(C.c_void_p(comm)))


class Exchange:
    """d2fe_exchange_*: pack -> ONE all-gather -> gate -> remote matchKNN -> D2H per ticket of a StereoPipe, queued on the stream of the lane that produced the
    ticket (own_stream=False) or on one stream of its own.  comm: an ncclComm_t address (rccl_comm_init_rank, or D2SLAM's own) or None with `all_gather` =
    a Python callable (user, d_send, d_recv, bytes_per_rank, stream) -> 0 (tests over gloo)."""

    def __init__(self, pipe, comm=None, world=1, rank=0, wire="fp32", loopback=False, slots=4, own_stream=False, timing=False, gate_thres=0.8, ratio=0.8, all_gather=None):
        self._lib = pipe._lib
        self._pipe = pipe
        c = _ExchangeConfig()
        self._lib.d2fe_exchange_default_config(C.byref(c))
        c.world, c.rank, c.wire, c.loopback, c.slots = int(world), int(rank), WIRE[wire], int(bool(loopback)), int(slots)
        c.own_stream, c.timing, c.gate_thres, c.ratio = int(bool(own_stream)), int(bool(timing)), float(gate_thres), float(ratio)
        self._cb = ALL_GATHER_FN(all_gather) if all_gather is not None else ALL_GATHER_FN()
        c.all_gather = self._cb
        self._x = C.c_void_p()
        _check(self._lib.d2fe_exchange_create(pipe._p, C.c_void_p(comm) if comm else None, C.byref(c), C.byref(self._x)))
        self.npairs = int(self._lib.d2fe_exchange_pairs(self._x))
        self.block_bytes = int(self._lib.d2fe_exchange_block_bytes(self._x))
        self.slots, self.timing = int(slots), bool(timing)
        self._res = _ExchangeResult()

    @property
    def stream(self):
        """own_stream=True: that hipStream_t (int), else None"""
        return self._lib.d2fe_exchange_stream(self._x)

    def enqueue(self, ticket, slot):
        _check(self._lib.d2fe_exchange_enqueue(self._x, int(ticket), int(slot)))

    def collect(self, slot):
        """blocks until the slot's results are in host memory; numpy VIEWS into the pinned slot (valid until the slot is enqueued again)"""
        r = self._res
        _check(self._lib.d2fe_exchange_collect(self._x, int(slot), C.byref(r)))
        n, cap = int(r.npairs), int(r.cap)

        def view(addr, ct, shape):
            if not addr:
                return None
            cnt = int(np.prod(shape)) if len(shape) else 1
            return np.ctypeslib.as_array(C.cast(addr, C.POINTER(ct)), shape=(max(cnt, 1),))[:cnt].reshape(shape)
        return {"ticket": int(r.ticket), "mq": view(r.q_idx, C.c_int32, (n, cap)), "mt": view(r.t_idx, C.c_int32, (n, cap)), "md": view(r.dist, C.c_float, (n, cap)),
                "mn": view(r.n_match, C.c_int32, (n,)), "gate_pass": view(r.gate_pass, C.c_int32, (n,)), "sims": view(r.gate_sims, C.c_float, (n,)),
                "gate_n": int(r.gate_n), "phase_ms": [float(v) for v in r.phase_ms] if self.timing else None}

    def close(self):
        if getattr(self, "_x", None) and self._x.value:
            self._lib.d2fe_exchange_destroy(self._x)
            self._x = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
