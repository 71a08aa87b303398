"""ctypes binding of slam_amd/librgbdvo_hip.so (the C ABI declared in include/rvo.h).

The library is the product: there is no CPU fallback.  Loading fails loudly if
the shared object is missing; compute calls fail loudly if no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RVO_LIB", os.path.join(_HERE, "librgbdvo_hip.so"))


class Intrinsics(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("depth_width", C.c_int32),
                ("depth_height", C.c_int32)] + [(n, C.c_double) for n in
                                                ("fx", "fy", "cx", "cy", "k1", "k2", "k3", "p1", "p2")]


class Params(C.Structure):
    _fields_ = [("feature_window_size", C.c_int32), ("num_pyramid_levels", C.c_int32),
                ("fast_init_threshold", C.c_int32), ("fast_threshold_min", C.c_int32),
                ("fast_threshold_max", C.c_int32), ("fast_adaptive_gain", C.c_double),
                ("target_pixels_per_feature", C.c_int32), ("grid_width", C.c_int32),
                ("grid_height", C.c_int32), ("max_keypoints_per_grid", C.c_int32),
                ("inlier_max_reprojection_error", C.c_double), ("clique_inlier_threshold", C.c_double),
                ("min_features_for_estimate", C.c_int32), ("max_mean_reprojection_error", C.c_double),
                ("feature_search_window", C.c_double), ("reference_frame_change_threshold", C.c_int32),
                ("warmup_frames", C.c_int32)]


class PairResult(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("cov", C.c_double * 36), ("status", C.c_int32),
                ("num_matches", C.c_int32), ("num_inliers", C.c_int32),
                ("num_reprojection_failures", C.c_int32), ("mean_reprojection_error", C.c_double)]


class StreamResult(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("curr_to_ref", C.c_double * 12), ("cov", C.c_double * 36),
                ("rc", C.c_int32), ("fast_threshold", C.c_int32), ("keyframe_changed", C.c_int32),
                ("status", C.c_int32), ("num_matches", C.c_int32), ("num_inliers", C.c_int32),
                ("used_fallback", C.c_int32), ("pad_", C.c_int32)]


# every symbol include/rvo.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "rvo_default_vo_params", "rvo_default_fm_params", "rvo_create", "rvo_destroy", "rvo_last_error",
    "rvo_set_stream", "rvo_sync", "rvo_prepare_frames", "rvo_prepare_frames_dev", "rvo_match_pairs",
    "rvo_match_pairs_async", "rvo_match_frames_dev", "rvo_get_pair_results", "rvo_initial_rotation", "rvo_streams_reset",
    "rvo_stream_step", "rvo_stream_step_dev", "rvo_get_stream_results", "rvo_stream_results_dev",
    "rvo_get_stream_slots",
    "rvo_get_level_info", "rvo_get_level_image", "rvo_get_grid_keypoints", "rvo_get_keypoints",
    "rvo_stream_step_host_async", "rvo_wait_upload", "rvo_host_alloc", "rvo_host_free", "rvo_get_descriptors", "rvo_get_matches", "rvo_get_match_tracks", "rvo_get_match_consistency", "rvo_get_keypoint_tracks", "rvo_reset_tracks", "rvo_get_lut", "rvo_profile_enable", "rvo_profile_select", "rvo_profile_read",
    "rvo_selftest_stdalgo", "rvo_bgr_to_gray_dev", "rvo_depth_to_point_cloud_dev", "rvo_get_pair_information", "rvo_get_pair_information_noalloc", "rvo_set_pair_chunks", "rvo_wait_peer",
    "rvo_streams_record", "rvo_streams_set_source_index", "rvo_streams_depth_in_place", "rvo_trajectories_dev", "rvo_get_trajectories",
    "rvo_set_pipeline", "rvo_get_batch_results", "rvo_comm_get_unique_id", "rvo_comm_create", "rvo_comm_destroy", "rvo_gather_trajectories",
]
TRAJ_DOUBLES = 16   # RVO_TRAJ_DOUBLES
COMM_ID_BYTES = 128  # RVO_COMM_ID_BYTES

_LIB = None


def _share_one_hip_runtime():
    """A process must hold ONE copy of the HIP/HSA runtime.  librgbdvo_hip.so needs `libamdhip64.so.7` and would take
    /opt/rocm's; the PyTorch wheel bundles its own copy as torch/lib/libamdhip64.so (same soname) and asks for it by
    file name, so when this library is loaded BEFORE torch the process ends up with two runtimes and the second one to
    initialise finds no device (torch then reports "No HIP GPUs are available"; seen in round 1 and first blamed on
    rvo_destroy).  In the other order the dynamic linker binds this library to torch's copy by soname.  So: if no HIP
    runtime is mapped yet and a torch installation with a bundled runtime exists, map that copy first (without
    importing torch).  RVO_HIP_RUNTIME=system skips this (processes that never load torch lose nothing either way)."""
    if os.environ.get("RVO_HIP_RUNTIME", "") == "system":
        return
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return
    except OSError:
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for d in (spec.submodule_search_locations or []) if spec else []:
        p = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)
            return


def load():
    """Load the HIP library.  Raises (no fallback) if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run ./build.sh (python -c 'import __graft_entry__ as g; g.build()')")
    _share_one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.rvo_last_error.restype = C.c_char_p
    vp, ip = C.c_void_p, C.c_int
    L.rvo_create.argtypes = [C.POINTER(Intrinsics), C.POINTER(Params), ip, ip, ip, C.POINTER(vp)]
    L.rvo_destroy.argtypes = [vp]
    L.rvo_destroy.restype = None
    L.rvo_set_stream.argtypes = [vp, vp]
    L.rvo_sync.argtypes = [vp]
    L.rvo_set_pair_chunks.argtypes = [vp, ip]
    L.rvo_wait_peer.argtypes = [vp, vp]
    L.rvo_prepare_frames.argtypes = [vp, ip, vp, vp, vp, vp]
    L.rvo_prepare_frames_dev.argtypes = [vp, ip, vp, vp, vp, vp]
    L.rvo_match_pairs.argtypes = [vp, ip, vp, vp, vp, vp]
    L.rvo_match_pairs_async.argtypes = [vp, ip, vp, vp, vp]
    L.rvo_match_frames_dev.argtypes = [vp, ip, vp, vp, vp, vp, vp]
    L.rvo_bgr_to_gray_dev.argtypes = [vp, ip, vp, vp]
    L.rvo_depth_to_point_cloud_dev.argtypes = [vp, ip, vp, C.c_double, C.c_double, vp]
    L.rvo_get_pair_information.argtypes = [vp, ip, vp]
    L.rvo_get_pair_information_noalloc.argtypes = [vp, ip, vp]
    L.rvo_get_pair_results.argtypes = [vp, ip, vp]
    L.rvo_initial_rotation.argtypes = [vp, ip, vp, vp, vp]
    L.rvo_streams_reset.argtypes = [vp, ip]
    L.rvo_streams_depth_in_place.argtypes = [vp, ip]
    L.rvo_stream_step.argtypes = [vp, ip, vp, vp, vp]
    L.rvo_stream_step_dev.argtypes = [vp, ip, vp, vp]
    L.rvo_get_stream_results.argtypes = [vp, ip, vp]
    L.rvo_stream_results_dev.argtypes = [vp, C.POINTER(vp)]
    L.rvo_get_stream_slots.argtypes = [vp, ip, vp]
    L.rvo_get_level_info.argtypes = [vp, ip, ip, vp]
    L.rvo_get_level_image.argtypes = [vp, ip, ip, vp]
    L.rvo_get_grid_keypoints.argtypes = [vp, ip, ip, vp]
    L.rvo_get_keypoints.argtypes = [vp, ip, ip, vp, vp, vp]
    L.rvo_get_descriptors.argtypes = [vp, ip, ip, vp]
    L.rvo_get_matches.argtypes = [vp, ip, ip, vp, vp]
    L.rvo_get_match_tracks.argtypes = [vp, ip, ip, vp]
    L.rvo_get_match_consistency.argtypes = [vp, ip, ip, vp]
    L.rvo_stream_step_host_async.argtypes = [vp, ip, vp, vp]
    L.rvo_wait_upload.argtypes = [vp]
    L.rvo_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.rvo_host_free.argtypes = [vp]
    L.rvo_get_keypoint_tracks.argtypes = [vp, ip, ip, vp]
    L.rvo_reset_tracks.argtypes = [vp, ip, vp]
    L.rvo_get_lut.argtypes = [vp, vp, vp]
    L.rvo_profile_enable.argtypes = [vp, ip]
    L.rvo_profile_select.argtypes = [vp, C.c_char_p]
    L.rvo_profile_read.argtypes = [vp, ip, vp, vp, vp]
    L.rvo_selftest_stdalgo.argtypes = [ip, vp, ip, ip, ip, vp]
    L.rvo_set_pipeline.argtypes = [vp, ip]
    L.rvo_get_batch_results.argtypes = [vp, ip, ip, vp]
    L.rvo_streams_record.argtypes = [vp, ip]
    L.rvo_streams_set_source_index.argtypes = [vp, ip, vp]
    L.rvo_trajectories_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rvo_get_trajectories.argtypes = [vp, ip, ip, vp]
    L.rvo_comm_get_unique_id.argtypes = [vp]
    L.rvo_comm_create.argtypes = [ip, ip, ip, vp, C.POINTER(vp)]
    L.rvo_comm_destroy.argtypes = [vp]
    L.rvo_gather_trajectories.argtypes = [vp, vp, C.c_size_t, vp]
    L.rvo_default_vo_params.argtypes = [C.POINTER(Params)]
    L.rvo_default_fm_params.argtypes = [C.POINTER(Params)]
    _LIB = L
    return L
