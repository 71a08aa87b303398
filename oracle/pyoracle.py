"""ctypes binding of oracle/librvo_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


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


def default_params(frame_matcher: bool = False, **over) -> Params:
    """settings.hpp:11-52 defaults (VO: keyframe threshold 250, FrameMatcher: 150)."""
    p = Params(9, 3, 20, 5, 70, 0.005, 250, 80, 80, 25, 1.5, 0.1, 10, 10.0, 25.0,
               150 if frame_matcher else 250, 100)
    for k, v in over.items():
        setattr(p, k, v)
    return p


def make_intrinsics(d: dict) -> Intrinsics:
    return Intrinsics(**{k: d[k] for k, _ in Intrinsics._fields_})


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "librvo_oracle.so")
    src = os.path.join(_HERE, "rvo_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Intrinsics), C.POINTER(Params)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_vo_create.restype = C.c_void_p
        L.orc_vo_create.argtypes = [C.c_void_p, C.c_int]
        L.orc_vo_destroy.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bgr_to_gray(bgr: np.ndarray) -> np.ndarray:
    """ConvertBGR2Gray, /root/reference/helper.cpp:55-64.  bgr: [..., 3] u8 -> [...] u8"""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty(bgr.shape[:-1], np.uint8)
    L = lib()
    L.orc_bgr_to_gray.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_bgr_to_gray(_p(bgr), out.size, _p(out))
    return out


def information(cov: np.ndarray) -> np.ndarray:
    """info = cov.inverse(), pose_graph.cpp:265 (6x6 partial-pivot elimination)."""
    cov = np.ascontiguousarray(cov, np.float64).reshape(6, 6)
    out = np.zeros((6, 6))
    lib().orc_information(_p(cov), _p(out))
    return out


def depth_to_cloud(depth: np.ndarray, x_to_z: float, y_to_z: float) -> np.ndarray:
    """DepthToPointCloud::operator(), /root/reference/helper.cpp:30-53.  depth: [H, W] u16 -> [H, W, 3] f32"""
    depth = np.ascontiguousarray(depth, np.uint16)
    h, w = depth.shape
    out = np.empty((h, w, 3), np.float32)
    L = lib()
    L.orc_depth_to_cloud.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
    L.orc_depth_to_cloud(_p(depth), w, h, x_to_z, y_to_z, _p(out))
    return out


def set_perturbation(norm_order: int = 0, true_division: int = 0, q_noise_rad: float = 0.0, seed: int = 1, rot_float: int = 0):
    """Process-wide perturbation of the oracle's conventions (struct Perturb, rvo_oracle.cpp); all zero = the oracle."""
    L = lib()
    L.orc_set_perturbation.argtypes = [C.c_int, C.c_int, C.c_double, C.c_uint, C.c_int]
    L.orc_set_perturbation(norm_order, true_division, q_noise_rad, seed, rot_float)


class Oracle:
    def __init__(self, intr: dict, params: Params | None = None):
        self.L = lib()
        self.intr = make_intrinsics(intr)
        self.params = params or default_params()
        self.h = C.c_void_p(self.L.orc_create(C.byref(self.intr), C.byref(self.params)))
        self.W, self.H = intr["width"], intr["height"]

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lut(self):
        x = np.empty((self.H, self.W), np.float32)
        y = np.empty((self.H, self.W), np.float32)
        self.L.orc_get_lut(self.h, _p(x), _p(y))
        return x, y

    def prepare(self, slot: int, gray: np.ndarray, depth: np.ndarray, thr: int):
        gray = np.ascontiguousarray(gray, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        self.L.orc_prepare_frame(self.h, C.c_int(slot), _p(gray), _p(depth), C.c_int(thr))

    def level_info(self, slot, level):
        o = np.zeros(6, np.int32)
        self.L.orc_level_info(self.h, C.c_int(slot), C.c_int(level), _p(o))
        return dict(width=int(o[0]), height=int(o[1]), stride=int(o[2]), num_keypoints=int(o[3]),
                    num_detected=int(o[4]), num_after_grid=int(o[5]))

    def level_image(self, slot, level):
        i = self.level_info(slot, level)
        out = np.empty((i["height"], i["stride"]), np.uint8)
        self.L.orc_level_image(self.h, C.c_int(slot), C.c_int(level), _p(out))
        return out[:, :i["width"]]

    def grid_keypoints(self, slot, level):
        i = self.level_info(slot, level)
        out = np.empty((i["num_after_grid"], 3), np.float32)
        self.L.orc_level_grid_keypoints(self.h, C.c_int(slot), C.c_int(level), _p(out))
        return out

    def keypoints(self, slot, level):
        n = self.level_info(slot, level)["num_keypoints"]
        uvs = np.empty((n, 3), np.float32)
        rect = np.empty((n, 2), np.float64)
        xyz = np.empty((n, 3), np.float64)
        self.L.orc_level_keypoints(self.h, C.c_int(slot), C.c_int(level), _p(uvs), _p(rect), _p(xyz))
        return uvs, rect, xyz

    def descriptors(self, slot, level):
        n = self.level_info(slot, level)["num_keypoints"]
        st = self.L.orc_descriptor_stride(self.h)
        out = np.empty((n, st), np.uint8)
        self.L.orc_level_descriptors(self.h, C.c_int(slot), C.c_int(level), _p(out))
        return out

    def initial_rotation(self, prev_slot, cur_slot):
        q = np.zeros(4)
        self.L.orc_initial_rotation(self.h, C.c_int(prev_slot), C.c_int(cur_slot), _p(q))
        return q

    def estimate_motion(self, ref_slot, tgt_slot, init3x4):
        r = PairResult()
        init = np.ascontiguousarray(init3x4, np.float64).reshape(12)
        self.L.orc_estimate_motion(self.h, C.c_int(ref_slot), C.c_int(tgt_slot), _p(init), C.byref(r))
        return result_to_dict(r)

    def match_pair(self, ref_slot, tgt_slot):
        r = PairResult()
        self.L.orc_match_pair(self.h, C.c_int(ref_slot), C.c_int(tgt_slot), C.byref(r))
        return result_to_dict(r)

    MARGIN_NAMES = ("window", "refine_gate", "esm_step", "sse_tie", "clique", "reproj_gate", "mean_reproj", "gn_sse_rel",
                    "degree_gap", "n_window", "n_clique", "n_esm")

    def margins(self) -> dict:
        """Distances of the last EstimateMotion's decided quantities to their thresholds (struct Margins, rvo_oracle.cpp)."""
        out = np.zeros(12)
        self.L.orc_get_margins(self.h, _p(out))
        return dict(zip(self.MARGIN_NAMES, out.tolist()))

    def match_tracks(self, n):
        """track_id of the last EstimateMotion's matches (motion_estimate.cpp:380-391), rows as in matches()."""
        out = np.zeros(n, np.int32)
        self.L.orc_get_match_tracks(self.h, C.c_int(n), _p(out))
        return out

    def match_consistency(self, n):
        """consistency_vec of the last EstimateMotion's matches: [n, n] u8, rows as matches(), columns = ids."""
        out = np.zeros((n, n), np.uint8)
        self.L.orc_get_match_consistency(self.h, C.c_int(n), _p(out))
        return out

    def keypoint_tracks(self, slot, level):
        out = np.zeros(self.level_info(slot, level)["num_keypoints"], np.int32)
        self.L.orc_level_tracks(self.h, C.c_int(slot), C.c_int(level), _p(out))
        return out

    def num_tracks(self):
        return int(self.L.orc_num_tracks(self.h))

    def matches(self, n):
        ints = np.zeros((n, 7), np.int32)
        dbl = np.zeros((n, 8), np.float64)
        self.L.orc_get_matches(self.h, C.c_int(n), _p(ints), _p(dbl))
        return ints, dbl


class OracleVO:
    """RGBDVO::ProcessFrame (vo.cpp:55-165) on the oracle, one stream."""

    def __init__(self, oracle: Oracle, base_slot: int = 0):
        self.o = oracle
        self.h = C.c_void_p(oracle.L.orc_vo_create(oracle.h, C.c_int(base_slot)))

    def process(self, gray, depth):
        gray = np.ascontiguousarray(gray, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        oi = np.zeros(8, np.int32)
        od = np.zeros(24, np.float64)
        rc = self.o.L.orc_vo_process(self.h, _p(gray), _p(depth), _p(oi), _p(od))
        return dict(rc=int(rc), fast_thr=int(oi[1]), keyframe_changed=bool(oi[2]), status=int(oi[3]),
                    num_inliers=int(oi[4]), num_matches=int(oi[5]), used_fallback=bool(oi[6]), pose=od[:12].reshape(3, 4).copy(),
                    curr_to_ref=od[12:].reshape(3, 4).copy())

    def close(self):
        if self.h:
            self.o.L.orc_vo_destroy(self.h)
            self.h = None


def result_to_dict(r: PairResult) -> dict:
    return dict(pose=np.array(r.pose).reshape(3, 4), cov=np.array(r.cov).reshape(6, 6), status=int(r.status),
                num_matches=int(r.num_matches), num_inliers=int(r.num_inliers),
                num_reprojection_failures=int(r.num_reprojection_failures),
                mean_reprojection_error=float(r.mean_reprojection_error))
