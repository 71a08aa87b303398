"""Python mirror of the rgbdvo API on top of the C ABI (include/rvo.h).

`Context` is the batched engine (frame slots + pair work areas on one GPU);
`FrameMatcher` mirrors rgbdvo::FrameMatcher (/root/reference/rgbdvo/fm.hpp:14-44) and
`RGBDVOStreams` mirrors N independent rgbdvo::RGBDVO instances
(/root/reference/rgbdvo/vo.hpp:15-102).  All compute happens in the HIP library.
"""
from __future__ import annotations

import os
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Intrinsics, PairResult, Params, StreamResult

STATUS_NAMES = ["NO_DATA", "SUCCESS", "INSUFFICIENT_INLIERS", "OPTIMIZATION_FAILURE", "REPROJECTION_ERROR"]


class RvoError(RuntimeError):
    pass


def default_params(frame_matcher: bool = False, **over) -> Params:
    """GetRGBDVODefaultOptions / GetFrameMatcherDefaultOptions (settings.hpp:11-52)."""
    p = Params()
    L = _lib.load()
    (L.rvo_default_fm_params if frame_matcher else L.rvo_default_vo_params)(C.byref(p))
    for k, v in over.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def params_from_options(options: dict, frame_matcher: bool = False) -> Params:
    """Parse a reference-style string map (settings.hpp keys, helper.cpp:13-59 stoi/stod)."""
    p = default_params(frame_matcher)
    for key, val in options.items():
        name = key.replace("-", "_")
        if name == "min_feature_for_estimate":
            name = "min_features_for_estimate"
        if name == "feature_search_window":
            p.feature_search_window = float(val)
            continue
        if not hasattr(p, name):
            raise KeyError(f"unknown option {key!r}")
        cur = getattr(p, name)
        setattr(p, name, float(val) if isinstance(cur, float) else int(val))
    return p


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


class Context:
    def __init__(self, intr: dict, params: Params | None = None, max_frames: int = 8, max_pairs: int = 4,
                 device: int = 0):
        self.L = _lib.load()
        self.intr = Intrinsics(**{k: intr[k] for k, _ in Intrinsics._fields_})
        self.params = params or default_params()
        self.W, self.H = intr["width"], intr["height"]
        self.Wd, self.Hd = intr["depth_width"], intr["depth_height"]
        self.levels = self.params.num_pyramid_levels
        self.max_frames, self.max_pairs = max_frames, max_pairs
        h = C.c_void_p()
        self._chk(self.L.rvo_create(C.byref(self.intr), C.byref(self.params), device, max_frames, max_pairs,
                                    C.byref(h)))
        self.h = h

    def _chk(self, rc):
        if rc != 0:
            raise RvoError(f"rvo error {rc}: {self.L.rvo_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.rvo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        self._chk(self.L.rvo_set_stream(self.h, C.c_void_p(stream_ptr)))

    def sync(self):
        self._chk(self.L.rvo_sync(self.h))

    # -- frames
    def prepare_frames(self, gray, depth, fast_thr, slots):
        gray = np.ascontiguousarray(gray, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        slots = _i32(slots)
        n = len(slots)
        thr = _i32(np.broadcast_to(np.asarray(fast_thr, np.int32), (n,)))
        assert gray.size == n * self.W * self.H and depth.size == n * self.Wd * self.Hd
        self._chk(self.L.rvo_prepare_frames(self.h, n, _ptr(gray), _ptr(depth), _ptr(thr), _ptr(slots)))

    def prepare_frames_dev(self, gray_ptr: int, depth_ptr: int, fast_thr, slots):
        slots = _i32(slots)
        n = len(slots)
        thr = _i32(np.broadcast_to(np.asarray(fast_thr, np.int32), (n,)))
        self._chk(self.L.rvo_prepare_frames_dev(self.h, n, C.c_void_p(gray_ptr), C.c_void_p(depth_ptr), _ptr(thr),
                                                _ptr(slots)))

    # -- pairs
    def match_pairs(self, ref_slots, tgt_slots, init_pose=None):
        ref_slots, tgt_slots = _i32(ref_slots), _i32(tgt_slots)
        n = len(ref_slots)
        out = (PairResult * n)()
        ip = None if init_pose is None else np.ascontiguousarray(init_pose, np.float64).reshape(n, 12)
        self._chk(self.L.rvo_match_pairs(self.h, n, _ptr(ref_slots), _ptr(tgt_slots), _ptr(ip), out))
        return [pair_result_to_dict(r) for r in out]

    def match_pairs_async(self, ref_slots, tgt_slots, init_pose=None):
        ref_slots, tgt_slots = _i32(ref_slots), _i32(tgt_slots)
        n = len(ref_slots)
        ip = None if init_pose is None else np.ascontiguousarray(init_pose, np.float64).reshape(n, 12)
        self._chk(self.L.rvo_match_pairs_async(self.h, n, _ptr(ref_slots), _ptr(tgt_slots), _ptr(ip)))

    def match_frames_dev(self, gray_ptr, depth_ptr, fast_thr, slots, init_pose=None):
        """FrameMatcher::Match for len(slots)/2 pairs of device-resident images (frames 2i, 2i+1 -> pair i); async."""
        slots = _i32(slots)
        n = len(slots) // 2
        thr = _i32(np.broadcast_to(np.asarray(fast_thr, np.int32), (2 * n,)))
        ip = None if init_pose is None else np.ascontiguousarray(init_pose, np.float64).reshape(n, 12)
        self._chk(self.L.rvo_match_frames_dev(self.h, n, C.c_void_p(gray_ptr), C.c_void_p(depth_ptr), _ptr(thr), _ptr(slots), _ptr(ip)))

    def bgr_to_gray_dev(self, n, bgr_ptr, gray_ptr):
        """ConvertBGR2Gray (helper.cpp:55-64) for n frames in device memory; async."""
        self._chk(self.L.rvo_bgr_to_gray_dev(self.h, n, C.c_void_p(bgr_ptr), C.c_void_p(gray_ptr)))

    def depth_to_point_cloud_dev(self, n, depth_ptr, x_to_z, y_to_z, cloud_ptr):
        """DepthToPointCloud::operator() (helper.cpp:30-53) for n depth images in device memory; async."""
        self._chk(self.L.rvo_depth_to_point_cloud_dev(self.h, n, C.c_void_p(depth_ptr), float(x_to_z), float(y_to_z),
                                                      C.c_void_p(cloud_ptr)))

    def set_pair_chunks(self, chunks: int):
        self._chk(self.L.rvo_set_pair_chunks(self.h, chunks))

    def set_pipeline(self, staged: bool):
        """rvo_set_pipeline: staged = consecutive match_frames_dev calls overlap inside this context (stage per stream)."""
        self._chk(self.L.rvo_set_pipeline(self.h, 1 if staged else 0))

    def wait_peer(self, peer: "Context"):
        """What is enqueued next waits until the FAST stage of peer's batch in flight is done (rvo_wait_peer)."""
        self._chk(self.L.rvo_wait_peer(self.h, peer.h))

    def pair_information(self, n):
        """info = cov.inverse() of the first n pair results (pose_graph.cpp:265), [n, 6, 6]; zeros where not SUCCESS."""
        out = np.zeros((n, 6, 6))
        self._chk(self.L.rvo_get_pair_information(self.h, n, _ptr(out)))
        return out

    def pair_information_noalloc(self, n):
        """pair_information() through the entry point that allocates nothing (fixed staging array in the library)."""
        out = np.zeros((n, 6, 6))
        self._chk(self.L.rvo_get_pair_information_noalloc(self.h, n, _ptr(out)))
        return out

    def pair_results(self, n):
        out = (PairResult * n)()
        self._chk(self.L.rvo_get_pair_results(self.h, n, out))
        return [pair_result_to_dict(r) for r in out]

    def initial_rotation(self, prev_slots, cur_slots):
        prev_slots, cur_slots = _i32(prev_slots), _i32(cur_slots)
        q = np.zeros((len(prev_slots), 4))
        self._chk(self.L.rvo_initial_rotation(self.h, len(prev_slots), _ptr(prev_slots), _ptr(cur_slots), _ptr(q)))
        return q

    # -- inspection
    def level_info(self, slot, level):
        o = np.zeros(6, np.int32)
        self._chk(self.L.rvo_get_level_info(self.h, slot, level, _ptr(o)))
        return dict(width=int(o[0]), height=int(o[1]), stride=int(o[2]), num_keypoints=int(o[3]),
                    num_detected=int(o[4]), num_after_grid=int(o[5]))

    def level_image(self, slot, level):
        i = self.level_info(slot, level)
        out = np.empty((i["height"], i["stride"]), np.uint8)
        self._chk(self.L.rvo_get_level_image(self.h, slot, level, _ptr(out)))
        return out[:, :i["width"]]

    def grid_keypoints(self, slot, level):
        n = self.level_info(slot, level)["num_after_grid"]
        out = np.empty((n, 3), np.float32)
        self._chk(self.L.rvo_get_grid_keypoints(self.h, slot, level, _ptr(out)))
        return out

    def keypoints(self, slot, level):
        n = self.level_info(slot, level)["num_keypoints"]
        uvs = np.empty((n, 3), np.float32)
        rect = np.empty((n, 2), np.float64)
        xyz = np.empty((n, 3), np.float64)
        self._chk(self.L.rvo_get_keypoints(self.h, slot, level, _ptr(uvs), _ptr(rect), _ptr(xyz)))
        return uvs, rect, xyz

    def descriptors(self, slot, level):
        n = self.level_info(slot, level)["num_keypoints"]
        out = np.empty((n, 80), np.uint8)
        self._chk(self.L.rvo_get_descriptors(self.h, slot, level, _ptr(out)))
        return out

    def matches(self, pair, n):
        ints = np.zeros((n, 7), np.int32)
        dbl = np.zeros((n, 8), np.float64)
        rc = self.L.rvo_get_matches(self.h, pair, n, _ptr(ints), _ptr(dbl))
        if rc < 0:
            self._chk(rc)
        return ints, dbl

    def match_tracks(self, pair, n):
        """FeatureMatch::track_id of the rows matches(pair, n) returns (motion_estimate.cpp:380-391)."""
        out = np.zeros(n, np.int32)
        rc = self.L.rvo_get_match_tracks(self.h, pair, n, _ptr(out))
        if rc < 0:
            self._chk(rc)
        return out

    def match_consistency(self, pair, n):
        """FeatureMatch::consistency_vec of the rows matches(pair, n) returns: [n, n] u8, column = the other match's id."""
        out = np.zeros((n, n), np.uint8)
        rc = self.L.rvo_get_match_consistency(self.h, pair, n * n, _ptr(out))
        if rc < 0:
            self._chk(rc)
        return out

    def keypoint_tracks(self, slot, level):
        """KeypointData::track_id of a level's keypoints (-1 = none)."""
        out = np.zeros(self.level_info(slot, level)["num_keypoints"], np.int32)
        rc = self.L.rvo_get_keypoint_tracks(self.h, slot, level, _ptr(out))
        if rc < 0:
            self._chk(rc)
        return out

    def reset_tracks(self, slots):
        """track_id = -1 on every keypoint of these slots, as after Frame::PrepareFrame (frame.cpp:138); async."""
        slots = _i32(slots)
        self._chk(self.L.rvo_reset_tracks(self.h, len(slots), _ptr(slots)))

    def lut(self):
        x = np.empty((self.H, self.W), np.float32)
        y = np.empty((self.H, self.W), np.float32)
        self._chk(self.L.rvo_get_lut(self.h, _ptr(x), _ptr(y)))
        return x, y

    # -- streams
    def streams_reset(self, n_streams):
        self._chk(self.L.rvo_streams_reset(self.h, n_streams))

    def stream_step(self, gray, depth, n_streams):
        gray = np.ascontiguousarray(gray, np.uint8)
        depth = np.ascontiguousarray(depth, np.uint16)
        out = (StreamResult * n_streams)()
        self._chk(self.L.rvo_stream_step(self.h, n_streams, _ptr(gray), _ptr(depth), out))
        return [stream_result_to_dict(r) for r in out]

    def stream_step_host_async(self, gray, depth, n_streams):
        """rvo_stream_step_host_async: returns once the step is enqueued; gray / depth (ideally from host_array) must stay
        untouched until wait_upload()."""
        assert gray.dtype == np.uint8 and depth.dtype == np.uint16 and gray.flags.c_contiguous and depth.flags.c_contiguous
        self._chk(self.L.rvo_stream_step_host_async(self.h, n_streams, _ptr(gray), _ptr(depth)))

    def wait_upload(self):
        self._chk(self.L.rvo_wait_upload(self.h))

    def stream_step_dev(self, gray_ptr: int, depth_ptr: int, n_streams):
        self._chk(self.L.rvo_stream_step_dev(self.h, n_streams, C.c_void_p(gray_ptr), C.c_void_p(depth_ptr)))

    def streams_set_source_index(self, n_sources: int, index):
        """Stream s reads image index[s] of the (then n_sources-image) arrays given to stream_step_dev; None clears."""
        if index is None:
            self._chk(self.L.rvo_streams_set_source_index(self.h, 0, None))
        else:
            index = _i32(index)
            self._chk(self.L.rvo_streams_set_source_index(self.h, n_sources, _ptr(index)))

    def streams_depth_in_place(self, on: bool = True):
        """rvo_streams_depth_in_place: stream_step_dev reads the depth images where they lie (keep them until the step is done)."""
        self._chk(self.L.rvo_streams_depth_in_place(self.h, 1 if on else 0))

    def stream_results(self, n_streams):
        out = (StreamResult * n_streams)()
        self._chk(self.L.rvo_get_stream_results(self.h, n_streams, out))
        return [stream_result_to_dict(r) for r in out]

    def streams_record(self, max_steps: int):
        """Keep the trajectories on the device: every later stream step appends one row per stream (rvo_streams_record)."""
        self._chk(self.L.rvo_streams_record(self.h, max_steps))

    def trajectories(self, n_streams: int, steps: int) -> np.ndarray:
        """[n_streams, steps, 16]: pose 3x4, status, num_inliers, rc, used_fallback per recorded step; synchronises."""
        out = np.zeros((n_streams, steps, _lib.TRAJ_DOUBLES))
        self._chk(self.L.rvo_get_trajectories(self.h, n_streams, steps, _ptr(out)))
        return out

    def trajectories_dev(self):
        """(device pointer, steps recorded, max_steps) of the [stream][max_steps][16] trajectory buffer."""
        p, n, cap = C.c_void_p(), C.c_int32(), C.c_int32()
        self._chk(self.L.rvo_trajectories_dev(self.h, C.byref(p), C.byref(n), C.byref(cap)))
        return p.value, n.value, cap.value

    def stream_results_dev_ptr(self) -> int:
        p = C.c_void_p()
        self._chk(self.L.rvo_stream_results_dev(self.h, C.byref(p)))
        return p.value

    # -- profiling
    def profile_enable(self, on=True, only: str | None = None):
        """HIP events around every launch, or only around launches of the kernel `only`."""
        self._chk(self.L.rvo_profile_select(self.h, only.encode() if only else None))
        self._chk(self.L.rvo_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        calls = (C.c_int32 * cap)()
        n = self.L.rvo_profile_read(self.h, cap, names, ms, calls)
        if n < 0:
            self._chk(n)
        return {names[i].decode(): dict(total_ms=float(ms[i]), calls=int(calls[i])) for i in range(min(n, cap))}


def pair_result_to_dict(r: PairResult) -> dict:
    return dict(pose=np.array(r.pose).reshape(3, 4), cov=np.array(r.cov).reshape(6, 6), status=int(r.status),
                num_matches=int(r.num_matches), num_inliers=int(r.num_inliers),
                num_reprojection_failures=int(r.num_reprojection_failures),
                mean_reprojection_error=float(r.mean_reprojection_error))


def stream_result_to_dict(r: StreamResult) -> dict:
    return dict(pose=np.array(r.pose).reshape(3, 4), curr_to_ref=np.array(r.curr_to_ref).reshape(3, 4),
                cov=np.array(r.cov).reshape(6, 6), rc=int(r.rc), fast_thr=int(r.fast_threshold),
                keyframe_changed=bool(r.keyframe_changed), status=int(r.status), num_matches=int(r.num_matches),
                num_inliers=int(r.num_inliers), used_fallback=bool(r.used_fallback))


class FrameMatcher:
    """rgbdvo::FrameMatcher (fm.hpp:14-44) at batch size 1: SetRefFrame / Match / GetMotionEstimate / GetCov, plus the
    batched loop closure of SURVEY.md section 8f row 1 (CacheKeyFrame / MatchCachedKeyFrames): keyframes stay prepared
    on the device and PoseGraph::LinearLoopClosure's candidate loop (pose_graph.cpp:85-145) is ONE rvo_match_pairs."""

    def __init__(self, intr: dict, options: dict | None = None, device: int = 0, max_keyframes: int = 0):
        self.ctx = Context(intr, params_from_options(options or {}, frame_matcher=True), max_frames=2 + max_keyframes,
                           max_pairs=max(1, max_keyframes), device=device)
        self.max_keyframes = max_keyframes
        self.thr = self.ctx.params.fast_init_threshold
        self.result = None

    def SetRefFrame(self, gray, raw_depth):  # fm.cpp:22-30
        self.ctx.prepare_frames(gray, raw_depth, self.thr, [0])

    def Match(self, gray, raw_depth) -> bool:  # fm.cpp:32-58
        self.ctx.prepare_frames(gray, raw_depth, self.thr, [1])
        self.result = self.ctx.match_pairs([0], [1])[0]
        return self.result["status"] == 1

    def GetMotionEstimate(self):
        m = np.eye(4)
        m[:3] = self.result["pose"]
        return m

    def GetCov(self):
        return self.result["cov"]

    def CacheKeyFrame(self, kf_id: int, gray, raw_depth):
        assert 0 <= kf_id < self.max_keyframes
        self.ctx.prepare_frames(gray, raw_depth, self.thr, [2 + kf_id])

    def MatchCachedKeyFrames(self, ref_id: int, target_ids):
        """[(target_id, valid, motion 4x4, cov 6x6, information 6x6)] for every candidate, in one batch."""
        for i in [ref_id, *target_ids]:  # ids map to slots 2 + id; -1 / -2 would name the live target / reference frame
            if not 0 <= int(i) < self.max_keyframes:
                raise RvoError(f"keyframe id {i} out of range")
        t = [2 + int(i) for i in target_ids]
        # SetRefFrame + one Match per candidate work on freshly prepared frames (fm.cpp:22-58): no track ids yet
        self.ctx.reset_tracks(sorted({2 + ref_id, *t}))
        res = self.ctx.match_pairs([2 + ref_id] * len(t), t)
        info = self.ctx.pair_information(len(t))
        out = []
        for k, r in enumerate(res):
            m = np.eye(4)
            m[:3] = r["pose"]
            out.append((int(target_ids[k]), r["status"] == 1, m, r["cov"], info[k]))
        return out


class HostArray:
    """A numpy array in page-locked host memory (rvo_host_alloc); free() or drop the last reference."""

    def __init__(self, shape, dtype):
        self.L = _lib.load()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        rc = self.L.rvo_host_alloc(n, C.byref(p))
        if rc != 0:
            raise RuntimeError(f"rvo_host_alloc: {rc}")
        self.ptr = p
        self.array = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)).view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.L.rvo_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PipelinedFrameMatcher:
    """Batches of FrameMatcher::Match (fm.cpp:32-58) kept in flight.  Each submit() is ONE rvo_match_frames_dev call;
    nothing is synchronised until results() / sync().  Inputs are device pointers and are read in place (include/rvo.h).

    contexts = 1 (default): the library's staged pipeline (rvo_set_pipeline) -- one context, whose four streams run the
    frame stages of batch k+3 beside the initial rotation + matching of batch k+2, the pair stages of batch k+1 and the
    latency-bound tail (clique, Gauss-Newton) of batch k; consecutive batches rotate through five sets of frame slots
    (a batch's frames are read by its pair stages while later batches' frames are being prepared).
    contexts >= 2: the round-1 arrangement -- `contexts` contexts take alternate batches, each as one chain
    (double buffering by the caller), with rvo_wait_peer before every submit."""

    def __init__(self, intr: dict, n_pairs: int, params: Params | None = None, device: int = 0, depth: int | None = None,
                 contexts: int = 1):
        if depth is not None:  # the old keyword: number of contexts taking alternate batches
            contexts = depth
        self.n_pairs = n_pairs
        p = params or default_params(frame_matcher=True)
        # RVO_STAGED=0 with one context: one chain per batch, nothing overlaps (tools/alone_times.py)
        self.staged = contexts == 1 and os.environ.get("RVO_STAGED", "1") != "0"
        if self.staged:
            nsets = int(os.environ.get("RVO_SLOT_SETS", "5"))  # frame-slot sets consecutive batches rotate through
            self.ctxs = [Context(intr, p, max_frames=2 * nsets * n_pairs, max_pairs=n_pairs, device=device)]
            self.ctxs[0].set_pipeline(True)
            self.slot_sets = [np.arange(2 * n_pairs * i, 2 * n_pairs * (i + 1), dtype=np.int32) for i in range(nsets)]
        else:
            self.ctxs = [Context(intr, p, max_frames=2 * n_pairs, max_pairs=n_pairs, device=device) for _ in range(contexts)]
            for c in self.ctxs:
                c.set_pipeline(False)
                c.set_pair_chunks(1)
            self.slot_sets = [np.arange(2 * n_pairs, dtype=np.int32)]
        self.k = 0
        self.stagger = os.environ.get("RVO_STAGGER", "1") != "0"  # rvo_wait_peer before every submit (+2 %)

    def slots_of(self, ticket: int) -> np.ndarray:
        """Frame slots batch `ticket` was prepared into (frames 2i, 2i+1 -> pair i)."""
        return self.slot_sets[ticket % len(self.slot_sets)]

    def submit(self, gray_ptr: int, depth_ptr: int, fast_thr, init_pose=None) -> int:
        """Enqueue one batch (frames 2i, 2i+1 -> pair i).  Returns the ticket results() takes."""
        c = self.ctxs[self.k % len(self.ctxs)]
        if not self.staged and self.stagger and len(self.ctxs) > 1:
            c.wait_peer(self.ctxs[(self.k - 1) % len(self.ctxs)])
        c.match_frames_dev(gray_ptr, depth_ptr, fast_thr, self.slots_of(self.k), init_pose)
        self.k += 1
        return self.k - 1

    def context_of(self, ticket: int) -> Context:
        keep = 2 if self.staged else len(self.ctxs)
        if not (self.k - keep <= ticket < self.k):
            raise ValueError("the results of that batch have been overwritten by a later one")
        return self.ctxs[ticket % len(self.ctxs)]

    def results(self, ticket: int):
        c = self.context_of(ticket)
        if not self.staged:
            return c.pair_results(self.n_pairs)
        out = (PairResult * self.n_pairs)()
        c._chk(c.L.rvo_get_batch_results(c.h, self.k - 1 - ticket, self.n_pairs, out))
        return [pair_result_to_dict(r) for r in out]

    def sync(self):
        for c in self.ctxs:
            c.sync()

    def close(self):
        for c in self.ctxs:
            c.close()


class RGBDVOStreams:
    """N independent rgbdvo::RGBDVO instances (vo.hpp:15-102) advanced in lock-step on one GPU."""

    def __init__(self, intr: dict, n_streams: int, options: dict | None = None, device: int = 0, params: Params | None = None):
        p = params or params_from_options(options or {})
        self.n = n_streams
        self.ctx = Context(intr, p, max_frames=3 * n_streams, max_pairs=n_streams, device=device)
        self.ctx.streams_reset(n_streams)

    def ProcessFrame(self, gray, raw_depth):
        """gray [n,H,W] u8, raw_depth [n,Hd,Wd] u16 -> list of per-stream result dicts."""
        return self.ctx.stream_step(gray, raw_depth, self.n)
