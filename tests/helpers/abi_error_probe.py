"""Run as a process of its own by tests/test_gpu_errors.py: every entry point of the C ABI with a bad argument -- a NULL
context, NULL buffers, counts and indices out of range -- must come back with a negative RVO_E_* code and a message, not
crash; a crash ends this process with a signal and the test names the call it died in (the last line printed)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slam_amd import _lib, synth  # noqa: E402
from slam_amd.api import Context, default_params  # noqa: E402

L = _lib.load()
intr = synth.default_intrinsics(320, 240)
ctx = Context(intr, default_params(frame_matcher=True, num_pyramid_levels=2), max_frames=4, max_pairs=2)
h = ctx.h
NULL = None
i32 = lambda *v: np.asarray(v, np.int32).ctypes.data_as(C.c_void_p)
buf = np.zeros(4 * 320 * 240 * 12, np.uint8)
B = buf.ctypes.data_as(C.c_void_p)
bad = 0


def expect_error(name, *args):
    global bad
    print("calling", name, flush=True)
    rc = getattr(L, name)(*args)
    msg = L.rvo_last_error().decode()
    if not (isinstance(rc, int) and rc < 0 and msg):
        print("  NOT REFUSED:", name, "rc", rc, "message", repr(msg), flush=True)
        bad += 1


# a NULL context everywhere
for name, args in [
    ("rvo_set_pair_chunks", (NULL, 1)), ("rvo_set_pipeline", (NULL, 1)), ("rvo_sync", (NULL,)), ("rvo_set_stream", (NULL, NULL)),
    ("rvo_prepare_frames", (NULL, 1, B, B, i32(10), i32(0))), ("rvo_prepare_frames_dev", (NULL, 1, B, B, i32(10), i32(0))),
    ("rvo_match_pairs", (NULL, 1, i32(0), i32(1), NULL, B)), ("rvo_match_pairs_async", (NULL, 1, i32(0), i32(1), NULL)),
    ("rvo_match_frames_dev", (NULL, 1, B, B, i32(10, 10), i32(0, 1), NULL)), ("rvo_get_pair_results", (NULL, 1, B)),
    ("rvo_get_batch_results", (NULL, 0, 1, B)), ("rvo_initial_rotation", (NULL, 1, i32(0), i32(1), B)),
    ("rvo_streams_reset", (NULL, 1)), ("rvo_stream_step", (NULL, 1, B, B, B)), ("rvo_stream_step_dev", (NULL, 1, B, B)),
    ("rvo_get_stream_results", (NULL, 1, B)), ("rvo_streams_set_source_index", (NULL, 1, i32(0))), ("rvo_streams_depth_in_place", (NULL, 1)),
    ("rvo_streams_record", (NULL, 4)), ("rvo_get_trajectories", (NULL, 1, 1, B)), ("rvo_get_level_info", (NULL, 0, 0, B)),
    ("rvo_get_level_image", (NULL, 0, 0, B)), ("rvo_get_grid_keypoints", (NULL, 0, 0, B)), ("rvo_get_keypoints", (NULL, 0, 0, B, B, B)),
    ("rvo_get_descriptors", (NULL, 0, 0, B)), ("rvo_get_matches", (NULL, 0, 1, B, B)), ("rvo_get_match_tracks", (NULL, 0, 1, B)),
    ("rvo_get_match_consistency", (NULL, 0, 1, B)), ("rvo_get_keypoint_tracks", (NULL, 0, 0, B)), ("rvo_reset_tracks", (NULL, 1, i32(0))),
    ("rvo_get_lut", (NULL, B, B)), ("rvo_get_pair_information", (NULL, 1, B)), ("rvo_get_pair_information_noalloc", (NULL, 1, B)),
    ("rvo_bgr_to_gray_dev", (NULL, 1, B, B)), ("rvo_profile_enable", (NULL, 1)), ("rvo_wait_upload", (NULL,)),
]:
    expect_error(name, *args)

# a live context, bad arguments
for name, args in [
    ("rvo_set_pair_chunks", (h, 0)), ("rvo_set_pair_chunks", (h, 99)),
    ("rvo_prepare_frames", (h, 1, NULL, B, i32(10), i32(0))), ("rvo_prepare_frames", (h, 1, B, NULL, i32(10), i32(0))),
    ("rvo_prepare_frames", (h, 1, B, B, NULL, i32(0))), ("rvo_prepare_frames", (h, 1, B, B, i32(10), NULL)),
    ("rvo_prepare_frames", (h, 5, B, B, i32(1, 1, 1, 1, 1), i32(0, 1, 2, 3, 0))),      # more frames than slots
    ("rvo_prepare_frames", (h, 1, B, B, i32(10), i32(4))), ("rvo_prepare_frames", (h, 1, B, B, i32(10), i32(-1))),
    ("rvo_prepare_frames", (h, 1, B, B, i32(300), i32(0))), ("rvo_prepare_frames", (h, -1, B, B, i32(10), i32(0))),
    ("rvo_prepare_frames", (h, 2, B, B, i32(10, 10), i32(1, 1))),                          # one slot twice
    ("rvo_match_pairs", (h, 1, NULL, i32(1), NULL, B)),  # (out = NULL is legal there: estimate and synchronise, results fetched later)
    ("rvo_match_pairs", (h, 3, i32(0, 1, 2), i32(1, 2, 3), NULL, B)),                      # more pairs than work areas
    ("rvo_match_pairs", (h, 1, i32(0), i32(9), NULL, B)), ("rvo_match_pairs", (h, 1, i32(-2), i32(1), NULL, B)),
    ("rvo_match_frames_dev", (h, 1, NULL, B, i32(10, 10), i32(0, 1), NULL)), ("rvo_match_frames_dev", (h, 3, B, B, i32(1, 1, 1, 1, 1, 1), i32(0, 1, 2, 3, 0, 1), NULL)),
    ("rvo_match_frames_dev", (h, 1, B, B, i32(10, 10), i32(0, 0), NULL)), ("rvo_match_frames_dev", (h, 1, B, B, i32(10, -5), i32(0, 1), NULL)),
    ("rvo_get_pair_results", (h, 3, B)), ("rvo_get_pair_results", (h, 1, NULL)), ("rvo_get_batch_results", (h, 7, 1, B)),
    ("rvo_initial_rotation", (h, 1, i32(0), i32(8), B)), ("rvo_initial_rotation", (h, 1, i32(0), i32(1), NULL)),
    ("rvo_streams_reset", (h, 0)), ("rvo_streams_reset", (h, 99)), ("rvo_stream_step", (h, 1, NULL, B, B)),
    ("rvo_get_level_info", (h, 9, 0, B)), ("rvo_get_level_info", (h, 0, 7, B)), ("rvo_get_level_info", (h, 0, 0, NULL)),
    ("rvo_get_level_image", (h, -1, 0, B)), ("rvo_get_keypoints", (h, 0, 5, B, B, B)), ("rvo_get_descriptors", (h, 4, 0, B)),
    ("rvo_get_matches", (h, 5, 1, B, B)), ("rvo_get_matches", (h, -1, 1, B, B)), ("rvo_get_match_consistency", (h, 2, 1, B)),
    ("rvo_get_keypoint_tracks", (h, 0, 2, B)), ("rvo_reset_tracks", (h, 1, i32(4))), ("rvo_reset_tracks", (h, 1, NULL)),
    ("rvo_get_lut", (h, NULL, B)), ("rvo_get_pair_information", (h, 3, B)), ("rvo_get_pair_information", (h, 1, NULL)),
    ("rvo_get_pair_information_noalloc", (h, 99999, B)), ("rvo_get_trajectories", (h, 1, 1, NULL)),
    ("rvo_streams_record", (h, -3)),
    ("rvo_bgr_to_gray_dev", (h, 1, NULL, B)), ("rvo_depth_to_point_cloud_dev", (h, 1, NULL, C.c_double(1.0), C.c_double(1.0), B)),
    ("rvo_wait_peer", (h, h)), ("rvo_wait_peer", (h, NULL)),
]:
    expect_error(name, *args)

# the context still works after all that
g0, d0, g1, d1, _ = synth.make_pair(3, intr, synth.PlaneScene(seed=1))
ctx.prepare_frames(np.stack([g0, g1]), np.stack([d0, d1]), 15, [0, 1])
r = ctx.match_pairs([0], [1])[0]
print("after the probes: status", r["status"], "matches", r["num_matches"], flush=True)
ok = r["status"] == 1 and r["num_matches"] > 50
ctx.close()
print("PROBE DONE bad =", bad, "ok =", ok, flush=True)
sys.exit(0 if bad == 0 and ok else 3)
