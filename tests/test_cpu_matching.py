"""Candidate windows + mutual SAD matching restated independently in numpy from the
reference text (motion_estimate.cpp:307-345, feature.cpp:6239-6335, feature.hpp:352-390) and compared with the oracle's
match list on its own keypoints -- VERDICT round 2, Next #5d.  A transcription slip in the oracle's matcher would otherwise
be invisible: the HIP path is only ever compared with the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle.pyoracle import Oracle, default_params
from slam_amd import synth
from golden_util import load_pair_fixture


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def numpy_matches(orc, levels, window=25.0):
    """[(level, ref index, target index)] in the order matchFeatures appends them: level by level, reference index ascending"""
    cam = np.zeros(4)
    orc.L.orc_rect_camera(orc.h, cam.ctypes.data_as(C.c_void_p))
    K = np.array([[cam[0], 0, cam[2], 0], [0, cam[1], cam[3], 0], [0, 0, 1, 0]])
    q0 = orc.initial_rotation(0, 1)
    init = np.eye(4)
    init[:3, :3] = quat_to_R(q0)            # fm.cpp:41-47: Identity rotated by the initial rotation
    P = K @ np.linalg.inv(init)             # reproj_mat = K * motion_estimate^-1, motion_estimate.cpp:309-313
    out = []
    for l in range(levels):
        _, rrect, rxyz = orc.keypoints(0, l)
        _, trect, _ = orc.keypoints(1, l)
        rd = orc.descriptors(0, l)[:, :80].astype(np.int32)
        td = orc.descriptors(1, l)[:, :80].astype(np.int32)
        if len(rxyz) == 0 or len(trect) == 0:
            continue
        h = (P @ np.concatenate([rxyz, np.ones((len(rxyz), 1))], 1).T).T
        proj = h[:, :2] / h[:, 2:3]
        t32 = trect.astype(np.float32).astype(np.float64)  # the const float getters, pyramid.hpp:95-101
        dist = np.sqrt(((t32[None, :, :] - proj[:, None, :]) ** 2).sum(-1))  # [ref, tgt]
        cand = dist < window
        sad = np.abs(rd[:, None, :] - td[None, :, :]).sum(-1)  # SAD over the 80 bytes, feature.hpp:364-381
        big = 80 * 255 + 1
        s = np.where(cand, sad, big)
        best_t = s.argmin(1)                 # strict <, first wins: the lowest target index among the minima
        best_r = s.argmin(0)                 # ... and the lowest reference index for a target (iteration order ref-major)
        for r in range(len(rxyz)):
            t = best_t[r]
            if s[r, t] < big and best_r[t] == r:
                out.append((l, r, int(t)))
    return out


def oracle_matches(orc, res):
    ints, _ = orc.matches(res["num_matches"])
    order = np.argsort(ints[:, 5])           # id = position before the degree sort (motion_estimate.cpp:183-185)
    return [(int(ints[i, 2]), int(ints[i, 0]), int(ints[i, 1])) for i in order]


def test_matcher_on_the_committed_fixture():
    z, intr, prm, thr = load_pair_fixture()
    orc = Oracle(intr, default_params(frame_matcher=True, **prm))
    orc.prepare(0, z["gray0"], z["depth0"], thr)
    orc.prepare(1, z["gray1"], z["depth1"], thr)
    res = orc.match_pair(0, 1)
    mine = numpy_matches(orc, prm["num_pyramid_levels"])
    assert len(mine) > 30 and mine == oracle_matches(orc, res)


@pytest.mark.parametrize("seed,invalid", [(0, 0.0), (5, 0.1)])
def test_matcher_on_640x480_pairs(scene, intr640, seed, invalid):
    orc = Oracle(intr640, default_params(frame_matcher=True))
    g0, d0, g1, d1, _ = synth.make_pair(seed, intr640, scene, invalid_frac=invalid)
    orc.prepare(0, g0, d0, 20)
    orc.prepare(1, g1, d1, 20)
    res = orc.match_pair(0, 1)
    mine = numpy_matches(orc, 3)
    theirs = oracle_matches(orc, res)
    assert len(mine) > 500 and mine == theirs
