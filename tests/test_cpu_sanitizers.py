"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host code only, no GPU).

oracle/rvo_oracle.cpp is the only pin of most parity rows, so a stray read in it would be invisible to every GPU test.  It is
built whole into tests/native/oracle_sanitized.cpp with -fsanitize=address,undefined -O1 and run over: the committed
160x120 fixture pair, 640x480 pairs, the adversarial set (low texture, 30 % invalid depth, FAST threshold 5: the
NaN-clamp and refinement-failure branches), a no-feature pair, a distorted camera with half-resolution depth, and a
20-frame RGBDVO sequence.  Zero sanitizer reports, and every number it prints equals the regular oracle library's."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from oracle.pyoracle import Oracle, OracleVO, default_params
from slam_amd import synth
from golden_util import load_pair_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sanitized(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "oracle_sanitized")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-msse3", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
                           "-o", exe, os.path.join(ROOT, "tests", "native", "oracle_sanitized.cpp")])
    return exe


def write_case(path, intr, prm, mode, thr, frames):
    with open(path, "wb") as f:
        f.write(struct.pack("4i", intr["width"], intr["height"], intr["depth_width"], intr["depth_height"]))
        f.write(struct.pack("9d", *[intr[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "k3", "p1", "p2")]))
        f.write(bytes(prm))
        f.write(struct.pack("3i", mode, thr, len(frames)))
        for g, d in frames:
            f.write(np.ascontiguousarray(g, np.uint8).tobytes())
            f.write(np.ascontiguousarray(d, np.uint16).tobytes())


def run(exe, case):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([exe, case], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr \
        and "LeakSanitizer" not in p.stderr, p.stderr[-3000:]
    return [l.split() for l in p.stdout.splitlines()]


def check_pairs(lines, intr, prm, thr, frames):
    orc = Oracle(intr, prm)
    rows = [l for l in lines if l[0] == "pair"]
    assert len(rows) == len(frames) // 2
    for i, row in enumerate(rows):
        orc.prepare(0, *frames[2 * i], thr)
        orc.prepare(1, *frames[2 * i + 1], thr)
        r = orc.match_pair(0, 1)
        assert [int(x) for x in row[2:6]] == [r["status"], r["num_matches"], r["num_inliers"], r["num_reprojection_failures"]], (i, row[:6])
        pose = np.array([float.fromhex(x) for x in row[6:18]]).reshape(3, 4)
        assert np.array_equal(pose, r["pose"]), i  # -O1 + sanitizers and -O3: the same IEEE operations in the same order
    orc.close()
    return [int(r[2]) for r in rows]


def test_oracle_pairs_under_asan_ubsan(sanitized, tmp_path, scene, intr640):
    prm = default_params(frame_matcher=True)
    frames, thr = [], 20
    for seed in (0, 5):
        g0, d0, g1, d1, _ = synth.make_pair(seed, intr640, scene)
        frames += [(g0, d0), (g1, d1)]
    frames += [(np.full((480, 640), 90, np.uint8), np.full((480, 640), 2000, np.uint16))] * 2  # no features at all
    case = str(tmp_path / "pairs.bin")
    write_case(case, intr640, prm, 0, thr, frames)
    st = check_pairs(run(sanitized, case), intr640, prm, thr, frames)
    assert st == [1, 1, 2]


def test_oracle_adversarial_and_fixture_under_asan_ubsan(sanitized, tmp_path, intr640):
    sc = synth.PlaneScene(contrast=0.12)
    prm = default_params(frame_matcher=True)
    frames = []
    for seed in range(4):
        g0, d0, g1, d1, _ = synth.make_pair(40 + seed, intr640, sc, invalid_frac=0.3)
        frames += [(g0, d0), (g1, d1)]
    case = str(tmp_path / "adv.bin")
    write_case(case, intr640, prm, 0, 5, frames)
    check_pairs(run(sanitized, case), intr640, prm, 5, frames)
    # the committed fixture pair (its own grid / level parameters)
    z, intr, p, thr = load_pair_fixture()
    prm = default_params(frame_matcher=True, **p)
    frames = [(z["gray0"], z["depth0"]), (z["gray1"], z["depth1"])]
    case = str(tmp_path / "fix.bin")
    write_case(case, intr, prm, 0, thr, frames)
    rows = run(sanitized, case)
    check_pairs(rows, intr, prm, thr, frames)
    assert [int(x) for x in rows[0][2:6]] == list(z["counts"])


def test_oracle_distorted_camera_half_res_depth_under_asan_ubsan(sanitized, tmp_path, scene):
    """The lens model's LUT, keypoints that round onto the image edge and the depth-scale paths (camera.cpp:16-67,
    depth.hpp:41-43): the places where the reference itself reads out of bounds."""
    intr = synth.default_intrinsics(320, 240)
    intr.update(depth_width=160, depth_height=120, k1=-0.28, k2=0.09, p1=0.001, p2=-0.0005, k3=0.0)
    prm = default_params(frame_matcher=True, num_pyramid_levels=2)
    frames = []
    for seed in (3, 4):
        g0, d0, g1, d1, _ = synth.make_pair(seed, intr, scene)
        frames += [(g0, d0), (g1, d1)]
    case = str(tmp_path / "dist.bin")
    write_case(case, intr, prm, 0, 12, frames)
    check_pairs(run(sanitized, case), intr, prm, 12, frames)


def test_oracle_vo_sequence_under_asan_ubsan(sanitized, tmp_path, scene, intr640):
    prm = default_params(warmup_frames=2)
    seq = [(g, d) for g, d, _ in synth.make_sequence(20, seed=9, intr=intr640, scene=scene, invalid_frac=0.05)]
    case = str(tmp_path / "vo.bin")
    write_case(case, intr640, prm, 1, 0, seq)
    rows = [l for l in run(sanitized, case) if l[0] == "frame"]
    assert len(rows) == 20
    orc = Oracle(intr640, prm)
    vo = OracleVO(orc)
    for i, (g, d) in enumerate(seq):
        r = vo.process(g, d)
        got = [int(x) for x in rows[i][2:9]]
        assert got == [r["rc"], r["fast_thr"], int(r["keyframe_changed"]), r["status"], r["num_inliers"], r["num_matches"], int(r["used_fallback"])], i
        assert np.array_equal(np.array([float.fromhex(x) for x in rows[i][9:21]]).reshape(3, 4), r["pose"]), i
    assert r["status"] == 1
    vo.close()
    orc.close()
