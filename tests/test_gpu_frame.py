"""GPU parity: Frame::PrepareFrame on the HIP path vs the CPU oracle, bit-exact (integer/byte work).

Covers SURVEY.md section 8a rows a1-a9: pyramid images, FAST+NMS counts, grid-filtered
lists (libstdc++ nth_element order), keypoint records, descriptors, XYZ.
"""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def _compare_frame(ctx, orc, slot_g, slot_o, levels):
    for l in range(levels):
        ig, io = ctx.level_info(slot_g, l), orc.level_info(slot_o, l)
        assert np.array_equal(ctx.level_image(slot_g, l), orc.level_image(slot_o, l)), f"level {l} image"
        assert ig == io, f"level {l} info {ig} vs {io}"
        assert np.array_equal(ctx.grid_keypoints(slot_g, l), orc.grid_keypoints(slot_o, l)), f"level {l} grid list"
        ug, rg, xg = ctx.keypoints(slot_g, l)
        uo, ro, xo = orc.keypoints(slot_o, l)
        assert np.array_equal(ug, uo), f"level {l} keypoints"
        assert np.array_equal(rg, ro), f"level {l} rect uv"
        assert np.array_equal(xg, xo), f"level {l} xyz (bitwise)"
        assert np.array_equal(ctx.descriptors(slot_g, l), orc.descriptors(slot_o, l)[:, :80]), f"level {l} descriptors"


@pytest.mark.parametrize("thr", [5, 20, 45])
def test_prepare_frame_640(scene, intr640, thr):
    from oracle.pyoracle import Oracle
    from slam_amd.api import Context
    ctx = Context(intr640, max_frames=4, max_pairs=1)
    orc = Oracle(intr640)
    g0, d0, g1, d1, _ = synth.make_pair(3, intr640, scene, invalid_frac=0.1)
    ctx.prepare_frames(np.stack([g0, g1]), np.stack([d0, d1]), thr, [2, 0])
    orc.prepare(0, g0, d0, thr)
    orc.prepare(1, g1, d1, thr)
    _compare_frame(ctx, orc, 2, 0, 3)
    _compare_frame(ctx, orc, 0, 1, 3)
    ctx.close()


@pytest.mark.parametrize("thr", [5, 20])
def test_prepare_frame_noise_overflowing_buckets(intr640, thr):
    """Dense noise: every bucket overflows max-keypoints-per-grid, so the nth_element replay decides the sets; more than
    half of the pixels pass FAST's compass pre-test, so k_fast_grid's candidate queue overflows into its dense pass."""
    from oracle.pyoracle import Oracle
    from slam_amd.api import Context
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    d = rng.integers(0, 5000, (480, 640)).astype(np.uint16)
    d[rng.random((480, 640)) < 0.3] = 0
    ctx = Context(intr640, max_frames=1, max_pairs=1)
    orc = Oracle(intr640)
    ctx.prepare_frames(g[None], d[None], thr, [0])
    orc.prepare(0, g, d, thr)
    assert orc.level_info(0, 0)["num_detected"] > 5000
    _compare_frame(ctx, orc, 0, 0, 3)
    ctx.close()


def test_prepare_frame_odd_geometry(scene):
    """Non-multiple-of-16 width, non-default grid, 2 levels, distorted camera, half-res depth."""
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    intr = synth.default_intrinsics(360, 250)
    intr.update(k1=0.08, k2=-0.02, p1=0.001, p2=-0.0005, depth_width=180, depth_height=125)
    g0, d0 = scene.render(intr, synth.make_pose([0.1, 0, 0], [0.02, 0.01, 0.1]))
    kw = dict(num_pyramid_levels=2, grid_width=50, grid_height=36, max_keypoints_per_grid=7)
    ctx = Context(intr, default_params(**kw), max_frames=1, max_pairs=1)
    orc = Oracle(intr, odef(**kw))
    assert np.array_equal(ctx.lut()[0], orc.lut()[0]) and np.array_equal(ctx.lut()[1], orc.lut()[1])
    ctx.prepare_frames(g0[None], d0[None], 12, [0])
    orc.prepare(0, g0, d0, 12)
    _compare_frame(ctx, orc, 0, 0, 2)
    ctx.close()


def test_empty_frame(intr640):
    """Constant image: zero corners everywhere, zero keypoints, no crash."""
    from slam_amd.api import Context
    ctx = Context(intr640, max_frames=1, max_pairs=1)
    g = np.full((480, 640), 77, np.uint8)
    d = np.full((480, 640), 1500, np.uint16)
    ctx.prepare_frames(g[None], d[None], 20, [0])
    for l in range(3):
        i = ctx.level_info(0, l)
        assert i["num_keypoints"] == 0 and i["num_detected"] == 0
    ctx.close()


def test_batch_of_19_frames_matches_oracle_frame_by_frame(scene, intr640):
    """The frame kernels deal their workgroups XCD-aware in groups of 8 frames (rvo_dev.h:xcd_map) and keep the natural
    mapping for a tail of fewer than 8: 19 distinct frames in ONE call (two full groups + a tail of 3), permuted slots,
    every frame compared with the oracle."""
    from oracle.pyoracle import Oracle
    from slam_amd.api import Context
    n = 19
    frames = [synth.make_pair(900 + i, intr640, scene, invalid_frac=0.05 * (i % 3))[:2] for i in range(n)]
    gray = np.stack([f[0] for f in frames])
    depth = np.stack([f[1] for f in frames])
    slots = np.array([(7 * i + 3) % n for i in range(n)], np.int32)  # a permutation of 0..18
    thr = np.array([10 + (i % 4) * 10 for i in range(n)], np.int32)
    ctx = Context(intr640, max_frames=n, max_pairs=1)
    ctx.prepare_frames(gray, depth, thr, slots)
    orc = Oracle(intr640)
    for i in range(n):
        orc.prepare(0, gray[i], depth[i], int(thr[i]))
        _compare_frame(ctx, orc, int(slots[i]), 0, 3)
    ctx.close()


def test_distorted_camera_keypoints_that_round_onto_the_image_edge():
    """Barrel distortion (k1 < 0) maps some pixels to rect_u in [W-0.5, W) / rect_v in [H-0.5, H): (int)(x+0.5) is then W
    or H.  du = W names column 0 of the next row for the depth pixel and the ray (depth.hpp:41-43, depth.cpp:72): kept,
    bit-equal to the oracle.  Past the last row the reference reads outside its buffers (undefined): such keypoints are
    dropped on both sides (deviations).  Dense noise puts a corner candidate on almost every pixel."""
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    W, H = 320, 240
    intr = synth.default_intrinsics(W, H)
    intr["k1"] = -0.1
    kw = dict(num_pyramid_levels=2, max_keypoints_per_grid=60)
    ctx = Context(intr, default_params(**kw), max_frames=1, max_pairs=1)
    orc = Oracle(intr, odef(**kw))
    assert np.array_equal(ctx.lut()[0], orc.lut()[0]) and np.array_equal(ctx.lut()[1], orc.lut()[1])
    on_edge = 0
    for seed in range(4):
        rng = np.random.default_rng(seed)
        g = rng.integers(0, 256, (H, W), dtype=np.uint8)
        d = rng.integers(500, 4000, (H, W)).astype(np.uint16)
        ctx.prepare_frames(g[None], d[None], 20, [0])
        orc.prepare(0, g, d, 20)
        _compare_frame(ctx, orc, 0, 0, 2)
        for l in range(2):
            rect = orc.keypoints(0, l)[1]
            on_edge += int(((rect[:, 0] >= W - 0.5) | (rect[:, 1] >= H - 0.5)).sum())
    assert on_edge > 0, "no keypoint exercised the edge rounding"
    ctx.close()


def test_duplicate_slots_in_one_batch_are_rejected(intr640):
    from slam_amd.api import Context, RvoError
    ctx = Context(intr640, max_frames=4, max_pairs=2)
    g = np.zeros((2, 480, 640), np.uint8)
    d = np.zeros((2, 480, 640), np.uint16)
    with pytest.raises(RvoError, match="twice"):
        ctx.prepare_frames(g, d, 20, [1, 1])
    with pytest.raises(RvoError, match="range"):
        ctx.prepare_frames(g, d, 20, [1, 4])
    ctx.prepare_frames(g, d, 20, [1, 3])
    ctx.close()
