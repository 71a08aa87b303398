"""GPU parity over unusual configurations, so that the special-cased fast paths of the kernels (cells on multiples of 4,
rows of 16 bytes, <= 1024 matches, images with the slot layout, ...) and their general fallbacks both meet the oracle.
Whole pair pipeline, same comparison as tests/test_gpu_pair.py (bit-exact lists, pose to 1e-9)."""
import numpy as np
import pytest

from slam_amd import synth
from test_gpu_pair import compare_pair

pytestmark = pytest.mark.gpu

CONFIGS = [
    # image size, parameter overrides, FAST threshold, what it exercises
    ((324, 244), dict(grid_width=50, grid_height=38, max_keypoints_per_grid=9, num_pyramid_levels=2), 15,
     "width not a multiple of 16 (gray copied, stride 336), cells not on multiples of 4, 2 levels"),
    ((640, 480), dict(grid_width=64, grid_height=48, max_keypoints_per_grid=40, num_pyramid_levels=3), 10,
     "4000+ keypoint slots: > 1024 matches (16 lanes per row in the clique, several consistency tiles)"),
    ((480, 360), dict(grid_width=96, grid_height=120, max_keypoints_per_grid=25, num_pyramid_levels=4,
                      feature_search_window=40.0, clique_inlier_threshold=0.05), 25,
     "big cells, 4 levels (coarsest 60x45), wide search window, tight clique threshold"),
    ((352, 288), dict(grid_width=33, grid_height=31, max_keypoints_per_grid=3, num_pyramid_levels=1,
                      inlier_max_reprojection_error=0.8, min_features_for_estimate=5), 20,
     "odd cells, one level (initial rotation on level 0), few keypoints per cell, harsh reprojection pruning"),
    ((362, 250), dict(grid_width=45, grid_height=40, max_keypoints_per_grid=10, num_pyramid_levels=3), 15,
     "odd level sizes (181x125 -> 90x62: the odd-length edge formulas of the pyramid in both directions)"),
    ((396, 314), dict(grid_width=116, grid_height=27, max_keypoints_per_grid=7, num_pyramid_levels=4), 12,
     "coarse levels ONE bucket wide and several high (99x78 -> 1 x 3 buckets: bucket row = index / 1; tools/fuzz_parity.py found it)"),
    ((320, 240), dict(grid_width=40, grid_height=40, max_keypoints_per_grid=1, num_pyramid_levels=2, min_features_for_estimate=5), 10,
     "one keypoint per bucket (bucket of a keypoint slot = slot / 1)"),
]


@pytest.mark.parametrize("size,over,thr,why", CONFIGS, ids=[c[3].split(":")[0].split(",")[0] for c in CONFIGS])
def test_pair_pipeline_unusual_configs(scene, size, over, thr, why):
    import torch
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    W, H = size
    intr = synth.default_intrinsics(W, H)
    orc = Oracle(intr, odef(frame_matcher=True, **over))
    n = 3
    pairs = [synth.make_pair(700 + 13 * i + W, intr, scene, invalid_frac=0.1 * i) for i in range(n)]
    gray = np.stack([p[j] for p in pairs for j in (0, 2)])
    depth = np.stack([p[j] for p in pairs for j in (1, 3)])
    # two-call path on host buffers ...
    a = Context(intr, default_params(frame_matcher=True, **over), max_frames=2 * n, max_pairs=n)
    a.prepare_frames(gray, depth, thr, np.arange(2 * n))
    ra = a.match_pairs(np.arange(0, 2 * n, 2), np.arange(1, 2 * n, 2))
    # ... and the fused device path
    b = Context(intr, default_params(frame_matcher=True, **over), max_frames=2 * n, max_pairs=n)
    dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()
    torch.cuda.synchronize()
    b.match_frames_dev(dg.data_ptr(), dd.data_ptr(), thr, np.arange(2 * n))
    rb = b.pair_results(n)
    some_success = False
    for i in range(n):
        orc.prepare(0, gray[2 * i], depth[2 * i], thr)
        orc.prepare(1, gray[2 * i + 1], depth[2 * i + 1], thr)
        ro = orc.match_pair(0, 1)
        compare_pair(a, orc, i, ra[i], ro)
        compare_pair(b, orc, i, rb[i], ro)
        some_success |= ro["status"] == 1
    assert some_success, why
    if "> 1024 matches" in why:
        assert max(r["num_matches"] for r in rb) > 1024, [r["num_matches"] for r in rb]
    a.close()
    b.close()


def test_largest_frame_the_pair_kernels_hold_and_the_first_one_beyond(scene):
    """Maximum sizes: 1600x1200 with the default 80x80 grid is 400 buckets = 10 000 keypoint slots per frame, the most whose
    sort keys + scratch fit the 160 KB of LDS k_clique works in (clique_lds_bytes, kernels_pair.h); one more column of
    buckets is refused by rvo_create with RVO_E_UNSUPPORTED and a message instead of a launch failure later."""
    import ctypes as C
    import torch
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd import _lib
    from slam_amd.api import Context, default_params
    W, H, thr = 1600, 1200, 10
    intr = synth.default_intrinsics(W, H)
    p = synth.make_pair(901, intr, scene)
    gray, depth = np.stack([p[0], p[2]]), np.stack([p[1], p[3]])
    orc = Oracle(intr, odef(frame_matcher=True))
    orc.prepare(0, gray[0], depth[0], thr)
    orc.prepare(1, gray[1], depth[1], thr)
    ro = orc.match_pair(0, 1)
    assert ro["status"] == 1 and ro["num_matches"] > 2048  # beyond k_clique_walk's register rows: the walk inside k_clique
    ctx = Context(intr, default_params(frame_matcher=True), max_frames=2, max_pairs=1)
    dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()
    torch.cuda.synchronize()
    ctx.match_frames_dev(dg.data_ptr(), dd.data_ptr(), thr, np.arange(2))
    compare_pair(ctx, orc, 0, ctx.pair_results(1)[0], ro)
    ctx.close()
    # 1680x1200: 21 x 15 + 11 x 8 + 6 x 4 = 427 buckets
    L = _lib.load()
    intr2 = synth.default_intrinsics(1680, 1200)
    i2 = _lib.Intrinsics(**{k: intr2[k] for k, _ in _lib.Intrinsics._fields_})
    prm = default_params(frame_matcher=True)
    h = C.c_void_p()
    rc = L.rvo_create(C.byref(i2), C.byref(prm), 0, 2, 1, C.byref(h))
    assert rc == -2 and h.value is None, rc  # RVO_E_UNSUPPORTED
    assert b"LDS" in L.rvo_last_error()
