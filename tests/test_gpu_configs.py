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
