"""GPU parity of the front-of-path conversions against the CPU oracle: bit-exact,
including the signed zeros DepthToPointCloud produces for raw depth 0.  Parity unpinned (the reference holds no
fixtures for helper.cpp); the oracle restates helper.cpp:30-64 line by line."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bgr_to_gray_and_point_cloud(intr640):
    import torch
    from oracle import pyoracle
    from slam_amd.api import Context, default_params
    rng = np.random.default_rng(11)
    n = 3
    bgr = rng.integers(0, 256, (n, 480, 640, 3), dtype=np.uint8)
    bgr[0, 0, :8] = [[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1], [254, 255, 255], [128, 128, 128]]
    depth = rng.integers(0, 65536, (n, 480, 640), dtype=np.uint16)
    depth[1, :40] = 0
    ctx = Context(intr640, default_params(), max_frames=2, max_pairs=1)
    d_bgr = torch.from_numpy(bgr).cuda()
    d_gray = torch.empty((n, 480, 640), dtype=torch.uint8, device="cuda")
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
    d_cloud = torch.empty((n, 480, 640, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    xtoz, ytoz = 2 * math.tan(1.0226 / 2), 2 * math.tan(0.7966 / 2)
    ctx.bgr_to_gray_dev(n, d_bgr.data_ptr(), d_gray.data_ptr())
    ctx.depth_to_point_cloud_dev(n, d_depth.data_ptr(), xtoz, ytoz, d_cloud.data_ptr())
    ctx.sync()
    assert np.array_equal(d_gray.cpu().numpy(), pyoracle.bgr_to_gray(bgr))
    got = d_cloud.cpu().numpy()
    for f in range(n):
        want = pyoracle.depth_to_cloud(depth[f], xtoz, ytoz)
        assert np.array_equal(got[f].view(np.uint32), want.view(np.uint32))
    # the converted image feeds the path unchanged: FrameMatcher on the device-converted gray
    ctx.close()


@pytest.mark.parametrize("size,n", [((321, 243), 2), ((162, 121), 1), ((333, 100), 3)])
def test_conversions_at_sizes_that_leave_a_tail(size, n):
    """Pixel counts that are no multiple of 16 or 4 (the 16-byte and dword groups of k_bgr_to_gray end in a tail of single
    pixels), odd widths in the point cloud's x / W."""
    import torch
    from oracle import pyoracle
    from slam_amd import synth
    from slam_amd.api import Context, default_params
    W, H = size
    rng = np.random.default_rng(W * 1000 + H)
    bgr = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    depth = rng.integers(0, 65536, (n, H, W), dtype=np.uint16)
    depth[0, :3] = 0
    ctx = Context(synth.default_intrinsics(W, H), default_params(num_pyramid_levels=2), max_frames=2, max_pairs=1)
    d_bgr = torch.from_numpy(bgr).cuda()
    d_gray = torch.full((n * H * W + 64,), 77, dtype=torch.uint8, device="cuda")  # guard bytes behind the images
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
    d_cloud = torch.full((n * H * W * 3 + 64,), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    xtoz, ytoz = 2 * math.tan(1.0226 / 2), 2 * math.tan(0.7966 / 2)
    ctx.bgr_to_gray_dev(n, d_bgr.data_ptr(), d_gray.data_ptr())
    ctx.depth_to_point_cloud_dev(n, d_depth.data_ptr(), xtoz, ytoz, d_cloud.data_ptr())
    ctx.sync()
    g = d_gray.cpu().numpy()
    assert np.array_equal(g[:n * H * W].reshape(n, H, W), pyoracle.bgr_to_gray(bgr)) and (g[n * H * W:] == 77).all()
    c = d_cloud.cpu().numpy()
    assert (c[n * H * W * 3:] == 7.0).all()
    got = c[:n * H * W * 3].reshape(n, H, W, 3)
    for f in range(n):
        assert np.array_equal(got[f].view(np.uint32), pyoracle.depth_to_cloud(depth[f], xtoz, ytoz).view(np.uint32))
    ctx.close()
