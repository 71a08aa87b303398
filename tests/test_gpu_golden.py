"""GPU: the HIP path against the committed golden fixture (tests/golden/pair_160x120.npz), bit-exact."""
import numpy as np
import pytest

from golden_util import check_against_fixture, load_pair_fixture

pytestmark = pytest.mark.gpu


def test_hip_path_matches_fixture():
    from slam_amd.api import Context, default_params
    z, intr, prm, thr = load_pair_fixture()
    ctx = Context(intr, default_params(frame_matcher=True, **prm), max_frames=2, max_pairs=1)
    ctx.prepare_frames(np.stack([z["gray0"], z["gray1"]]), np.stack([z["depth0"], z["depth1"]]), thr, [0, 1])
    check_against_fixture(ctx, z, prm["num_pyramid_levels"])
    np.testing.assert_allclose(ctx.initial_rotation([0], [1])[0], z["q0"], atol=1e-9, rtol=0)
    r = ctx.match_pairs([0], [1])[0]
    assert [r["status"], r["num_matches"], r["num_inliers"], r["num_reprojection_failures"]] == list(z["counts"])
    ints, dbl = ctx.matches(0, r["num_matches"])
    assert np.array_equal(ints, z["m_ints"])
    assert np.array_equal(dbl[:, :7], z["m_dbl"][:, :7], equal_nan=True)
    assert np.abs(r["pose"] - z["pose"]).max() < 1e-9
    np.testing.assert_allclose(r["cov"], z["cov"], rtol=1e-6, atol=1e-12)
    ctx.close()
