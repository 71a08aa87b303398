"""GPU: the entry point that allocates nothing (rvo_get_pair_information_noalloc, include/rvo.h) against the existing
rvo_get_pair_information, through the C ABI: bit-equal on two calls in a row, the second with a smaller input (the
library's static staging array still holds the first call's values), and the size limit is enforced.

Collected last on purpose (pytest -x would otherwise hide every test after a failure here)."""
import ctypes as C

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def test_noalloc_pair_information_equals_the_allocating_entry_point(scene):
    from slam_amd.api import Context, default_params
    intr = synth.default_intrinsics(320, 240)
    seq = list(synth.make_sequence(5, seed=21, intr=intr, scene=scene))
    ctx = Context(intr, default_params(frame_matcher=True, num_pyramid_levels=2), max_frames=5, max_pairs=4)
    ctx.prepare_frames(np.stack([s[0] for s in seq]), np.stack([s[1] for s in seq]), 20, list(range(5)))

    # first call: four pairs
    res = ctx.match_pairs([0, 0, 0, 0], [1, 2, 3, 4])
    assert sum(r["status"] == 1 for r in res) >= 3, [r["status"] for r in res]
    want4 = ctx.pair_information(4)
    got4 = ctx.pair_information_noalloc(4)
    assert np.array_equal(want4, got4)  # same kernel, same inputs: bit-equal
    assert want4.any()

    # second call right after it, smaller input (two other pairs): nothing of the first call may leak through
    res = ctx.match_pairs([4, 2], [3, 1])
    want2 = ctx.pair_information(2)
    got2 = ctx.pair_information_noalloc(2)
    assert np.array_equal(want2, got2)
    assert not np.array_equal(want2, want4[:2])  # the inputs really changed
    for k, r in enumerate(res):
        if r["status"] == 1:
            np.testing.assert_allclose(got2[k] @ r["cov"], np.eye(6), atol=1e-6)
        else:
            assert not got2[k].any()

    # n = 0 is a no-op; n beyond the context's pairs is rejected like in the existing entry point
    assert ctx.L.rvo_get_pair_information_noalloc(ctx.h, 0, None) == -1  # null output pointer
    out = np.zeros((5, 6, 6))
    assert ctx.L.rvo_get_pair_information_noalloc(ctx.h, 0, out.ctypes.data_as(C.c_void_p)) == 0
    assert ctx.L.rvo_get_pair_information_noalloc(ctx.h, 5, out.ctypes.data_as(C.c_void_p)) == -1
    ctx.close()


def test_noalloc_rejects_more_pairs_than_its_static_array_holds():
    from slam_amd.api import Context, default_params
    limit = 4096  # RVO_NOALLOC_MAX_PAIRS
    intr = synth.default_intrinsics(64, 64)
    ctx = Context(intr, default_params(frame_matcher=True, num_pyramid_levels=1), max_frames=2, max_pairs=limit + 1)
    out = np.ones((limit + 1, 6, 6))
    p = out.ctypes.data_as(C.c_void_p)
    assert ctx.L.rvo_get_pair_information(ctx.h, limit + 1, p) == 0  # the allocating entry point takes it
    assert ctx.L.rvo_get_pair_information_noalloc(ctx.h, limit + 1, p) == -1
    assert b"4096" in ctx.L.rvo_last_error()
    assert not out.any()  # no pair has been estimated: status NO_DATA -> zeros
    out[:] = 1
    assert ctx.L.rvo_get_pair_information_noalloc(ctx.h, limit, p) == 0  # the limit itself is fine
    assert not out[:limit].any() and out[limit:].all()
    ctx.close()
