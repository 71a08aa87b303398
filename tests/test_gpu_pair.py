"""GPU parity: FrameMatcher::Match / MotionEstimator::EstimateMotion on the HIP path vs the CPU oracle.

Bit-exact: match lists (ref/target index, level, status), refined sub-pixel positions, rectified
positions, refined XYZ, consistency degrees, the std::sort order and the inlier set.
Tolerance (fp64 reductions are tree-ordered on the GPU, sequential in the oracle; sin/cos/atan2 come
from different libms): pose 1e-9 here, far inside north_star's 1e-5 m / 1e-6 rad.
"""
import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-9


def rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1)))


def compare_pair(ctx, orc, pair, rg, ro, exact_dbl=True):
    assert rg["status"] == ro["status"]
    assert rg["num_matches"] == ro["num_matches"]
    assert rg["num_inliers"] == ro["num_inliers"]
    assert rg["num_reprojection_failures"] == ro["num_reprojection_failures"]
    n = ro["num_matches"]
    ig, dg = ctx.matches(pair, n)
    io, do = orc.matches(n)
    assert np.array_equal(ig, io), "match list / sort order / inlier set differ"
    if exact_dbl:
        assert np.array_equal(dg[:, :7], do[:, :7], equal_nan=True), "refined uv / rect / xyz differ"
    np.testing.assert_allclose(dg[:, 7], do[:, 7], rtol=0, atol=1e-7)
    if 0 < n <= 1200:  # FeatureMatch::consistency_vec + compatibility_degree (feature.hpp:262-270), the whole N x N relation
        cg, co = ctx.match_consistency(pair, n), orc.match_consistency(n)
        assert np.array_equal(cg, co), "consistency vectors differ"
        assert np.array_equal(cg.sum(1), ig[:, 6]) and not cg[np.arange(n), ig[:, 5]].any()
    assert np.abs(rg["pose"][:, 3] - ro["pose"][:, 3]).max() < POSE_TOL
    assert rot_angle(rg["pose"][:, :3], ro["pose"][:, :3]) < 1e-7  # arccos near 1 has ~1e-8 resolution
    assert np.abs(rg["pose"] - ro["pose"]).max() < POSE_TOL
    if ro["status"] == 1:
        np.testing.assert_allclose(rg["cov"], ro["cov"], rtol=1e-6, atol=1e-12)
        assert abs(rg["mean_reprojection_error"] - ro["mean_reprojection_error"]) < 1e-9


@pytest.mark.parametrize("seed", [0, 1, 2, 5])
def test_match_pair_640(scene, intr640, seed):
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    ctx = Context(intr640, default_params(frame_matcher=True), max_frames=2, max_pairs=1)
    orc = Oracle(intr640, odef(frame_matcher=True))
    g0, d0, g1, d1, T = synth.make_pair(seed, intr640, scene)
    ctx.prepare_frames(np.stack([g0, g1]), np.stack([d0, d1]), 20, [0, 1])
    orc.prepare(0, g0, d0, 20)
    orc.prepare(1, g1, d1, 20)
    qg = ctx.initial_rotation([0], [1])[0]
    qo = orc.initial_rotation(0, 1)
    np.testing.assert_allclose(qg, qo, rtol=0, atol=1e-9)
    rg = ctx.match_pairs([0], [1])[0]
    ro = orc.match_pair(0, 1)
    assert ro["status"] == 1 and ro["num_inliers"] > 100
    compare_pair(ctx, orc, 0, rg, ro)
    # and the estimate is right: ground truth within depth-quantisation noise
    assert np.abs(rg["pose"] - T[:3]).max() < 2e-3
    ctx.close()


def test_match_batch_with_given_init(scene, intr640):
    """Several pairs in one call, EstimateMotion with an explicit initial estimate (the VO call form, vo.cpp:118-122)."""
    from oracle.pyoracle import Oracle
    from slam_amd.api import Context
    n = 6
    ctx = Context(intr640, max_frames=2 * n, max_pairs=n)
    orc = Oracle(intr640)
    grays, depths, inits = [], [], []
    for s in range(n):
        g0, d0, g1, d1, T = synth.make_pair(20 + s, intr640, scene, invalid_frac=0.15 if s % 2 else 0.0,
                                            max_t=0.05, max_rot_deg=1.5)
        grays += [g0, g1]
        depths += [d0, d1]
        noisy = T.copy()
        noisy[:3, 3] += 0.01 * (s - 2)
        inits.append(noisy[:3])
        orc.prepare(2 * s, g0, d0, 15)
        orc.prepare(2 * s + 1, g1, d1, 15)
    ctx.prepare_frames(np.stack(grays), np.stack(depths), 15, list(range(2 * n)))
    refs, tgts = list(range(0, 2 * n, 2)), list(range(1, 2 * n, 2))
    res = ctx.match_pairs(refs, tgts, np.stack(inits))
    for s in range(n):
        ro = orc.estimate_motion(2 * s, 2 * s + 1, inits[s])
        compare_pair(ctx, orc, s, res[s], ro)
    ctx.close()


def test_adversarial_low_texture_invalid_depth(intr640):
    """BASELINE config 5: ~5 grey levels of contrast and 30 % invalid depth: sparse matches, refinement failures,
    NaN-clamp quirk, INSUFFICIENT_INLIERS."""
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    sc = synth.PlaneScene(contrast=0.12)
    statuses = set()
    ctx = Context(intr640, default_params(frame_matcher=True), max_frames=2, max_pairs=1)
    orc = Oracle(intr640, odef(frame_matcher=True))
    for seed in range(4):
        g0, d0, g1, d1, _ = synth.make_pair(40 + seed, intr640, sc, invalid_frac=0.3)
        thr = 5
        ctx.prepare_frames(np.stack([g0, g1]), np.stack([d0, d1]), thr, [0, 1])
        orc.prepare(0, g0, d0, thr)
        orc.prepare(1, g1, d1, thr)
        rg = ctx.match_pairs([0], [1])[0]
        ro = orc.match_pair(0, 1)
        statuses.add(ro["status"])
        compare_pair(ctx, orc, 0, rg, ro)
    ctx.close()
    print("statuses seen:", statuses)


def test_no_features_pair(intr640):
    """Constant images: zero keypoints, zero matches -> INSUFFICIENT_INLIERS, pose = initial estimate."""
    from slam_amd.api import Context
    ctx = Context(intr640, max_frames=2, max_pairs=1)
    g = np.full((2, 480, 640), 90, np.uint8)
    d = np.full((2, 480, 640), 2000, np.uint16)
    ctx.prepare_frames(g, d, 20, [0, 1])
    r = ctx.match_pairs([0], [1])[0]
    assert r["status"] == 2 and r["num_matches"] == 0 and r["num_inliers"] == 0
    assert np.allclose(r["pose"], np.eye(4)[:3])
    ctx.close()


@pytest.mark.parametrize("thr,per_grid,lo,hi", [(20, 25, 2049, 10**6), (20, 9, 1025, 2048), (20, 4, 300, 1024)])
def test_large_frames_config4(scene, thr, per_grid, lo, hi):
    """BASELINE config 4: 1280x960, 4 pyramid levels (large consistency matrix / clique stress).  The three cases put
    the number of matches into the three ranges the greedy walk treats differently: k_clique's own walk (rows staged in
    LDS, N > 2048), k_clique_walk with 64-word rows in registers, and with double-buffered 32-word rows."""
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    intr = synth.default_intrinsics(1280, 960)
    kw = dict(num_pyramid_levels=4, max_keypoints_per_grid=per_grid)
    ctx = Context(intr, default_params(frame_matcher=True, **kw), max_frames=2, max_pairs=1)
    orc = Oracle(intr, odef(frame_matcher=True, **kw))
    g0, d0, g1, d1, T = synth.make_pair(7, intr, scene)
    ctx.prepare_frames(np.stack([g0, g1]), np.stack([d0, d1]), thr, [0, 1])
    orc.prepare(0, g0, d0, thr)
    orc.prepare(1, g1, d1, thr)
    rg = ctx.match_pairs([0], [1])[0]
    ro = orc.match_pair(0, 1)
    assert lo <= ro["num_matches"] <= hi, ro["num_matches"]
    compare_pair(ctx, orc, 0, rg, ro)
    ctx.close()


def test_match_pair_without_the_split_clique_walk(scene, intr640, monkeypatch):
    """RVO_CLIQUE_SPLIT=0: k_clique sorts AND walks (the form before k_clique_walk existed); same results."""
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    monkeypatch.setenv("RVO_CLIQUE_SPLIT", "0")
    ctx = Context(intr640, default_params(frame_matcher=True), max_frames=2, max_pairs=1)
    orc = Oracle(intr640, odef(frame_matcher=True))
    g0, d0, g1, d1, T = synth.make_pair(3, intr640, scene)
    ctx.prepare_frames(np.stack([g0, g1]), np.stack([d0, d1]), 20, [0, 1])
    orc.prepare(0, g0, d0, 20)
    orc.prepare(1, g1, d1, 20)
    rg = ctx.match_pairs([0], [1])[0]
    ro = orc.match_pair(0, 1)
    compare_pair(ctx, orc, 0, rg, ro)
    ctx.close()


def test_match_frames_dev_equals_two_call_path(scene, intr640):
    """rvo_match_frames_dev (fused, chunked over HIP streams) == rvo_prepare_frames_dev + rvo_match_pairs, bit for bit,
    and pair 0 / the last pair also equal the oracle."""
    import torch
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    n = 70  # > 64 so that the batch is cut into two chunks
    pairs = [synth.make_pair(300 + i, intr640, scene, invalid_frac=0.05 * (i % 3)) for i in range(6)]
    gray = np.stack([pairs[i % 6][j] for i in range(n) for j in (0, 2)])
    depth = np.stack([pairs[i % 6][j] for i in range(n) for j in (1, 3)])
    dg = torch.from_numpy(gray).cuda()
    dd = torch.from_numpy(depth.view(np.int16)).cuda()
    torch.cuda.synchronize()
    slots = np.arange(2 * n, dtype=np.int32)
    a = Context(intr640, default_params(frame_matcher=True), max_frames=2 * n, max_pairs=n)
    b = Context(intr640, default_params(frame_matcher=True), max_frames=2 * n, max_pairs=n)
    a.prepare_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
    ra = a.match_pairs(slots[0::2], slots[1::2])
    b.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
    rb = b.pair_results(n)
    for i in range(n):
        assert ra[i]["status"] == rb[i]["status"] and ra[i]["num_matches"] == rb[i]["num_matches"]
        assert ra[i]["num_inliers"] == rb[i]["num_inliers"]
        assert np.array_equal(ra[i]["pose"], rb[i]["pose"]) and np.array_equal(ra[i]["cov"], rb[i]["cov"])
        if i in (0, 37, n - 1):
            ia, da = a.matches(i, ra[i]["num_matches"])
            ib, db = b.matches(i, rb[i]["num_matches"])
            assert np.array_equal(ia, ib) and np.array_equal(da, db, equal_nan=True)
    orc = Oracle(intr640, odef(frame_matcher=True))
    for i in (0, n - 1):
        orc.prepare(0, gray[2 * i], depth[2 * i], 20)
        orc.prepare(1, gray[2 * i + 1], depth[2 * i + 1], 20)
        compare_pair(b, orc, i, rb[i], orc.match_pair(0, 1))
    a.close()
    b.close()


def test_batches_in_flight_on_two_contexts_are_independent_and_deterministic(scene, intr640):
    """bench.py's arrangement: two contexts take alternate batches without any synchronisation in between.  Every batch
    must give the same bits as a context working alone, whatever is in flight beside it."""
    import torch
    from slam_amd.api import Context, default_params
    n = 48
    pairs = [synth.make_pair(500 + i, intr640, scene, invalid_frac=0.1 * (i % 2)) for i in range(4)]
    batches = []
    for shift in range(2):  # two different batches
        gray = np.stack([pairs[(i + shift) % 4][j] for i in range(n) for j in (0, 2)])
        depth = np.stack([pairs[(i + shift) % 4][j] for i in range(n) for j in (1, 3)])
        batches.append((torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()))
    torch.cuda.synchronize()
    slots = np.arange(2 * n, dtype=np.int32)

    def snapshot(ctx):
        res = ctx.pair_results(n)
        return [(r["status"], r["num_matches"], r["num_inliers"], r["pose"].tobytes(), r["cov"].tobytes()) for r in res]

    alone = []
    for dg, dd in batches:
        c = Context(intr640, default_params(frame_matcher=True), max_frames=2 * n, max_pairs=n)
        c.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
        alone.append(snapshot(c))
        c.close()
    ctxs = [Context(intr640, default_params(frame_matcher=True), max_frames=2 * n, max_pairs=n) for _ in range(2)]
    for c in ctxs:
        c.set_pair_chunks(1)
    for k in range(6):  # batches 0,1,0,1,... on contexts 0,1,0,1,... back to back
        dg, dd = batches[k % 2]
        ctxs[k % 2].match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
    for k in range(2):
        assert snapshot(ctxs[k]) == alone[k]
    # the same with rvo_wait_peer before every submit (a batch starts when the other context's FAST stage is done)
    for k in range(6):
        dg, dd = batches[(k + 1) % 2]
        ctxs[k % 2].wait_peer(ctxs[(k + 1) % 2])
        ctxs[k % 2].match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
    for k in range(2):
        assert snapshot(ctxs[k]) == alone[(k + 1) % 2]
    with pytest.raises(Exception):
        ctxs[0].wait_peer(ctxs[0])  # a context cannot wait for itself
    with pytest.raises(Exception):  # FAST thresholds outside 0..255 are rejected before anything is enqueued
        ctxs[0].match_frames_dev(batches[0][0].data_ptr(), batches[0][1].data_ptr(), 300, slots)
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("slot_sets", [5, 2, 1])
def test_staged_pipeline_in_one_context_equals_a_context_working_alone(scene, intr640, slot_sets, monkeypatch):
    """rvo_set_pipeline(1): one context keeps the frame stages of batch k+1, the pair stages of batch k and the tail of
    batch k-1 on its four streams.  Every batch must give the bits of a context that runs it alone as one chain; with a
    single set of frame slots (slot_sets = 1) the library itself has to hold batch k+1's frame stages back until batch
    k's pair stages have read the slots.  rvo_get_batch_results reaches the last two batches; rvo_get_matches the last."""
    import torch
    from slam_amd.api import Context, PipelinedFrameMatcher, default_params
    monkeypatch.setenv("RVO_SLOT_SETS", str(slot_sets))
    n = 40
    pairs = [synth.make_pair(700 + i, intr640, scene, invalid_frac=0.1 * (i % 2)) for i in range(5)]
    batches = []
    for shift in range(3):  # three different batches
        gray = np.stack([pairs[(i + shift) % 5][j] for i in range(n) for j in (0, 2)])
        depth = np.stack([pairs[(i + shift) % 5][j] for i in range(n) for j in (1, 3)])
        batches.append((torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()))
    torch.cuda.synchronize()
    slots = np.arange(2 * n, dtype=np.int32)

    def key(res):
        return [(r["status"], r["num_matches"], r["num_inliers"], r["pose"].tobytes(), r["cov"].tobytes()) for r in res]

    alone, alone_matches, alone_tracks = [], [], []
    for dg, dd in batches:
        c = Context(intr640, default_params(frame_matcher=True), max_frames=2 * n, max_pairs=n)
        c.set_pipeline(False)
        c.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
        res = c.pair_results(n)
        alone.append(key(res))
        alone_matches.append(c.matches(n - 1, res[n - 1]["num_matches"]))
        alone_tracks.append(c.match_tracks(n - 1, res[n - 1]["num_matches"]))
        c.close()
    pipe = PipelinedFrameMatcher(intr640, n, contexts=1)
    assert len(pipe.slot_sets) == slot_sets
    tickets = [pipe.submit(*[t.data_ptr() for t in batches[k % 3]], 20) for k in range(8)]
    assert key(pipe.results(tickets[-2])) == alone[6 % 3]  # while batch 7 may still be running
    assert key(pipe.results(tickets[-1])) == alone[7 % 3]
    with pytest.raises(ValueError):
        pipe.results(tickets[-3])  # overwritten by batch 7
    ctx = pipe.ctxs[0]
    im, dm = ctx.matches(n - 1, alone[7 % 3][n - 1][1])
    assert np.array_equal(im, alone_matches[7 % 3][0]) and np.array_equal(dm, alone_matches[7 % 3][1], equal_nan=True)
    # track ids: the context has counted the tracks of eight batches, the lone context those of one; the frames were
    # fresh both times (also with ONE set of slots, where the next batch's frame stages must not reset the keypoints'
    # ids before this batch's tail has written them), so the ids differ by a constant
    tr = ctx.match_tracks(n - 1, alone[7 % 3][n - 1][1])
    assert np.array_equal(tr - tr.min(), alone_tracks[7 % 3] - alone_tracks[7 % 3].min()) and tr.min() > 7 * 30000
    # given initial poses (no EstimateInitialRotation on the R stream), then back to one chain per call in the same context
    res_alone = None
    init = np.tile(np.eye(4)[:3].reshape(1, 12), (n, 1))
    for k in range(3):
        t = pipe.submit(*[x.data_ptr() for x in batches[k]], 20, init)
    staged_init = key(pipe.results(t))
    ctx.set_pipeline(False)
    ctx.match_frames_dev(batches[2][0].data_ptr(), batches[2][1].data_ptr(), 20, slots, init)
    res_alone = key(ctx.pair_results(n))
    assert staged_init == res_alone
    ctx.match_frames_dev(batches[1][0].data_ptr(), batches[1][1].data_ptr(), 20, slots)
    assert key(ctx.pair_results(n)) == alone[1]
    pipe.close()


def test_track_ids_of_a_batch_equal_one_estimator_working_through_the_pairs(scene, intr640):
    """motion_estimate.cpp:380-391 / 216-222: FeatureMatch::track_id and KeypointData::track_id.  A batch counts tracks
    like ONE MotionEstimator (FrameMatcher's, fm.hpp:41) that is handed its pairs in order; matching the same prepared
    frames again reuses the ids the reference keypoints already carry."""
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    n = 5
    pairs = [synth.make_pair(900 + i, intr640, scene, invalid_frac=0.1 * (i % 2)) for i in range(n)]
    gray = np.stack([pairs[i][j] for i in range(n) for j in (0, 2)])
    depth = np.stack([pairs[i][j] for i in range(n) for j in (1, 3)])
    slots = np.arange(2 * n, dtype=np.int32)
    ctx = Context(intr640, default_params(frame_matcher=True), max_frames=2 * n, max_pairs=n)
    orc = Oracle(intr640, odef(frame_matcher=True))
    ctx.prepare_frames(gray, depth, 20, slots)
    for s in range(2 * n):
        orc.prepare(s, gray[s], depth[s], 20)
    for rnd in range(2):  # second round: the same frames again, ids are already on the keypoints
        res = ctx.match_pairs(slots[0::2], slots[1::2])
        for i in range(n):
            ro = orc.match_pair(2 * i, 2 * i + 1)
            N = ro["num_matches"]
            assert res[i]["num_matches"] == N and N > 300
            tg, to = ctx.match_tracks(i, N), orc.match_tracks(N)
            assert np.array_equal(tg, to), (rnd, i)
            assert to.min() >= 0 and len(set(to.tolist())) == N  # one track per match
            for lvl in range(3):
                for s in (2 * i, 2 * i + 1):
                    assert np.array_equal(ctx.keypoint_tracks(s, lvl), orc.keypoint_tracks(s, lvl)), (rnd, i, s, lvl)
            ig, _ = ctx.matches(i, N)
            tgt_tr = ctx.keypoint_tracks(2 * i + 1, 0)
            lv0 = ig[:, 2] == 0
            inl = ig[:, 4] == 1
            assert np.array_equal(tgt_tr[ig[lv0 & inl, 1]], tg[lv0 & inl])       # inliers hand their id to the target keypoint
            assert (tgt_tr[ig[lv0 & ~inl, 1]] == -1).all()                       # outliers do not
        if rnd == 0:
            first_total = orc.num_tracks()
            assert first_total == sum(r["num_matches"] for r in res)            # pair i starts where pair i-1 stopped
    assert orc.num_tracks() >= first_total
    ctx.close()


def test_staged_pipeline_ragged_batches_and_mixed_entry_points(scene, intr640):
    """Batches of different sizes (1, 19, 40, 7 pairs) back to back through the staged pipeline, on the same slots, then
    the other entry points of the same context (rvo_match_pairs on the frames already prepared, rvo_get_pair_information,
    a stream step): every result equals what a fresh context computes for that batch alone."""
    import torch
    from slam_amd.api import Context, default_params
    cap = 40
    pairs = [synth.make_pair(1200 + i, intr640, scene, invalid_frac=0.1 * (i % 3 == 0)) for i in range(6)]
    gray = np.stack([pairs[i % 6][j] for i in range(cap) for j in (0, 2)])
    depth = np.stack([pairs[i % 6][j] for i in range(cap) for j in (1, 3)])
    dg = torch.from_numpy(gray).cuda()
    dd = torch.from_numpy(depth.view(np.int16)).cuda()
    torch.cuda.synchronize()

    def key(res):
        return [(r["status"], r["num_matches"], r["num_inliers"], r["pose"].tobytes(), r["cov"].tobytes()) for r in res]

    sizes = [1, 19, 40, 7]
    alone = {}
    for n in set(sizes):
        c = Context(intr640, default_params(frame_matcher=True), max_frames=2 * cap, max_pairs=cap)
        c.set_pipeline(False)
        c.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, np.arange(2 * n, dtype=np.int32))
        alone[n] = key(c.pair_results(n))
        c.close()
    ctx = Context(intr640, default_params(frame_matcher=True), max_frames=3 * cap, max_pairs=cap)  # staged by default
    from slam_amd.api import PairResult, pair_result_to_dict
    for rep in range(2):
        for n in sizes:
            ctx.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, np.arange(2 * n, dtype=np.int32))  # same slots every time
        out = (PairResult * sizes[-1])()
        ctx._chk(ctx.L.rvo_get_batch_results(ctx.h, 0, sizes[-1], out))
        assert key([pair_result_to_dict(r) for r in out]) == alone[sizes[-1]]
        out = (PairResult * sizes[-2])()
        ctx._chk(ctx.L.rvo_get_batch_results(ctx.h, 1, sizes[-2], out))
        assert key([pair_result_to_dict(r) for r in out]) == alone[sizes[-2]]
    # the frames of the last batch are still prepared in slots 0..13: match them again through rvo_match_pairs
    n = sizes[-1]
    res = ctx.match_pairs(np.arange(0, 2 * n, 2), np.arange(1, 2 * n, 2))
    assert key(res) == alone[n]
    info = ctx.pair_information(n)
    for i in range(n):
        if res[i]["status"] == 1:
            np.testing.assert_allclose(info[i] @ res[i]["cov"], np.eye(6), atol=1e-6)
    # zero pairs is a no-op, a batch beyond max_pairs an error; neither disturbs the pipeline
    ctx.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, np.zeros(0, np.int32))
    with pytest.raises(Exception):
        ctx.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, np.arange(2 * (cap + 1), dtype=np.int32))
    ctx.match_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, np.arange(2 * 19, dtype=np.int32))
    assert key(ctx.pair_results(19)) == alone[19]
    ctx.close()
