"""Randomised parity sweep of the STREAM path ON THE GPU BOX (not part of the test suite): random image sizes, grids, pyramid
depths, warm-up lengths, numbers of streams and of interleaved stream groups; sequences with invalid depth, occasional
blank or foreign frames (lost tracking, the fall-back to the previous frame, vo.cpp:131-152) -- every step of every stream
compared with OracleVO as tests/test_gpu_stream.py does (return code, FAST threshold, keyframe flag, counts and status exact,
poses to 1e-8, track ids exact).
    python tools/fuzz_streams.py [n_configs] [seed]       exit 1 on a mismatch"""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.pyoracle import Oracle, OracleVO, default_params as odef  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd.api import RGBDVOStreams, default_params  # noqa: E402


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    scenes = [synth.PlaneScene(seed=s) for s in range(3)]
    bad, steps_done, est, retried = 0, 0, 0, 0
    t0 = time.time()
    for c in range(n_cfg):
        W, H = int(rng.integers(200, 660)), int(rng.integers(160, 500))  # odd sizes too (rows then are not 16-byte multiples: the copying ingest)
        levels = int(rng.integers(1, 5))
        while levels > 1 and min(W, H) >> (levels - 1) < 24:
            levels -= 1
        warm = int(rng.integers(1, 4))
        over = dict(grid_width=int(rng.integers(32, 121)), grid_height=int(rng.integers(32, 121)),
                    max_keypoints_per_grid=int(rng.integers(4, 41)), num_pyramid_levels=levels, warmup_frames=warm)
        n_streams = int(rng.integers(2, 7))
        groups = int(rng.choice([1, 2, 4]))
        n_frames = warm + int(rng.integers(6, 12))
        seeds = [int(rng.integers(0, 10 ** 6)) for _ in range(n_streams)]
        inval = [float(rng.choice([0.0, 0.1, 0.3])) for _ in range(n_streams)]
        # disturbances: (frame, stream, kind) -- a blank frame (no features) or a frame of another place (tracking lost)
        dist = {(int(rng.integers(warm + 1, n_frames)), int(rng.integers(0, n_streams))): int(rng.integers(0, 2)) for _ in range(int(rng.integers(0, 4)))}
        path = int(rng.integers(0, 3))  # 0: host buffers (rvo_stream_step), 1: device buffers, 2: device buffers with the depth images read in place
        tag = f"cfg {c}: {W}x{H} {('host', 'dev', 'dev+depth in place')[path]} L{levels} grid {over['grid_width']}x{over['grid_height']}x{over['max_keypoints_per_grid']} warm {warm} streams {n_streams} groups {groups} frames {n_frames} disturbed {sorted(dist)}"
        if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != c:
            continue
        intr = synth.default_intrinsics(W, H)
        os.environ["RVO_STREAM_GROUP_MIN"] = "2"
        os.environ["RVO_STREAM_GROUPS"] = str(groups)
        try:
            vo = RGBDVOStreams(intr, n_streams, params=default_params(**over))
            if path == 2:
                vo.ctx.streams_depth_in_place(True)
        except Exception as e:
            print(tag, "-> refused:", str(e)[:80], flush=True)
            continue
        ovos = []
        try:
            orcs = [Oracle(intr, odef(**over)) for _ in range(n_streams)]
            ovos = [OracleVO(o) for o in orcs]
            seqs = [list(synth.make_sequence(n_frames, seed=seeds[s], intr=intr, scene=scenes[s % 3], invalid_frac=inval[s])) for s in range(n_streams)]
            far = list(synth.make_sequence(2, seed=999, intr=intr, scene=synth.PlaneScene(seed=7)))[1]
            rcs = []
            for f in range(n_frames):
                gs, ds = [], []
                for s in range(n_streams):
                    g, d = seqs[s][f][0], seqs[s][f][1]
                    if (f, s) in dist:
                        g, d = (np.full_like(g, 90), d) if dist[(f, s)] == 0 else (far[0], far[1])
                    gs.append(g); ds.append(d)
                gray, depth = np.stack(gs), np.stack(ds)
                assert gray.shape == (n_streams, H, W) and depth.shape == (n_streams, H, W)
                if path == 0:
                    res = vo.ProcessFrame(gray, depth)
                else:  # device-resident images; with depth in place every step's buffers stay alive until the results are read
                    dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()
                    torch.cuda.synchronize()
                    vo.ctx.stream_step_dev(dg.data_ptr(), dd.data_ptr(), n_streams)
                    res = vo.ctx.stream_results(n_streams)
                for s in range(n_streams):
                    ro, rg = ovos[s].process(gray[s], depth[s]), res[s]
                    where = (f, s)
                    assert rg["rc"] == ro["rc"], (where, "rc", rg["rc"], ro["rc"])
                    assert rg["fast_thr"] == ro["fast_thr"], (where, "fast_thr", rg["fast_thr"], ro["fast_thr"])
                    assert rg["keyframe_changed"] == ro["keyframe_changed"], (where, "keyframe")
                    if ro["rc"] == 1 and f >= warm:  # an estimate ran and succeeded
                        assert (rg["num_matches"], rg["num_inliers"], rg["status"]) == (ro["num_matches"], ro["num_inliers"], ro["status"]), \
                            (where, "counts", rg["num_matches"], rg["num_inliers"], rg["status"], ro["num_matches"], ro["num_inliers"], ro["status"])
                    assert np.abs(rg["pose"] - ro["pose"]).max() < 1e-8, (where, "pose", float(np.abs(rg["pose"] - ro["pose"]).max()))
                    assert np.abs(rg["curr_to_ref"] - ro["curr_to_ref"]).max() < 1e-8, (where, "curr_to_ref")
                    if ro["rc"] == 1 and f >= warm:
                        n = ro["num_matches"]
                        assert np.array_equal(vo.ctx.match_tracks(s, n), orcs[s].match_tracks(n)), (where, "track ids")
                        est += 1
                    steps_done += 1
                    rcs.append(ro["rc"])
            print(tag, "-> ok, return codes", sorted(set(rcs)), flush=True)
        except AssertionError:
            bad += 1
            print(tag, "-> MISMATCH", flush=True)
            traceback.print_exc()
        finally:
            for v in ovos:
                v.close()
            vo.ctx.close()
    print(f"{steps_done} stream steps compared ({est} estimates), {bad} configurations with a mismatch, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
