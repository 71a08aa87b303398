"""Randomised parity sweep ON THE GPU BOX (not part of the test suite): random image sizes, grids, pyramid depths, thresholds,
motions, invalid-depth fractions and contrasts through the fused device path, every pair compared with the CPU oracle by
tests/test_gpu_pair.py::compare_pair (bit-exact lists, refined positions, consistency vectors; pose to 1e-9).
    python tools/fuzz_parity.py [n_configs] [seed]        prints one line per configuration and a summary; exit 1 on a mismatch"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from oracle.pyoracle import Oracle, default_params as odef  # noqa: E402
from slam_amd import synth  # noqa: E402
from slam_amd.api import Context, default_params  # noqa: E402
from test_gpu_frame import _compare_frame  # noqa: E402
from test_gpu_pair import compare_pair  # noqa: E402


def run(n_cfg: int, seed: int):
    """-> (configurations with a mismatch, pairs compared, pairs with a tolerance-class difference only)"""
    rng = np.random.default_rng(seed)
    scenes = [synth.PlaneScene(seed=s) for s in range(3)]
    bad, pairs_done, succ, soft = 0, 0, 0, 0
    t0 = time.time()
    for c in range(n_cfg):
        W = int(rng.integers(160, 660))  # odd sizes too: level strides, the pyramid's odd-length edge formulas
        H = int(rng.integers(120, 500))
        levels = int(rng.integers(1, 5))
        while levels > 1 and min(W, H) >> (levels - 1) < 24:
            levels -= 1
        over = dict(grid_width=int(rng.integers(24, 121)), grid_height=int(rng.integers(24, 121)),
                    max_keypoints_per_grid=int(rng.integers(2, 41)), num_pyramid_levels=levels,
                    feature_search_window=float(rng.choice([10.0, 25.0, 40.0])),
                    clique_inlier_threshold=float(rng.choice([0.05, 0.1, 0.2])),
                    inlier_max_reprojection_error=float(rng.choice([0.8, 1.5, 3.0])),
                    min_features_for_estimate=int(rng.choice([5, 10, 20])))
        thr = int(rng.integers(5, 41))
        if os.environ.get("FUZZ_BIG"):  # many keypoints: more than 1024 / 2048 matches per pair (the clique kernels' other row regimes)
            W, H, thr = int(rng.integers(500, 1000)), int(rng.integers(400, 760)), int(rng.integers(5, 16))
            over.update(grid_width=int(rng.integers(24, 61)), grid_height=int(rng.integers(24, 61)), max_keypoints_per_grid=int(rng.integers(10, 41)))
        if os.environ.get("FUZZ_EXTREME"):  # the ends of the option ranges: tiny and huge cells, 1 or 100 keypoints per cell, thresholds
            lv = int(rng.integers(1, 7))      # near 0 and far above the texture's contrast, up to 6 levels, tiny and huge search windows
            while lv > 1 and min(W, H) >> (lv - 1) < 24:
                lv -= 1
            levels = lv
            gsel = lambda: int(rng.integers(8, 24)) if rng.random() < 0.5 else int(rng.integers(121, 254))
            over.update(grid_width=gsel(), grid_height=gsel(), num_pyramid_levels=lv,
                        max_keypoints_per_grid=int(rng.choice([1, 2, 3, 50, 80, 120])),
                        feature_search_window=float(rng.choice([1.0, 5.0, 100.0, 400.0])),
                        min_features_for_estimate=int(rng.choice([6, 8, 50])))  # (with 1-3 inliers the 6-parameter normal equations are
            # singular or nearly so: both sides then amplify their rounding without bound, poses differ by centimetres, nothing to compare)
            thr = int(rng.choice([1, 2, 3, 4, 45, 60, 90, 120]))
        intr = synth.default_intrinsics(W, H)
        kind = rng.random(4)
        if kind[0] < 0.3:  # a distorted input camera: the undistortion LUT, keypoints that leave the rectified image
            intr.update(k1=float(rng.uniform(-0.12, 0.12)), k2=float(rng.uniform(-0.02, 0.02)), p1=float(rng.uniform(-2e-3, 2e-3)),
                        p2=float(rng.uniform(-2e-3, 2e-3)))
        half_depth = kind[1] < 0.2  # depth image at half the resolution (Depth::x_scale_ / y_scale_, depth.cpp:14-15)
        if half_depth:
            intr.update(depth_width=W // 2, depth_height=H // 2)
        host_path = kind[2] < 0.35    # rvo_prepare_frames + rvo_match_pairs on host buffers instead of the fused device call
        with_init = kind[3] < 0.3     # EstimateMotion with a given initial estimate (the VO call form) instead of the initial rotation
        n = 3
        tag = f"cfg {c}: {W}x{H}{' distorted' if intr['k1'] else ''}{' depth/2' if half_depth else ''}{' host' if host_path else ''}{' init' if with_init else ''} L{levels} grid {over['grid_width']}x{over['grid_height']}x{over['max_keypoints_per_grid']} thr {thr}"
        # every random draw of the configuration is made here, so that FUZZ_ONLY=<index> replays one configuration alone
        draws = [(int(rng.integers(0, 10 ** 6)), int(rng.integers(0, 3)), float(rng.choice([0.0, 0.1, 0.3])),
                  float(rng.choice([0.01, 0.03, 0.08])), float(rng.choice([0.3, 1.0, 3.0]))) for _ in range(n)]
        low_contrast = rng.random() < 0.3
        noise = rng.random() < 0.12   # images of (block) noise instead of the rendered scene: a corner candidate on almost every pixel
        noise_seed, noise_block, noise_shift = int(rng.integers(0, 10 ** 6)), int(rng.choice([1, 1, 2, 3])), (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
        if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != c:
            continue
        try:
            ctx = Context(intr, default_params(frame_matcher=True, **over), max_frames=2 * n, max_pairs=n)
        except Exception as e:  # a configuration this build refuses (LDS, slots): not a parity question
            print(tag, "-> refused:", str(e)[:80], flush=True)
            continue
        try:
            orc = Oracle(intr, odef(frame_matcher=True, **over))
            pairs = [synth.make_pair(sd, intr, scenes[sc], invalid_frac=inv, max_t=mt, max_rot_deg=mr) for sd, sc, inv, mt, mr in draws]
            gray = np.stack([p[j] for p in pairs for j in (0, 2)])
            depth = np.stack([p[j] for p in pairs for j in (1, 3)])
            if noise:  # FAST's candidate queue overflows into its dense pass, every bucket overflows, SAD scores tie: the second
                nr = np.random.default_rng(noise_seed)  # frame of a pair is the first one shifted by a few pixels
                for i in range(n):
                    base = nr.integers(0, 256, ((H + noise_block - 1) // noise_block, (W + noise_block - 1) // noise_block), dtype=np.uint8)
                    g0 = np.kron(base, np.ones((noise_block, noise_block), np.uint8))[:H, :W]
                    gray[2 * i], gray[2 * i + 1] = g0, np.roll(g0, noise_shift, axis=(0, 1))
                    depth[2 * i + 1] = np.roll(depth[2 * i], (noise_shift[0] * depth.shape[1] // H, noise_shift[1] * depth.shape[2] // W), axis=(0, 1))
            # (synth renders the depth image at the intrinsics' depth resolution)
            assert gray.shape == (2 * n, H, W) and depth.shape == (2 * n, intr["depth_height"], intr["depth_width"])  # the device entry points trust their pointers
            inits = None
            if with_init:  # the true motion, slightly off
                inits = np.stack([(p[4] @ synth.make_pose([0.004, -0.003, 0.002], [0.002, -0.001, 0.0015]))[:3] for p in pairs])
            if low_contrast:  # few, weak corners
                gray = (128 + (gray.astype(np.float32) - 128) * 0.15).astype(np.uint8)
            if host_path:
                ctx.prepare_frames(gray, depth, thr, np.arange(2 * n))
                rb = ctx.match_pairs(np.arange(0, 2 * n, 2), np.arange(1, 2 * n, 2), init_pose=inits)
            else:
                dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()
                torch.cuda.synchronize()
                ctx.match_frames_dev(dg.data_ptr(), dd.data_ptr(), thr, np.arange(2 * n), init_pose=inits)
                rb = ctx.pair_results(n)
            st = []
            for i in range(n):
                orc.prepare(0, gray[2 * i], depth[2 * i], thr)
                orc.prepare(1, gray[2 * i + 1], depth[2 * i + 1], thr)
                _compare_frame(ctx, orc, 2 * i, 0, levels)  # the frames first: a mismatch names its level and stage
                _compare_frame(ctx, orc, 2 * i + 1, 1, levels)
                ro = orc.estimate_motion(0, 1, inits[i]) if with_init else orc.match_pair(0, 1)
                try:
                    compare_pair(ctx, orc, i, rb[i], ro)
                except AssertionError:
                    rg = rb[i]
                    print("  pair", i, "gpu", {k: rg[k] for k in ("status", "num_matches", "num_inliers", "num_reprojection_failures", "mean_reprojection_error")},
                          "oracle", {k: ro[k] for k in ("status", "num_matches", "num_inliers", "num_reprojection_failures", "mean_reprojection_error")},
                          "max |pose diff|", float(np.abs(rg["pose"] - ro["pose"]).max()), flush=True)
                    # the Gauss-Newton values are tolerance-class (tree-ordered sums on the GPU, north_star asks for 1e-5 m /
                    # 1e-6 rad); compare_pair's 1e-9 is tight enough to trip on a badly conditioned estimate (a dozen inliers):
                    # such a pair is reported, and counted as a mismatch only beyond 1e-7 or when an exact quantity differs
                    nm = ro["num_matches"]
                    exact = all(rg[k] == ro[k] for k in ("status", "num_matches", "num_inliers", "num_reprojection_failures")) and \
                        np.array_equal(ctx.matches(i, nm)[0], orc.matches(nm)[0])
                    if not (exact and np.abs(rg["pose"] - ro["pose"]).max() < 1e-7 and
                            abs(rg["mean_reprojection_error"] - ro["mean_reprojection_error"]) < 1e-7):
                        raise
                    print("  (tolerance-class difference only)", flush=True)
                    soft += 1
                st.append((ro["status"], ro["num_matches"], ro["num_inliers"]))
                pairs_done += 1
                succ += ro["status"] == 1
            print(tag + (" NOISE" if noise else ""), "->", st, flush=True)
        except AssertionError:
            bad += 1
            print(tag, "-> MISMATCH", flush=True)
            traceback.print_exc()
        finally:
            ctx.close()
    print(f"{pairs_done} pairs compared ({succ} with status SUCCESS), {bad} configurations with a mismatch, "
          f"{soft} pairs beyond 1e-9 but within 1e-7 in tolerance-class values, {time.time() - t0:.0f} s")
    return bad, pairs_done, soft


def main():
    bad, _, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
