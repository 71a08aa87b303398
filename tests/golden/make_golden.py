"""Generates tests/golden/pair_160x120.npz: inputs + per-stage expected outputs of the CPU oracle
(oracle/rvo_oracle.cpp) on one small synthetic pair.  The reference itself cannot be built here (no Eigen), so these
vectors pin the oracle against regressions and let the GPU box check the HIP path against committed data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, default_params  # noqa: E402
from slam_amd import synth  # noqa: E402

W, H, LEVELS, THR = 160, 120, 2, 15


def main():
    intr = synth.default_intrinsics(W, H)
    intr.update(k1=0.05, p1=0.0005)
    scene = synth.PlaneScene()
    g0, d0, g1, d1, T = synth.make_pair(77, intr, scene, invalid_frac=0.05)
    prm = dict(num_pyramid_levels=LEVELS, grid_width=40, grid_height=40, max_keypoints_per_grid=12)
    orc = Oracle(intr, default_params(frame_matcher=True, **prm))
    orc.prepare(0, g0, d0, THR)
    orc.prepare(1, g1, d1, THR)
    out = dict(gray0=g0, depth0=d0, gray1=g1, depth1=d1, T_ref_tgt=T[:3],
               intr=np.array([intr[k] for k in ("width", "height", "depth_width", "depth_height", "fx", "fy", "cx", "cy",
                                                "k1", "k2", "k3", "p1", "p2")], np.float64),
               params=np.array([LEVELS, 40, 40, 12, THR], np.int32))
    for s in (0, 1):
        for l in range(LEVELS):
            out[f"img_{s}_{l}"] = orc.level_image(s, l)
            i = orc.level_info(s, l)
            out[f"info_{s}_{l}"] = np.array([i["num_keypoints"], i["num_detected"], i["num_after_grid"]], np.int32)
            out[f"grid_{s}_{l}"] = orc.grid_keypoints(s, l)
            u, r, x = orc.keypoints(s, l)
            out[f"kp_{s}_{l}"], out[f"rect_{s}_{l}"], out[f"xyz_{s}_{l}"] = u, r, x
            out[f"desc_{s}_{l}"] = orc.descriptors(s, l)[:, :80]
    out["q0"] = orc.initial_rotation(0, 1)
    r = orc.match_pair(0, 1)
    ints, dbl = orc.matches(r["num_matches"])
    out["m_ints"], out["m_dbl"] = ints, dbl
    out["pose"], out["cov"] = r["pose"], r["cov"]
    out["counts"] = np.array([r["status"], r["num_matches"], r["num_inliers"], r["num_reprojection_failures"]], np.int32)
    out["mean_reproj"] = np.array([r["mean_reprojection_error"]])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pair_160x120.npz"), **out)
    print("wrote pair_160x120.npz:", r["status"], r["num_matches"], r["num_inliers"])


if __name__ == "__main__":
    main()
