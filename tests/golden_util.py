import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_pair_fixture():
    z = np.load(os.path.join(GOLD, "pair_160x120.npz"))
    keys = ("width", "height", "depth_width", "depth_height", "fx", "fy", "cx", "cy", "k1", "k2", "k3", "p1", "p2")
    intr = {k: (int(v) if i < 4 else float(v)) for i, (k, v) in enumerate(zip(keys, z["intr"]))}
    levels, gw, gh, mp, thr = [int(v) for v in z["params"]]
    prm = dict(num_pyramid_levels=levels, grid_width=gw, grid_height=gh, max_keypoints_per_grid=mp)
    return z, intr, prm, thr


def check_against_fixture(eng, z, levels, slot_of=(0, 1), desc_cols=80):
    """eng: oracle or HIP context exposing level_image/level_info/grid_keypoints/keypoints/descriptors."""
    for s in (0, 1):
        for l in range(levels):
            sl = slot_of[s]
            assert np.array_equal(eng.level_image(sl, l), z[f"img_{s}_{l}"])
            i = eng.level_info(sl, l)
            assert [i["num_keypoints"], i["num_detected"], i["num_after_grid"]] == list(z[f"info_{s}_{l}"])
            assert np.array_equal(eng.grid_keypoints(sl, l), z[f"grid_{s}_{l}"])
            u, r, x = eng.keypoints(sl, l)
            assert np.array_equal(u, z[f"kp_{s}_{l}"]) and np.array_equal(r, z[f"rect_{s}_{l}"])
            assert np.array_equal(x, z[f"xyz_{s}_{l}"])
            assert np.array_equal(eng.descriptors(sl, l)[:, :desc_cols], z[f"desc_{s}_{l}"])
