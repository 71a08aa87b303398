"""CPU suite (no GPU): the oracle against golden vectors and independent numpy restatements, host logic,
the libstdc++ replay, and the C-ABI surface."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from oracle.pyoracle import Oracle, OracleVO, default_params
from slam_amd import synth
from golden_util import GOLD, check_against_fixture, load_pair_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def test_fast_tree_report_is_clean():
    """oracle/check_fast_tree.py evaluated both reference decision trees on all 3^16 ring states."""
    rep = json.load(open(os.path.join(GOLD, "fast_tree_check.json")))
    assert rep["mismatches"] == 0 and rep["states_checked_per_tree"] == 3 ** 16


def test_fast_golden_patches():
    """Golden vectors produced by evaluating the reference's own trees (feature.cpp:67-2995, 3033-5953)."""
    z = np.load(os.path.join(GOLD, "fast_patches.npz"))
    L = pyoracle.lib()
    import ctypes as C
    patches, thr, is_corner, score = z["patches"], z["thr"], z["is_corner"], z["score"]
    sc = C.c_int()
    for i in range(0, len(thr), 3):
        p = np.ascontiguousarray(patches[i])
        c = L.orc_fast_pixel(p.ctypes.data_as(C.c_void_p), 7, 3, 3, int(thr[i]), C.byref(sc))
        assert bool(c) == bool(is_corner[i]) and sc.value == score[i], i


def _dense_fast_nms(img, thr):
    """Independent numpy restatement: segment test, score = A-1, dense non-strict 8-neighbour NMS."""
    h, w = img.shape
    I = img.astype(np.int32)
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - c for dx, dy in RING])
    d2 = np.concatenate([d, d[:8]])
    lo = np.stack([d2[i:i + 9].min(0) for i in range(16)]).max(0)
    hi = np.stack([d2[i:i + 9].max(0) for i in range(16)]).min(0)
    A = np.maximum(lo, -hi)
    score = np.zeros((h, w), np.int32)
    score[3:h - 3, 3:w - 3] = np.where(A > thr, A, 0)  # score+1
    pad = np.pad(score, 1)
    nb = np.stack([pad[1 + dy:h + 1 + dy, 1 + dx:w + 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx or dy)])
    keep = (score > 0) & (nb < score[None]).all(0)
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys, score[ys, xs] - 1], 1).astype(np.float32)


@pytest.mark.parametrize("thr", [5, 20, 60])
def test_fast_nms_equals_dense_rule(thr):
    import ctypes as C
    rng = np.random.default_rng(thr)
    img = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    img[20:60, 30:90] = (img[20:60, 30:90] // 8) + 100  # low-contrast block: ties in scores
    stride = 144
    buf = np.zeros((97, stride), np.uint8)
    buf[:, :131] = img
    out = np.zeros((20000, 3), np.float32)
    n = pyoracle.lib().orc_fast(buf.ctypes.data_as(C.c_void_p), 131, 97, stride, thr, out.ctypes.data_as(C.c_void_p), 20000)
    assert np.array_equal(out[:n], _dense_fast_nms(img, thr))


def _np_gauss(line_src, n):
    s = line_src.astype(np.int32)
    out = np.zeros(s.shape[:-1] + (n // 2,), np.int32)
    out[..., 0] = (s[..., 0] + 4 * s[..., 1] + 3 * s[..., 2]) >> 3
    last = n // 2 - 1
    for r in range(1, last):
        out[..., r] = (s[..., 2 * r - 2] + 4 * s[..., 2 * r - 1] + 6 * s[..., 2 * r] + 4 * s[..., 2 * r + 1] + s[..., 2 * r + 2]) >> 4
    o = 2 * (last - 1)
    if n & 1:
        out[..., last] = (3 * s[..., o] + 4 * s[..., o + 1] + s[..., o + 2]) >> 3
    else:
        out[..., last] = (s[..., o] + 4 * s[..., o + 1] + 7 * s[..., o + 2] + 4 * s[..., o + 3]) >> 4
    return out


@pytest.mark.parametrize("w,h", [(64, 48), (50, 37), (33, 64)])
def test_pyramid_against_numpy(w, h):
    import ctypes as C
    rng = np.random.default_rng(w * h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    dst = np.zeros((h // 2, w // 2), np.uint8)
    pyoracle.lib().orc_pyr_down(img.ctypes.data_as(C.c_void_p), w, w, h, dst.ctypes.data_as(C.c_void_p), w // 2)
    mid = _np_gauss(img, w)                       # [h, w/2]
    ref = _np_gauss(mid.T.copy(), h).T            # [h/2, w/2]
    assert np.array_equal(dst, ref.astype(np.uint8))


def test_oracle_matches_committed_fixture():
    z, intr, prm, thr = load_pair_fixture()
    orc = Oracle(intr, default_params(frame_matcher=True, **prm))
    orc.prepare(0, z["gray0"], z["depth0"], thr)
    orc.prepare(1, z["gray1"], z["depth1"], thr)
    check_against_fixture(orc, z, prm["num_pyramid_levels"])
    r = orc.match_pair(0, 1)
    assert [r["status"], r["num_matches"], r["num_inliers"], r["num_reprojection_failures"]] == list(z["counts"])
    ints, dbl = orc.matches(r["num_matches"])
    assert np.array_equal(ints, z["m_ints"]) and np.array_equal(dbl, z["m_dbl"], equal_nan=True)
    assert np.array_equal(r["pose"], z["pose"])


def test_descriptor_semantics():
    """80-byte patch = 9x9 raster minus the last pixel, mean-normalised twice with saturation."""
    intr = synth.default_intrinsics(160, 120)
    orc = Oracle(intr, default_params(num_pyramid_levels=1))
    g, d = synth.PlaneScene().render(intr, synth.make_pose([0, 0, 0], [0, 0, 0]))
    orc.prepare(0, g, d, 15)
    uvs, _, _ = orc.keypoints(0, 0)
    desc = orc.descriptors(0, 0)
    assert len(uvs) > 20
    for (u, v, _), dsc in list(zip(uvs, desc))[:40]:
        p = g[int(v) - 4:int(v) + 5, int(u) - 4:int(u) + 5].reshape(-1)[:80].astype(np.int32)
        for _ in range(2):
            m = int(p.sum()) // 80
            p = np.clip(p + (128 - m), 0, 255) if m != 128 else p
        assert np.array_equal(dsc[:80], p.astype(np.uint8))


def test_pair_pose_accuracy_and_symmetry(scene, intr640):
    """Size-independent properties: estimate close to ground truth; swapping ref/target gives the inverse."""
    orc = Oracle(intr640, default_params(frame_matcher=True))
    g0, d0, g1, d1, T = synth.make_pair(4, intr640, scene)
    orc.prepare(0, g0, d0, 20)
    orc.prepare(1, g1, d1, 20)
    a = orc.match_pair(0, 1)
    ints, _ = orc.matches(a["num_matches"])
    b = orc.match_pair(1, 0)
    assert a["status"] == 1 and b["status"] == 1
    assert np.abs(a["pose"] - T[:3]).max() < 2e-3
    Ta, Tb = np.eye(4), np.eye(4)
    Ta[:3], Tb[:3] = a["pose"], b["pose"]
    assert np.abs(Ta @ Tb - np.eye(4)).max() < 2e-3
    assert sorted(ints[:, 5]) == list(range(a["num_matches"]))          # ids are a permutation
    assert (np.diff(ints[:, 6]) <= 0).all()                              # sorted by degree, descending
    assert ints[:, 4].sum() == a["num_inliers"]


def test_vo_state_machine_short_sequence(scene, intr640):
    orc = Oracle(intr640, default_params(warmup_frames=2))
    vo = OracleVO(orc)
    seq = list(synth.make_sequence(8, seed=3, intr=intr640, scene=scene))
    rcs = [vo.process(g, d) for g, d, _ in seq]
    assert [r["rc"] for r in rcs] == [0, 1, 1, 1, 1, 1, 1, 1]
    assert rcs[1]["keyframe_changed"]
    T_gt = np.linalg.inv(seq[1][2]) @ seq[-1][2]
    assert np.abs(rcs[-1]["pose"] - T_gt[:3]).max() < 5e-3
    vo.close()


def test_stdalgo_replays_libstdcxx(tmp_path):
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "test_stdalgo.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert "stdalgo OK" in out


def test_c_abi_exports_every_declared_symbol():
    from slam_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    hdr = open(os.path.join(ROOT, "include", "rvo.h")).read()
    declared = sorted(set(re.findall(r"\b(rvo_[a-z_0-9]+)\s*\(", hdr)))
    assert sorted(_lib.SYMBOLS) == declared
    import ctypes as C
    L = C.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s


def test_defaults_match_reference_settings():
    """settings.hpp:11-52 values (no GPU needed: rvo_default_*_params is host code)."""
    from slam_amd.api import default_params as dp, params_from_options
    p, f = dp(), dp(frame_matcher=True)
    assert (p.feature_window_size, p.num_pyramid_levels, p.fast_init_threshold, p.fast_threshold_min, p.fast_threshold_max) == (9, 3, 20, 5, 70)
    assert (p.grid_width, p.grid_height, p.max_keypoints_per_grid, p.target_pixels_per_feature) == (80, 80, 25, 250)
    assert (p.inlier_max_reprojection_error, p.clique_inlier_threshold, p.min_features_for_estimate) == (1.5, 0.1, 10)
    assert (p.max_mean_reprojection_error, p.feature_search_window, p.fast_adaptive_gain) == (10.0, 25.0, 0.005)
    assert p.reference_frame_change_threshold == 250 and f.reference_frame_change_threshold == 150
    q = params_from_options({"num-pyramid-levels": "4", "fast-adaptive-gain": "0.01", "min-feature-for-estimate": "12"})
    assert q.num_pyramid_levels == 4 and q.fast_adaptive_gain == 0.01 and q.min_features_for_estimate == 12
    with pytest.raises(KeyError):
        params_from_options({"no-such-option": "1"})


def test_front_of_path_conversions_known_answers():
    """ConvertBGR2Gray / DepthToPointCloud restatements (helper.cpp:30-64) against values worked out by hand in
    Python floats (IEEE double, same operation order)."""
    from oracle import pyoracle
    px = np.array([[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77]], np.uint8)
    want = [int((0.114 * float(b) + 0.587 * float(g)) + 0.299 * float(r)) for b, g, r in px]
    assert pyoracle.bgr_to_gray(px).tolist() == want
    assert want[1] == 0 and want[2] == 29 and want[3] == 149 and want[4] == 76
    d = np.array([[0, 1000], [2000, 65535]], np.uint16)
    c = pyoracle.depth_to_cloud(d, 1.2, 0.9)
    z = np.float32(1000 * 0.001)
    assert c[0, 1, 2] == z and c[0, 1, 0] == np.float32((1 / 2 - 0.5) * float(z) * 1.2)
    assert c[0, 1, 1] == np.float32((0.5 - 0 / 2) * float(z) * 0.9)
    assert c[0, 0, 2] == 0 and np.signbit(c[0, 0, 0])  # (0/W - 0.5) * 0 = -0.0, kept


def test_rvo_create_rejects_bad_arguments_before_touching_the_gpu():
    """Nothing may throw, divide by zero or exit across the C ABI (include/rvo.h): bad options come back as error codes."""
    import ctypes as C
    from slam_amd import _lib, synth
    from slam_amd.api import default_params
    L = _lib.load()

    def create(intr_over=None, **prm_over):
        intr = synth.default_intrinsics(640, 480)
        intr.update(intr_over or {})
        i = _lib.Intrinsics(**{k: intr[k] for k, _ in _lib.Intrinsics._fields_})
        p = default_params(**prm_over)
        h = C.c_void_p()
        rc = L.rvo_create(C.byref(i), C.byref(p), 0, 3, 1, C.byref(h))
        assert h.value is None
        return rc, L.rvo_last_error().decode()

    assert create(target_pixels_per_feature=0)[0] == -1          # vo.cpp:42-43 divides by it
    assert create(dict(depth_width=0))[0] == -1
    assert create(dict(depth_height=-3))[0] == -1
    assert create(dict(fx=0.0))[0] == -1
    assert create(dict(fy=float("nan")))[0] == -1
    assert create(fast_threshold_min=80)[0] == -1               # min > max
    assert create(fast_init_threshold=300)[0] == -1
    assert create(feature_window_size=11)[0] == -2              # RVO_E_UNSUPPORTED
    assert create(num_pyramid_levels=9)[0] == -2
    assert create(dict(width=4000))[0] == -2
    rc, msg = create(grid_width=2)
    assert rc == -2 and msg
