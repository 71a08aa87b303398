"""EstimateInitialRotation AS THE REFERENCE WRITES IT, in numpy float32 -- VERDICT round 2, Next #5c.

The oracle (and the HIP kernel) solve the 3-parameter ESM step with fp64 normal equations and a closed-form exponential;
the reference uses a float JacobiSVD pseudo-inverse (motion_estimate.cpp:1454-1470) and Matrix3f::exp() (:1528-1536).
Eigen is not available here, so the reference's solver cannot be run; what CAN be measured is the distance between the
oracle's solver and ANOTHER float32 SVD pseudo-inverse + matrix exponential (LAPACK's sgesdd through numpy, scipy's Pade
expm) following the reference text line by line (column-major flattening, float accumulations, float error norm,
the stop rules of :1416-1487).  Two backward-stable float32 SVDs differ from each other by the same order as either differs
from the exact solution, so the measured |q0_oracle - q0_svd| is the size of the exposure DESIGN.md section 3.1 could only
estimate before.  Asserted: below the 2e-5 rad the decided sets are known to survive, on every pair input of the GPU
parity tests, and no iteration-count flip."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
from scipy.linalg import expm

from oracle.pyoracle import Oracle, default_params
from slam_amd import synth

f32 = np.float32


def gray_to_eigen(img, ds):
    """grayToEigen, motion_estimate.cpp:1338-1367 (float accumulation; the sums are exact)"""
    h, w = img.shape
    if ds == 0:
        return img.astype(f32)
    rows, cols = h >> ds, w >> ds
    out = np.zeros((rows, cols), f32)
    for y in range(h):
        ey = y >> ds
        if ey >= rows:
            continue
        for k in range(1 << ds):
            xs = np.arange(k, (cols << ds), 1 << ds)
            out[ey] += img[y, xs].astype(f32)
    pf = f32(float(1 << ds) * float(1 << ds))
    return (out / pf).astype(f32)


def gradient(im):
    """computeGradient, :1494-1517"""
    dx = np.zeros_like(im)
    dy = np.zeros_like(im)
    dx[:, 1:-1] = im[:, 2:] - im[:, :-2]
    dy[1:-1, :] = im[2:, :] - im[:-2, :]
    dx = (dx / f32(2.0)).astype(f32)
    dy = (dy / f32(2.0)).astype(f32)
    dx[:, 0] = im[:, 1] - im[:, 0]
    dx[:, -1] = im[:, -1] - im[:, -2]
    dy[0, :] = im[1, :] - im[0, :]
    dy[-1, :] = im[-1, :] - im[-2, :]
    return dx, dy


def warp(src, H, rows, cols):
    """constructWarpedImage, :1538-1571; points in column-major order (templatePoints_, :1396-1404)"""
    x = np.repeat(np.arange(cols, dtype=f32), rows)   # flattenMatrix of the column-major x grid
    y = np.tile(np.arange(rows, dtype=f32), cols)
    hx = (H[0, 0] * x + H[0, 1] * y) + H[0, 2] * f32(1)
    hy = (H[1, 0] * x + H[1, 1] * y) + H[1, 2] * f32(1)
    hw = (H[2, 0] * x + H[2, 1] * y) + H[2, 2] * f32(1)
    px, py = (hx / hw).astype(f32), (hy / hw).astype(f32)
    fx0, fy0 = np.floor(px).astype(np.int64), np.floor(py).astype(np.int64)
    cx0, cy0 = np.ceil(px).astype(np.int64), np.ceil(py).astype(np.int64)
    val = np.full(len(px), 128.0)
    inside = (0 <= px) & (px < cols - 1) & (0 <= py) & (py < rows - 1)
    fok = (0 <= fx0) & (fx0 < cols) & (0 <= fy0) & (fy0 < rows)
    cok = (0 <= cx0) & (cx0 < cols) & (0 <= cy0) & (cy0 < rows)
    s = src.astype(np.float64)
    m = ~inside & ~fok & cok
    val[m] = s[cy0[m], cx0[m]]
    m = ~inside & fok
    val[m] = s[fy0[m], fx0[m]]
    m = inside
    x1 = px[m].astype(np.float64) - fx0[m]
    y1 = py[m].astype(np.float64) - fy0[m]
    x2, y2 = 1 - x1, 1 - y1
    val[m] = x2 * y2 * s[fy0[m], fx0[m]] + x1 * y2 * s[fy0[m], fx0[m] + 1] + x2 * y1 * s[fy0[m] + 1, fx0[m]] + x1 * y1 * s[fy0[m] + 1, fx0[m] + 1]
    return val.astype(f32).reshape(cols, rows).T  # warped(i % rows, i / rows)


def track_as_written(tmpl, test, n_iters=8):
    """InitialHomographyEstimator::Track, :1416-1492, float32 throughout; returns (bestH, iterations run, best iteration)"""
    rows, cols = tmpl.shape
    tdx, tdy = gradient(tmpl)
    flat = lambda m: np.ascontiguousarray(m.T).reshape(-1)  # flattenMatrix: column-major
    xx = np.repeat(np.arange(cols, dtype=f32), rows) + f32(1)
    yy = np.tile(np.arange(rows, dtype=f32), cols) + f32(1)
    H = np.eye(3, dtype=f32)
    bestH = H.copy()
    min_error, last_rms = np.inf, np.inf
    best_iter = last_improved = 0
    ran = 0
    for it in range(n_iters):
        ran = it + 1
        warped = warp(test, H, rows, cols)
        err = flat((warped - tmpl).astype(f32))
        rms = float(np.sqrt(np.dot(err, err))) / np.sqrt(float(rows * cols))  # float norm / sqrt(double N), :1411-1414
        if rms < min_error:
            min_error, bestH, best_iter = rms, H.copy(), it
        wdx, wdy = gradient(warped)
        dx, dy = flat(tdx) + flat(wdx), flat(tdy) + flat(wdy)
        Jt = np.stack([dx, dy, dx * yy - dy * xx], 1).astype(f32)
        U, sig, Vt = np.linalg.svd(Jt, full_matrices=False)  # float32 in, float32 out (sgesdd)
        assert U.dtype == f32
        r = 0
        inv = sig.copy()
        while r < len(sig) and not sig[r] < 1e-7:
            inv[r] = f32(1.0) / sig[r]
            r += 1
        if r == 0:
            Jt_plus = np.zeros((3, len(err)), f32)
        else:
            Jt_plus = ((Vt[:r].T * inv[:r]) @ U[:, :r].T).astype(f32)
        lie = (f32(-2) * (Jt_plus @ err)).astype(f32)
        M = np.array([[0, lie[2], lie[0]], [-lie[2], 0, lie[1]], [0, 0, 0]], f32)
        H = (H @ expm(M).astype(f32)).astype(f32)
        if rms < last_rms:
            last_improved = it
        lnorm = float(np.sqrt(np.dot(lie, lie)))
        if lnorm < 1e-6 or (rms - min_error > 3 and it - best_iter > 2) or it - best_iter > 4 or it - last_improved > 2:
            break
        last_rms = rms
    return bestH, ran, best_iter


def q0_from_H(H, fx, irl=4):
    """motion_estimate.cpp:288-303 (S H S^-1 in float, angles in double, RPYToQuat helper.hpp:59-81)"""
    sf = f32(1 << irl)
    S = np.diag([sf, sf, f32(1)]).astype(f32)
    Hs = ((S @ H).astype(f32) @ np.linalg.inv(S).astype(f32)).astype(f32)
    roll, pitch, yaw = np.arcsin(float(Hs[1, 2]) / fx), -np.arcsin(float(Hs[0, 2]) / fx), -np.arctan2(float(Hs[1, 0]), float(Hs[0, 0]))
    r, p, y = roll / 2, pitch / 2, yaw / 2
    return np.array([np.cos(r) * np.cos(p) * np.cos(y) + np.sin(r) * np.sin(p) * np.sin(y),
                     np.sin(r) * np.cos(p) * np.cos(y) - np.cos(r) * np.sin(p) * np.sin(y),
                     np.cos(r) * np.sin(p) * np.cos(y) + np.sin(r) * np.cos(p) * np.sin(y),
                     np.cos(r) * np.cos(p) * np.sin(y) - np.sin(r) * np.sin(p) * np.cos(y)])


def quat_angle(a, b):
    d = abs(float(np.dot(a, b)))
    return 2 * np.arcsin(min(1.0, np.linalg.norm(a - np.sign(np.dot(a, b)) * b) / 2)) if d > 0.5 else 2 * np.arccos(d)


def measure(case_list=None):
    """[(case name, pairs, max angle between the two q0 [rad], iteration flips, max |q0| rotation [rad])]"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stability_margins
    rows = []
    for name, intr, prm, pairs, thr in (case_list or stability_margins.cases()):
        orc = Oracle(intr, prm)
        irl = 4
        pyr = min(prm.num_pyramid_levels - 1, irl)
        worst, flips, turn = 0.0, 0, 0.0
        for g0, d0, g1, d1 in pairs:
            orc.prepare(0, g0, d0, thr)
            orc.prepare(1, g1, d1, thr)
            q_orc = orc.initial_rotation(0, 1)
            info = np.zeros(2, np.int32)
            orc.L.orc_initial_rotation_info(orc.h, info.ctypes.data_as(C.c_void_p))
            tmpl = gray_to_eigen(orc.level_image(0, pyr), irl - pyr)
            test = gray_to_eigen(orc.level_image(1, pyr), irl - pyr)
            H, ran, best = track_as_written(tmpl, test)
            q_svd = q0_from_H(H, intr["fx"])
            worst = max(worst, quat_angle(q_orc, q_svd))
            turn = max(turn, quat_angle(q_orc, np.array([1.0, 0, 0, 0])))
            flips += int((ran, best) != (int(info[0]), int(info[1])))
        orc.close()
        rows.append((name, len(pairs), worst, flips, turn))
    return rows


def test_oracle_solver_against_the_reference_text_with_a_float32_svd():
    rows = measure()
    for name, n, worst, flips, turn in rows:
        print(f"{name}: {n} pairs, max |q0_oracle - q0_svd| = {worst:.3g} rad, iteration flips {flips}, |q0| up to {turn:.3g} rad")
        assert worst < 2e-5, (name, worst)   # the decided sets survive 2e-5 rad on these inputs (tests/test_cpu_margins.py)
        assert flips == 0, (name, flips)
        assert turn > 1e-4                    # the estimator does turn: the comparison is not between two identities


def test_restatement_pieces_on_known_answers():
    """The numpy pieces above against hand-checkable cases: the identity homography reproduces the image; a pure shift moves
    it; exp of the generator is a rotation + shift; gradients of a ramp."""
    im = np.arange(12, dtype=f32).reshape(3, 4) * f32(2)
    assert np.array_equal(warp(im, np.eye(3, dtype=f32), 3, 4), im)
    Hs = np.eye(3, dtype=f32)
    Hs[0, 2] = 1  # sample one column to the right
    w = warp(im, Hs, 3, 4)
    assert np.array_equal(w[:, :3], im[:, 1:]) and np.all(w[:, 3] == 128)  # x = 4 is outside: the default value, :1566
    dx, dy = gradient(im)
    assert np.all(dx == 2) and np.all(dy == 8)
    a = 0.3
    E = expm(np.array([[0, a, 0.5], [-a, 0, -0.2], [0, 0, 0]]))
    assert np.allclose(E[:2, :2], [[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
    g = gray_to_eigen(np.arange(64, dtype=np.uint8).reshape(8, 8), 1)
    assert g.shape == (4, 4) and g[0, 0] == (0 + 1 + 8 + 9) / 4
