"""Synthetic RGB-D frames for parity tests and the benchmark.

A textured plane (fixed 2048x2048 u8 texture: multi-octave value noise plus
random rectangles) is viewed by a pinhole camera in the style of
/root/reference/sensor.cpp:147-151 (fx = fy, principal point at the image
centre, no distortion).  Gray is the bilinearly sampled texture, depth is the
exact camera-frame z of the plane in millimetres (u16, 0 = invalid), i.e. the
format RGBDVO::ProcessFrame takes (vo.hpp:20).

Everything is numpy and seeded, so the GPU box regenerates identical inputs.
"""
from __future__ import annotations

import numpy as np

_TEX_CACHE: dict = {}


def make_texture(seed: int = 1, size: int = 2048, contrast: float = 1.0) -> np.ndarray:
    key = (seed, size, contrast)
    if key in _TEX_CACHE:
        return _TEX_CACHE[key]
    rng = np.random.default_rng(seed)
    tex = np.full((size, size), 128.0, dtype=np.float32)
    # multi-octave value noise
    amp = 40.0
    cells = 8
    while cells <= size // 4:
        g = rng.random((cells + 1, cells + 1), dtype=np.float32) - 0.5
        idx = np.linspace(0, cells, size, endpoint=False, dtype=np.float32)
        i0 = idx.astype(np.int32)
        f = idx - i0
        f = f * f * (3 - 2 * f)
        rows = g[i0][:, i0] * (1 - f)[None, :] + g[i0][:, i0 + 1] * f[None, :]
        rows1 = g[i0 + 1][:, i0] * (1 - f)[None, :] + g[i0 + 1][:, i0 + 1] * f[None, :]
        tex += amp * (rows * (1 - f)[:, None] + rows1 * f[:, None])
        amp *= 0.6
        cells *= 4
    # random rectangles (corners for FAST)
    n_rect = 6000
    xs = rng.integers(0, size, n_rect)
    ys = rng.integers(0, size, n_rect)
    ws = rng.integers(6, 90, n_rect)
    hs = rng.integers(6, 90, n_rect)
    vals = rng.integers(0, 256, n_rect)
    for x, y, w, h, v in zip(xs, ys, ws, hs, vals):
        tex[y:y + h, x:x + w] = 0.5 * tex[y:y + h, x:x + w] + 0.5 * v
    tex = 128.0 + (tex - 128.0) * contrast
    tex = np.clip(tex, 0, 255).astype(np.float32)
    _TEX_CACHE[key] = tex
    return tex


def default_intrinsics(width: int = 640, height: int = 480) -> dict:
    f = 525.0 * width / 640.0
    return dict(width=width, height=height, depth_width=width, depth_height=height,
                fx=f, fy=f, cx=width / 2.0, cy=height / 2.0,
                k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0)


def rpy_to_R(roll: float, pitch: float, yaw: float) -> np.ndarray:
    cr, sr = np.cos(roll), np.sin(roll)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cy, sy = np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_pose(t, rpy) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = rpy_to_R(*rpy)
    T[:3, 3] = t
    return T


class PlaneScene:
    """Plane through (0,0,dist) with a tilted normal, textured over `extent` metres."""

    def __init__(self, seed: int = 1, dist: float = 2.0, tilt_deg=(8.0, -5.0), extent: float = 7.0,
                 contrast: float = 1.0):
        self.tex = make_texture(seed, contrast=contrast)
        self.size = self.tex.shape[0]
        R = rpy_to_R(np.deg2rad(tilt_deg[0]), np.deg2rad(tilt_deg[1]), 0.0)
        self.e1, self.e2, self.n = R[:, 0], R[:, 1], R[:, 2]
        self.p0 = np.array([0.0, 0.0, dist])
        self.extent = extent

    def render(self, intr: dict, T_wc: np.ndarray, invalid_frac: float = 0.0, rng=None):
        """Return (gray u8 [H,W], depth u16 [H,W]) for camera-to-world pose T_wc."""
        W, H = intr["width"], intr["height"]
        u = (np.arange(W, dtype=np.float64) - intr["cx"]) / intr["fx"]
        v = (np.arange(H, dtype=np.float64) - intr["cy"]) / intr["fy"]
        R, t = T_wc[:3, :3], T_wc[:3, 3]
        # ray direction in world = R @ (u, v, 1)
        dx = R[0, 0] * u[None, :] + R[0, 1] * v[:, None] + R[0, 2]
        dy = R[1, 0] * u[None, :] + R[1, 1] * v[:, None] + R[1, 2]
        dz = R[2, 0] * u[None, :] + R[2, 1] * v[:, None] + R[2, 2]
        denom = self.n[0] * dx + self.n[1] * dy + self.n[2] * dz
        s = float(self.n @ (self.p0 - t)) / denom  # camera-frame z (ray has z_c = 1)
        X = t[0] + s * dx - self.p0[0]
        Y = t[1] + s * dy - self.p0[1]
        Z = t[2] + s * dz - self.p0[2]
        a = X * self.e1[0] + Y * self.e1[1] + Z * self.e1[2]
        b = X * self.e2[0] + Y * self.e2[1] + Z * self.e2[2]
        tx = (a / self.extent + 0.5) * self.size
        ty = (b / self.extent + 0.5) * self.size
        tx = np.clip(tx, 0, self.size - 1.001)
        ty = np.clip(ty, 0, self.size - 1.001)
        x0 = tx.astype(np.int32)
        y0 = ty.astype(np.int32)
        fx = (tx - x0).astype(np.float32)
        fy = (ty - y0).astype(np.float32)
        T = self.tex
        g = (T[y0, x0] * (1 - fx) * (1 - fy) + T[y0, x0 + 1] * fx * (1 - fy)
             + T[y0 + 1, x0] * (1 - fx) * fy + T[y0 + 1, x0 + 1] * fx * fy)
        gray = np.clip(np.rint(g), 0, 255).astype(np.uint8)
        depth = np.clip(np.rint(s * 1000.0), 0, 65535).astype(np.uint16)
        depth[s <= 0] = 0
        if invalid_frac > 0:
            rng = rng or np.random.default_rng(0)
            depth = depth.copy()
            # blobs
            n_blobs = max(1, int(invalid_frac * 0.8 * W * H / (40 * 40)))
            bx = rng.integers(0, W, n_blobs)
            by = rng.integers(0, H, n_blobs)
            for x, y in zip(bx, by):
                depth[max(0, y - 20):y + 20, max(0, x - 20):x + 20] = 0
            # salt noise
            salt = rng.random((H, W)) < invalid_frac * 0.2
            depth[salt] = 0
        if intr.get("depth_width", W) != W or intr.get("depth_height", H) != H:
            dw, dh = intr["depth_width"], intr["depth_height"]
            ys = (np.arange(dh) * H) // dh
            xs = (np.arange(dw) * W) // dw
            depth = depth[ys][:, xs]
        return np.ascontiguousarray(gray), np.ascontiguousarray(depth)


def random_small_motion(rng, max_t: float = 0.02, max_rot_deg: float = 0.5) -> np.ndarray:
    t = rng.uniform(-max_t, max_t, 3)
    r = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg, 3))
    return make_pose(t, r)


def make_pair(seed: int, intr: dict | None = None, scene: PlaneScene | None = None,
              invalid_frac: float = 0.0, max_t: float = 0.02, max_rot_deg: float = 0.5):
    """One independent frame pair (BASELINE config 2 unit).

    Returns (gray_ref, depth_ref, gray_tgt, depth_tgt, T_ref_tgt) where T_ref_tgt maps
    target-camera points into the reference camera frame -- the quantity
    MotionEstimator::GetMotionEstimate() returns (vo.cpp:125-130).
    """
    intr = intr or default_intrinsics()
    scene = scene or PlaneScene()
    rng = np.random.default_rng(1000 + seed)
    T_w_ref = make_pose(rng.uniform(-0.8, 0.8, 3) * np.array([1, 1, 0.25]),
                        np.deg2rad(rng.uniform(-4, 4, 3)))
    T_w_tgt = T_w_ref @ random_small_motion(rng, max_t, max_rot_deg)
    g0, d0 = scene.render(intr, T_w_ref, invalid_frac, rng)
    g1, d1 = scene.render(intr, T_w_tgt, invalid_frac, rng)
    return g0, d0, g1, d1, np.linalg.inv(T_w_ref) @ T_w_tgt


def sequence_poses(n_frames: int, seed: int = 0) -> tuple:
    """Camera poses T_w_c of make_sequence's trajectory (a pure function of the frame index, so frames can be rendered
    independently, e.g. by a process pool).  Returns (poses, the generator make_sequence goes on to render with)."""
    rng = np.random.default_rng(5000 + seed)
    ph = rng.uniform(0, 2 * np.pi, 6)
    start = rng.uniform(-0.5, 0.5, 3) * np.array([1, 1, 0.2])
    out = []
    for i in range(n_frames):
        s = i / 30.0
        t = start + np.array([0.35 * np.sin(0.9 * s + ph[0]), 0.25 * np.sin(0.7 * s + ph[1]),
                              0.10 * np.sin(0.5 * s + ph[2])])
        r = np.deg2rad(np.array([3.0 * np.sin(0.8 * s + ph[3]), 3.0 * np.sin(0.6 * s + ph[4]),
                                 4.0 * np.sin(0.5 * s + ph[5])]))
        out.append(make_pose(t, r))
    return out, rng


def make_sequence(n_frames: int, seed: int = 0, intr: dict | None = None, scene: PlaneScene | None = None,
                  invalid_frac: float = 0.0):
    """Smooth trajectory (BASELINE config 1 / one stream of config 3): <= 2 cm and <= 0.5 deg per frame.

    Yields (gray, depth, T_w_c).
    """
    intr = intr or default_intrinsics()
    scene = scene or PlaneScene()
    poses, rng = sequence_poses(n_frames, seed)
    for T in poses:
        g, d = scene.render(intr, T, invalid_frac, rng)
        yield g, d, T
