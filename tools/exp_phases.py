"""Ablation run (not part of the product): wall time of the prepare call with k_fast_grid cut short after each phase.
Build:  hipcc ... -DRVO_PHASE_TIMING -o slam_amd/abl_ph.so slam_amd/csrc/rvo.hip ;  RVO_LIB=slam_amd/abl_ph.so python tools/exp_phases.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from slam_amd import _lib, synth  # noqa: E402
from slam_amd.api import Context, default_params  # noqa: E402

n_pairs = 256
gray, depth, T = bench.make_batch(n_pairs, 0)
dev = torch.device("cuda", 0)
dg = torch.from_numpy(gray).to(dev)
dd = torch.from_numpy(depth.view(np.int16)).to(dev)
intr = synth.default_intrinsics(640, 480)
ctx = Context(intr, default_params(frame_matcher=True), max_frames=2 * n_pairs, max_pairs=n_pairs, device=0)
slots = np.arange(2 * n_pairs, dtype=np.int32)
L = _lib.load()
names = ["stage", "phaseA", "phaseB", "nms", "compact", "nth", "lists"]
import time
REPS = int(os.environ.get('EXP_REPS', '20'))
have_abl = hasattr(L, "rvo_debug_fast_stop")  # only in -DRVO_PHASE_TIMING builds
for stop in ((0, 1, 2, 3, 4, 5, 6, 99) if have_abl else (99,)):  # ablation: k_fast_grid returns after phase `stop`; wall time of the whole prepare call
    if have_abl:
        L.rvo_debug_fast_stop(stop)
    for _ in range(3):
        ctx.prepare_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(REPS):
        ctx.prepare_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
    ctx.sync()
    print("stop after", names[stop] if stop < 7 else "end", f"{(time.perf_counter() - t0) / REPS * 1e3:.3f} ms per prepare call", flush=True)
ctx.close()
