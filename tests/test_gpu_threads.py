"""Several contexts on one GPU, each driven by a host thread of its own at the same time (the header's rule: one context per
(device, host thread); ctypes releases the GIL inside every call, so the calls really overlap).  Every thread's results
must equal the oracle's: nothing in the library may be shared between contexts except read-only tables."""
import threading

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu


def test_three_contexts_on_three_threads(scene):
    import torch
    from oracle.pyoracle import Oracle, default_params as odef
    from slam_amd.api import Context, default_params
    W, H, thr, n = 320, 240, 15, 8
    intr = synth.default_intrinsics(W, H)
    over = dict(num_pyramid_levels=3, grid_width=40, grid_height=40, max_keypoints_per_grid=12)
    pairs = [synth.make_pair(900 + i, intr, scene, invalid_frac=0.1 * (i % 2)) for i in range(n)]
    orc = Oracle(intr, odef(frame_matcher=True, **over))
    want = []
    for p in pairs:
        orc.prepare(0, p[0], p[1], thr)
        orc.prepare(1, p[2], p[3], thr)
        want.append(orc.match_pair(0, 1))
    assert sum(r["status"] == 1 for r in want) >= n - 1
    gray = np.stack([p[j] for p in pairs for j in (0, 2)])
    depth = np.stack([p[j] for p in pairs for j in (1, 3)])
    errors = []

    def worker(k):
        try:
            torch.cuda.set_device(0)
            dg, dd = torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()
            torch.cuda.synchronize()
            ctx = Context(intr, default_params(frame_matcher=True, **over), max_frames=4 * n, max_pairs=n)
            for it in range(25):
                slots = np.arange(2 * n, dtype=np.int32) + 2 * n * (it % 2)
                if (it + k) % 3 == 0:  # the two-call form on host buffers between the fused device batches
                    ctx.prepare_frames(gray, depth, thr, slots)
                    res = ctx.match_pairs(slots[0::2], slots[1::2])
                else:
                    ctx.match_frames_dev(dg.data_ptr(), dd.data_ptr(), thr, slots)
                    res = ctx.pair_results(n)
                for i in range(n):
                    a, b = res[i], want[i]
                    if (a["status"], a["num_matches"], a["num_inliers"]) != (b["status"], b["num_matches"], b["num_inliers"]) or \
                            np.abs(a["pose"] - b["pose"]).max() > 1e-8:
                        errors.append((k, it, i, a["status"], a["num_matches"], a["num_inliers"], b["status"], b["num_matches"], b["num_inliers"]))
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
