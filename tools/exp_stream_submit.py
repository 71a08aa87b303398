"""Host time to enqueue one 512-stream step (rvo_stream_step_dev returns when everything is enqueued) against the step's
GPU time: is the stream path bound by the submitting thread?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from slam_amd import synth

if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    seqs = bench.make_leg_inputs.__globals__["_render_stream_seq"]
    from concurrent.futures import ProcessPoolExecutor
    synth.make_texture()
    with ProcessPoolExecutor(8) as ex:
        sq = list(ex.map(seqs, [(900 + u, 6) for u in range(16)]))
    import torch
    from slam_amd.api import Context, default_params
    intr = synth.default_intrinsics(640, 480)
    U, T = 16, 6
    ug = torch.from_numpy(np.stack([np.stack([sq[u][t][0] for u in range(U)]) for t in range(T)])).cuda()
    ud = torch.from_numpy(np.stack([np.stack([sq[u][t][1] for u in range(U)]) for t in range(T)]).view(np.int16)).cuda()
    ctx = Context(intr, default_params(warmup_frames=1), max_frames=3 * S, max_pairs=S)
    ctx.streams_reset(S)
    ctx.streams_set_source_index(U, np.arange(S) % U)
    order = list(range(T)) + list(range(T - 2, 0, -1))
    for k in range(6):
        ctx.stream_step_dev(ug[order[k % 10]].data_ptr(), ud[order[k % 10]].data_ptr(), S)
    ctx.sync()
    n = 60
    sub = 0.0
    t0 = time.perf_counter()
    for k in range(6, 6 + n):
        a = time.perf_counter()
        ctx.stream_step_dev(ug[order[k % 10]].data_ptr(), ud[order[k % 10]].data_ptr(), S)
        sub += time.perf_counter() - a
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"streams={S} submit ms/step={sub / n * 1e3:.3f} loop ms/step={(t1 - t0) / n * 1e3:.3f} total ms/step={(t2 - t0) / n * 1e3:.3f} drain after the loop ms={(t2 - t1) * 1e3:.2f}")
