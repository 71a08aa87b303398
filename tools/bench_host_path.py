"""Host-buffer entry point: rvo_stream_step with images in host memory (pageable numpy / pinned torch), n streams per step.
python tools/bench_host_path.py [streams] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from slam_amd import synth
from slam_amd.api import RGBDVOStreams, default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gray, depth, _ = bench.make_batch(n // 2, 0)  # 2 * (n/2) = n frames
intr = synth.default_intrinsics(640, 480)
for kind in ("pageable", "pinned"):
    vo = RGBDVOStreams(intr, n, params=default_params(warmup_frames=0))
    if kind == "pinned":
        g = torch.from_numpy(gray).pin_memory().numpy()
        d = torch.from_numpy(depth.view(np.int16)).pin_memory().numpy().view(np.uint16)
    else:
        g, d = gray, depth
    for _ in range(3):
        vo.ProcessFrame(g, d)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = vo.ProcessFrame(g, d)
    dt = (time.perf_counter() - t0) / steps
    mb = (g.nbytes + d.nbytes) / 1e6
    print(f"{kind:9s}: {dt * 1e3:.2f} ms per step of {n} frames = {n / dt:.0f} frames/s, {mb / dt / 1e3:.1f} GB/s of images; ok {sum(x['status'] == 1 for x in r)}", flush=True)
    if kind == "pinned":  # uploads of step k+1 beside the kernels of step k: two pinned sets, no result read-back per step
        from slam_amd.api import HostArray
        hg = [HostArray(g.shape, np.uint8) for _ in range(2)]
        hd = [HostArray(d.shape, np.uint16) for _ in range(2)]
        for i in range(2):
            hg[i].array[...] = g
            hd[i].array[...] = d
        for k in range(3):
            vo.ctx.stream_step_host_async(hg[k & 1].array, hd[k & 1].array, n)
        vo.ctx.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            vo.ctx.stream_step_host_async(hg[k & 1].array, hd[k & 1].array, n)
        vo.ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        r = vo.ctx.stream_results(n)
        print(f"pinned, rvo_stream_step_host_async: {dt * 1e3:.2f} ms per step = {n / dt:.0f} frames/s, {mb / dt / 1e3:.1f} GB/s; ok {sum(x['status'] == 1 for x in r)}", flush=True)
    vo.ctx.close()
