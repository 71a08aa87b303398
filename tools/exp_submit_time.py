"""One-off: how long does the HOST take to enqueue staged batches (is the pipeline submission-bound or blocked)?"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
import torch
from slam_amd import synth
from slam_amd.api import PipelinedFrameMatcher
n_pairs = int(os.environ.get("PAIRS", "256"))
gray, depth, _ = bench.make_batch(n_pairs, 0)
dg = torch.from_numpy(gray).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
pipe = PipelinedFrameMatcher(synth.default_intrinsics(640, 480), n_pairs, contexts=1)
torch.cuda.synchronize()
for k in range(30):
    pipe.submit(dg.data_ptr(), dd.data_ptr(), 20)
pipe.sync()
for steps in (1, 2, 4, 8, 50):
    t0 = time.perf_counter()
    ts = []
    for k in range(steps):
        pipe.submit(dg.data_ptr(), dd.data_ptr(), 20)
        ts.append(time.perf_counter() - t0)
    t1 = time.perf_counter()
    pipe.sync()
    t2 = time.perf_counter()
    print(f"{steps} batches: submit loop {1e3 * (t1 - t0):.3f} ms, until done {1e3 * (t2 - t0):.3f} ms; per-submit return times (ms):",
          " ".join(f"{1e3 * x:.2f}" for x in ts[:10]), flush=True)
