import sys, numpy as np, os
sys.path.insert(0,'.')
import bench
gray, depth, T = bench.make_batch(128, 0)
import torch
from slam_amd import synth
from slam_amd.api import Context, default_params
dg = torch.from_numpy(gray).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
ctx = Context(synth.default_intrinsics(), default_params(frame_matcher=True), max_frames=256, max_pairs=1)
slots = np.arange(256, dtype=np.int32)
for i in range(3): ctx.prepare_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
ctx.sync(); ctx.profile_enable(True)
for i in range(10):
    ctx.prepare_frames_dev(dg.data_ptr(), dd.data_ptr(), 20, slots)
p = ctx.profile_read()
print(os.environ.get('RVO_LIB','base'), {k: round(v['total_ms']/v['calls'],4) for k,v in p.items()})
