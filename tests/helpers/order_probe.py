"""Helper of tests/test_gpu_launch.py: a fresh process that uses the library BEFORE torch touches the GPU, then torch, then
the library again (the order that failed in round 1: two HIP runtimes in one process, see slam_amd/_lib.py)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from slam_amd import synth
from slam_amd.api import Context, default_params
intr = synth.default_intrinsics(320, 240)
c = Context(intr, default_params(frame_matcher=True, num_pyramid_levels=2), max_frames=2, max_pairs=1)
g0,d0,g1,d1,T = synth.make_pair(3, intr, synth.PlaneScene())
c.prepare_frames(np.stack([g0,g1]), np.stack([d0,d1]), 20, [0,1])
print(c.match_pairs([0],[1])[0]["status"])
c.close()
import torch
print("avail", torch.cuda.is_available())
torch.cuda.init()
x = torch.arange(10, device="cuda")
print(x.sum().item())
c = Context(intr, default_params(frame_matcher=True, num_pyramid_levels=2), max_frames=2, max_pairs=1)
c.close()
print("ok")
libs = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
print("hip runtimes mapped:", len(libs), sorted(libs))
