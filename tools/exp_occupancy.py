"""One-off: resident workgroups per kernel over time in the pipelined benchmark, from in-kernel time stamps (no profiler).
Builds the library with -DRVO_OCC into /tmp, runs `steps` batches on `contexts` contexts, prints a text timeline
(10 us buckets) and per-kernel statistics.   python tools/exp_occupancy.py [contexts] [steps] [extra -D flags]"""
import ctypes as C, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
extra = sys.argv[3].split() if len(sys.argv) > 3 else []
so = os.environ.get("RVO_OCC_LIB", "/tmp/librgbdvo_occ.so")  # a library built beforehand with -DRVO_OCC, or build one now
if "RVO_OCC_LIB" not in os.environ:
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-DRVO_OCC", "-Wno-unused-result", *extra,
                       "-o", so, os.path.join(root, "slam_amd/csrc/rvo.hip")])
os.environ["RVO_LIB"] = so
import bench
import torch
from slam_amd import synth, _lib
from slam_amd.api import PipelinedFrameMatcher
n_pairs = 256
gray, depth, _ = bench.make_batch(n_pairs, 0)
dg = torch.from_numpy(gray).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
pipe = PipelinedFrameMatcher(synth.default_intrinsics(640, 480), n_pairs, contexts=nctx)
torch.cuda.synchronize()
L = _lib.load()
for k in range(6):
    pipe.submit(dg.data_ptr(), dd.data_ptr(), 20)
pipe.sync()
L.rvo_debug_occupancy(1, None)
import time
t0 = time.perf_counter()
for k in range(steps):
    pipe.submit(dg.data_ptr(), dd.data_ptr(), 20)
pipe.sync()
dt = time.perf_counter() - t0
K, B = 16, 65536
occ = np.zeros((K, B), np.int32)
assert L.rvo_debug_occupancy(0, occ.ctypes.data_as(C.c_void_p)) == 0
names = {1: "pyr", 2: "fast", 3: "buildkp", 4: "desc", 5: "rot", 6: "match", 7: "refine", 8: "finish", 9: "consist", 10: "clique", 11: "gn", 12: "walk"}
act = np.abs(occ).sum(0) > 0
idx = np.nonzero(act)[0]
# buckets wrap modulo 65536: rotate so that the run is contiguous
lo = idx.min(); hi = idx.max()
if hi - lo > B // 2:  # wrapped
    occ = np.roll(occ, B // 2, axis=1); idx = np.nonzero(np.abs(occ).sum(0) > 0)[0]; lo, hi = idx.min(), idx.max()
seg = occ[:, lo:hi + 1]
res = np.cumsum(seg, axis=1)
print(f"{nctx} contexts, {steps} batches in {dt * 1e3:.2f} ms = {dt / steps * 1e3:.4f} ms/batch; timeline {hi - lo + 1} buckets of 10 us")
hdr = "  t(us) " + " ".join(f"{names[k]:>7s}" for k in sorted(names))
print(hdr)
b0 = int(os.environ.get("OCC_START", "0"))
for b in range(b0, seg.shape[1]):
    if b > b0 + 400: break
    print(f"{b * 10:7d} " + " ".join(f"{res[k, b]:7d}" for k in sorted(names)))
print("resident-WG x time (WG*ms per batch):", {names[k]: round(res[k].sum() * 0.01 / steps, 2) for k in sorted(names)})
busy = {names[k]: float((res[k] > 0).mean()) for k in sorted(names)}
print("fraction of time with >=1 WG resident:", {k: round(v, 3) for k, v in busy.items()})
