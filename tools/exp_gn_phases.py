"""One-off: builds librgbdvo_hip with -DRVO_GN_TIMING into /tmp and runs a few batches; k_gn prints its phase cycles."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/librgbdvo_gn_timing.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-DRVO_GN_TIMING", "-Wno-unused-result", *os.environ.get("RVO_EXTRA_FLAGS", "").split(),
                       "-o", so, os.path.join(root, "slam_amd/csrc/rvo.hip")])
env = dict(os.environ, RVO_LIB=so, RVO_BENCH_CONTEXTS="1", RVO_PAIR_CHUNKS="1", RVO_BENCH_NOPROF="1")
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu", "--steps", "2", "--warmup", "1", "--pairs", sys.argv[1] if len(sys.argv) > 1 else "256"],
                     env=env, capture_output=True, text=True)
for l in (out.stdout + out.stderr).splitlines():
    if l.startswith("k_gn"):
        print(l)
