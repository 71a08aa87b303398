#!/bin/bash
# Runs ON THE GPU BOX: the library's HOST code under AddressSanitizer + UBSan (the device code is built as usual), driven by
# the C++ facade demo (tests/native/facade_demo.cpp: RGBDVO::ProcessFrame over a sequence, FrameMatcher, cached keyframes).
# Everything is built into /tmp/san; prints the sanitizers' reports, if any, and the demo's exit code.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p /tmp/san $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result \
  -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
  -o /tmp/san/librgbdvo_hip.so $root/slam_amd/csrc/rvo.hip || exit 1
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fno-gpu-sanitize -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer \
  -o /tmp/san/facade_demo $root/tests/native/facade_demo.cpp -L/tmp/san -lrgbdvo_hip -Wl,-rpath,/tmp/san || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O1 -g -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
  -o /tmp/san/pipeline_san $root/tests/native/pipeline_san.cpp -L/tmp/san -lrgbdvo_hip -Wl,-rpath,/tmp/san || exit 1
python3 - <<PY || exit 1
import sys
sys.path.insert(0, "$root")
from slam_amd import synth
W, H, N = 320, 240, 9
intr = synth.default_intrinsics(W, H)
with open("/tmp/san/frames.raw", "wb") as f:
    for g, d, _ in synth.make_sequence(N, seed=9, intr=intr, scene=synth.PlaneScene(seed=1)):
        f.write(g.tobytes()); f.write(d.tobytes())
PY
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  timeout -k 10 200 /tmp/san/facade_demo /tmp/san/frames.raw 320 240 9 100 > $out/san_demo.out 2> $out/san_demo.err
rc=$?
echo "demo exit code $rc, $(wc -l < $out/san_demo.out) output lines"
echo "sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' $out/san_demo.err)"
grep -A12 'ERROR: AddressSanitizer\|runtime error:' $out/san_demo.err | head -60
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  timeout -k 10 200 /tmp/san/pipeline_san /tmp/san/frames.raw 320 240 9 > $out/san_pipe.out 2> $out/san_pipe.err
rc2=$?
cat $out/san_pipe.out
echo "pipeline driver exit code $rc2"
echo "sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' $out/san_pipe.err)"
grep -A12 'ERROR: AddressSanitizer\|runtime error:' $out/san_pipe.err | head -60
tail -3 $out/san_pipe.err
[ $rc -eq 0 ] && [ $rc2 -eq 0 ]
