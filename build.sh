#!/bin/bash
# Builds librgbdvo_hip.so (the product library) for gfx950 and the CPU oracle (test infrastructure).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fPIC -shared -Wall -Wno-unused-function -Wno-unused-result \
  -o slam_amd/librgbdvo_hip.so slam_amd/csrc/rvo.hip
make -C oracle -s
# C++ facade demo (links the C ABI; run by tests/test_gpu_facade.py on the GPU box)
g++ -std=c++17 -O2 -Wall -o tests/native/facade_demo tests/native/facade_demo.cpp -Lslam_amd -lrgbdvo_hip -Wl,-rpath,'$ORIGIN/../../slam_amd'
