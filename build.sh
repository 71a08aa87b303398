#!/bin/bash
# Builds librgbdvo_hip.so (the product library) for gfx950 and the CPU oracle (test infrastructure).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fPIC -shared -Wall -Wno-unused-function -Wno-unused-result \
  -o slam_amd/librgbdvo_hip.so slam_amd/csrc/rvo.hip
make -C oracle -s
