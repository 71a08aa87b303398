#!/bin/bash
# One-off A/B ON THE GPU BOX: a variant library with the experiment knobs compiled in, a knob from the environment set to each value in turn
#   bash tools/exp_cmd.sh RVO_FUSE_BUILD 0 1 0 1
so=/tmp/librgbdvo_variant.so
knob=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result -DRVO_EXPERIMENT_KNOBS -o $so slam_amd/csrc/rvo.hip || exit 1
for v in "$@"; do
  env $knob=$v RVO_LIB=$so RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs $knob', $v, round(d['value']), round(d['ms_per_step'],4))" || exit 1
  env $knob=$v RVO_LIB=$so timeout -k 10 150 python tools/bench_streams.py 512 1 2>/dev/null | grep -o "contexts=.*frames/s=[0-9]*" || exit 1
done
