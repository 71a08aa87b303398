#!/bin/bash
# One-off experiment ON THE GPU BOX: k_fast_grid with fewer workgroups per CU (RVO_FAST_LDS_MIN pads its LDS), pair bench and stream bench
so=/tmp/librgbdvo_variant.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result -DRVO_EXPERIMENT_KNOBS -o $so slam_amd/csrc/rvo.hip || exit 1
for lds in ${@:-0 24000 0 24000 0 24000}; do
  RVO_FAST_LDS_MIN=$lds RVO_LIB=$so RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs lds', $lds, round(d['value']), round(d['ms_per_step'],4))" || exit 1
  RVO_FAST_LDS_MIN=$lds RVO_LIB=$so timeout -k 10 150 python tools/bench_streams.py 512 1 2>/dev/null | grep -o "lds.*\|contexts=.*frames/s=[0-9]*" || exit 1
done
