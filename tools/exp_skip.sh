#!/bin/bash
# ON THE GPU BOX: what does each kernel cost the staged pipeline?  Builds the library with RVO_SKIP_EXPERIMENT and runs
# bench.py with one launch (or several, comma separated) left out after the first 12 batches (results go stale, the timing stays meaningful).
so=/tmp/librgbdvo_skip.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result -DRVO_SKIP_EXPERIMENT -o $so slam_amd/csrc/rvo.hip 2>/dev/null || exit 1
for k in "$@"; do
  RVO_SKIP=$k RVO_LIB=$so RVO_BENCH_CONTEXTS=1 RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip $k', round(d['value']), round(d['ms_per_step'],4))" || exit 1
done
