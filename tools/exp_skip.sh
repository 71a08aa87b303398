#!/bin/bash
# One-off experiment ON THE GPU BOX: what each launch costs the staged pipeline -- the batch time with that launch left out
# (results go stale after the 12th batch; RVO_SKIP_EXPERIMENT build), beside tools/alone_times.py's time of the kernel alone.
so=/tmp/librgbdvo_variant.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result -DRVO_SKIP_EXPERIMENT -o $so slam_amd/csrc/rvo.hip || exit 1
for k in none init_rotation match,match_finish gauss_newton clique clique,clique_walk track_ids consistency refine refine,refine_finish build_keypoints pyr_down none "init_rotation,gauss_newton,clique,track_ids" "init_rotation,gauss_newton,clique,track_ids,match,match_finish"; do
  RVO_SKIP=$k RVO_LIB=$so RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip $k', round(d['value']), round(d['ms_per_step'],4))" || exit 1
done
