#!/bin/bash
# One-off ON THE GPU BOX: compile-time knobs of the pair kernels, each built into /tmp and benched twice (pairs/s, 100 steps)
so=/tmp/librgbdvo_variant.so
for fl in "-DRVO_NOOP=1" "$@" "-DRVO_NOOP=2"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result $fl -o $so slam_amd/csrc/rvo.hip 2>/dev/null || { echo "$fl: build failed"; continue; }
  r=""
  for i in 1 2; do
    v=$(RVO_LIB=$so RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1000,1), 'ok' if d['check']['pairs_success']==256 else 'BAD')") || exit 1
    r="$r $v"
  done
  echo "$fl:$r"
done
