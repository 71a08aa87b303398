#!/bin/bash
# One-off experiments ON THE GPU BOX: build a variant of the library with extra -D flags into /tmp and bench it.
#   bash tools/exp_variant.sh "-DMATCH_THREADS=256" [contexts...]      (RVO_SRC=dir/rvo.hip builds another source tree)
flags=$1; shift
so=/tmp/librgbdvo_variant.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result $flags -o $so ${RVO_SRC:-slam_amd/csrc/rvo.hip} || exit 1
for n in ${@:-2}; do
  RVO_LIB=$so RVO_BENCH_CONTEXTS=$n RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --no-legs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$flags', $n, round(d['value']), round(d['ms_per_step'],4), d['check'])" || exit 1
done
