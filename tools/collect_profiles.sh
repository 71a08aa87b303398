#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh v6'): the bench line, the rocprofv3 kernel-trace summary of
# the same command and the two PMC passes (FETCH_SIZE / WRITE_SIZE / SQ issue counters, each a run of its own) that profiles/ holds.
# tools/summarise_profiles.py <tag> then copies the summaries from gpurun_out/ into profiles/.
set -o pipefail
tag=${1:-vX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout -k 10 400 python3 bench.py > $out/bench_$tag.json 2> $out/bench_$tag.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o $tag -- python3 $root/bench.py --no-cpu --no-legs > $out/prof_$tag.log 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- python3 $root/bench.py --no-cpu --no-legs --steps 4 --warmup 2 --prewarm 0 > $out/pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- python3 $root/bench.py --no-cpu --no-legs --steps 4 --warmup 2 --prewarm 0 > $out/pmc_write.log 2>&1 || exit 4
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq -o s -- python3 $root/bench.py --no-cpu --no-legs --steps 4 --warmup 2 --prewarm 0 > $out/pmc_sq.log 2>&1 || exit 5
echo done
