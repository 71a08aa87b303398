#!/bin/bash
# Experiment ON THE GPU BOX: shader clock and power while the bench pipeline is running (is 2.4 GHz the clock the
# VALU-issue roofline should be priced at?).
cd ${GRAFT_REPO_ROOT:-.}
RVO_BENCH_NOPROF=1 timeout -k 10 100 python bench.py --no-cpu --steps 15000 --warmup 10 > gpurun_out/clk_bench.json 2>/dev/null &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" | tr '\n' ';'
  echo
  sleep 1
done
wait $pid
cut -c1-200 gpurun_out/clk_bench.json
