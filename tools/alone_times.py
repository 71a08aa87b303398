"""Per-kernel launch durations with ONE context (launches do not overlap): RVO_BENCH_CONTEXTS=1 bench.py --no-cpu, kernels table only."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, RVO_BENCH_CONTEXTS="1", RVO_PAIR_CHUNKS="1", RVO_STAGED="0")
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu", "--no-legs"], env=env, capture_output=True, text=True)
d = json.loads(out.stdout.strip().splitlines()[-1])
print(f"{d['value']:.0f} pairs/s, {d['ms_per_step']:.3f} ms/step (one context)")
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["calls"]):
    print(f"  {k:16s} {v['avg_ms'] * v['calls'] / d['kernels_steps']:.3f} ms/step ({v['calls'] // d['kernels_steps']} launches)")
