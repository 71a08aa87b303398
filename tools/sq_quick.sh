#!/bin/bash
# Runs ON THE GPU BOX: only the SQ issue-counter pass of tools/collect_profiles.sh, summarised on stdout (an iteration aid).
# bash tools/sq_quick.sh [script.py args...]   (default: bench.py, 4 steps); prints the sum over ALL launches too.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sqq -o s -- python3 ${@:-$root/bench.py --no-cpu --no-legs --steps 4 --warmup 2 --prewarm 0} > $out/pmc_sqq.log 2>&1 || exit 5
python3 - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$out/pmc_sqq/s_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if not m.get("SQ_WAVES"): continue
    ms = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.4e9 * 1e3
    n = len(v["SQ_WAVES"]) // max(1, len(acc["k_fast_grid"]["SQ_WAVES"]))
    tot += ms * n
    grand = globals().get("grand", 0.0) + ms * len(v["SQ_WAVES"])
    print(f"{k:20s} x{n} waves {m['SQ_WAVES']:8.0f} valu/wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:7.0f} issue_ms {ms:.4f}")
print("issue ms per batch", round(tot, 4), " sum over all launches of the run", round(grand, 3), " fast_grid launches", len(acc["k_fast_grid"]["SQ_WAVES"]))
PY
