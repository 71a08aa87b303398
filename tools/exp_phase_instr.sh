#!/bin/bash
# One-off ON THE GPU BOX: VALU instructions per wave of k_fast_grid by phase (ablation build, SQ counters per dispatch)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result -DRVO_PHASE_TIMING -o /tmp/abl_ph.so $root/slam_amd/csrc/rvo.hip || exit 1
cd /tmp && export TMPDIR=/tmp RVO_LIB=/tmp/abl_ph.so EXP_REPS=2
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_ph -o s -- python3 $root/tools/exp_phases.py > $out/pmc_ph.log 2>&1 || exit 5
python3 - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("$out/pmc_ph/s_counter_collection.csv")) if "k_fast_grid" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows: by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
d = list(by.values())
names = ["stage", "phaseA", "phaseB", "nms", "compact", "nth", "lists", "end"]
per = len(d) // 8
prev = 0
for i, n in enumerate(names):
    g = d[i * per:(i + 1) * per]
    v = sum(x["SQ_INSTS_VALU"] / x["SQ_WAVES"] for x in g) / len(g)
    ms = sum(x["SQ_ACTIVE_INST_VALU"] for x in g) / len(g) * 4 / 1024 / 2.4e9 * 1e3
    print(f"{n:8s} cumulative {v:7.1f} VALU/wave (+{v - prev:6.1f})  issue {ms:.4f} ms")
    prev = v
PY
