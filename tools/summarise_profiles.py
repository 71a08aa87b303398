"""Copies the summaries tools/collect_profiles.sh left under gpurun_out/ into profiles/ (run here, after the gpurun call):
python tools/summarise_profiles.py v11 [r02]"""
import collections
import os
import csv
import json
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
shutil.copy(f"gpurun_out/prof_{tag}/{tag}_kernel_stats.csv", f"profiles/{rnd}_{tag}_kernel_stats.csv")
line = open(f"gpurun_out/bench_{tag}.json").read().strip().splitlines()[-1]
open(f"profiles/{rnd}_{tag}_bench.json", "w").write(line + "\n")

out = {}
for ctr, f in (("FETCH_SIZE", "gpurun_out/pmc_fetch/f_counter_collection.csv"),
               ("WRITE_SIZE", "gpurun_out/pmc_write/w_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[name].append(float(r["Counter_Value"]))
    out[ctr] = {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}

names = {"k_ingest": "ingest", "k_stream_post": "stream_post", "k_stream_post2": "stream_post2", "k_pyr_down": "pyr_down", "k_fast_grid": "fast_grid", "k_build_keypoints": "build_keypoints",
         "k_descriptors": "descriptors", "k_init_rotation": "init_rotation", "k_match": "match", "k_match_finish": "match_finish", "k_refine": "refine",
         "k_refine_finish": "refine_finish", "k_consistency": "consistency", "k_clique": "clique", "k_clique_walk": "clique_walk", "k_gn": "gauss_newton", "k_track_count": "track_count", "k_track_assign": "track_assign"}
res = json.load(open(f"profiles/{rnd}_pmc_traffic.json")) if os.path.exists(f"profiles/{rnd}_pmc_traffic.json") else dict(json.load(open("profiles/r02_pmc_traffic.json")), source_tag=tag)
res["kernels"] = {}
for k, n in names.items():
    if k not in out["FETCH_SIZE"]:
        continue
    cnt, f = out["FETCH_SIZE"][k]
    w = out["WRITE_SIZE"][k][1]
    res["kernels"][n] = {"launches": cnt, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                         "traffic_bytes": round((2 * f + w) * 1024)}
res["source_tag"] = tag
json.dump(res, open(f"profiles/{rnd}_pmc_traffic.json", "w"), indent=1)
for n, v in res["kernels"].items():
    print(n, v["launches"], v["traffic_bytes"])
d = json.loads(line)
print(d["value"], d["ms_per_step"], d["roofline"], d["cpu_baseline"]["value"])

# SQ issue counters (a third --pmc pass, optional): where each kernel's wave cycles go
import os
sq = "gpurun_out/pmc_sq/s_counter_collection.csv"
if os.path.exists(sq):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sq)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    nb = len(acc["k_fast_grid"]["SQ_WAVE_CYCLES"])  # batches in the run
    with open(f"profiles/{rnd}_{tag}_sq_counters.csv", "w") as f:
        f.write("kernel,launches_per_batch,SQ_WAVES,wave_instr_per_wave,SQ_INSTS_VALU,SQ_WAVE_CYCLES,valu_active_frac,inst_active_frac,wait_any_frac,wait_inst_frac,lds_active_frac,valu_issue_ms_at_2.4GHz\n")
        for k, v in acc.items():
            if k not in names:
                continue
            m = {c: sum(x) / len(x) for c, x in v.items()}
            wc = m["SQ_WAVE_CYCLES"]
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs: x4 cycles / 1024 SIMDs / 2.4 GHz
            ms = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.4e9 * 1e3
            f.write(f"{names[k]},{len(v['SQ_WAVE_CYCLES']) // nb},{m.get('SQ_WAVES', 0):.0f},{(m['SQ_INSTS_VALU'] / m['SQ_WAVES']) if m.get('SQ_WAVES') else 0:.0f},{m['SQ_INSTS_VALU']:.0f},{wc:.0f},{m['SQ_ACTIVE_INST_VALU'] / wc:.3f},"
                    f"{m['SQ_ACTIVE_INST_ANY'] / wc:.3f},{m['SQ_WAIT_ANY'] / wc:.3f},{m['SQ_WAIT_INST_ANY'] / wc:.3f},"
                    f"{m['SQ_ACTIVE_INST_LDS'] / wc:.3f},{ms:.4f}\n")
    shutil.copy(f"profiles/{rnd}_{tag}_sq_counters.csv", f"profiles/{rnd}_sq_counters.csv")  # the copy bench.py reads
    print(open(f"profiles/{rnd}_{tag}_sq_counters.csv").read())
