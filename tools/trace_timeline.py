"""Reads a rocprofv3 --kernel-trace CSV (gpurun_out/<dir>/*_kernel_trace.csv) and prints, for the steady state, each
kernel's average duration, the average gap before it on its queue, and how much of the time the GPU ran 0 / 1 / 2+ kernels."""
import csv, sys, collections, glob
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/trace*/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
lo, hi = t0 + (t1 - t0) * 0.4, t0 + (t1 - t0) * 0.9  # steady state
sel = [e for e in ev if e[0] >= lo and e[1] <= hi]
dur = collections.defaultdict(list)
for s, e, n, q, st in sel:
    dur[n].append(e - s)
print(f"{f}: {len(sel)} dispatches in {(hi - lo) / 1e6:.2f} ms")
nb = len(dur.get("k_fast_grid", [1]))
print(f"batches: {nb}, ms per batch {(hi - lo) / 1e6 / nb:.4f}")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:22s} n {len(v):5d} avg {sum(v) / len(v) / 1e3:8.1f} us  per batch {sum(v) / nb / 1e3:8.1f} us")
# concurrency profile
pts = []
for s, e, n, q, st in sel:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
cur, last, acc = 0, lo, collections.Counter()
for t, d in pts:
    acc[min(cur, 4)] += t - last
    last = t; cur += d
tot = sum(acc.values())
print("kernels in flight:", {k: f"{v / tot:.2%}" for k, v in sorted(acc.items())})
# per-queue gaps
byq = collections.defaultdict(list)
for e in sel:
    byq[e[3]].append(e)
for q, L in byq.items():
    gaps = collections.defaultdict(list)
    for a, b in zip(L, L[1:]):
        gaps[b[2]].append(b[0] - a[1])
    print(f"queue {q}: {len(L)} dispatches;", ", ".join(f"{n}: {sum(v) / len(v) / 1e3:.1f}" for n, v in gaps.items()))
