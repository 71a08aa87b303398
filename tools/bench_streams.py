"""configs[2] shape on ONE GPU: S independent RGB-D streams advanced in lock-step by rvo_stream_step_dev (the batched
RGBDVO::ProcessFrame, vo.cpp:55-165); inputs resident in HBM.  Prints frames/s (= stream steps x S / time)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 6  # distinct frames per stream, cycled forward/backward so that consecutive frames always overlap
from slam_amd import synth
intr = synth.default_intrinsics(640, 480)
from concurrent.futures import ProcessPoolExecutor

def _seq(seed):
    sc = synth.PlaneScene(seed=seed % 8)
    return [(g, d) for g, d, _ in synth.make_sequence(T, seed=seed, intr=intr, scene=sc)]

if __name__ == "__main__":
    synth.make_texture()
    with ProcessPoolExecutor(8) as ex:
        seqs = list(ex.map(_seq, range(min(S, 32))))
    import torch
    from slam_amd.api import Context, default_params
    gray = np.stack([np.stack([seqs[s % len(seqs)][t][0] for s in range(S)]) for t in range(T)])
    depth = np.stack([np.stack([seqs[s % len(seqs)][t][1] for s in range(S)]) for t in range(T)])
    dg = torch.from_numpy(gray).cuda(); dd = torch.from_numpy(depth.view(np.int16)).cuda()
    p = default_params(); p.warmup_frames = 1
    # the streams are split over NC contexts (own HIP streams each): the latency-bound pair stages of one group of
    # streams overlap the frame stages of the other
    NC = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    per = S // NC
    ctxs = [Context(intr, p, max_frames=3 * per, max_pairs=per) for _ in range(NC)]
    for c in ctxs:
        c.streams_reset(per)
        if os.environ.get('RVO_DEPTH_IN_PLACE', '1') != '0': c.streams_depth_in_place(True)
        c.set_pair_chunks(1 if NC > 1 else int(os.environ.get('RVO_PAIR_CHUNKS', '2')))
    torch.cuda.synchronize()
    order = list(range(T)) + list(range(T - 2, 0, -1))
    def step(k):
        t = order[k % len(order)]
        for i, c in enumerate(ctxs):
            if NC > 1:
                c.wait_peer(ctxs[i - 1])  # starts when the other group's FAST stage is done
            c.stream_step_dev(dg[t, i * per:].data_ptr(), dd[t, i * per:].data_ptr(), per)
    for k in range(6): step(k)
    for c in ctxs: c.sync()
    PROF = bool(os.environ.get("RVO_STREAMS_PROF"))  # HIP events around every launch cost several percent
    for c in ctxs: c.profile_enable(PROF)
    n = 60
    t0 = time.perf_counter()
    for k in range(6, 6 + n): step(k)
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    prof = {}
    for c in ctxs:
        for name, v in c.profile_read().items():
            a = prof.setdefault(name, dict(total_ms=0.0, calls=0)); a["total_ms"] += v["total_ms"]; a["calls"] += v["calls"]
    res = [r for c in ctxs for r in c.stream_results(per)]
    S = per * NC
    ok = sum(r["status"] == 1 for r in res)
    print(f"contexts={NC} streams={S} ms/step={dt / n * 1e3:.3f} frames/s={S * n / dt:.0f} last-step success={ok}/{S} "
          f"mean inliers={np.mean([r['num_inliers'] for r in res]):.0f} fallbacks={sum(r['used_fallback'] for r in res)}")
    print({k: (round(v["total_ms"] / n, 3), v["calls"] // n) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])})
    print("mean matches", np.mean([r["num_matches"] for r in res]))
