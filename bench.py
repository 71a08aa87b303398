#!/usr/bin/env python3
"""Benchmark of the rgbdvo front-end on MI355X (BASELINE.json metric: frame-pairs/sec at 640x480).

Workload (configs[1]): a batch of 256 independent synthetic 640x480 gray+depth frame pairs per GPU, 3-level
pyramid, default feature grid.  One step = Frame::PrepareFrame on all 512 frames + FrameMatcher::Match on the
256 pairs (initial rotation, matching, refinement, clique, Gauss-Newton), inputs resident in HBM.
N > 1: one process per GPU -- `python bench.py --gpus N` starts the N ranks itself (slam_amd/launch.py); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the launcher's ranks are used -- every rank runs its own 256 pairs (weak scaling, no collective in the
timed loop); the per-pair poses are gathered once at the end over RCCL (all_gather) and timed separately.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"


def _render_pair(args):
    seed, = args
    from slam_amd import synth
    intr = synth.default_intrinsics(640, 480)
    g0, d0, g1, d1, T = synth.make_pair(seed, intr, synth.PlaneScene())
    return g0, d0, g1, d1, T[:3]


def make_batch(n_pairs: int, seed0: int):
    from concurrent.futures import ProcessPoolExecutor
    from slam_amd import synth
    synth.make_texture()  # build once before forking
    world = int(os.environ.get("WORLD_SIZE", "1"))
    workers = int(os.environ.get("RVO_BENCH_WORKERS", "8" if world <= 2 else "6"))
    with ProcessPoolExecutor(max_workers=workers) as ex:
        out = list(ex.map(_render_pair, [(seed0 + i,) for i in range(n_pairs)], chunksize=4))
    gray = np.empty((2 * n_pairs, 480, 640), np.uint8)
    depth = np.empty((2 * n_pairs, 480, 640), np.uint16)
    for i, (g0, d0, g1, d1, _) in enumerate(out):
        gray[2 * i], gray[2 * i + 1] = g0, g1
        depth[2 * i], depth[2 * i + 1] = d0, d1
    return gray, depth, np.stack([o[4] for o in out])


def _render_seq_frame(args):
    T, = args
    from slam_amd import synth
    intr = synth.default_intrinsics(640, 480)
    return synth.PlaneScene().render(intr, T, 0.0, None)


def make_sequence_frames(n_frames: int, seed: int = 0):
    """configs[0]: one 640x480 stream over a textured plane (slam_amd.synth.make_sequence's trajectory), rendered by a pool."""
    from concurrent.futures import ProcessPoolExecutor
    from slam_amd import synth
    synth.make_texture()
    poses, _ = synth.sequence_poses(n_frames, seed)
    workers = int(os.environ.get("RVO_BENCH_WORKERS", "8"))
    with ProcessPoolExecutor(max_workers=workers) as ex:
        out = list(ex.map(_render_seq_frame, [(T,) for T in poses], chunksize=4))
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), poses


def _render_any_pair(args):
    seed, size, contrast, invalid = args
    from slam_amd import synth
    intr = synth.default_intrinsics(*size)
    g0, d0, g1, d1, T = synth.make_pair(seed, intr, synth.PlaneScene(contrast=contrast), invalid_frac=invalid)
    return g0, d0, g1, d1, T[:3]


def _render_stream_seq(args):
    seed, n = args
    from slam_amd import synth
    intr = synth.default_intrinsics(640, 480)
    return [(g, d) for g, d, _ in synth.make_sequence(n, seed=seed, intr=intr, scene=synth.PlaneScene(seed=seed % 4))]


LEG_STREAMS, LEG_UNIQUE, LEG_SEQ_T = 512, 16, 6      # configs[2] per GPU: 512 streams replaying 16 sequences of 6 frames
LEG_BIG_PAIRS = 32                                     # configs[3]: 1280x960, 4 levels, pairs per batch
LEG_ADV_PAIRS = 256                                    # configs[4]: low texture + 30 % invalid depth


def make_leg_inputs():
    """Synthetic inputs of the configs[2] / [3] / [4] legs (rendered by a pool before anything touches the GPU)."""
    from concurrent.futures import ProcessPoolExecutor
    from slam_amd import synth
    synth.make_texture()
    workers = int(os.environ.get("RVO_BENCH_WORKERS", "8"))
    with ProcessPoolExecutor(max_workers=workers) as ex:
        seqs = list(ex.map(_render_stream_seq, [(900 + u, LEG_SEQ_T) for u in range(LEG_UNIQUE)]))
        big = list(ex.map(_render_any_pair, [(7000 + i, (1280, 960), 1.0, 0.0) for i in range(LEG_BIG_PAIRS)]))
        adv = list(ex.map(_render_any_pair, [(40 + i, (640, 480), 0.12, 0.3) for i in range(LEG_ADV_PAIRS)], chunksize=4))

    def stack(pairs):
        g = np.stack([x for p in pairs for x in (p[0], p[2])])
        d = np.stack([x for p in pairs for x in (p[1], p[3])])
        return g, d, np.stack([p[4] for p in pairs])
    return dict(seqs=seqs, big=stack(big), adv=stack(adv))


def path_bytes(W, H, L, K, N):
    """(B_frame, B_pair, B_step) at K keypoints per frame and N matches per pair."""
    G = W * H
    P = sum((W >> l) * (H >> l) for l in range(1, L))
    b_frame = G + 2 * P + 128 * K + 4 * K
    return b_frame, 2 * b_frame + 160 * K + N * N / 4, b_frame + 160 * K + N * N / 4


def run_pair_leg(name, intr, params, gray, depth, T_gt, thr, dev_index, steps, warmup, log, oracle_pairs):
    """One FrameMatcher::Match batch per step through the staged pipeline (the configs[1] arrangement) on other inputs:
    pairs/s with the inputs resident in HBM, result statistics, and a count check of a few pairs against the oracle."""
    import torch
    from slam_amd.api import PipelinedFrameMatcher, STATUS_NAMES
    n_pairs = len(T_gt)
    dev = torch.device("cuda", dev_index)
    d_gray = torch.from_numpy(gray).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)
    pipe = PipelinedFrameMatcher(intr, n_pairs, params=params, device=dev_index)
    torch.cuda.synchronize()
    for _ in range(warmup):
        pipe.submit(d_gray.data_ptr(), d_depth.data_ptr(), thr)
    pipe.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(d_gray.data_ptr(), d_depth.data_ptr(), thr)
    pipe.sync()
    dt = time.perf_counter() - t0
    last = pipe.k - 1
    res = pipe.results(last)
    ctx = pipe.context_of(last)
    L = params.num_pyramid_levels
    kp = [sum(ctx.level_info(int(s), l)["num_keypoints"] for l in range(L)) for s in pipe.slots_of(last)[::max(1, 2 * n_pairs // 16)]]
    K, N = float(np.mean(kp)), float(np.mean([r["num_matches"] for r in res]))
    ok = [i for i, r in enumerate(res) if r["status"] == 1]
    hist = {}
    for r in res:
        hist[STATUS_NAMES[r["status"]]] = hist.get(STATUS_NAMES[r["status"]], 0) + 1
    same = None
    if oracle_pairs:
        from oracle.pyoracle import Oracle, default_params as odef
        orc = Oracle(intr, odef(frame_matcher=True, num_pyramid_levels=L))
        same = 0
        for i in range(oracle_pairs):
            orc.prepare(0, gray[2 * i], depth[2 * i], thr)
            orc.prepare(1, gray[2 * i + 1], depth[2 * i + 1], thr)
            ro = orc.match_pair(0, 1)
            same += int((ro["status"], ro["num_matches"], ro["num_inliers"]) == (res[i]["status"], res[i]["num_matches"], res[i]["num_inliers"])
                        and float(np.abs(ro["pose"] - res[i]["pose"]).max()) < 1e-8)
        orc.close()
    pipe.close()
    _, b_pair, _ = path_bytes(intr["width"], intr["height"], L, K, N)
    rate = n_pairs * steps / dt
    log(f"{name} leg done")
    return dict(value=rate, unit="frame-pairs/s", pairs_per_batch=n_pairs, steps=steps, warmup=warmup, ms_per_step=dt / steps * 1e3,
                mean_kp_per_frame=K, mean_matches=N, mean_inliers=float(np.mean([r["num_inliers"] for r in res])),
                status_histogram=hist,
                max_abs_pose_err_vs_truth=float(max(np.abs(res[i]["pose"] - T_gt[i]).max() for i in ok)) if ok else None,
                bytes_per_pair=b_pair, roofline_path_frac=rate * b_pair / HBM_PEAK,
                oracle_check=f"{same}/{oracle_pairs} pairs with the oracle's status, match and inlier counts and pose (1e-8)" if oracle_pairs else None,
                oracle_pairs_identical=same, oracle_pairs_checked=oracle_pairs)


def run_stream_leg(seqs, dev_index, steps, warmup, log, oracle_streams=2, oracle_steps=10):
    """configs[2] per GPU: LEG_STREAMS lock-step streams through rvo_stream_step_dev (RGBDVO::ProcessFrame, vo.cpp:55-165)
    on ONE context (the library cuts a step into four groups of 128), images resident in HBM once per distinct sequence
    (rvo_streams_set_source_index), trajectories recorded on the device, nothing synchronised between the steps."""
    import torch
    from slam_amd import synth
    from slam_amd.api import Context, default_params
    S, U, T = LEG_STREAMS, len(seqs), len(seqs[0])
    intr = synth.default_intrinsics(640, 480)
    dev = torch.device("cuda", dev_index)
    ug = torch.from_numpy(np.stack([np.stack([seqs[u][t][0] for u in range(U)]) for t in range(T)])).to(dev)  # [T, U, H, W]
    ud = torch.from_numpy(np.stack([np.stack([seqs[u][t][1] for u in range(U)]) for t in range(T)]).view(np.int16)).to(dev)
    p = default_params(warmup_frames=1)  # the reference discards 99 frames first (vo.cpp:58-69); the synthetic streams start at once
    ctx = Context(intr, p, max_frames=3 * S, max_pairs=S, device=dev_index)
    ctx.streams_reset(S)
    ctx.streams_set_source_index(U, np.arange(S) % U)
    ctx.streams_depth_in_place(True)  # every frame's depth image stays resident for the whole leg
    ctx.streams_record(warmup + steps)
    order = list(range(T)) + list(range(T - 2, 0, -1))  # forward and back: consecutive frames always overlap
    torch.cuda.synchronize()
    for k in range(warmup):
        t = order[k % len(order)]
        ctx.stream_step_dev(ug[t].data_ptr(), ud[t].data_ptr(), S)
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        t = order[k % len(order)]
        ctx.stream_step_dev(ug[t].data_ptr(), ud[t].data_ptr(), S)
    ctx.sync()
    dt = time.perf_counter() - t0
    traj = ctx.trajectories(S, warmup + steps)  # [S, steps, 16]: pose 3x4, status, inliers, rc, used_fallback
    final = ctx.stream_results(S)
    kp = [sum(ctx.level_info(3 * s + j, l)["num_keypoints"] for l in range(3)) for s in range(0, S, 64) for j in range(3)]
    K, N = float(np.mean(kp)), float(np.mean([r["num_matches"] for r in final]))
    replicas_equal = all(np.array_equal(traj[s], traj[s % U]) for s in range(U, S))
    from oracle.pyoracle import Oracle, OracleVO, default_params as odef
    same, total = 0, 0
    for s in range(oracle_streams):
        ovo = OracleVO(Oracle(intr, odef(warmup_frames=1)))
        for k in range(min(oracle_steps, warmup + steps)):
            t = order[k % len(order)]
            ro = ovo.process(seqs[s % U][t][0], seqs[s % U][t][1])
            total += 1
            same += int(traj[s, k, 14] == ro["rc"] and (k == 0 or (traj[s, k, 12], traj[s, k, 13]) == (ro["status"], ro["num_inliers"]))
                        and float(np.abs(traj[s, k, :12].reshape(3, 4) - ro["pose"]).max()) < 1e-8)
        ovo.close()
    ctx.close()
    _, _, b_step = path_bytes(640, 480, 3, K, N)
    rate = S * steps / dt
    log("configs2 leg done")
    return dict(value=rate, unit="stream-frames/s", streams=S, contexts=1, unique_sequences=U, steps=steps, warmup=warmup,
                ms_per_step=dt / steps * 1e3, mean_kp_per_frame=K, mean_matches=N,
                mean_inliers=float(np.mean([r["num_inliers"] for r in final])),
                estimates_ok=int((traj[:, 1:, 12] == 1).sum()), estimates=S * (warmup + steps - 1),
                fallbacks=int(traj[:, :, 15].sum()), replicas_bit_equal=bool(replicas_equal),
                bytes_per_step=b_step, roofline_path_frac=rate * b_step / HBM_PEAK,
                oracle_check=f"{same}/{total} stream steps (streams 0..{oracle_streams - 1}, first {oracle_steps} steps) with OracleVO's rc, status, inlier count and pose (1e-8)",
                oracle_steps_identical=same, oracle_steps_checked=total)


def latest_profile(name: str) -> str:
    """profiles/rNN_<name> of the newest round that has one (the committed rocprofv3 --pmc summaries of this command)."""
    for rnd in ("r03", "r02", "r01"):
        p = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", f"r03_{name}")


def algorithmic_bytes(geom: dict, stats: dict) -> dict:
    """Algorithmic HBM bytes per LAUNCH of each kernel kind (definitions: DESIGN.md section 5)."""
    W, H, L = geom["W"], geom["H"], geom["levels"]
    nf, npair = stats["n_frames"], stats["n_pairs"]
    lv = [(W >> l) * (H >> l) for l in range(L)]
    K, N = stats["mean_kp_per_frame"], stats["mean_matches"]
    B = {}
    B["pyr_down"] = nf * sum(lv[l - 1] + lv[l] for l in range(1, L)) / max(1, L - 1)
    B["fast_grid"] = nf * (sum(lv) + geom["total_buckets"] * (geom["max_per"] * (4 + 80) + 8))
    # bucket lists in, keypoint records out (the descriptor rows stay in bucket order since round 3: kp_src names them)
    B["build_keypoints"] = nf * (geom["total_buckets"] * (geom["max_per"] * 4 + 8) + K * (36 + 8 + 4 + 2 + 8) + K * 2)
    B["init_rotation"] = npair * 2 * lv[L - 1]
    B["match"] = npair * (2 * K * 80 + 2 * K * (32 + 2) + N * 8)
    B["refine"] = npair * N * (2 * 100 + 20)
    B["refine_finish"] = npair * N * (20 + 100)
    B["consistency"] = npair * (N * N / 8 + N * 48)
    B["clique"] = npair * (N * N / 8 + N * 12)   # degrees (the bit matrix once), sort keys in and out
    B["clique_walk"] = npair * (N * N / 8 + N * 5)  # every candidate's row once, its key, the inlier flags
    B["gauss_newton"] = npair * N * 88
    return B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=256, help="independent frame pairs per GPU per step")
    ap.add_argument("--cpu-pairs", type=int, default=256, help="pairs timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="passes over those pairs (about 10 s of CPU work in all)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--seq-frames", type=int, default=300, help="frames of the configs[0] sequence timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-legs", action="store_true", help="skip the configs[2] / configs[3] / configs[4] legs (rank 0, N=1 only)")
    ap.add_argument("--prewarm", type=float, default=float(os.environ.get("RVO_BENCH_PREWARM_S", "1.0")),
                    help="seconds of untimed work before the warm-up steps (clocks of an idle box)")
    a = ap.parse_args()

    # `python bench.py --gpus N` without an outer launcher: start the N rank processes ourselves (one per GPU) and relay
    # rank 0's JSON line.  Under torch.distributed.run (RANK / WORLD_SIZE already set) this process IS a rank.
    from slam_amd.launch import launched_by_parent, spawn_ranks
    if a.gpus > 1 and not launched_by_parent():
        sys.exit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus))

    # stdout carries exactly one JSON line: library banners (gloo, RCCL, HIP) that print to fd 1 are sent to stderr
    from slam_amd.launch import reserve_stdout
    json_fd = reserve_stdout()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_pairs = a.pairs
    t_start = time.perf_counter()

    def log(msg):  # progress on stderr (a long run must not look hung; a stuck phase names itself)
        sys.stderr.write(f"[bench r{rank} +{time.perf_counter() - t_start:6.1f}s] {msg}\n")
        sys.stderr.flush()

    # synthetic inputs first (forks workers; nothing has touched the GPU yet)
    gray, depth, T_gt = make_batch(n_pairs, seed0=100000 * rank)
    seq = None
    if world == 1 and not a.no_cpu and a.seq_frames > 0:
        seq = make_sequence_frames(a.seq_frames)

    legs_in = make_leg_inputs() if (world == 1 and not a.no_legs) else None
    log("inputs made")
    import torch
    import torch.distributed as dist
    from slam_amd import synth
    from slam_amd.api import PipelinedFrameMatcher

    # one process per GPU; on a box with fewer GPUs than ranks (rehearsals) ranks share devices
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if world > 1:
        # gloo carries the control plane (barriers, the max-over-ranks clock); nccl (= RCCL over xGMI) carries the
        # one data collective of the path, the final trajectory gather
        if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # RCCL bootstrap only; the data path is xGMI
        dist.init_process_group("cpu:gloo,cuda:nccl", timeout=__import__("datetime").timedelta(seconds=180))
    log("process group up")
    dev = torch.device("cuda", dev_index)
    d_gray = torch.from_numpy(gray).to(dev)
    d_depth = torch.from_numpy(depth.view(np.int16)).to(dev)

    intr = synth.default_intrinsics(640, 480)
    # Every step is one whole batch of n_pairs pairs through ONE C-ABI call on ONE context; the library's staged pipeline
    # (rvo_set_pipeline, the default) runs the stages of successive batches on four streams, so the latency-bound tail of
    # batch k overlaps the frame stages of batch k+1.  RVO_BENCH_CONTEXTS=2 selects round 1's arrangement instead (two
    # contexts taking alternate batches, each batch one chain).
    n_ctx = int(os.environ.get("RVO_BENCH_CONTEXTS", "1"))
    pipe = PipelinedFrameMatcher(intr, n_pairs, device=dev_index, contexts=n_ctx)
    ctxs = pipe.ctxs
    if os.environ.get("RVO_PAIR_CHUNKS"):
        for c in ctxs:
            c.set_pair_chunks(int(os.environ["RVO_PAIR_CHUNKS"]))
    log("contexts created")
    thr = ctxs[0].params.fast_init_threshold
    torch.cuda.synchronize()  # inputs are resident before anything is enqueued on the contexts' own streams

    def step(k):
        # FrameMatcher::Match for the whole batch: Frame::Prepare of both frames + EstimateMotion of every pair
        pipe.submit(d_gray.data_ptr(), d_depth.data_ptr(), thr)

    def sync_all():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()

    def read_prof():
        acc = {}
        for c in ctxs:
            for name, v in c.profile_read().items():
                e = acc.setdefault(name, dict(total_ms=0.0, calls=0))
                e["total_ms"] += v["total_ms"]
                e["calls"] += v["calls"]
        return acc

    # Warm-up.  With HIP events around all ~15 launches of a batch the pipeline runs about 7 % slower, so the warm-up
    # (events around every launch) only finds the dominant kernel; the timed region then brackets that kernel alone.
    noprof = bool(os.environ.get("RVO_BENCH_NOPROF"))
    # A box that has been idle runs its first ~0.5 s of work at lower clocks (first run after a lease: 230k instead of
    # 290k pairs/s with 10 warm-up steps = 9 ms): keep the GPU busy for a second before the W warm-up steps.  Untimed.
    prewarm_s = a.prewarm
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < prewarm_s:
        for k in range(8):
            step(k)
        sync_all()
    log("clocks warm")
    if not noprof:
        for c in ctxs:
            c.profile_enable(True)
    for k in range(a.warmup):
        step(k)
    sync_all()
    log("warm-up done")
    dom = None
    if not noprof:
        wp = read_prof()
        dom = max(wp, key=lambda k: wp[k]["total_ms"]) if wp else None
        # Inside the pipeline every launch is stretched by whatever runs beside it (k_refine: 0.15 ms alone, 0.5 ms here),
        # so "dominant" is decided by the work a kernel does, not by how long it was resident: the largest VALU issue
        # time per batch in the committed SQ counters of this command, if that file is there
        sp0 = latest_profile("sq_counters.csv")
        if wp and os.path.exists(sp0):
            import csv
            issue = {r["kernel"]: float(r["valu_issue_ms_at_2.4GHz"]) * int(r["launches_per_batch"]) for r in csv.DictReader(open(sp0))}
            cand = [k for k in wp if k in issue]
            if cand:
                dom = max(cand, key=lambda k: issue[k])
        for c in ctxs:
            c.profile_enable(dom is not None, only=dom)  # also clears the accumulated records

    def barrier():
        if world > 1:
            dist.all_reduce(torch.zeros(1))  # CPU tensor -> gloo

    barrier()
    sync_all()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k)
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    log("timed region done")
    prof_dom = read_prof() if not noprof else {}  # the dominant kernel over the timed region, on its own stream
    # the other kernels' durations (the `kernels` table): a separate pass after the timed region, events around every launch
    prof = {}
    post_steps = min(a.steps, 20)
    if not noprof:
        for c in ctxs:
            c.profile_enable(True)
        for k in range(post_steps):
            step(a.warmup + a.steps + k)
        sync_all()
        prof = read_prof()
        for c in ctxs:
            c.profile_enable(False)
    log("kernel table pass done")
    last = pipe.k - 1
    ctx = pipe.context_of(last)  # holds the last batch's results
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # results + final trajectory gather (RCCL all_gather of the per-pair 3x4 poses and status)
    res = pipe.results(last)
    sync_all()
    gather_ms = None
    gather_backend = None
    gather_ranks = None
    gather_total_ms = None
    verdict = dict(gather_hung=False, gather_ok=True, gather_over_rccl=False, exit_code=0)
    if world > 1:
        from slam_amd.dist import gather_trajectories_auto
        local = np.stack([np.concatenate([r["pose"].reshape(12), [r["status"], r["num_inliers"]]]) for r in res])
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        # RCCL through the C ABI (rvo_gather_trajectories) -> torch's nccl -> gloo, agreed between the ranks
        allp, gather_backend, gather_t = gather_trajectories_auto(local, dev_index)
        gather_ms = gather_t.get("gather_ms")
        gather_total_ms = (time.perf_counter() - g0) * 1e3  # incl. communicator set-up and transports that were refused
        gather_ranks = dist.get_world_size()
        assert allp.shape == (gather_ranks * n_pairs, 14)
        mine = allp[rank * n_pairs:(rank + 1) * n_pairs]
        assert np.array_equal(mine, local)  # every rank finds its own block where its rank says
        from slam_amd.dist import gather_verdict
        verdict = gather_verdict(gather_backend, world, bool(gather_t.get("distinct_gpus")))

    log("results read, gather done")
    ok = sum(r["status"] == 1 for r in res)
    err = max(np.abs(r["pose"] - T_gt[i]).max() for i, r in enumerate(res) if r["status"] == 1) if ok else float("nan")
    kp = [sum(ctx.level_info(int(s), l)["num_keypoints"] for l in range(3)) for s in pipe.slots_of(last)[::37]]
    stats = dict(n_frames=2 * n_pairs, n_pairs=n_pairs, mean_kp_per_frame=float(np.mean(kp)),
                 mean_matches=float(np.mean([r["num_matches"] for r in res])),
                 mean_inliers=float(np.mean([r["num_inliers"] for r in res])))

    if rank == 0:
        total_pairs = world * n_pairs * a.steps
        value = total_pairs / dt
        geom = dict(W=640, H=480, levels=3, total_buckets=64, max_per=25)
        B = algorithmic_bytes(geom, stats)  # per launch if the whole batch is one launch
        for k in list(B):  # the batch runs as stream chunks: scale to what ONE launch of the kernel processes
            if k in prof:
                base = (geom["levels"] - 1) if k == "pyr_down" else 1
                B[k] = B[k] / max(1.0, prof[k]["calls"] / (post_steps * base))
        roof = None
        if dom and dom in B and dom in prof_dom:
            avg_s = prof_dom[dom]["total_ms"] / prof_dom[dom]["calls"] * 1e-3
            ach = B[dom] / avg_s
            # HBM traffic per launch: not measurable from inside this process; taken from the committed rocprofv3
            # --pmc passes of this same command (profiles/rNN_pmc_traffic.json), valid for the default launch shape
            traffic, traffic_src = None, None
            tp = latest_profile("pmc_traffic.json")
            if os.path.exists(tp) and n_pairs == 256 and n_ctx == 1 and not os.environ.get("RVO_PAIR_CHUNKS"):
                tj = json.load(open(tp))
                if dom in tj["kernels"]:
                    traffic = tj["kernels"][dom]["traffic_bytes"]
                    traffic_src = f"profiles/{os.path.basename(tp)}: (2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch, separate --pmc passes"
            roof = dict(bound="hbm", kernel=dom, achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                        frac=ach / HBM_PEAK, traffic=traffic, traffic_source=traffic_src, avg_launch_ms=avg_s * 1e3,
                        launches_timed=prof_dom[dom]["calls"],
                        algorithmic_bytes_per_launch=B[dom],
                        note="launch = one whole batch (512 frames / 256 pairs); the launches of four batches are in flight at a time, "
                             "so a launch's duration includes contention (0.285 ms alone); the kernel is VALU-bound, not "
                             "HBM-bound (DESIGN.md section 5); dominant = largest VALU issue time per batch")
        # VALU issue roofline (the bound that actually holds for this path): issue time per batch from
        # the committed SQ counters of this same command (a --pmc pass of its own) against the live batch time
        roof_valu = None
        sp = latest_profile("sq_counters.csv")
        if os.path.exists(sp) and n_pairs == 256 and n_ctx == 1 and not os.environ.get("RVO_PAIR_CHUNKS"):
            import csv
            issue_ms = sum(float(r["valu_issue_ms_at_2.4GHz"]) * int(r["launches_per_batch"]) for r in csv.DictReader(open(sp)))
            roof_valu = dict(bound="valu-issue", issue_ms_per_batch=issue_ms, frac=issue_ms / (dt / a.steps * 1e3),
                             source=f"profiles/{os.path.basename(sp)}: SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz per launch, "
                                    "summed over the launches of one batch")
        kernels = {k: dict(avg_ms=v["total_ms"] / v["calls"], calls=v["calls"],
                           GBps=(B[k] / (v["total_ms"] / v["calls"] * 1e-3) / 1e9) if k in B else None)
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        # whole-path roofline with SURVEY 8(d)'s per-pair figure evaluated at the measured K and N
        K, N = stats["mean_kp_per_frame"], stats["mean_matches"]
        b_frame = 640 * 480 + 2 * (320 * 240 + 160 * 120) + 128 * K + 4 * K
        b_pair = 2 * b_frame + 160 * K + N * N / 4
        cpu = None
        cpu_multi = None
        if world == 1 and not a.no_cpu:
            from oracle.pyoracle import Oracle, default_params as odef
            orc = Oracle(intr, odef(frame_matcher=True))
            m = min(a.cpu_pairs, n_pairs)
            c0 = time.perf_counter()
            same = 0
            for rep in range(a.cpu_reps):
                for i in range(m):
                    orc.prepare(0, gray[2 * i], depth[2 * i], thr)
                    orc.prepare(1, gray[2 * i + 1], depth[2 * i + 1], thr)
                    ro = orc.match_pair(0, 1)
                    if rep == 0:
                        same += (ro["num_matches"] == res[i]["num_matches"] and ro["num_inliers"] == res[i]["num_inliers"]
                                 and ro["status"] == res[i]["status"])
            cdt = time.perf_counter() - c0
            # the same oracle on the host cores this process may use (independent pairs, one Oracle per thread; the
            # library calls release the GIL): SURVEY 8(d)(ii)'s "N independent streams on N cores" leg
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
            oracles = [Oracle(intr, odef(frame_matcher=True)) for _ in range(nthr)]

            mreps = 8 * a.cpu_reps  # about as much wall time as the single-thread leg

            def cpu_worker(t):
                o = oracles[t]
                for rep in range(mreps):
                    for i in range(t, m, nthr):
                        o.prepare(0, gray[2 * i], depth[2 * i], thr)
                        o.prepare(1, gray[2 * i + 1], depth[2 * i + 1], thr)
                        o.match_pair(0, 1)
            m0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(cpu_worker, range(nthr)))
            mdt = time.perf_counter() - m0
            cpu_multi = dict(value=m * mreps / mdt, unit="frame-pairs/s", cores=nthr, kind="port",
                             sample=f"the same {m} pairs x {mreps} passes split over {nthr} threads ({mdt:.1f} s), one oracle instance per thread")
            cpu = dict(value=m * a.cpu_reps / cdt, unit="frame-pairs/s", cores=1, kind="port",
                       sample=f"first {m} of the same {n_pairs} pairs x {a.cpu_reps} passes ({cdt:.1f} s), oracle/librvo_oracle.so single thread; "
                              f"{same}/{m} pairs have identical match/inlier counts and status to the GPU result")
        cpu_seq = None
        if seq is not None:
            # configs[0]: the reference's own CPU-runnable case -- RGBDVO::ProcessFrame over one 300-frame stream (100
            # warm-up frames that only adapt the FAST threshold, vo.cpp:58-69, then one EstimateMotion per frame), on the
            # oracle with one core, and the same stream through the library (rvo_stream_step, one stream, host images in)
            from oracle.pyoracle import Oracle, OracleVO, default_params as odef
            from slam_amd.api import RGBDVOStreams
            log("cpu sequence leg")
            sg, sd, sposes = seq
            so = Oracle(intr, odef())
            vo = OracleVO(so)
            s0 = time.perf_counter()
            ro_seq = [vo.process(sg[i], sd[i]) for i in range(len(sg))]
            sdt = time.perf_counter() - s0
            vo.close()
            so.close()
            gs = RGBDVOStreams(intr, 1, device=dev_index)
            g0 = time.perf_counter()
            rg_seq = [gs.ProcessFrame(sg[i:i + 1], sd[i:i + 1])[0] for i in range(len(sg))]
            gdt = time.perf_counter() - g0
            gs.ctx.close()
            same_seq = sum(int(a_["status"] == b_["status"] and a_["num_inliers"] == b_["num_inliers"] and
                               a_["num_matches"] == b_["num_matches"] and a_["keyframe_changed"] == b_["keyframe_changed"])
                           for a_, b_ in zip(ro_seq, rg_seq))
            dpose = max(float(np.abs(a_["pose"] - b_["pose"]).max()) for a_, b_ in zip(ro_seq, rg_seq))
            k0 = next(i for i, r_ in enumerate(ro_seq) if r_["rc"] == 1)  # first frame after the warm-up: pose = Identity there
            terr = float(np.abs(ro_seq[-1]["pose"][:, 3] - (np.linalg.inv(sposes[k0]) @ sposes[-1])[:3, 3]).max())
            cpu_seq = dict(value=len(sg) / sdt, unit="frames/s", cores=1, kind="port",
                           sample=f"configs[0]: one {len(sg)}-frame synthetic 640x480 stream through OracleVO (vo.cpp:55-165), "
                                  f"{sdt:.1f} s, single thread",
                           estimates_success=sum(r_["status"] == 1 for r_ in ro_seq),
                           gpu_same_stream_frames_per_s=len(sg) / gdt,
                           gpu_note="the same stream alone through rvo_stream_step (host images in, one frame per call): "
                                    "latency-bound, the GPU is built for many streams (configs[2])",
                           frames_identical_to_gpu=same_seq, max_abs_pose_diff_vs_gpu=dpose,
                           drift_after_all_frames_m=terr)
        legs = {}
        if legs_in is not None:
            # BASELINE.json configs[2..4] as driver-run numbers, after the timed region, on the same GPU;
            # the configs[1] contexts are closed first (their work areas are not needed any more)
            from slam_amd.api import default_params as dparams
            for c in ctxs:
                c.close()
            ctxs = []
            legs["configs2_per_gpu"] = run_stream_leg(legs_in["seqs"], dev_index, steps=60, warmup=6, log=log)
            g3, d3, T3 = legs_in["big"]
            legs["configs3"] = run_pair_leg("configs3", synth.default_intrinsics(1280, 960), dparams(frame_matcher=True, num_pyramid_levels=4),
                                            g3, d3, T3, thr, dev_index, steps=40, warmup=5, log=log, oracle_pairs=2 if not a.no_cpu else 0)
            legs["configs3"]["workload"] = "configs[3]: 1280x960, 4 pyramid levels, 32 independent pairs per batch, device-resident inputs"
            g4, d4, T4 = legs_in["adv"]
            legs["configs4"] = run_pair_leg("configs4", intr, dparams(frame_matcher=True), g4, d4, T4, 5, dev_index, steps=40, warmup=5,
                                            log=log, oracle_pairs=8 if not a.no_cpu else 0)
            legs["configs4"]["workload"] = ("configs[4]: adversarial 640x480 pairs (texture contrast 0.12 = about 5 grey levels, 30 % of the depth "
                                            "invalid in blobs and salt noise, FAST threshold at its floor 5), 256 pairs per batch")
            legs["configs2_per_gpu"]["workload"] = ("configs[2] per-GPU shard: 512 lock-step 640x480 streams through rvo_stream_step_dev on one "
                                                    "context, 16 distinct sequences replayed through rvo_streams_set_source_index, device-resident inputs, "
                                                    "depth sampled in place (rvo_streams_depth_in_place)")
        line = dict(metric="frame-pairs/sec (whole node) at 640x480", value=value, unit="frame-pairs/s",
                    n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=dt / a.steps * 1e3,
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8/int32/f64", data="synthetic",
                    config=dict(workload=f"configs[1]: {n_pairs} independent synthetic 640x480 gray+depth frame pairs "
                                         f"per GPU, 3-level pyramid, default feature grid (80x80x25), device-resident inputs",
                                pairs_per_gpu=n_pairs, contexts=n_ctx, pipeline="staged (4 streams, 4 work-area sets, 5 frame-slot sets)" if n_ctx == 1 else f"{n_ctx} contexts, alternate batches",
                                parallelism=f"dp{world} (independent pairs, no collective in loop)"),
                    roofline=roof,
                    roofline_path=dict(bytes_per_pair=b_pair, frac=value / world * b_pair / HBM_PEAK,
                                       note="SURVEY 8(d) B_pair at the measured K and N, per GPU"),
                    roofline_valu=roof_valu, cpu_baseline=cpu, cpu_baseline_all_cores=cpu_multi, cpu_baseline_sequence=cpu_seq, kernels=kernels,
                    kernels_steps=post_steps,
                    kernels_note=f"separate pass of {post_steps} steps after the timed region with HIP events around every launch "
                                 "(that instrumentation costs about 7 %); the timed region brackets only the dominant kernel", stats=stats,
                    check=dict(pairs_success=ok, max_abs_pose_err_vs_truth=float(err)),
                    gather_ms=gather_ms, gather_total_ms=gather_total_ms, gather_backend=gather_backend, gather_ranks=gather_ranks,
                    gather_hung=verdict["gather_hung"], gather_ok=verdict["gather_ok"], gather_over_rccl=verdict["gather_over_rccl"],
                    **legs)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        # a hung collective, or a gather that bypassed RCCL although every rank has a GPU of its own, must not look green:
        # the line above is out, the exit code says what happened (slam_amd/dist.py: EXIT_GATHER_HUNG / _NOT_RCCL)
        from slam_amd import dist as rdist
        if rdist.HUNG:
            rdist.finish_after_gather(verdict["exit_code"])
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()
        if verdict["exit_code"]:
            log(f"gather verdict: {verdict}")
            sys.exit(verdict["exit_code"])


if __name__ == "__main__":
    main()
