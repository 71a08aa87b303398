"""configs[2] end to end: S independent RGB-D streams sharded over the ranks of one node (one process per GPU,
slam_amd.dist.shard_range), every rank advancing its streams in lock-step with rvo_stream_step_dev on one context (the library
interleaves two groups of streams itself; --contexts 2 adds round 1's two-context arrangement on top), no collective in the loop; at the end ONE gather of the trajectories (RCCL, gloo fallback) and rank 0 writes them in
the pose graph's vertex file format (slam_amd/vertex_io.py) and prints one JSON line.

  python tools/run_streams.py --streams 64 --frames 12
  python tools/run_streams.py --gpus 8 --streams 4096 --frames 300      (starts its own 8 ranks, slam_amd/launch.py)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_streams.py --streams 4096 --frames 300
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import synth

INTR = synth.default_intrinsics(640, 480)
UNIQUE = 16  # distinct synthetic sequences; stream s replays sequence s % UNIQUE


def _seq(args):
    seed, T = args
    return [(g, d, P) for g, d, P in synth.make_sequence(T, seed=seed, intr=INTR, scene=synth.PlaneScene(seed=seed % 8))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--out", default="")
    ap.add_argument("--contexts", type=int, default=1, choices=(1, 2), help="contexts per GPU (2: streams split over two contexts with rvo_wait_peer)")
    ap.add_argument("--gpus", type=int, default=1, help="started without torchrun: launch this many rank processes (one per GPU)")
    a = ap.parse_args()
    from slam_amd.launch import launched_by_parent, spawn_ranks
    if a.gpus > 1 and not launched_by_parent():
        sys.exit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from concurrent.futures import ProcessPoolExecutor
    synth.make_texture()
    with ProcessPoolExecutor(8) as ex:
        seqs = list(ex.map(_seq, [(s, a.frames) for s in range(UNIQUE)]))

    import torch
    import torch.distributed as dist
    from slam_amd.api import Context, default_params
    from slam_amd.dist import shard_range
    from slam_amd import vertex_io
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if world > 1:
        if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # RCCL bootstrap only; the data path is xGMI
        dist.init_process_group("cpu:gloo,cuda:nccl", timeout=__import__("datetime").timedelta(seconds=180))
    mine = shard_range(a.streams, rank, world)
    S = len(mine)
    # The UNIQUE sequences live on the device once ([T, UNIQUE, H, W]: 1.5 + 2.9 GB at 300 frames, whatever the number of
    # streams); stream s reads image s % UNIQUE of step t's block through rvo_streams_set_source_index -- no
    # [T, S, H, W] array exists anywhere (at 512 streams x 300 frames it would be 141 GB per rank).
    ug = torch.from_numpy(np.stack([np.stack([seqs[u][t][0] for u in range(UNIQUE)]) for t in range(a.frames)])).cuda()
    ud = torch.from_numpy(np.stack([np.stack([seqs[u][t][1] for u in range(UNIQUE)]) for t in range(a.frames)]).view(np.int16)).cuda()
    torch.cuda.synchronize()
    p = default_params()
    p.warmup_frames = 1  # the reference discards 99 frames first (vo.cpp:58-69); the synthetic sequences start at once
    halves = [range(0, S // 2), range(S // 2, S)] if (S >= 2 and a.contexts == 2) else [range(0, S)]
    ctxs = []
    for h in halves:
        c = Context(INTR, p, max_frames=3 * len(h), max_pairs=len(h), device=dev_index)
        c.streams_reset(len(h))
        c.streams_set_source_index(UNIQUE, [mine[k] % UNIQUE for k in h])
        c.streams_depth_in_place(True)  # all frames of the run are resident: depth is sampled where it lies
        c.set_pair_chunks(1)
        ctxs.append(c)
    for c, h in zip(ctxs, halves):
        c.streams_record(a.frames)  # trajectories stay on the device: no host read-back (and no sync) inside the loop
    t0 = time.perf_counter()
    for t in range(a.frames):
        for k, (c, h) in enumerate(zip(ctxs, halves)):
            if len(ctxs) > 1:
                c.wait_peer(ctxs[1 - k])
            c.stream_step_dev(ug[t].data_ptr(), ud[t].data_ptr(), len(h))
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0

    def read_local():  # [S, frames, 16] on the host (the fall-back transports and the world of one)
        return np.concatenate([c.trajectories(len(h), a.frames) for c, h in zip(ctxs, halves)])
    backend, verdict = None, dict(gather_hung=False, gather_ok=True, gather_over_rccl=False, exit_code=0)
    if world > 1:
        from slam_amd.dist import gather_trajectories_auto, gather_verdict
        # one context: its device-side trajectory buffer IS the [S, frames, 16] block -> gathered where it lies
        # (rvo_trajectories_dev -> rvo_gather_trajectories), the result read back once
        local_dev = None
        if len(ctxs) == 1:
            ptr, rec, cap = ctxs[0].trajectories_dev()
            if rec == cap == a.frames:
                local_dev = (ptr, S * a.frames * 16 * 8)
        g0 = time.perf_counter()
        allt, backend, ginfo = gather_trajectories_auto(read_local, dev_index, local_dev=local_dev, local_shape=(S, a.frames, 16))
        gather_s = time.perf_counter() - g0
        verdict = gather_verdict(backend, world, bool(ginfo.get("distinct_gpus")))
        traj = allt[mine.start:mine.stop] if len(set(len(shard_range(a.streams, r, world)) for r in range(world))) == 1 else read_local()
    else:
        allt = traj = read_local()
        gather_s = 0.0
    # accuracy against the synthetic ground truth: pose_t maps frame t into frame 0
    err = 0.0
    for k, s in enumerate(mine):
        T0 = seqs[s % UNIQUE][0][2]
        Tn = seqs[s % UNIQUE][a.frames - 1][2]
        gt = np.linalg.inv(T0) @ Tn
        err = max(err, float(np.abs(traj[k, -1, :12].reshape(3, 4) - gt[:3]).max()))
    import resource
    rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6  # KiB -> GB
    if world > 1:
        stats = torch.tensor([dt, err, rss_gb], dtype=torch.float64)
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dt, err, rss_gb = float(stats[0]), float(stats[1]), float(stats[2])
    if rank == 0:
        assert allt.shape == (a.streams, a.frames, 16)
        if a.out:
            os.makedirs(a.out, exist_ok=True)
            for s in range(min(a.streams, 8)):
                vertex_io.save_vertices(os.path.join(a.out, f"stream_{s:04d}.vertices"), np.arange(a.frames), allt[s, :, :12].reshape(-1, 3, 4))
            np.save(os.path.join(a.out, "trajectories_first8.npy"), allt[:8])
        ok = int((allt[:, 1:, 12] == 1).sum())
        print(json.dumps(dict(streams=a.streams, frames=a.frames, n_gpus=world, stream_frames_per_s=a.streams * a.frames / dt,
                              estimates_ok=ok, estimates=a.streams * (a.frames - 1), max_abs_pose_err_vs_truth=err,
                              gather_backend=backend, gather_ranks=world if world > 1 else None, gather_s=gather_s,
                              gather_hung=verdict["gather_hung"], gather_ok=verdict["gather_ok"],
                              peak_host_rss_gb=rss_gb, unique_sequences=UNIQUE)), flush=True)
    if world > 1:
        from slam_amd import dist as rdist
        if rdist.HUNG:  # the line is out; a helper thread still blocks inside RCCL: leave with the failure code
            rdist.finish_after_gather(verdict["exit_code"])
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()
        if verdict["exit_code"]:
            sys.exit(verdict["exit_code"])


if __name__ == "__main__":
    main()
