"""Helper: two ranks on ONE GPU try to build the C ABI's RCCL communicator.  RCCL refuses two ranks on one device; what
matters is that every rank gets an error code back (no hang, no abort) -- or, on a node with two GPUs, a working gather."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from slam_amd.launch import launched_by_parent, reserve_stdout, spawn_ranks  # noqa: E402


def main():
    if not launched_by_parent():
        sys.exit(spawn_ranks([os.path.abspath(__file__)], 2, timeout_s=150))
    fd = reserve_stdout()
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist
    from slam_amd.dist import gather_trajectories_rccl, make_rccl_comm
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    ndev = torch.cuda.device_count()
    dev = rank % max(1, ndev)
    torch.cuda.set_device(dev)
    out = dict(rank=rank, devices=ndev)
    try:
        comm = make_rccl_comm(dev)
        local = torch.full((4, 3), float(rank), dtype=torch.float64, device="cuda")
        allt = gather_trajectories_rccl(comm, local).cpu().numpy()
        out["gathered"] = allt[:, 0].tolist()
        comm.close()
    except RuntimeError as e:
        out["error"] = str(e)[:200]
    outs = [None] * world
    dist.all_gather_object(outs, out)
    if rank == 0:
        os.write(fd, (json.dumps(outs) + "\n").encode())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
