"""Helper of tests/test_cpu_launch.py: a script with bench.py's launch contract (`--gpus N` starts N ranks; rank 0
prints ONE JSON line).  The ranks meet over gloo and all_gather their rank numbers."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from slam_amd.launch import launched_by_parent, reserve_stdout, spawn_ranks  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--fail-rank", type=int, default=-1)
    a = ap.parse_args()
    if a.gpus > 1 and not launched_by_parent():
        sys.exit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus, timeout_s=120))
    json_fd = reserve_stdout()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if rank == a.fail_rank:
        sys.exit(7)
    ranks = [rank]
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", timeout=__import__("datetime").timedelta(seconds=60))
        t = torch.tensor([rank], dtype=torch.int64)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        ranks = [int(o.item()) for o in out]
        dist.destroy_process_group()
    print("noise from rank", rank, file=sys.stderr)
    if rank != 0:
        os.write(json_fd, b"stdout of a non-zero rank must not reach the parent's stdout\n")
    if rank == 0:
        os.write(json_fd, (json.dumps(dict(n_gpus=world, ranks=ranks, local_rank=int(os.environ.get("LOCAL_RANK", "0")))) + "\n").encode())


if __name__ == "__main__":
    main()
