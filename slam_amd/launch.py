"""One process per GPU without an outer torchrun: `script --gpus N` started as a plain `python script ...` re-launches
itself as N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, the same
variables torch.distributed.run sets) and relays rank 0's standard output.  The reference has no counterpart (it is a
single process, rgbdvo/vo.cpp:55); this is plumbing for BASELINE.json's "1, 2, 4 and 8 GPUs".

The parent never touches the GPU (children are fresh processes, nothing is exec'ed over a process that initialised HIP).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time


def launched_by_parent() -> bool:
    """True inside a rank process (ours or torch.distributed.run's)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def reserve_stdout() -> int:
    """Returns a private duplicate of fd 1 and points fd 1 at stderr: banners that libraries (gloo, RCCL, HIP) print
    to standard output then cannot get between the caller and its ONE JSON line (os.write(fd, line))."""
    sys.stdout.flush()
    fd = os.dup(1)
    os.dup2(2, 1)
    return fd


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(argv: list[str], n: int, timeout_s: float | None = None, extra_env: dict | None = None) -> int:
    """Run `python argv...` as n rank processes on this node; returns 0 iff every rank exited with 0.

    Rank 0's stdout is passed through to ours (the one JSON line); the other ranks' stdout goes to stderr.  If a rank
    fails, the others are terminated (exact PIDs) and the first non-zero exit code is returned.
    """
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=out))
    t0 = time.monotonic()
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (timeout_s is not None and time.monotonic() - t0 > timeout_s):
            if rc == 0:
                rc = 124
            for p in alive:
                p.terminate()
            for p in alive:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    return rc
