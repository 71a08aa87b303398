"""Multi-GPU plumbing: one process per GPU, independent streams/pairs sharded over ranks, and the one
collective of the path -- the final trajectory gather.  The reference has no
counterpart (it is single-process); torch.distributed is used only as the RCCL/gloo front-end.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_units: int, rank: int, world: int) -> range:
    """Block distribution of n_units independent streams/pairs over `world` ranks (remainder to the first ranks)."""
    base, rem = divmod(n_units, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_trajectories(local: np.ndarray, device=None) -> np.ndarray:
    """all_gather of per-rank [units, ...] float64 arrays (poses / statuses) -> [sum(units), ...] on every rank.

    Shards may be ragged: sizes are exchanged first, then the padded blocks (one all_gather each, nothing in
    the inner loop).  backend nccl (= RCCL over xGMI) when a process group with GPUs exists, gloo on CPU.
    """
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, np.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.copy()
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + local.shape[1:], dtype=torch.float64, device=dev)
    pad[: local.shape[0]] = torch.from_numpy(local).to(dev)
    out = torch.empty((world * m,) + local.shape[1:], dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, pad)
    out = out.cpu().numpy().reshape((world, m) + local.shape[1:])
    return np.concatenate([out[r, : sizes[r]] for r in range(world)], axis=0)


class RcclComm:
    """The C ABI's communicator (include/rvo.h: rvo_comm_*): ncclAllGather over RCCL / xGMI without torch in the data
    path.  A world of one never opens librccl."""

    def __init__(self, device: int, rank: int, world: int, unique_id: bytes | None):
        import ctypes as C
        from . import _lib
        self.L = _lib.load()
        self.rank, self.world, self.device = rank, world, device
        h = C.c_void_p()
        idb = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id) if unique_id is not None else None
        rc = self.L.rvo_comm_create(device, rank, world, idb, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"rvo_comm_create: {rc}: {self.L.rvo_last_error().decode()}")
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        L = _lib.load()
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        rc = L.rvo_comm_get_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"rvo_comm_get_unique_id: {rc}: {L.rvo_last_error().decode()}")
        return bytes(buf)

    def all_gather_dev(self, local_ptr: int, bytes_per_rank: int, all_ptr: int):
        import ctypes as C
        rc = self.L.rvo_gather_trajectories(self.h, C.c_void_p(local_ptr), bytes_per_rank, C.c_void_p(all_ptr))
        if rc != 0:
            raise RuntimeError(f"rvo_gather_trajectories: {rc}: {self.L.rvo_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.rvo_comm_destroy(self.h)
            self.h = None


def draw_rccl_id():
    """Collective on the control plane (torch.distributed, main thread only): rank 0 draws the RCCL unique id and hands
    it round.  Returns (id bytes | None, error text): every rank gets the same answer, so all go on or none does."""
    import torch.distributed as dist
    box = [None, ""]
    if dist.get_rank() == 0:
        try:
            box[0] = RcclComm.unique_id()
        except Exception as e:  # noqa: BLE001 -- e.g. no loadable librccl: RVO_E_UNSUPPORTED
            box[1] = f"{type(e).__name__}: {e}"
    dist.broadcast_object_list(box, src=0)
    return box[0], box[1]


def make_rccl_comm(device_index: int) -> RcclComm:
    """Collective: rank 0 draws the RCCL unique id, the control plane (torch.distributed, gloo) hands it round, every
    rank joins.  Without a process group: a world of one."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return RcclComm(device_index, 0, 1, None)
    uid, err = draw_rccl_id()
    if uid is None:
        raise RuntimeError(f"rvo_comm_get_unique_id on rank 0: {err}")
    return RcclComm(device_index, dist.get_rank(), dist.get_world_size(), uid)


def gather_trajectories_rccl(comm: RcclComm, local_dev, out_dev=None):
    """all_gather of equal-sized per-rank device tensors (torch tensors only as device memory) through the C ABI;
    returns the [world * units, ...] device tensor."""
    import torch
    local_dev = local_dev.contiguous()
    if out_dev is None:
        out_dev = torch.empty((comm.world * local_dev.shape[0],) + tuple(local_dev.shape[1:]), dtype=local_dev.dtype,
                              device=local_dev.device)
    torch.cuda.synchronize(local_dev.device)
    comm.all_gather_dev(local_dev.data_ptr(), local_dev.numel() * local_dev.element_size(), out_dev.data_ptr())
    return out_dev


HUNG = False  # a collective attempt timed out in this process: its helper thread still blocks inside RCCL and cannot be
#               joined, so the process must end through finish_after_gather() (os._exit with a NON-ZERO code)


def _with_timeout(fn, timeout_s: float):
    """fn() in a helper thread; TimeoutError if it has not returned in time (RCCL calls block for ever when a peer is missing)."""
    import threading
    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as e:  # noqa: BLE001 -- handed to the caller
            box["e"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        global HUNG
        HUNG = True
        raise TimeoutError(f"no answer within {timeout_s:.0f} s")
    if "e" in box:
        raise box["e"]
    return box["v"]


def _all_agree(ok: bool, device_index: int | None = None) -> bool:
    """True iff every rank says ok: one tiny all_reduce on the control plane (a CPU tensor -> gloo); a process group
    without a CPU backend (nccl only) gets an object collective instead.  Main thread only."""
    import torch
    import torch.distributed as dist
    try:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())
    except RuntimeError:
        votes = [None] * dist.get_world_size()
        dist.all_gather_object(votes, bool(ok))
        return all(votes)


def gather_trajectories_auto(local, device_index: int, timeout_s: float = 60.0, local_dev=None, local_shape=None):
    """The final trajectory gather with the best transport that works on this node, agreed between the ranks after
    every attempt so that nobody waits in a collective the others gave up on:
      1. the C ABI's RCCL communicator (rvo_gather_trajectories: ncclAllGather over xGMI, equal-sized shards),
      2. torch.distributed's nccl backend (= RCCL), if the process group has one,
      3. gloo (ranks sharing one GPU in rehearsals: RCCL refuses two ranks on a device).
    local: this rank's [units, ...] float64 array, or a callable that returns it (read only if a transport needs host
    data).  local_dev = (device pointer, bytes) of the same block already in device memory, complete (the producing
    context synchronised): leg 1 then gathers it where it lies, nothing goes through the host first; local_shape is
    the block's shape then.
    Returns (all [sum(units), ...] float64, backend string, dict: communicator set-up and gather times in ms, and
    distinct_gpus = every rank has a GPU of its own; pass backend and that flag to gather_verdict)."""
    import time
    import torch
    import torch.distributed as dist

    def host():
        nonlocal local
        if callable(local):
            local = local()
        local = np.ascontiguousarray(local, np.float64)
        return local

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return host().copy(), "none (world of one)", dict(distinct_gpus=True)
    world = dist.get_world_size()
    shape = tuple(local_shape) if local_shape is not None else tuple(host().shape)
    notes = []
    # shards must be equal-sized for ncclAllGather: agree on that first (ragged shards go through the padded torch path)
    n = torch.tensor([shape[0], 1 if local_dev is not None else 0], dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    equal = len({int(x[0].item()) for x in sizes}) == 1
    in_place = all(int(x[1].item()) for x in sizes)  # every rank's block is already in device memory
    # one node, one rank per GPU: RCCL refuses two ranks on one device (after a slow failed bootstrap), so a rehearsal in
    # which ranks share a card goes straight to gloo
    d = torch.tensor([device_index], dtype=torch.int64)
    devs = [torch.zeros_like(d) for _ in range(world)]
    dist.all_gather(devs, d)
    distinct = len({int(x.item()) for x in devs}) == world
    if not distinct:
        notes.append("ranks share a GPU: RCCL not attempted")
    if equal and distinct and not HUNG:
        # Only the RCCL calls themselves (ncclCommInitRank, ncclAllGather: they block for ever when a peer is missing)
        # run in a helper thread with a time limit; every torch.distributed collective stays on this thread, so the
        # ranks' control-plane collectives can never be matched out of order by a helper that is still running.
        t0 = time.perf_counter()
        uid, err = draw_rccl_id()
        comm = None
        ok = uid is not None
        if ok:
            rank = dist.get_rank()
            try:
                comm = _with_timeout(lambda: RcclComm(device_index, rank, world, uid), timeout_s)
            except Exception as e:  # noqa: BLE001
                ok, err = False, f"{type(e).__name__}: {e}"
        out = None
        if _all_agree(ok):  # every rank holds a communicator
            dev = torch.device("cuda", device_index)
            t1 = time.perf_counter()
            try:
                if in_place:  # the device block as the context left it -> ncclAllGather -> one read-back of the result
                    out_dev = torch.empty((world * shape[0],) + shape[1:], dtype=torch.float64, device=dev)
                    torch.cuda.synchronize(dev)

                    def run():
                        comm.all_gather_dev(local_dev[0], local_dev[1], out_dev.data_ptr())
                        return out_dev.cpu().numpy()
                else:
                    src = torch.from_numpy(host()).to(dev)

                    def run():
                        return gather_trajectories_rccl(comm, src).cpu().numpy()
                out = _with_timeout(run, timeout_s)
                ok2, err = True, ""
            except Exception as e:  # noqa: BLE001
                ok2, err = False, f"{type(e).__name__}: {e}"
            ok2 = _all_agree(ok2)
            t2 = time.perf_counter()
            if not HUNG:  # a helper still inside RCCL holds the communicator: leave it to the process's end
                comm.close()
            if ok2:
                return out, "rccl (C ABI rvo_gather_trajectories, ncclAllGather" + (", device block in place)" if in_place else ")"), dict(comm_init_ms=(t1 - t0) * 1e3, gather_ms=(t2 - t1) * 1e3, distinct_gpus=distinct)
        elif comm is not None:
            comm.close()
        notes.append(f"C-ABI rccl: {err[:160] or 'failed on another rank'}")
    has_nccl = False
    try:
        has_nccl = "nccl" in str(dist.get_backend_config())
    except Exception:  # noqa: BLE001
        pass
    if has_nccl and distinct and not HUNG:
        t1 = time.perf_counter()
        try:
            h = host()
            out = _with_timeout(lambda: gather_trajectories(h, torch.device("cuda", device_index)), timeout_s)
            ok, err = True, ""
        except Exception as e:  # noqa: BLE001
            ok, err, out = False, f"{type(e).__name__}: {e}", None
        if _all_agree(ok):
            return out, "nccl (torch.distributed)" + (" after " + "; ".join(notes) if notes else ""), dict(gather_ms=(time.perf_counter() - t1) * 1e3, distinct_gpus=distinct)
        notes.append(f"torch nccl: {err[:120] or 'failed on another rank'}")
    t1 = time.perf_counter()
    out = gather_trajectories(host(), None)
    return out, "gloo (" + "; ".join(notes) + ")", dict(gather_ms=(time.perf_counter() - t1) * 1e3, distinct_gpus=distinct)


EXIT_GATHER_HUNG = 3      # a collective attempt timed out (the JSON line, with "gather_hung": true, is already out)
EXIT_GATHER_NOT_RCCL = 4  # every rank had a GPU of its own and the gather still did not go over RCCL


def gather_verdict(backend: str | None, world: int, distinct_gpus: bool) -> dict:
    """Fields for the JSON line + the exit code a multi-rank run must end with: a dead or bypassed RCCL must not look
    green.  Ranks that share a GPU (rehearsals on a one-GPU box) cannot use RCCL and are allowed gloo."""
    over_rccl = bool(backend) and (backend.startswith("rccl") or backend.startswith("nccl"))
    code = 0
    if world > 1 and HUNG:
        code = EXIT_GATHER_HUNG
    elif world > 1 and distinct_gpus and not over_rccl:
        code = EXIT_GATHER_NOT_RCCL
    return dict(gather_hung=bool(HUNG), gather_ok=(world == 1 or (not HUNG and (over_rccl or not distinct_gpus))),
                gather_over_rccl=over_rccl, exit_code=code)


def finish_after_gather(code: int):
    """End a rank after its JSON line is written.  With a helper thread stuck inside RCCL the interpreter cannot shut
    down normally (the thread never returns), hence os._exit -- with the failure code, never 0."""
    import os
    import sys
    if HUNG:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code or EXIT_GATHER_HUNG)
    if code:
        sys.exit(code)
