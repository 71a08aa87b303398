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


def make_rccl_comm(device_index: int) -> RcclComm:
    """Collective: rank 0 draws the RCCL unique id, the control plane (torch.distributed, gloo) hands it round, every
    rank joins.  Without a process group: a world of one."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return RcclComm(device_index, 0, 1, None)
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [RcclComm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return RcclComm(device_index, rank, world, box[0])


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


HUNG = False  # a collective attempt timed out in this process: finish with os._exit (a blocked helper thread cannot be joined)


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


def _all_agree(ok: bool) -> bool:
    """True iff every rank says ok (one tiny all_reduce on the control plane: CPU tensor -> gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def gather_trajectories_auto(local: np.ndarray, device_index: int, timeout_s: float = 60.0):
    """The final trajectory gather with the best transport that works on this node, agreed between the ranks after
    every attempt so that nobody waits in a collective the others gave up on:
      1. the C ABI's RCCL communicator (rvo_gather_trajectories: ncclAllGather over xGMI, equal-sized shards),
      2. torch.distributed's nccl backend (= RCCL), if the process group has one,
      3. gloo (ranks sharing one GPU in rehearsals: RCCL refuses two ranks on a device).
    Returns (all [sum(units), ...] float64, backend string, timings dict in ms: communicator set-up and the gather itself)."""
    import time
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, np.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.copy(), "none (world of one)", {}
    world = dist.get_world_size()
    notes = []
    # shards must be equal-sized for ncclAllGather: agree on that first (ragged shards go through the padded torch path)
    n = torch.tensor([local.shape[0]], dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    equal = len({int(x.item()) for x in sizes}) == 1
    # one node, one rank per GPU: RCCL refuses two ranks on one device (after a slow failed bootstrap), so a rehearsal in
    # which ranks share a card goes straight to gloo
    d = torch.tensor([device_index], dtype=torch.int64)
    devs = [torch.zeros_like(d) for _ in range(world)]
    dist.all_gather(devs, d)
    distinct = len({int(x.item()) for x in devs}) == world
    if not distinct:
        notes.append("ranks share a GPU: RCCL not attempted")
    if equal and distinct and not HUNG:
        state = {}

        def attempt():
            state["comm"] = make_rccl_comm(device_index)
            return True
        t0 = time.perf_counter()
        try:
            ok = _with_timeout(attempt, timeout_s)
            err = ""
        except Exception as e:  # noqa: BLE001
            ok, err = False, f"{type(e).__name__}: {e}"
        if _all_agree(ok):
            dev = torch.device("cuda", device_index)
            t1 = time.perf_counter()
            try:
                out = _with_timeout(lambda: gather_trajectories_rccl(state["comm"], torch.from_numpy(local).to(dev)).cpu().numpy(), timeout_s)
                ok2, err = True, ""
            except Exception as e:  # noqa: BLE001
                ok2, err, out = False, f"{type(e).__name__}: {e}", None
            if _all_agree(ok2):
                t2 = time.perf_counter()
                state["comm"].close()
                return out, "rccl (C ABI rvo_gather_trajectories, ncclAllGather)", dict(comm_init_ms=(t1 - t0) * 1e3, gather_ms=(t2 - t1) * 1e3)
        elif ok:
            state["comm"].close()
        notes.append(f"C-ABI rccl: {err[:120] or 'failed on another rank'}")
    has_nccl = False
    try:
        has_nccl = "nccl" in str(dist.get_backend_config())
    except Exception:  # noqa: BLE001
        pass
    if has_nccl and distinct and not HUNG:
        t1 = time.perf_counter()
        try:
            out = _with_timeout(lambda: gather_trajectories(local, torch.device("cuda", device_index)), timeout_s)
            ok, err = True, ""
        except Exception as e:  # noqa: BLE001
            ok, err, out = False, f"{type(e).__name__}: {e}", None
        if _all_agree(ok):
            return out, "nccl (torch.distributed)" + (" after " + "; ".join(notes) if notes else ""), dict(gather_ms=(time.perf_counter() - t1) * 1e3)
        notes.append(f"torch nccl: {err[:120] or 'failed on another rank'}")
    t1 = time.perf_counter()
    out = gather_trajectories(local, None)
    return out, "gloo (" + "; ".join(notes) + ")", dict(gather_ms=(time.perf_counter() - t1) * 1e3)
