"""N > 1 path on CPU: world_size-2 gloo run of the sharding + trajectory gather used by bench.py."""
import os
import sys

import pytest
import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from slam_amd.dist import gather_trajectories, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_streams, T = 7, 5  # ragged: 4 + 3
    mine = shard_range(n_streams, rank, world)
    local = np.stack([np.arange(T * 12, dtype=np.float64).reshape(T, 12) + 1000 * s for s in mine])
    allp = gather_trajectories(local)
    q.put((rank, list(mine), allp))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = np.stack([np.arange(60, dtype=np.float64).reshape(5, 12) + 1000 * s for s in range(7)])
    shards = sorted(got)
    assert shards[0][1] + shards[1][1] == list(range(7))
    for _, _, allp in got:
        assert np.array_equal(allp, want)


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from slam_amd.dist import shard_range
    for n in (0, 1, 7, 4096):
        for w in (1, 2, 3, 8):
            idx = [i for r in range(w) for i in shard_range(n, r, w)]
            assert idx == list(range(n))


def test_vertex_file_format_roundtrip(tmp_path):
    """struct Vertex {uint32 id; Matrix4d pose} records (pose_graph.hpp:51-57): 144 bytes, id at 0, column-major pose at 16."""
    import struct
    import numpy as np
    from slam_amd import vertex_io
    rng = np.random.default_rng(3)
    poses = rng.normal(size=(5, 3, 4))
    ids = [0, 1, 2, 7, 9]
    p = str(tmp_path / "v.bin")
    vertex_io.save_vertices(p, ids[:3], poses[:3])
    vertex_io.save_vertices(p, ids[3:], poses[3:], append=True)  # SaveVertexById appends
    raw = open(p, "rb").read()
    assert len(raw) == 5 * 144
    # decode record 3 by hand, the way the C++ struct lies in memory
    rid, = struct.unpack_from("<I", raw, 3 * 144)
    col_major = struct.unpack_from("<16d", raw, 3 * 144 + 16)
    assert rid == 7
    assert col_major[4 * 3 + 1] == poses[3, 1, 3]      # element (row 1, col 3) at index col*4 + row
    assert col_major[3::4] == (0.0, 0.0, 0.0, 1.0)     # last row of the homogeneous matrix
    got_ids, got = vertex_io.load_vertices(p)
    assert got_ids.tolist() == ids and np.array_equal(got[:, :3], poses)
    assert np.array_equal(vertex_io.load_vertex_by_id(p, 9)[:3], poses[4]) and vertex_io.load_vertex_by_id(p, 5) is None


def test_vertex_file_cpp_and_python_agree(tmp_path):
    """include/rgbdvo/vertex_io.hpp (the C++ caller's side of pose_graph.cpp:147-199) writes what slam_amd/vertex_io.py
    reads, byte for byte, and reads what it writes."""
    import subprocess
    import numpy as np
    from slam_amd import vertex_io
    exe = str(tmp_path / "test_vertex_io")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "test_vertex_io.cpp")])
    n = 6
    want = np.empty((n, 3, 4))
    for i in range(n):
        for k in range(12):
            want[i].flat[k] = (i + 1) * 0.125 + k * 1.0 / 3.0 - (7.0 / (i + 2) if k % 4 == 3 else 0.0)
    ids = [3 * i + 1 for i in range(n)]
    p_cpp, p_py = str(tmp_path / "cpp.bin"), str(tmp_path / "py.bin")
    subprocess.check_call([exe, "write", p_cpp, str(n)])
    got_ids, got = vertex_io.load_vertices(p_cpp)
    assert got_ids.tolist() == ids and np.array_equal(got[:, :3], want) and np.array_equal(got[:, 3], np.tile([0, 0, 0, 1.0], (n, 1)))
    vertex_io.save_vertices(p_py, ids, want)
    assert open(p_py, "rb").read() == open(p_cpp, "rb").read()
    lines = subprocess.check_output([exe, "read", p_py], text=True).strip().splitlines()
    assert lines[-1] == "by_id 1 0"
    back = np.array([[float(x) for x in ln.split()[1:]] for ln in lines[:-1]]).reshape(n, 3, 4)
    assert [int(ln.split()[0]) for ln in lines[:-1]] == ids and np.array_equal(back, want)
    # a truncated file is refused by both sides
    open(p_py, "ab").write(b"\0" * 10)
    assert subprocess.call([exe, "read", p_py]) == 1
    with pytest.raises(ValueError):
        vertex_io.load_vertices(p_py)


def test_missing_rccl_is_an_error_code_not_a_crash():
    """A box without a loadable librccl: rvo_comm_get_unique_id and rvo_comm_create(world > 1) must come back with
    RVO_E_UNSUPPORTED and a message (the fall-back to torch's nccl / gloo in gather_trajectories_auto depends on it).
    Runs in a child process: the library caches its dlopen verdict.  No GPU needed -- the loader is asked before any device call."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "from slam_amd import _lib\n"
        "L = _lib.load()\n"
        "buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()\n"
        "rc = L.rvo_comm_get_unique_id(buf)\n"
        "msg = L.rvo_last_error().decode()\n"
        "print(rc, '|', msg)\n"
        "rc2 = L.rvo_comm_get_unique_id(buf)\n"
        "print(rc2, '|', L.rvo_last_error().decode())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RVO_RCCL_LIB="/nonexistent/librccl.so.1", RVO_HIP_RUNTIME="system")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    for line in lines[-2:]:
        rc, msg = line.split("|", 1)
        assert int(rc) == -2, line  # RVO_E_UNSUPPORTED
        assert "librccl not found" in msg and "/nonexistent/librccl.so.1" in msg, line
