"""On-disk trajectory format of the reference's pose graph.

PoseGraph::SaveVertices / SaveVertexById / LoadVertices / LoadVertexById (pose_graph.cpp:147-199) write and read raw
`struct Vertex { uint32_t id; Eigen::Matrix4d pose; }` records (pose_graph.hpp:43-57).  Eigen::Matrix4d is a
16-byte-aligned fixed-size type (Eigen 3.2 on x86-64, the version DESIGN.md assumes), so the record is
  offset 0   uint32 id
  offset 4   12 bytes of padding (whatever the writer's stack held; written as zeros here, ignored on read)
  offset 16  16 doubles, COLUMN-major 4x4
= 144 bytes, little-endian, no header, records in vertex-map order.  Gathered trajectories written with
save_vertices() are read by the reference's own LoadVertices and vice versa.  Host code only; parity unpinned (the
reference ships no vertex file).
"""
from __future__ import annotations

import numpy as np

VERTEX_DTYPE = np.dtype({"names": ["id", "pose"], "formats": ["<u4", ("<f8", (16,))], "offsets": [0, 16], "itemsize": 144})


def poses_to_records(ids, poses) -> np.ndarray:
    """ids [n]; poses [n,3,4] or [n,4,4] row-major (what rvo results hold) -> Vertex records."""
    ids = np.asarray(ids, np.uint32)
    poses = np.asarray(poses, np.float64)
    n = len(ids)
    m = np.tile(np.eye(4), (n, 1, 1))
    m[:, : poses.shape[1], :] = poses
    rec = np.zeros(n, VERTEX_DTYPE)
    rec["id"] = ids
    rec["pose"] = m.transpose(0, 2, 1).reshape(n, 16)  # column-major
    return rec


def save_vertices(path: str, ids, poses, append: bool = False) -> None:
    """SaveVertices (pose_graph.cpp:174-183); append=True is SaveVertexById (pose_graph.cpp:147-154)."""
    with open(path, "ab" if append else "wb") as f:
        f.write(poses_to_records(ids, poses).tobytes())


def load_vertices(path: str):
    """LoadVertices (pose_graph.cpp:185-199): -> (ids [n] uint32, poses [n,4,4] row-major)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size % VERTEX_DTYPE.itemsize:
        raise ValueError(f"{path}: size {raw.size} is not a multiple of sizeof(Vertex) = {VERTEX_DTYPE.itemsize}")
    rec = raw.view(VERTEX_DTYPE)
    return rec["id"].copy(), rec["pose"].reshape(-1, 4, 4).transpose(0, 2, 1).copy()


def load_vertex_by_id(path: str, vid: int):
    """LoadVertexById (pose_graph.cpp:156-172): first record with that id, or None."""
    ids, poses = load_vertices(path)
    hit = np.nonzero(ids == vid)[0]
    return None if len(hit) == 0 else poses[hit[0]]
