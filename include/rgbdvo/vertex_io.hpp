// vertex_io.hpp -- the on-disk trajectory format of the reference's pose graph, for C++ callers of the facade
// (SURVEY.md section 8, row f4).  Replaces PoseGraph::SaveVertices / SaveVertexById / LoadVertices / LoadVertexById
// (pose_graph.cpp:147-199), which write and read raw `struct Vertex { VertexId id; Eigen::Matrix4d pose; }` records
// (pose_graph.hpp:43-57): with Eigen 3.2 on x86-64 the 4x4 double matrix is 16-byte aligned, so a record is
//   offset 0    uint32 id
//   offset 4    12 bytes of padding (the reference writes whatever its stack held; zeros here, ignored on read)
//   offset 16   16 doubles, column-major
// = 144 bytes, little-endian, no header.  Files written here are read by the reference's LoadVertices and the other
// way round; slam_amd/vertex_io.py is the Python side of the same format (tests/test_cpu_dist.py checks the two
// against each other byte for byte).  Host code only.  Parity unpinned: the reference ships no vertex file.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "eigen_compat.hpp"

namespace rgbdvo {

struct VertexRecord {
  uint32_t id;
  uint8_t pad[12];
  double pose[16];  // column-major 4x4
};
static_assert(sizeof(VertexRecord) == 144, "record layout of pose_graph.hpp:51-57");

// row-major 3x4 (what rvo_get_trajectories / rvo_gather_trajectories hold per frame) -> record; last row 0 0 0 1
inline VertexRecord MakeVertexRecord(uint32_t id, const double* pose3x4) {
  VertexRecord v;
  std::memset(&v, 0, sizeof(v));
  v.id = id;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) v.pose[c * 4 + r] = pose3x4[r * 4 + c];
  v.pose[15] = 1.0;
  return v;
}
inline VertexRecord MakeVertexRecord(uint32_t id, const Eigen::Isometry3d& pose) {
  VertexRecord v;
  std::memset(&v, 0, sizeof(v));
  v.id = id;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) v.pose[c * 4 + r] = pose.matrix()(r, c);
  return v;
}
inline Eigen::Isometry3d PoseOf(const VertexRecord& v) {
  Eigen::Isometry3d m = Eigen::Isometry3d::Identity();
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) m.matrix()(r, c) = v.pose[c * 4 + r];
  return m;
}

// SaveVertices (pose_graph.cpp:174-183) truncates; append = true is SaveVertexById (pose_graph.cpp:147-154).
// Returns false when the file cannot be opened or written (the reference exits the process there).
inline bool SaveVertices(const std::string& fn, const std::vector<VertexRecord>& vs, bool append = false) {
  std::FILE* f = std::fopen(fn.c_str(), append ? "ab" : "wb");
  if (!f) return false;
  const size_t n = vs.empty() ? 0 : std::fwrite(vs.data(), sizeof(VertexRecord), vs.size(), f);
  const bool ok = std::fclose(f) == 0 && n == vs.size();
  return ok;
}
inline bool SaveVertexById(const std::string& fn, const VertexRecord& v) { return SaveVertices(fn, {v}, true); }

// One stream's trajectory (n_frames row-major 3x4 poses, e.g. a row of rvo_gather_trajectories' output) as vertices
// first_id, first_id + 1, ...
inline bool SaveTrajectory(const std::string& fn, const double* poses3x4, int n_frames, uint32_t first_id = 0, bool append = false) {
  std::vector<VertexRecord> vs;
  vs.reserve(n_frames > 0 ? n_frames : 0);
  for (int i = 0; i < n_frames; i++) vs.push_back(MakeVertexRecord(first_id + (uint32_t)i, poses3x4 + (size_t)i * 12));
  return SaveVertices(fn, vs, append);
}

// LoadVertices (pose_graph.cpp:185-199).  The reference reads whole records until the file size is covered; a file
// whose size is not a multiple of 144 is refused here instead of yielding a partly uninitialised last record.
inline bool LoadVertices(const std::string& fn, std::vector<VertexRecord>* out) {
  out->clear();
  std::FILE* f = std::fopen(fn.c_str(), "rb");
  if (!f) return false;
  VertexRecord v;
  size_t got;
  while ((got = std::fread(&v, 1, sizeof(v), f)) == sizeof(v)) out->push_back(v);
  std::fclose(f);
  return got == 0;
}

// LoadVertexById (pose_graph.cpp:156-172): the first record with that id
inline bool LoadVertexById(const std::string& fn, uint32_t id, VertexRecord* out) {
  std::vector<VertexRecord> vs;
  if (!LoadVertices(fn, &vs)) return false;
  for (const auto& v : vs)
    if (v.id == id) { *out = v; return true; }
  return false;
}

}  // namespace rgbdvo
