// CPU-only check of include/rgbdvo/vertex_io.hpp against slam_amd/vertex_io.py (driven by tests/test_cpu_dist.py):
//   test_vertex_io write <file> <n>   writes n vertices (ids 3*i+1, poses from a fixed formula), the last two appended
//   test_vertex_io read <file>        prints "id p0 ... p11" (row-major 3x4, %.17g) per record, then by-id lookups
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "rgbdvo/vertex_io.hpp"

static void pose_of(int i, double* p) {
  for (int k = 0; k < 12; k++) p[k] = (i + 1) * 0.125 + k * 1.0 / 3.0 - (k % 4 == 3 ? 7.0 / (i + 2) : 0.0);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string fn = argv[2];
  if (!std::strcmp(argv[1], "write")) {
    const int n = argc > 3 ? std::atoi(argv[3]) : 5;
    std::vector<double> traj((size_t)n * 12);
    for (int i = 0; i < n; i++) pose_of(i, traj.data() + (size_t)i * 12);
    std::vector<rgbdvo::VertexRecord> vs;
    for (int i = 0; i < n - 2; i++) vs.push_back(rgbdvo::MakeVertexRecord(3u * i + 1u, traj.data() + (size_t)i * 12));
    if (!rgbdvo::SaveVertices(fn, vs)) return 1;
    // through the Isometry3d overload and the appending form
    const Eigen::Isometry3d iso = rgbdvo::detail::iso_from_3x4(traj.data() + (size_t)(n - 2) * 12);
    if (!rgbdvo::SaveVertexById(fn, rgbdvo::MakeVertexRecord(3u * (n - 2) + 1u, iso))) return 1;
    if (!rgbdvo::SaveTrajectory(fn, traj.data() + (size_t)(n - 1) * 12, 1, 3u * (n - 1) + 1u, true)) return 1;
    return 0;
  }
  if (!std::strcmp(argv[1], "read")) {
    std::vector<rgbdvo::VertexRecord> vs;
    if (!rgbdvo::LoadVertices(fn, &vs)) return 1;
    for (const auto& v : vs) {
      std::printf("%u", v.id);
      const Eigen::Isometry3d m = rgbdvo::PoseOf(v);
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) std::printf(" %.17g", m.matrix()(r, c));
      std::printf("\n");
    }
    rgbdvo::VertexRecord one;
    std::printf("by_id %d %d\n", (int)rgbdvo::LoadVertexById(fn, vs.empty() ? 0u : vs.back().id, &one), (int)rgbdvo::LoadVertexById(fn, 4000000000u, &one));
    return 0;
  }
  return 2;
}
