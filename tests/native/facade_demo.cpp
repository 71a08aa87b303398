// Exercises the C++ facade (include/rgbdvo/) the way the reference's callers do:
//   hand_held_demo.cpp:236-278 (RGBDVO::Create / ProcessFrame / GetPose / IsKeyFrameChanged / inlier drawing) and
//   pose_graph.cpp:104-139 (FrameMatcher::SetRefFrame / Match / GetMotionEstimate / GetCov).
// Input: a raw file of N frames (W*H u8 gray then W*H u16 depth each).  Output: one text line per frame.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/rgbdvo/fm.hpp"
#include "../../include/rgbdvo/vo.hpp"

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s frames.raw W H N warmup\n", argv[0]); return 2; }
  const int W = atoi(argv[2]), H = atoi(argv[3]), N = atoi(argv[4]), warm = atoi(argv[5]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<std::vector<uint8_t>> gray(N, std::vector<uint8_t>((size_t)W * H));
  std::vector<std::vector<uint16_t>> depth(N, std::vector<uint16_t>((size_t)W * H));
  for (int i = 0; i < N; i++) {
    if (fread(gray[i].data(), 1, (size_t)W * H, f) != (size_t)W * H) return 2;
    if (fread(depth[i].data(), 2, (size_t)W * H, f) != (size_t)W * H) return 2;
  }
  fclose(f);

  rgbdvo::IntrinsicParameters p;  // hand_held_demo.cpp:289-298 style
  p.width = p.depth_width = W; p.height = p.depth_height = H;
  p.fx = p.fy = 525.0 * W / 640.0; p.cx = W / 2.0; p.cy = H / 2.0;
  rgbdvo::Camera camera(p);
  rgbdvo::Options opt = rgbdvo::GetRGBDVODefaultOptions();
  (void)warm;
  rgbdvo::RGBDVO::Create(camera, opt);
  rgbdvo::RGBDVO* vo = rgbdvo::RGBDVO::instance();
  vo->SetInitPose();
  // the reference discards the first 99 frames (vo.cpp:58-69): feed the first frame 99 extra times
  for (int i = 0; i < 99; i++) vo->ProcessFrame(gray[0].data(), depth[0].data());
  for (int i = 0; i < N; i++) {
    vo->ProcessFrame(gray[i].data(), depth[i].data());
    const Eigen::Isometry3d& T = vo->GetPose();
    const rgbdvo::MotionEstimator* me = vo->GetMotionEstimator();
    int drawn = 0;
    const rgbdvo::FeatureMatch* m = me->GetMatches();
    for (int k = 0; k < me->GetNumMatches(); k++)
      if (m[k].inlier && m[k].ref_keypoint->base_uv(0) >= 0 && m[k].target_keypoint->base_uv(1) >= 0) drawn++;
    printf("VO %d %d %d %d %d", i, (int)vo->IsKeyFrameChanged(), me->GetNumMatches(), me->GetNumInliers(), drawn);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) printf(" %.17g", T.matrix()(r, c));
    printf("\n");
  }
  rgbdvo::FrameMatcher fm(camera);
  fm.SetRefFrame(gray[0].data(), depth[0].data());
  for (int i = 1; i < N; i += 3) {
    const bool ok = fm.Match(gray[i].data(), depth[i].data());
    const Eigen::Isometry3d& T = fm.GetMotionEstimate();
    printf("FM %d %d %d", i, (int)ok, fm.GetCov().rows());
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) printf(" %.17g", T.matrix()(r, c));
    printf(" %.17g\n", fm.GetCov()(0, 0));
  }
  // batched loop closure on cached keyframes (pose_graph.cpp:85-145 as one call) + edge information (pose_graph.cpp:265)
  rgbdvo::FrameMatcher lc(camera, rgbdvo::GetFrameMatcherDefaultOptions(), 0, N);
  for (int i = 0; i < N; i++) lc.CacheKeyFrame(i, gray[i].data(), depth[i].data());
  std::vector<int> cand;
  for (int i = 0; i + 1 < N; i++) cand.push_back(i);
  std::vector<rgbdvo::LoopClosureEdge> edges;
  const int nok = lc.MatchCachedKeyFrames(N - 1, cand, &edges);
  printf("LC %d %d\n", (int)edges.size(), nok);
  for (const auto& e : edges) {
    printf("LCE %d %d %d", e.target_id, (int)e.valid, e.num_inliers);
    for (int k = 0; k < 12; k++) printf(" %.17g", e.motion[k]);
    printf(" %.17g %.17g\n", e.cov[0], e.information[0]);
  }
  return 0;
}
