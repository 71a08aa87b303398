// rgbdvo::FrameMatcher (rgbdvo/fm.hpp:14-44, fm.cpp:6-58) on the GPU library: slot 0 = reference, slot 1 = target.
// Extension: keyframes cached prepared on the device (slots 2..) and the candidate
// loop of PoseGraph::LinearLoopClosure (pose_graph.cpp:85-145) submitted as ONE rvo_match_pairs call, each accepted
// edge with its information matrix cov.inverse() (pose_graph.cpp:265).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../rvo.h"
#include "camera.hpp"
#include "motion_estimate.hpp"
#include "settings.hpp"

namespace rgbdvo {

// One candidate of a batched loop closure: what PoseGraph::LinearLoopClosure reads after Match() (pose_graph.cpp:132-139)
struct LoopClosureEdge {
  int target_id = -1;
  bool valid = false;        // Match() would have returned true
  int status = 0, num_matches = 0, num_inliers = 0;
  double motion[12] = {};    // GetMotionEstimate(), 3x4 row-major
  double cov[36] = {};       // GetCov()
  double information[36] = {};  // cov.inverse(), what PoseGraphOptimizer::AddEdge feeds g2o
};

class FrameMatcher {
 public:
  FrameMatcher(const Camera& camera, const Options& options = GetFrameMatcherDefaultOptions(), int device = 0,
               int max_keyframes = 0)
      : camera_(camera.MakeCopy()), max_keyframes_(max_keyframes) {
    rvo_default_fm_params(&prm_);
    detail::params_from_options(options, &prm_);
    const rvo_intrinsics in = camera_->GetInputCameraParameters().toC();
    if (rvo_create(&in, &prm_, device, 2 + max_keyframes, max_keyframes > 0 ? max_keyframes : 1, &ctx_)) {
      fprintf(stderr, "FrameMatcher: %s\n", rvo_last_error());
      abort();
    }
    cov_ = Eigen::MatrixXd(6, 6);
  }
  ~FrameMatcher() { rvo_destroy(ctx_); }

  void SetRefFrame(const uint8_t* gray, const uint16_t* raw_depth) {  // fm.cpp:22-30
    const int32_t thr = prm_.fast_init_threshold, slot = 0;
    if (rvo_prepare_frames(ctx_, 1, gray, raw_depth, &thr, &slot)) fprintf(stderr, "FrameMatcher: %s\n", rvo_last_error());
  }

  bool Match(const uint8_t* gray, const uint16_t* raw_depth) {  // fm.cpp:32-58
    const int32_t thr = prm_.fast_init_threshold, slot = 1, ref = 0;
    rvo_pair_result r;
    if (rvo_prepare_frames(ctx_, 1, gray, raw_depth, &thr, &slot) || rvo_match_pairs(ctx_, 1, &ref, &slot, nullptr, &r)) {
      fprintf(stderr, "FrameMatcher: %s\n", rvo_last_error());
      return false;
    }
    est_.bind(ctx_, 0, 0, 1, prm_.num_pyramid_levels, r);
    return est_.IsMotionEstimateValid();
  }

  const Eigen::Isometry3d& GetMotionEstimate() const { return est_.GetMotionEstimate(); }
  const Eigen::MatrixXd& GetCov() const { return est_.GetMotionEstimateCov(); }
  const IntrinsicParameters& GetInputCameraParameters() const { return camera_->GetInputCameraParameters(); }
  const MotionEstimator* GetMotionEstimator() const { return &est_; }

  // ---- batched loop closure on cached keyframes
  bool CacheKeyFrame(int id, const uint8_t* gray, const uint16_t* raw_depth) {
    if (id < 0 || id >= max_keyframes_) return false;
    const int32_t thr = prm_.fast_init_threshold, slot = 2 + id;
    return rvo_prepare_frames(ctx_, 1, gray, raw_depth, &thr, &slot) == 0;
  }
  // every candidate against keyframe ref_id in one call; returns the number of valid edges
  int MatchCachedKeyFrames(int ref_id, const std::vector<int>& target_ids, std::vector<LoopClosureEdge>* edges) {
    const int n = (int)target_ids.size();
    edges->assign(n, LoopClosureEdge());
    if (n == 0) return 0;
    if (ref_id < 0 || ref_id >= max_keyframes_) { fprintf(stderr, "FrameMatcher: keyframe id %d out of range\n", ref_id); return 0; }
    for (int i = 0; i < n; i++)  // ids map to slots 2 + id: -1 / -2 would name the live target / reference frame
      if (target_ids[i] < 0 || target_ids[i] >= max_keyframes_) { fprintf(stderr, "FrameMatcher: keyframe id %d out of range\n", target_ids[i]); return 0; }
    std::vector<int32_t> ref(n, 2 + ref_id), tgt(n);
    for (int i = 0; i < n; i++) tgt[i] = 2 + target_ids[i];
    std::vector<rvo_pair_result> r(n);
    std::vector<double> info((size_t)n * 36);
    // SetRefFrame + one Match per candidate work on freshly prepared frames (fm.cpp:22-58): the cached ones start
    // without track ids, and the batch then counts tracks like those calls in order (rvo_get_match_tracks)
    std::vector<int32_t> fresh(tgt);
    fresh.push_back(2 + ref_id);
    std::sort(fresh.begin(), fresh.end());
    fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
    if (rvo_reset_tracks(ctx_, (int)fresh.size(), fresh.data())) { fprintf(stderr, "FrameMatcher: %s\n", rvo_last_error()); return 0; }
    if (rvo_match_pairs(ctx_, n, ref.data(), tgt.data(), nullptr, r.data()) || rvo_get_pair_information(ctx_, n, info.data())) {
      fprintf(stderr, "FrameMatcher: %s\n", rvo_last_error());
      return 0;
    }
    int ok = 0;
    for (int i = 0; i < n; i++) {
      LoopClosureEdge& e = (*edges)[i];
      e.target_id = target_ids[i];
      e.valid = r[i].status == RVO_SUCCESS;
      e.status = r[i].status; e.num_matches = r[i].num_matches; e.num_inliers = r[i].num_inliers;
      memcpy(e.motion, r[i].pose, sizeof(e.motion));
      memcpy(e.cov, r[i].cov, sizeof(e.cov));
      memcpy(e.information, &info[(size_t)i * 36], sizeof(e.information));
      ok += e.valid;
    }
    return ok;
  }

  FrameMatcher(const FrameMatcher&) = delete;
  FrameMatcher& operator=(const FrameMatcher&) = delete;

 private:
  std::unique_ptr<Camera> camera_;
  int max_keyframes_ = 0;
  rvo_params prm_;
  rvo_ctx* ctx_ = nullptr;
  MotionEstimator est_;
  Eigen::MatrixXd cov_;
};

}  // namespace rgbdvo
