// rgbdvo::FrameMatcher (rgbdvo/fm.hpp:14-44, fm.cpp:6-58) on the GPU library: slot 0 = reference, slot 1 = target.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "../rvo.h"
#include "camera.hpp"
#include "motion_estimate.hpp"
#include "settings.hpp"

namespace rgbdvo {

class FrameMatcher {
 public:
  FrameMatcher(const Camera& camera, const Options& options = GetFrameMatcherDefaultOptions(), int device = 0)
      : camera_(camera.MakeCopy()) {
    rvo_default_fm_params(&prm_);
    detail::params_from_options(options, &prm_);
    const rvo_intrinsics in = camera_->GetInputCameraParameters().toC();
    if (rvo_create(&in, &prm_, device, 2, 1, &ctx_)) { fprintf(stderr, "FrameMatcher: %s\n", rvo_last_error()); abort(); }
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

  FrameMatcher(const FrameMatcher&) = delete;
  FrameMatcher& operator=(const FrameMatcher&) = delete;

 private:
  std::unique_ptr<Camera> camera_;
  rvo_params prm_;
  rvo_ctx* ctx_ = nullptr;
  MotionEstimator est_;
  Eigen::MatrixXd cov_;
};

}  // namespace rgbdvo
