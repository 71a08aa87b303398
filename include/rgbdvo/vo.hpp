// rgbdvo::RGBDVO (rgbdvo/vo.hpp:15-102, vo.cpp:11-165): the singleton streaming front-end, one GPU stream of the
// batched engine.  Error convention kept: when both motion estimates fail the process exits (vo.cpp:154-158).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>

#include "../rvo.h"
#include "camera.hpp"
#include "motion_estimate.hpp"
#include "settings.hpp"

namespace rgbdvo {

class RGBDVO {
 public:
  static void Create(const Camera& camera, const Options& options) {  // vo.cpp:11-15 (idempotent)
    if (inst()) return;
    inst() = new RGBDVO(camera, options);
  }
  static RGBDVO* instance() {
    if (!inst()) { fprintf(stderr, "RGBDVO::instance() before Create()\n"); abort(); }  // vo.cpp:18 assert
    return inst();
  }

  void ProcessFrame(const uint8_t* gray, const uint16_t* raw_depth) {  // vo.cpp:55-165
    if (init_cnt_ < (size_t)prm_.warmup_frames) {  // progress output of vo.cpp:58-67
      if (init_cnt_ == 0) std::cerr << "Init RGBVO.";
      else if (init_cnt_ % 20 == 0) std::cerr << ".";
      ++init_cnt_;
      if (init_cnt_ == (size_t)prm_.warmup_frames) std::cerr << "start." << std::endl;
    }
    rvo_stream_result r;
    if (rvo_stream_step(ctx_, 1, gray, raw_depth, &r)) { fprintf(stderr, "RGBDVO: %s\n", rvo_last_error()); exit(-1); }
    if (r.rc == 0) return;
    integrated_ = detail::iso_from_3x4(r.pose);
    pose_ = compose(init_pose_, integrated_);
    curr_to_ref_ = detail::iso_from_3x4(r.curr_to_ref);
    cov_ = detail::mat6_from(r.cov);
    keyframe_changed_ = r.keyframe_changed != 0;
    int32_t slots[3];
    rvo_get_stream_slots(ctx_, 0, slots);
    rvo_pair_result pr;
    rvo_get_pair_results(ctx_, 1, &pr);
    if (r.status == 0) est_.set_counts(0, 0, 0);
    else est_.bind(ctx_, 0, r.used_fallback ? slots[1] : slots[0], slots[2], prm_.num_pyramid_levels, pr);
    if (r.rc < 0) {
      std::cerr << "Cannot get valid motion estimation, exit." << std::endl;
      exit(-1);
    }
  }

  const Eigen::Isometry3d& GetPose() const { return pose_; }
  const Eigen::Isometry3d& GetCurrToRefMotionEstimate() const { return curr_to_ref_; }
  const Eigen::MatrixXd& GetCurrToRefCov() const { return cov_; }
  bool IsKeyFrameChanged() const { return keyframe_changed_; }
  const MotionEstimator* GetMotionEstimator() const { return &est_; }
  // vo.hpp:42-45: pose_ = pose, at any time.  The engine integrates from identity (`integrated_`), the facade keeps
  // pose_ = init_pose_ * integrated_; setting init_pose_ = pose * integrated_^-1 makes GetPose() equal `pose` now and
  // continue the reference's pose_ = pose_ * curr_to_prev chain from there, also when called in the middle of a run.
  const void SetInitPose(Eigen::Isometry3d pose = Eigen::Isometry3d::Identity()) {
    init_pose_ = compose(pose, rigid_inverse(integrated_));
    pose_ = pose;
  }

 private:
  RGBDVO(const Camera& camera, const Options& options) : camera_(camera.MakeCopy()) {
    rvo_default_vo_params(&prm_);
    detail::params_from_options(options, &prm_);
    const rvo_intrinsics in = camera_->GetInputCameraParameters().toC();
    int device = 0;  // the reference has no notion of a device; RVO_DEVICE selects the GPU of the singleton
    if (const char* e = getenv("RVO_DEVICE")) device = atoi(e);
    if (rvo_create(&in, &prm_, device, 3, 1, &ctx_) || rvo_streams_reset(ctx_, 1)) { fprintf(stderr, "RGBDVO: %s\n", rvo_last_error()); abort(); }
    cov_ = Eigen::MatrixXd(6, 6);
    // the reference leaves pose_ uninitialised until SetInitPose (vo.hpp:42-45): Identity here
    pose_ = Eigen::Isometry3d::Identity(); init_pose_ = Eigen::Isometry3d::Identity();
    integrated_ = Eigen::Isometry3d::Identity(); curr_to_ref_ = Eigen::Isometry3d::Identity();
  }
  ~RGBDVO() { rvo_destroy(ctx_); }
  static RGBDVO*& inst() { static RGBDVO* p = nullptr; return p; }
  static Eigen::Isometry3d compose(const Eigen::Isometry3d& a, const Eigen::Isometry3d& b) {
    Eigen::Isometry3d r;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += a.matrix()(i, k) * b.matrix()(k, j);
        r.matrix()(i, j) = s;
      }
    return r;
  }

  static Eigen::Isometry3d rigid_inverse(const Eigen::Isometry3d& a) {  // [R t]^-1 = [R' -R't]
    Eigen::Isometry3d r = Eigen::Isometry3d::Identity();
    for (int i = 0; i < 3; i++) {
      double s = 0;
      for (int j = 0; j < 3; j++) { r.matrix()(i, j) = a.matrix()(j, i); s += a.matrix()(j, i) * a.matrix()(j, 3); }
      r.matrix()(i, 3) = -s;
    }
    return r;
  }

  std::unique_ptr<Camera> camera_;
  rvo_params prm_;
  rvo_ctx* ctx_ = nullptr;
  size_t init_cnt_ = 0;
  bool keyframe_changed_ = false;
  Eigen::Isometry3d pose_, init_pose_, integrated_, curr_to_ref_;
  Eigen::MatrixXd cov_;
  MotionEstimator est_;
};

}  // namespace rgbdvo
