// rgbdvo::MotionEstimator read surface (rgbdvo/motion_estimate.hpp:40-74), filled from the GPU results.
#pragma once
#include <cmath>
#include <vector>

#include "../rvo.h"
#include "feature.hpp"

namespace rgbdvo {

enum MotionEstimateStatusCode { NO_DATA, SUCCESS, INSUFFICIENT_INLIERS, OPTIMIZATION_FAILURE, REPROJECTION_ERROR };

class MotionEstimator {
 public:
  bool IsMotionEstimateValid() const { return status_ == SUCCESS; }
  MotionEstimateStatusCode GetMotionEstimateStatus() const { return status_; }
  const Eigen::Isometry3d& GetMotionEstimate() const { return motion_; }
  const Eigen::MatrixXd& GetMotionEstimateCov() const { return cov_; }
  const FeatureMatch* GetMatches() const { load(); return matches_.data(); }
  int32_t GetNumMatches() const { return num_matches_; }
  int32_t GetNumInliers() const { return num_inliers_; }
  int32_t GetNumReprojectionFailures() const { return num_fail_; }
  double GetMeanInlierReprojectionError() const { return mean_err_; }

  // facade plumbing: (ctx, pair work area, slots) of the estimate this object describes
  void bind(rvo_ctx* ctx, int pair, int ref_slot, int tgt_slot, int levels, const rvo_pair_result& r) {
    ctx_ = ctx; pair_ = pair; ref_slot_ = ref_slot; tgt_slot_ = tgt_slot; levels_ = levels;
    status_ = (MotionEstimateStatusCode)r.status;
    num_matches_ = r.num_matches; num_inliers_ = r.num_inliers; num_fail_ = r.num_reprojection_failures;
    mean_err_ = r.mean_reprojection_error;
    motion_ = detail::iso_from_3x4(r.pose);
    cov_ = detail::mat6_from(r.cov);
    loaded_ = false;
  }
  void set_counts(int status, int matches, int inliers) {
    status_ = (MotionEstimateStatusCode)status; num_matches_ = matches; num_inliers_ = inliers; loaded_ = false;
  }

 private:
  void load() const {  // lazily pull the match list (post-sort order) and the keypoints it points into
    if (loaded_ || !ctx_) return;
    loaded_ = true;
    matches_.assign(num_matches_, FeatureMatch());
    for (int s = 0; s < 2; s++) {
      kps_[s].assign(levels_, {});
      for (int l = 0; l < levels_; l++) {
        int32_t info[6];
        const int slot = s ? tgt_slot_ : ref_slot_;
        if (rvo_get_level_info(ctx_, slot, l, info)) continue;
        const int n = info[3];
        std::vector<float> uvs(3 * (size_t)n);
        std::vector<double> rect(2 * (size_t)n), xyz(3 * (size_t)n);
        std::vector<int32_t> tracks((size_t)n, -1);
        rvo_get_keypoints(ctx_, slot, l, uvs.data(), rect.data(), xyz.data());
        rvo_get_keypoint_tracks(ctx_, slot, l, tracks.data());
        kps_[s][l].resize(n);
        for (int i = 0; i < n; i++) {
          KeypointData& k = kps_[s][l][i];
          k.kp.u = uvs[3 * i]; k.kp.v = uvs[3 * i + 1]; k.kp.score = uvs[3 * i + 2];
          k.base_uv = Eigen::Vector2d(k.kp.u * (1 << l), k.kp.v * (1 << l));
          k.rect_base_uv = Eigen::Vector2d(rect[2 * i], rect[2 * i + 1]);
          k.xyz = Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
          k.xyzw = Eigen::Vector4d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 1.0);
          k.pyramid_level = (uint8_t)l; k.keypoint_index = i; k.track_id = tracks[i];
        }
      }
    }
    std::vector<int32_t> ints(7 * (size_t)num_matches_);
    std::vector<double> dbl(8 * (size_t)num_matches_);
    std::vector<int32_t> mtracks((size_t)num_matches_, -1);
    const int n = rvo_get_matches(ctx_, pair_, num_matches_, ints.data(), dbl.data());
    rvo_get_match_tracks(ctx_, pair_, num_matches_, mtracks.data());
    for (int m = 0; m < n; m++) {
      const int32_t* a = &ints[7 * m];
      const double* d = &dbl[8 * m];
      FeatureMatch& f = matches_[m];
      f.ref_keypoint = &kps_[0][a[2]][a[0]];
      f.target_keypoint = &kps_[1][a[2]][a[1]];
      f.refined_target_keypoint = *f.target_keypoint;
      f.refined_target_keypoint.kp.u = (float)d[0]; f.refined_target_keypoint.kp.v = (float)d[1];
      f.refined_target_keypoint.base_uv = Eigen::Vector2d(d[0] * (1 << a[2]), d[1] * (1 << a[2]));
      f.refined_target_keypoint.rect_base_uv = Eigen::Vector2d(d[2], d[3]);
      f.refined_target_keypoint.xyz = Eigen::Vector3d(d[4], d[5], d[6]);
      f.refined_target_keypoint.xyzw = Eigen::Vector4d(d[4], d[5], d[6], 1.0);
      f.track_id = mtracks[m];
      f.status = (MatchStatusCode)a[3]; f.inlier = a[4] != 0; f.in_maximal_clique = f.inlier || d[7] > 0;
      f.id = a[5]; f.compatibility_degree = a[6]; f.reprojection_error = d[7];
    }
  }

  rvo_ctx* ctx_ = nullptr;
  int pair_ = 0, ref_slot_ = 0, tgt_slot_ = 0, levels_ = 0;
  MotionEstimateStatusCode status_ = NO_DATA;
  int32_t num_matches_ = 0, num_inliers_ = 0, num_fail_ = 0;
  double mean_err_ = 0;
  Eigen::Isometry3d motion_ = Eigen::Isometry3d::Identity();
  Eigen::MatrixXd cov_;
  mutable bool loaded_ = false;
  mutable std::vector<FeatureMatch> matches_;
  mutable std::vector<std::vector<KeypointData>> kps_[2];
};

}  // namespace rgbdvo
