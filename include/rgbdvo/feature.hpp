// The read-only inspection types the callers touch (rgbdvo/feature.hpp:23-31, 34-105, 188-308):
// hand_held_demo.cpp:140-155 reads FeatureMatch::inlier and {ref,target}_keypoint->base_uv.
#pragma once
#include <cstdint>

#include "eigen_compat.hpp"

namespace rgbdvo {

struct KeyPoint {
  float u, v, score;
  KeyPoint() : u(0), v(0), score(0) {}
};

class KeypointData {
 public:
  KeyPoint kp;
  Eigen::Vector4d xyzw;  // homogeneous coordinates (x, y, z, 1), feature.hpp:44-47
  Eigen::Vector3d xyz;
  bool has_depth = true;
  Eigen::Vector2d base_uv;
  Eigen::Vector2d rect_base_uv;
  uint8_t pyramid_level = 0;
  int32_t keypoint_index = 0;
  int32_t track_id = -1;  // feature.hpp:75: the track this keypoint belongs to (motion_estimate.cpp:380-391), -1 = none
};

enum MatchStatusCode { MATCH_NEEDS_DEPTH_REFINEMENT, MATCH_REFINEMENT_FAILED, MATCH_OK };

class FeatureMatch {
 public:
  KeypointData* target_keypoint = nullptr;
  KeypointData* ref_keypoint = nullptr;
  KeypointData refined_target_keypoint;
  int compatibility_degree = 0;
  bool in_maximal_clique = false;
  bool inlier = false;
  int id = 0;
  double reprojection_error = 0;
  MatchStatusCode status = MATCH_OK;
  int track_id = -1;  // feature.hpp:297-302
};

}  // namespace rgbdvo
