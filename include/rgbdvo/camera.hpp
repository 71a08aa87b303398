// rgbdvo::IntrinsicParameters / rgbdvo::Camera with the reference's surface (rgbdvo/camera.hpp:11-131).
// The undistortion LUT itself lives in the GPU context (built by rvo_create); Camera only carries parameters.
#pragma once
#include <cstdint>

#include "../rvo.h"
#include "eigen_compat.hpp"

namespace rgbdvo {

struct IntrinsicParameters {  // rgbdvo/camera.hpp:11-49
  IntrinsicParameters()
      : width(0), height(0), depth_width(0), depth_height(0), fx(0), fy(0), cx(0), cy(0), k1(0), k2(0), k3(0), p1(0),
        p2(0), h_fov(0.0), v_fov(0.0) {}
  int32_t width, height;
  int32_t depth_width, depth_height;
  double fx, fy;
  double cx, cy;
  double k1, k2, k3;
  double p1, p2;
  double h_fov, v_fov;
  rvo_intrinsics toC() const { return rvo_intrinsics{width, height, depth_width, depth_height, fx, fy, cx, cy, k1, k2, k3, p1, p2}; }
};

class Camera {  // rgbdvo/camera.hpp:52-131
 public:
  explicit Camera(const IntrinsicParameters& p) : input_(p), rect_(p) { rect_.k1 = rect_.k2 = rect_.k3 = rect_.p1 = rect_.p2 = 0; }
  const IntrinsicParameters& GetInputCameraParameters() const { return input_; }
  const IntrinsicParameters& GetRectifiedCameraParameters() const { return rect_; }
  Camera* MakeCopy() const { return new Camera(input_); }
  Camera(const Camera&) = delete;
  Camera& operator=(const Camera&) = delete;

 private:
  IntrinsicParameters input_, rect_;
};

}  // namespace rgbdvo
