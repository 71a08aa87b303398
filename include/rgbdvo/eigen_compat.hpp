// eigen_compat.hpp -- the facade's signatures use Eigen types, exactly like the reference headers
// (rgbdvo/vo.hpp:22-45, fm.hpp:31-37).  With Eigen3 on the include path the real types are used.  Without it
// (this build image has no Eigen) a minimal value-type subset is provided so that the facade itself can be compiled
// and exercised; it is NOT an Eigen replacement and is never used to build reference code.
#pragma once
#if defined(__has_include)
#if __has_include(<Eigen/Geometry>)
#define RGBDVO_HAVE_EIGEN 1
#endif
#endif

#ifdef RGBDVO_HAVE_EIGEN
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace rgbdvo {
namespace detail {
inline Eigen::Isometry3d iso_from_3x4(const double* p) {
  Eigen::Isometry3d m = Eigen::Isometry3d::Identity();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) m.matrix()(i, j) = p[i * 4 + j];
  return m;
}
inline Eigen::MatrixXd mat6_from(const double* p) {
  Eigen::MatrixXd m(6, 6);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) m(i, j) = p[i * 6 + j];
  return m;
}
}  // namespace detail
}  // namespace rgbdvo
#else
#include <cstring>
#include <vector>
namespace Eigen {
struct Vector2d {
  double v[2] = {0, 0};
  Vector2d() {}
  Vector2d(double a, double b) { v[0] = a; v[1] = b; }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
};
struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() {}
  Vector3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
struct Vector4d {
  double v[4] = {0, 0, 0, 0};
  Vector4d() {}
  Vector4d(double a, double b, double c, double d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
};
struct Matrix4d {
  double m[16];  // column-major, like Eigen
  double& operator()(int r, int c) { return m[c * 4 + r]; }
  double operator()(int r, int c) const { return m[c * 4 + r]; }
  const double* data() const { return m; }
};
struct Isometry3d {
  Matrix4d M;
  Isometry3d() { setIdentity(); }
  static Isometry3d Identity() { return Isometry3d(); }
  void setIdentity() { for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = (r == c); }
  Matrix4d& matrix() { return M; }
  const Matrix4d& matrix() const { return M; }
  Vector3d translation() const { return Vector3d(M(0, 3), M(1, 3), M(2, 3)); }
};
struct MatrixXd {
  int r = 0, c = 0;
  std::vector<double> d;  // column-major
  MatrixXd() {}
  MatrixXd(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, 0.0) {}
  int rows() const { return r; }
  int cols() const { return c; }
  double& operator()(int i, int j) { return d[(size_t)j * r + i]; }
  double operator()(int i, int j) const { return d[(size_t)j * r + i]; }
  const double* data() const { return d.data(); }
};
}  // namespace Eigen
namespace rgbdvo {
namespace detail {
inline Eigen::Isometry3d iso_from_3x4(const double* p) {
  Eigen::Isometry3d m;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) m.matrix()(i, j) = p[i * 4 + j];
  return m;
}
inline Eigen::MatrixXd mat6_from(const double* p) {
  Eigen::MatrixXd m(6, 6);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) m(i, j) = p[i * 6 + j];
  return m;
}
}  // namespace detail
}  // namespace rgbdvo
#endif
