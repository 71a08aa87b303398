// rvo_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Plain, single-threaded C++ restatement of the reference's rgbdvo hot path
// (alanzplus/SLAM, rgbdvo/*.cpp).  It is the *checker* for the HIP kernels in
// slam_amd/csrc: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product library never links or calls it.
//
// PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors for
// rgbdvo and cannot be compiled in this image (every
// rgbdvo translation unit includes Eigen3, which is neither installed nor
// vendored; writing a stand-in Eigen is not allowed).  This file therefore
// follows the reference *source text*; each function cites the file:line it
// restates.  The one piece of the path that can be checked mechanically against
// the reference text without compiling it -- the machine-generated FAST-9
// decision trees -- is pinned by oracle/check_fast_tree.py.
//
// Arithmetic conventions (they matter for bit parity):
//  * built with -O2 -ffp-contract=off on x86-64 (no FMA, no x87) like the
//    reference's -msse3 build (CMakeLists.txt:137);
//  * std::nth_element / std::sort are the real libstdc++ (g++ 11.4) ones: the
//    reference's results depend on them (pyramid.cpp:42, motion_estimate.cpp:526);
//  * Eigen fixed-size expressions are restated with Eigen 3.2.x evaluation
//    order: dot products/sums left-to-right, Vector3d::norm() as
//    sqrt(x*x + (y*y + z*z)), `v /= s` as `v *= (1/s)`.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

// ---------------------------------------------------------------- POD inputs
struct Intrinsics {  // camera.hpp:11-49
  int32_t width, height, depth_width, depth_height;
  double fx, fy, cx, cy, k1, k2, k3, p1, p2;
};

struct Params {  // settings.hpp:11-52 (same keys, parsed)
  int32_t feature_window_size;
  int32_t num_pyramid_levels;
  int32_t fast_init_threshold;
  int32_t fast_threshold_min;
  int32_t fast_threshold_max;
  double fast_adaptive_gain;
  int32_t target_pixels_per_feature;
  int32_t grid_width;
  int32_t grid_height;
  int32_t max_keypoints_per_grid;
  double inlier_max_reprojection_error;
  double clique_inlier_threshold;
  int32_t min_features_for_estimate;
  double max_mean_reprojection_error;
  double feature_search_window;
  int32_t reference_frame_change_threshold;
  int32_t warmup_frames;  // vo.cpp:58 hard-codes 100
};

struct PairResult {
  double pose[12];  // 3x4 row-major [R|t], target/current -> reference
  double cov[36];   // 6x6
  int32_t status;   // motion_estimate.hpp:13-20
  int32_t num_matches;
  int32_t num_inliers;
  int32_t num_reprojection_failures;
  double mean_reprojection_error;
};

enum { NO_DATA = 0, SUCCESS, INSUFFICIENT_INLIERS, OPTIMIZATION_FAILURE, REPROJECTION_ERROR };
enum { MATCH_NEEDS_DEPTH_REFINEMENT = 0, MATCH_REFINEMENT_FAILED, MATCH_OK };  // feature.hpp:188-201

inline int round_up(int x, int a) { int r = x % a; return r ? x + (a - r) : x; }  // helper.hpp:36-42

// ---------------------------------------------------------------- stability margins
// How far the quantities the path DECIDES on sit from their thresholds, and what happens to the decided sets when the
// parts of this restatement that are conventions (Eigen's evaluation order, the initial-rotation solver) are perturbed.
// Recorded per EstimateMotion call; read with orc_get_margins.  Test instrumentation only: it changes no result.
struct Margins {
  double window = INFINITY;       // | ||proj - kp|| - feature_search_window |            motion_estimate.cpp:335     [px]
  double refine_gate = INFINITY;  // | ||kp - refined|| - 1.5 |  (moved matches only)      motion_estimate.cpp:360-366 [px]
  double esm_step = INFINITY;     // | max(|dx|,|dy|) - 0.1 | over ESM iterations          motion_estimate.cpp:894     [px]
  double sse_tie = INFINITY;      // | sse - final_sse | over ESM iterations (integers)    motion_estimate.cpp:882
  double clique = INFINITY;       // | |dref - dtgt| - clique_inlier_threshold |           motion_estimate.cpp:507     [m]
  double reproj_gate = INFINITY;  // | e - inlier_max_reprojection_error | over inliers    motion_estimate.cpp:204-214 [px]
  double mean_reproj = INFINITY;  // | mean e - max_mean_reprojection_error |              motion_estimate.cpp:229     [px]
  double gn_sse_rel = INFINITY;   // | sse - final_sse | / final_sse over GN iterations    motion_estimate.cpp:1296
  double degree_gap = INFINITY;   // smallest gap between the degrees of two matches that are adjacent after the sort and
                                  // on different sides of the inlier decision (0: a tie decided by std::sort's order)
  double n_window = 0, n_clique = 0, n_esm = 0;  // number of decisions behind window / clique / esm_step
  void note(double* slot, double v) { if (v < *slot) *slot = v; }
};
// Perturbations of the conventions (orc_set_perturbation; process-wide, tests only)
struct Perturb {
  int norm_order = 0;    // bit 0: Vector3d::norm() as sqrt((x*x + y*y) + z*z) instead of sqrt(x*x + (y*y + z*z));
                         // bit 1: 4-term dot products (P * point) pairwise instead of left to right
  int true_division = 0; // 1: `uvw /= w` as true divisions instead of a multiplication by 1/w
  int rot_float = 0;     // 1: EstimateInitialRotation's 3x3 normal equations accumulated and solved in fp32 (a solver with
                         //    cond^2 * eps_float error: an upper bound for what the reference's float JacobiSVD can differ by)
  double q_noise = 0;    // rotate the initial-rotation quaternion by this angle [rad] about a pseudo-random axis
  unsigned seed = 1;
};
static Perturb g_pert;

// ---------------------------------------------------------------- isometry
struct Iso {
  double R[3][3];
  double t[3];
  void setIdentity() {
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[i][j] = (i == j); t[i] = 0; }
  }
};

// Eigen Transform<Isometry>::inverse(): linear^T, -(linear^T * translation)
Iso iso_inverse(const Iso& a) {
  Iso r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.R[i][j] = a.R[j][i];
  for (int i = 0; i < 3; i++) r.t[i] = -((r.R[i][0] * a.t[0] + r.R[i][1] * a.t[1]) + r.R[i][2] * a.t[2]);
  return r;
}

// a * b (affine-compact product)
Iso iso_mul(const Iso& a, const Iso& b) {
  Iso r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++)
      r.R[i][j] = (a.R[i][0] * b.R[0][j] + a.R[i][1] * b.R[1][j]) + a.R[i][2] * b.R[2][j];
    r.t[i] = ((a.R[i][0] * b.t[0] + a.R[i][1] * b.t[1]) + a.R[i][2] * b.t[2]) + a.t[i];
  }
  return r;
}

struct Quat { double w, x, y, z; };

// Eigen QuaternionBase::toRotationMatrix()
void quat_to_R(const Quat& q, double R[3][3]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
  R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}

// Eigen quaternion-from-rotation-matrix (quaternionbase_assign_impl<Other,3,3>)
Quat R_to_quat(const double m[3][3]) {
  Quat q;
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[2][1] - m[1][2]) * t;
    q.y = (m[0][2] - m[2][0]) * t;
    q.z = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m[k][j] - m[j][k]) * t;
    v[j] = (m[j][i] + m[i][j]) * t;
    v[k] = (m[k][i] + m[i][k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}

// Transform::rotate(q): linear = linear * R(q)
void iso_rotate(Iso* a, const Quat& q) {
  double Rq[3][3];
  quat_to_R(q, Rq);
  double L[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      L[i][j] = (a->R[i][0] * Rq[0][j] + a->R[i][1] * Rq[1][j]) + a->R[i][2] * Rq[2][j];
  memcpy(a->R, L, sizeof(L));
}

// helper.hpp:59-81
Quat rpy_to_quat(double roll, double pitch, double yaw) {
  double hr = roll / 2, hp = pitch / 2, hy = yaw / 2;
  double sr = sin(hr), sp = sin(hp), sy = sin(hy);
  double cr = cos(hr), cp = cos(hp), cy = cos(hy);
  Quat q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  return q;
}

// motion_estimate.cpp:903-930
Iso iso_from_params(const double p[6]) {
  Quat q = rpy_to_quat(p[3], p[4], p[5]);
  Iso r;
  r.setIdentity();
  r.t[0] = p[0]; r.t[1] = p[1]; r.t[2] = p[2];  // translate() on identity linear
  iso_rotate(&r, q);
  return r;
}

// motion_estimate.cpp:932-948
void iso_to_params(const Iso& M, double p[6]) {
  Quat q = R_to_quat(M.R);
  double roll_a = 2 * (q.w * q.x + q.y * q.z);
  double roll_b = 1 - 2 * (q.x * q.x + q.y * q.y);
  double pitch_sin = 2 * (q.w * q.y - q.z * q.x);
  double yaw_a = 2 * (q.w * q.z + q.x * q.y);
  double yaw_b = 1 - 2 * (q.y * q.y + q.z * q.z);
  p[0] = M.t[0]; p[1] = M.t[1]; p[2] = M.t[2];
  p[3] = atan2(roll_a, roll_b);
  p[4] = asin(pitch_sin);
  p[5] = atan2(yaw_a, yaw_b);
}

// ---------------------------------------------------------------- camera
struct Camera {  // camera.cpp:5-67
  Intrinsics in, rect;
  std::vector<float> Xmap, Ymap;
  explicit Camera(const Intrinsics& p) : in(p), rect(p) {
    rect.k1 = rect.k2 = rect.k3 = rect.p1 = rect.p2 = 0;
    int W = in.width, H = in.height;
    Xmap.resize((size_t)W * H);
    Ymap.resize((size_t)W * H);
    double fx = in.fx, fy = in.fy, cx = in.cx, cy = in.cy;
    double fxp = rect.fx, fyp = rect.fy, cxp = rect.cx, cyp = rect.cy;
    double k1 = in.k1, k2 = in.k2, k3 = in.k3, p1 = in.p1, p2 = in.p2;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        double x1 = (x - cx) / fx, y1 = (y - cy) / fy;
        double x0 = x1, y0 = y1;
        for (int j = 0; j < 5; ++j) {
          double r2 = x1 * x1 + y1 * y1;
          double icdist = 1. / (1. + ((k3 * r2 + k2) * r2 + k1) * r2);
          double deltaX = 2 * p1 * x1 * y1 + p2 * (r2 + 2 * x1 * x1);
          double deltaY = p1 * (r2 + 2 * y1 * y1) + 2 * p2 * x1 * y1;
          x1 = (x0 - deltaX) * icdist;
          y1 = (y0 - deltaY) * icdist;
        }
        x1 = x1 / 1.0; y1 = y1 / 1.0;
        x1 = x1 * fxp + cxp;
        y1 = y1 * fyp + cyp;
        Xmap[(size_t)y * W + x] = (float)x1;
        Ymap[(size_t)y * W + x] = (float)y1;
      }
  }
  // camera.hpp:85-117
  void BilinearLookup(const double uv[2], double out[2]) const {
    int32_t u = (int32_t)uv[0], v = (int32_t)uv[1];
    float wright = (float)(uv[0] - u), wbottom = (float)(uv[1] - v);
    float w[4] = {(1 - wright) * (1 - wbottom), wright * (1 - wbottom), (1 - wright) * wbottom,
                  wright * wbottom};
    int32_t ra = v * in.width + u;
    size_t idx[4] = {(size_t)ra, (size_t)ra + 1, (size_t)ra + in.width, (size_t)ra + in.width + 1};
    size_t n = Xmap.size();
    out[0] = 0; out[1] = 0;
    for (int i = 0; i < 4; ++i) {
      // the reference reads one row past the LUT when v == height-1; the ESM
      // bounds (pyramid.hpp:136-139) keep v <= height-2, so clamp defensively.
      size_t k = idx[i] < n ? idx[i] : n - 1;
      out[0] += w[i] * Xmap[k];  // float product, double accumulate
      out[1] += w[i] * Ymap[k];
    }
  }
};

// ---------------------------------------------------------------- keypoints
struct KeyPoint { float u, v, score; };  // feature.hpp:23-31

struct KeypointData {  // feature.hpp:34-105
  KeyPoint kp{0, 0, 0};
  double xyzw[4]{0, 0, 0, 1};
  bool has_depth = false;
  double xyz[3]{0, 0, 0};
  double base_uv[2]{0, 0};
  double rect_base_uv[2]{0, 0};
  uint8_t pyramid_level = 0;
  int32_t keypoint_index = 0;
  mutable int32_t track_id = -1;  // frame.cpp:138; written through the matches' const pointers (motion_estimate.cpp:380-391)
};

// ---------------------------------------------------------------- depth
struct Depth {  // depth.cpp / depth.hpp
  int rgb_w, rgb_h, dw, dh;
  float x_scale, y_scale;
  double fx_inv, fy_inv, neg_cx_div_fx, neg_cy_div_fy;
  std::vector<float> data;
  explicit Depth(const Intrinsics& p) {
    rgb_w = p.width; rgb_h = p.height; dw = p.depth_width; dh = p.depth_height;
    x_scale = (float)dw / (float)rgb_w;
    y_scale = (float)dh / (float)rgb_h;
    data.assign((size_t)dw * dh, NAN);
    fx_inv = 1.0 / p.fx; fy_inv = 1.0 / p.fy;
    neg_cx_div_fx = -p.cx * fx_inv;
    neg_cy_div_fy = -p.cy * fy_inv;
  }
  // vo.cpp:83-87 / fm.cpp:24-27: d ? d*1e-3 : NAN  (double product -> float)
  void SetRaw(const uint16_t* raw) {
    size_t n = (size_t)dw * dh;
    for (size_t i = 0; i < n; ++i) { uint16_t d = raw[i]; data[i] = d ? d * 1e-3 : NAN; }
  }
  int32_t RGB2DepthIdx(double u, double v) const {  // depth.hpp:41-43
    return (int32_t)(v * y_scale) * dw + (int32_t)(u * x_scale);
  }
  bool HaveXYZ(int u, int v) const { return !std::isnan(data[RGB2DepthIdx(u, v)]); }
  // depth.cpp:97-162
  bool GetXYZInterp(KeypointData* kp) const {
    float u_f = (float)kp->rect_base_uv[0], v_f = (float)kp->rect_base_uv[1];
    float v_f_d = v_f * y_scale, u_f_d = u_f * x_scale;
    int32_t v = (int32_t)v_f_d, u = (int32_t)u_f_d;
    float wright = (u_f_d - u), wbottom = (v_f_d - v);
    // depth.cpp:109 asserts these bounds (the reference aborts outside them); a batch API cannot
    // abort, so an out-of-range refined position is reported as a failed refinement instead.
    if (!(u >= 0 && v >= 0 && u < dw - 1 && v < dh - 1)) return false;
    float w[4] = {(1 - wright) * (1 - wbottom), wright * (1 - wbottom), (1 - wright) * wbottom,
                  wright * wbottom};
    int32_t di = v * dw + u;
    double depths[4] = {data[di], data[di + 1], data[di + dw], data[di + dw + 1]};
    int missing = 0;
    for (int i = 0; i < 4; i++) if (std::isnan(depths[i])) missing++;
    if (missing == 4) return false;
    double z;
    if (missing) {
      // depth.cpp:137-147 -- the reference tests isnan() (not !isnan()), so z is
      // always NaN here; reproduced on purpose.
      float wmax = -1;
      z = NAN;
      for (int i = 0; i < 4; i++)
        if (std::isnan(depths[i]) && w[i] > wmax) { z = depths[i]; wmax = w[i]; }
    } else {
      z = 0;
      for (int i = 0; i < 4; i++) z += depths[i] * w[i];
    }
    kp->xyz[0] = z * (u_f * fx_inv + neg_cx_div_fx);
    kp->xyz[1] = z * (v_f * fy_inv + neg_cy_div_fy);
    kp->xyz[2] = z;
    kp->xyzw[0] = kp->xyz[0]; kp->xyzw[1] = kp->xyz[1]; kp->xyzw[2] = kp->xyz[2]; kp->xyzw[3] = 1;
    return true;
  }
};

// ---------------------------------------------------------------- FAST-9
// feature.cpp:2997-3015 ring offsets (dx,dy), index 0 = (0,+3), clockwise as listed.
const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

uint8_t g_has9[65536];
bool g_has9_init = false;
void init_has9() {
  if (g_has9_init) return;
  for (int m = 0; m < 65536; m++) {
    uint32_t d = (uint32_t)m | ((uint32_t)m << 16);
    uint32_t r = d;
    for (int k = 1; k < 9; k++) r &= (d >> k);
    g_has9[m] = (r & 0xFFFF) != 0;
  }
  g_has9_init = true;
}

// Segment test that fast9Detect's decision tree (feature.cpp:3033-5953)
// implements: 9 contiguous ring pixels all > p+b or all < p-b.
inline bool fast9_is_corner(const uint8_t* p, const int off[16], int b) {
  int cb = *p + b, c_b = *p - b;
  uint32_t br = 0, dk = 0;
  for (int i = 0; i < 16; i++) {
    int v = p[off[i]];
    br |= (uint32_t)(v > cb) << i;
    dk |= (uint32_t)(v < c_b) << i;
  }
  return g_has9[br] || g_has9[dk];
}

// feature.cpp:67-2995: binary search over b with the same segment test.
inline int fast9_corner_score(const uint8_t* p, const int off[16], int bstart) {
  int bmin = bstart, bmax = 255, b = (bmax + bmin) / 2;
  for (;;) {
    if (fast9_is_corner(p, off, b)) bmin = b; else bmax = b;
    if (bmin == bmax - 1 || bmin == bmax) return bmin;
    b = (bmin + bmax) / 2;
  }
}

struct Pt { int x, y; };

// feature.cpp:6058-6077 with nonmax_suppression = true (the only mode used).
void FAST(const uint8_t* img, int width, int height, int stride, std::vector<KeyPoint>* out, int thr) {
  init_has9();
  int off[16];
  for (int i = 0; i < 16; i++) off[i] = RING_DX[i] + stride * RING_DY[i];
  std::vector<Pt> corners;
  for (int y = 3; y < height - 3; y++)
    for (int x = 3; x < width - 3; x++)
      if (fast9_is_corner(&img[y * stride + x], off, thr)) corners.push_back({x, y});
  std::vector<int> scores(corners.size());
  for (size_t n = 0; n < corners.size(); n++)
    scores[n] = fast9_corner_score(&img[corners[n].y * stride + corners[n].x], off, thr);

  // fastNonmaxSuppression, feature.cpp:5960-6056 (sparse walk, kept as written)
  out->clear();
  int num_corners = (int)corners.size();
  if (num_corners < 1) return;
  int point_above = 0, point_below = 0;
  int last_row = corners[num_corners - 1].y;
  std::vector<int> row_start(last_row + 1, -1);
  int prev_row = -1;
  for (int i = 0; i < num_corners; i++)
    if (corners[i].y != prev_row) { row_start[corners[i].y] = i; prev_row = corners[i].y; }
  for (int i = 0; i < num_corners; i++) {
    int score = scores[i];
    Pt pos = corners[i];
    if (i > 0 && corners[i - 1].x == pos.x - 1 && corners[i - 1].y == pos.y && scores[i - 1] >= score) continue;
    if (i < num_corners - 1 && corners[i + 1].x == pos.x + 1 && corners[i + 1].y == pos.y &&
        scores[i + 1] >= score)
      continue;
    bool suppressed = false;
    if (pos.y != 0 && row_start[pos.y - 1] != -1) {
      if (corners[point_above].y < pos.y - 1) point_above = row_start[pos.y - 1];
      for (; corners[point_above].y < pos.y && corners[point_above].x < pos.x - 1; point_above++) {}
      for (int j = point_above; corners[j].y < pos.y && corners[j].x <= pos.x + 1; j++) {
        int x = corners[j].x;
        if ((x == pos.x - 1 || x == pos.x || x == pos.x + 1) && scores[j] >= score) { suppressed = true; break; }
      }
      if (suppressed) continue;
    }
    if (pos.y != last_row && row_start[pos.y + 1] != -1 && point_below < num_corners) {
      if (corners[point_below].y < pos.y + 1) point_below = row_start[pos.y + 1];
      for (; point_below < num_corners && corners[point_below].y == pos.y + 1 &&
             corners[point_below].x < pos.x - 1; point_below++) {}
      for (int j = point_below; j < num_corners && corners[j].y == pos.y + 1 && corners[j].x <= pos.x + 1; j++) {
        int x = corners[j].x;
        if ((x == pos.x - 1 || x == pos.x || x == pos.x + 1) && scores[j] >= score) { suppressed = true; break; }
      }
      if (suppressed) continue;
    }
    out->push_back({(float)corners[i].x, (float)corners[i].y, (float)score});
  }
}

// ---------------------------------------------------------------- pyramid
// pyramid.cpp:67-99
void HorizontalGaussFilter(const uint8_t* src, int src_stride, int src_width, int src_height,
                           uint8_t* dest, int dest_stride) {
  int dst_max_row = src_width / 2 - 1;
  for (int dst_col = 0; dst_col < src_height; dst_col++) {
    const uint8_t* s = &src[dst_col * src_stride];
    uint16_t sum = s[0] + 4 * s[1] + 3 * s[2];
    dest[dst_col] = sum >> 3;
    int dst_row;
    for (dst_row = 1; dst_row < dst_max_row; dst_row++) {
      sum = s[0] + 4 * s[1] + 6 * s[2] + 4 * s[3] + s[4];
      dest[dst_row * dest_stride + dst_col] = sum / 16;
      s += 2;
    }
    if (src_width & 1) {
      sum = 3 * s[0] + 4 * s[1] + s[2];
      dest[dst_max_row * dest_stride + dst_col] = sum >> 3;
    } else {
      sum = s[0] + 4 * s[1] + 7 * s[2] + 4 * s[3];
      dest[dst_max_row * dest_stride + dst_col] = sum >> 4;
    }
  }
}

// pyramid.cpp:53-65
void BuildNextLevel(const uint8_t* src, int src_stride, int src_width, int src_height, uint8_t* dest,
                    int dst_stride, uint8_t* buffer) {
  HorizontalGaussFilter(src, src_stride, src_width, src_height, buffer, src_height);
  HorizontalGaussFilter(buffer, src_height, src_height, src_width / 2, dest, dst_stride);
}

// pyramid.cpp:15-51, pyramid.hpp:13-21
struct GridFilter {
  int grid_w, grid_h, max_per, rows, cols;
  std::vector<std::vector<KeyPoint>> grids;
  void init(int img_w, int img_h, int gw, int gh, int mp) {
    grid_w = gw; grid_h = gh; max_per = mp;
    rows = (int)ceil((float)img_h / gh);
    cols = (int)ceil((float)img_w / gw);
    grids.assign((size_t)rows * cols, {});
  }
  void Filter(std::vector<KeyPoint>* kps) {
    for (auto& g : grids) g.clear();
    for (auto& kp : *kps) {
      int32_t iy = kp.v / grid_h, ix = kp.u / grid_w;  // float / int -> float -> trunc
      grids[iy * cols + ix].push_back(kp);
    }
    kps->clear();
    for (auto& g : grids) {
      auto new_end = g.end();
      if (g.size() > (size_t)max_per) {
        new_end = g.begin() + max_per;
        std::nth_element(g.begin(), new_end, g.end(),
                         [](const KeyPoint& l, const KeyPoint& r) { return l.score > r.score; });
      }
      std::copy(g.begin(), new_end, std::back_inserter(*kps));
    }
  }
};

// ---------------------------------------------------------------- descriptor
struct Descriptor {  // feature.cpp:6080-6207
  int stride_img, win, len, stride, pad;
  std::vector<int> offs;
  void init(int raw_gray_stride, int window) {
    stride_img = raw_gray_stride; win = window;
    len = win * win;
    if ((len - 1) % 16 == 0) len -= 1;
    stride = round_up(len, 16);
    pad = stride - len;
    offs.resize(len);
    int i = 0, dr = (win - 1) / 2;
    for (int yo = -dr; yo <= dr; yo++)
      for (int xo = -dr; xo <= dr && i < len; xo++) offs[i++] = yo * stride_img + xo;
  }
  // feature.cpp:6158-6178; the SSE adds/subs run over `stride` bytes (pad included)
  void Normalized(uint8_t* d) const {
    uint32_t mean = std::accumulate(d, d + len, 0) / len;
    if (mean < 128) {
      int o = 128 - mean;
      for (int i = 0; i < stride; i++) { int v = d[i] + o; d[i] = v > 255 ? 255 : v; }
    } else if (mean > 128) {
      int o = mean - 128;
      for (int i = 0; i < stride; i++) { int v = d[i] - o; d[i] = v < 0 ? 0 : v; }
    }
  }
  // feature.cpp:6112-6138
  void ExtractInterp(const uint8_t* g, float x, float y, uint8_t* d) const {
    int32_t u = (int32_t)x, v = (int32_t)y;
    float wright = (x - u), wbottom = (y - v);
    float wa = (1. - wright) * (1. - wbottom);
    float wb = wright * (1. - wbottom);
    float wc = (1. - wright) * wbottom;
    float wd = wright * wbottom;
    int32_t init = v * stride_img + u;
    for (int i = 0; i < len; i++) {
      int32_t k = init + offs[i];
      d[i] = (uint8_t)(wa * g[k] + wb * g[k + 1] + wc * g[k + stride_img] + wd * g[k + stride_img + 1]);
    }
    Normalized(d);
    memset(d + len, 0, pad);
  }
  // feature.cpp:6140-6156
  void ExtractAligned(const uint8_t* g, int x, int y, uint8_t* d) const {
    int32_t start = y * stride_img + x;
    for (int i = 0; i < len; i++) d[i] = g[start + offs[i]];
    Normalized(d);
    memset(d + len, 0, pad);
  }
};

// feature.hpp:364-381
inline int32_t sad_score(const uint8_t* a, const uint8_t* b, int len) {
  int s = 0;
  int n = round_up(len, 16);
  for (int i = 0; i < n; i++) s += abs((int)a[i] - (int)b[i]);
  return s;
}

// ---------------------------------------------------------------- level/frame
struct Level {  // pyramid.cpp:101-147
  int width, height, stride, level_num;
  std::vector<uint8_t> gray, pyrbuf;
  std::vector<KeyPoint> initial;
  int num_detected = 0;
  GridFilter grid;
  std::vector<KeypointData> kps;
  std::vector<uint8_t> desc;
  Descriptor dx;
  int kp_min_x, kp_min_y, kp_max_x, kp_max_y;
  Level(int w, int h, int l, const Params& p) {
    width = w; height = h; level_num = l;
    stride = round_up(w, 16);
    gray.assign((size_t)stride * h, 0);
    pyrbuf.assign((size_t)(w * h) / 2 + 16, 0);
    grid.init(w, h, p.grid_width, p.grid_height, p.max_keypoints_per_grid);
    dx.init(stride, p.feature_window_size);
    kp_min_x = kp_min_y = p.feature_window_size;
    kp_max_x = w - p.feature_window_size - 2;
    kp_max_y = h - p.feature_window_size - 2;
  }
  bool isLegal(float x, float y) const {  // pyramid.hpp:136-139
    return x >= kp_min_x && x <= kp_max_x && y >= kp_min_y && y <= kp_max_y;
  }
  const uint8_t* GetDescriptor(int i) const { return desc.data() + (size_t)i * dx.stride; }
};

struct Frame {  // frame.cpp
  const Camera* cam;
  const Params* prm;
  int width, height, num_levels, win;
  std::vector<Level> levels;
  Depth depth;  // the reference keeps one Depth per VO; one per frame slot is equivalent
  Frame(const Camera* c, const Params* p) : cam(c), prm(p), depth(c->in) {
    width = c->in.width; height = c->in.height;
    num_levels = p->num_pyramid_levels; win = p->feature_window_size;
    for (int l = 0; l < num_levels; l++) levels.emplace_back(width >> l, height >> l, l, *p);
  }
  int NumKeypoints() const { int r = 0; for (auto& l : levels) r += (int)l.kps.size(); return r; }
  int NumDetected() const { int r = 0; for (auto& l : levels) r += l.num_detected; return r; }

  // frame.cpp:37-158
  void Prepare(const uint8_t* raw_gray, const uint16_t* raw_depth, int fast_threshold) {
    depth.SetRaw(raw_depth);
    Level& first = levels[0];
    for (int row = 0; row < height; row++)
      memcpy(first.gray.data() + (size_t)row * first.stride, raw_gray + (size_t)row * width, width);
    for (int ln = 0; ln < num_levels; ln++) {
      Level& lv = levels[ln];
      if (ln > 0) {
        Level& pv = levels[ln - 1];
        BuildNextLevel(pv.gray.data(), pv.stride, pv.width, pv.height, lv.gray.data(), lv.stride,
                       pv.pyrbuf.data());
      }
      lv.initial.clear();
      FAST(lv.gray.data(), lv.width, lv.height, lv.stride, &lv.initial, fast_threshold);
      lv.num_detected = (int)lv.initial.size();
      lv.grid.Filter(&lv.initial);
      lv.kps.clear();
      int mde = (win - 1) / 2 + 1;
      int min_x = mde, min_y = mde, max_x = lv.width - (mde + 1), max_y = lv.height - (mde + 1);
      for (auto& c : lv.initial) {
        if (c.u < min_x || c.u > max_x || c.v < min_y || c.v > max_y) continue;
        KeypointData k;
        k.kp = c;
        k.base_uv[0] = c.u * (1 << ln);
        k.base_uv[1] = c.v * (1 << ln);
        k.pyramid_level = ln;
        int32_t idx = (int32_t)(k.base_uv[1] * width + k.base_uv[0]);
        k.rect_base_uv[0] = cam->Xmap[idx];
        k.rect_base_uv[1] = cam->Ymap[idx];
        if (k.rect_base_uv[0] < 0 || k.rect_base_uv[0] >= width || k.rect_base_uv[1] < 0 ||
            k.rect_base_uv[1] >= height)
          continue;
        int32_t du = (int32_t)(k.rect_base_uv[0] + 0.5), dv = (int32_t)(k.rect_base_uv[1] + 0.5);
        // rect_u in [W-0.5, W) gives du = W (distorted cameras): depth index and ray index (depth.cpp:72, v*W+u) then name
        // column 0 of the next row, which is what the reference reads; past the last row both leave their buffers
        // (undefined in the reference, Eigen's assert aborts there) -- documented deviation: the keypoint is dropped.
        if (depth.RGB2DepthIdx(du, dv) >= depth.dw * depth.dh || dv * width + du >= width * height) continue;
        if (!depth.HaveXYZ(du, dv)) continue;
        k.xyzw[0] = k.xyzw[1] = k.xyzw[2] = k.xyzw[3] = NAN;
        k.has_depth = false;
        k.keypoint_index = (int)lv.kps.size();
        lv.kps.push_back(k);
      }
      // feature.cpp:6193-6207: ExtractAligned (normalises) then Normalized again
      lv.desc.assign(lv.kps.size() * (size_t)lv.dx.stride + 16, 0);
      for (size_t n = 0; n < lv.kps.size(); n++) {
        uint8_t* d = lv.desc.data() + n * lv.dx.stride;
        lv.dx.ExtractAligned(lv.gray.data(), (int)lv.kps[n].kp.u, (int)lv.kps[n].kp.v, d);
        lv.dx.Normalized(d);
      }
    }
    // depth.cpp:48-78
    for (auto& lv : levels)
      for (auto& k : lv.kps) {
        int32_t u = (int32_t)(k.rect_base_uv[0] + 0.5), v = (int32_t)(k.rect_base_uv[1] + 0.5);
        float z = depth.data[depth.RGB2DepthIdx(u, v)];
        if (std::isnan(z)) {
          k.has_depth = false;
          k.xyzw[0] = k.xyzw[1] = k.xyzw[2] = k.xyzw[3] = NAN;
          k.xyz[0] = k.xyz[1] = k.xyz[2] = NAN;
        } else {
          k.has_depth = true;
          const int32_t ri = v * width + u;  // rays_->col(v * rgb_width_ + u), depth.cpp:72
          double rx = (ri % width) * depth.fx_inv + depth.neg_cx_div_fx;  // depth.cpp:32-34 rays
          double ry = (ri / width) * depth.fy_inv + depth.neg_cy_div_fy;
          double zd = z;
          k.xyz[0] = zd * rx; k.xyz[1] = zd * ry; k.xyz[2] = zd * 1.0;
          k.xyzw[0] = k.xyz[0]; k.xyzw[1] = k.xyz[1]; k.xyzw[2] = k.xyz[2]; k.xyzw[3] = 1;
        }
      }
    // frame.cpp:160-192 FilterBadKeypoints (stable compaction + re-index)
    for (auto& lv : levels) {
      size_t w = 0;
      for (size_t r = 0; r < lv.kps.size(); r++) {
        if (!lv.kps[r].has_depth) continue;
        if (w != r) {
          lv.kps[w] = lv.kps[r];
          memcpy(lv.desc.data() + w * lv.dx.stride, lv.desc.data() + r * lv.dx.stride, lv.dx.stride);
        }
        w++;
      }
      lv.kps.resize(w);
      for (size_t i = 0; i < w; i++) lv.kps[i].keypoint_index = (int)i;
    }
  }
};

// ---------------------------------------------------------------- matches
struct FeatureMatch {  // feature.hpp:210-308
  int target_idx = -1, ref_idx = -1, level = 0;
  const KeypointData* target_keypoint = nullptr;
  const KeypointData* ref_keypoint = nullptr;
  KeypointData refined_target_keypoint;
  std::vector<int> consistency_vec;
  int compatibility_degree = 0;
  bool in_maximal_clique = false;
  bool inlier = false;
  int id = 0;
  double reprojection_error = 0;
  int status = MATCH_OK;
  int track_id = -1;
};

// motion_estimate.cpp:723-755 (exact int32 dot product)
inline int dot_i16(const int16_t* a, const int16_t* b, int n) {
  int s = 0;
  for (int i = 0; i < n; i++) s += (int)a[i] * (int)b[i];
  return s;
}

// 6x6 inverse.  Eigen's MatrixBase::inverse() for fixed sizes > 4 goes through
// PartialPivLU; restated as Gaussian elimination with partial pivoting.
bool inverse6(const double A[6][6], double inv[6][6]) {
  double a[6][12];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { a[i][j] = A[i][j]; a[i][6 + j] = (i == j); }
  for (int c = 0; c < 6; c++) {
    int piv = c;
    double best = fabs(a[c][c]);
    for (int r = c + 1; r < 6; r++) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
    if (piv != c) for (int j = 0; j < 12; j++) std::swap(a[c][j], a[piv][j]);
    double d = a[c][c];
    for (int r = c + 1; r < 6; r++) {
      double f = a[r][c] / d;
      for (int j = c; j < 12; j++) a[r][j] -= f * a[c][j];
    }
  }
  for (int c = 5; c >= 0; c--) {
    double d = a[c][c];
    for (int j = 6; j < 12; j++) {
      double s = a[c][j];
      for (int k = c + 1; k < 6; k++) s -= a[c][k] * a[k][j];
      a[c][j] = s / d;
    }
  }
  bool ok = true;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { inv[i][j] = a[i][6 + j]; if (!std::isfinite(inv[i][j])) ok = false; }
  return ok;
}

// motion_estimate.cpp:1053-1123 / 1125-1206: P, Ju, Jv, Jw for both directions
struct JacMats { double P[3][4], Ju[6][4], Jv[6][4], Jw[6][4]; };

void forward_jac_mats(const double prm[6], double fx, double px, double py, JacMats* m) {
  double tx = prm[0], ty = prm[1], tz = prm[2];
  double sr = sin(prm[3]), cr = cos(prm[3]), sp = sin(prm[4]), cp = cos(prm[4]), sy = sin(prm[5]),
         cy = cos(prm[5]);
  double P[3][4] = {
      {fx * cp * cy - px * sp, px * cp * sr - fx * (cr * sy - cy * sp * sr),
       fx * (sr * sy + cr * cy * sp) + px * cp * cr, fx * tx + px * tz},
      {fx * cp * sy - py * sp, fx * (cr * cy + sp * sr * sy) + py * cp * sr,
       py * cp * cr - fx * (cy * sr - cr * sp * sy), fx * ty + py * tz},
      {-sp, cp * sr, cp * cr, tz}};
  double Ju[6][4] = {{0, 0, 0, fx},
                     {0, 0, 0, 0},
                     {0, 0, 0, px},
                     {0, fx * (sr * sy + cr * cy * sp) + px * cp * cr, fx * (cr * sy - cy * sp * sr) - px * cp * sr, 0},
                     {-px * cp - fx * cy * sp, -sr * (px * sp - fx * cp * cy), -cr * (px * sp - fx * cp * cy), 0},
                     {-fx * cp * sy, -fx * cr * cy - fx * sp * sr * sy, fx * cy * sr - fx * cr * sp * sy, 0}};
  double Jv[6][4] = {{0, 0, 0, 0},
                     {0, 0, 0, fx},
                     {0, 0, 0, py},
                     {0, py * cp * cr - fx * (cy * sr - cr * sp * sy), -fx * (cr * cy + sp * sr * sy) - py * cp * sr, 0},
                     {-py * cp - fx * sp * sy, -sr * (py * sp - fx * cp * sy), -cr * (py * sp - fx * cp * sy), 0},
                     {fx * cp * cy, fx * cy * sp * sr - fx * cr * sy, fx * sr * sy + fx * cr * cy * sp, 0}};
  double Jw[6][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 1}, {0, cp * cr, -cp * sr, 0},
                     {-cp, -sp * sr, -cr * sp, 0}, {0, 0, 0, 0}};
  memcpy(m->P, P, sizeof(P)); memcpy(m->Ju, Ju, sizeof(Ju)); memcpy(m->Jv, Jv, sizeof(Jv));
  memcpy(m->Jw, Jw, sizeof(Jw));
}

void reverse_jac_mats(const double prm[6], double fx, double px, double py, JacMats* m) {
  double tx = prm[0], ty = prm[1], tz = prm[2];
  double sr = sin(prm[3]), cr = cos(prm[3]), sp = sin(prm[4]), cp = cos(prm[4]), sy = sin(prm[5]),
         cy = cos(prm[5]);
  double P[3][4] = {
      {px * (sr * sy + cr * cy * sp) + fx * cp * cy, fx * cp * sy - px * (cy * sr - cr * sp * sy),
       px * cp * cr - fx * sp,
       -px * (tx * (sr * sy + cr * cy * sp) - ty * (cy * sr - cr * sp * sy) + tz * cp * cr) -
           fx * (tx * cp * cy - tz * sp + ty * cp * sy)},
      {py * (sr * sy + cr * cy * sp) - fx * (cr * sy - cy * sp * sr),
       fx * (cr * cy + sp * sr * sy) - py * (cy * sr - cr * sp * sy), cp * (py * cr + fx * sr),
       -py * (tx * (sr * sy + cr * cy * sp) - ty * (cy * sr - cr * sp * sy) + tz * cp * cr) -
           fx * (ty * (cr * cy + sp * sr * sy) - tx * (cr * sy - cy * sp * sr) + tz * cp * sr)},
      {sr * sy + cr * cy * sp, cr * sp * sy - cy * sr, cp * cr,
       ty * (cy * sr - cr * sp * sy) - tx * (sr * sy + cr * cy * sp) - tz * cp * cr}};
  double Ju[6][4] = {
      {0, 0, 0, -px * (sr * sy + cr * cy * sp) - fx * cp * cy},
      {0, 0, 0, px * (cy * sr - cr * sp * sy) - fx * cp * sy},
      {0, 0, 0, fx * sp - px * cp * cr},
      {px * cr * sy - px * cy * sp * sr, -px * cr * cy - px * sp * sr * sy, -px * cp * sr,
       px * ty * (cr * cy + sp * sr * sy) - px * tx * (cr * sy - cy * sp * sr) + px * tz * cp * sr},
      {-cy * (fx * sp - px * cp * cr), -sy * (fx * sp - px * cp * cr), -fx * cp - px * cr * sp,
       fx * (tz * cp + tx * cy * sp + ty * sp * sy) - px * (tx * cp * cr * cy - tz * cr * sp + ty * cp * cr * sy)},
      {px * (cy * sr - cr * sp * sy) - fx * cp * sy, px * (sr * sy + cr * cy * sp) + fx * cp * cy, 0,
       -fx * (ty * cp * cy - tx * cp * sy) - px * (tx * (cy * sr - cr * sp * sy) + ty * (sr * sy + cr * cy * sp))}};
  double Jv[6][4] = {
      {0, 0, 0, fx * (cr * sy - cy * sp * sr) - py * (sr * sy + cr * cy * sp)},
      {0, 0, 0, py * (cy * sr - cr * sp * sy) - fx * (cr * cy + sp * sr * sy)},
      {0, 0, 0, -cp * (py * cr + fx * sr)},
      {fx * (sr * sy + cr * cy * sp) + py * (cr * sy - cy * sp * sr),
       -fx * (cy * sr - cr * sp * sy) - py * (cr * cy + sp * sr * sy), cp * (fx * cr - py * sr),
       py * (ty * (cr * cy + sp * sr * sy) - tx * (cr * sy - cy * sp * sr) + tz * cp * sr) -
           fx * (tx * (sr * sy + cr * cy * sp) - ty * (cy * sr - cr * sp * sy) + tz * cp * cr)},
      {cp * cy * (py * cr + fx * sr), cp * sy * (py * cr + fx * sr), -sp * (py * cr + fx * sr),
       -(py * cr + fx * sr) * (tx * cp * cy - tz * sp + ty * cp * sy)},
      {py * (cy * sr - cr * sp * sy) - fx * (cr * cy + sp * sr * sy),
       py * (sr * sy + cr * cy * sp) - fx * (cr * sy - cy * sp * sr), 0,
       fx * (tx * (cr * cy + sp * sr * sy) + ty * (cr * sy - cy * sp * sr)) -
           py * (tx * (cy * sr - cr * sp * sy) + ty * (sr * sy + cr * cy * sp))}};
  double Jw[6][4] = {
      {0, 0, 0, -sr * sy - cr * cy * sp},
      {0, 0, 0, cy * sr - cr * sp * sy},
      {0, 0, 0, -cp * cr},
      {cr * sy - cy * sp * sr, -cr * cy - sp * sr * sy, -cp * sr,
       ty * (cr * cy + sp * sr * sy) - tx * (cr * sy - cy * sp * sr) + tz * cp * sr},
      {cp * cr * cy, cp * cr * sy, -cr * sp, -cr * (tx * cp * cy - tz * sp + ty * cp * sy)},
      {cy * sr - cr * sp * sy, sr * sy + cr * cy * sp, 0,
       -tx * (cy * sr - cr * sp * sy) - ty * (sr * sy + cr * cy * sp)}};
  memcpy(m->P, P, sizeof(P)); memcpy(m->Ju, Ju, sizeof(Ju)); memcpy(m->Jv, Jv, sizeof(Jv));
  memcpy(m->Jw, Jw, sizeof(Jw));
}

inline double dot4(const double* r, const double* p) {
  if (g_pert.norm_order & 2) return (r[0] * p[0] + r[1] * p[1]) + (r[2] * p[2] + r[3] * p[3]);  // pairwise (vectorised Eigen)
  return ((r[0] * p[0] + r[1] * p[1]) + r[2] * p[2]) + r[3] * p[3];
}

// rows (2 per point) of the Jacobian: motion_estimate.cpp:1109-1122 / 1192-1205
void jac_rows(const JacMats& m, const double* pt, double row_u[6], double row_v[6]) {
  double u = dot4(m.P[0], pt), v = dot4(m.P[1], pt), w = dot4(m.P[2], pt);
  double w_inv_sq = 1 / (w * w);
  for (int k = 0; k < 6; k++) {
    double du = dot4(m.Ju[k], pt), dv = dot4(m.Jv[k], pt), dw = dot4(m.Jw[k], pt);
    row_u[k] = (du * w - dw * u) * w_inv_sq;
    row_v[k] = (dv * w - dw * v) * w_inv_sq;
  }
}

// K * motion.matrix(), K = [fx 0 px 0; 0 fy py 0; 0 0 1 0]
void proj_matrix(double fx, double fy, double px, double py, const Iso& m, double P[3][4]) {
  double K[3][4] = {{fx, 0, px, 0}, {0, fy, py, 0}, {0, 0, 1, 0}};
  double M[4][4] = {{m.R[0][0], m.R[0][1], m.R[0][2], m.t[0]},
                    {m.R[1][0], m.R[1][1], m.R[1][2], m.t[1]},
                    {m.R[2][0], m.R[2][1], m.R[2][2], m.t[2]},
                    {0, 0, 0, 1}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++)
      P[i][j] = ((K[i][0] * M[0][j] + K[i][1] * M[1][j]) + K[i][2] * M[2][j]) + K[i][3] * M[3][j];
}

// motion_estimate.cpp:950-972
void reproj_error(const std::vector<double>& pts, const std::vector<double>& proj, double fx, double px,
                  double py, const Iso& motion, double* err) {
  double P[3][4];
  proj_matrix(fx, fx, px, py, motion, P);  // NB: fx on both axes (motion_estimate.cpp:1221-1223)
  size_t n = pts.size() / 4;
  for (size_t i = 0; i < n; i++) {
    const double* p = &pts[4 * i];
    double u = dot4(P[0], p), v = dot4(P[1], p), w = dot4(P[2], p);
    err[2 * i + 0] = u / w - proj[2 * i + 0];
    err[2 * i + 1] = v / w - proj[2 * i + 1];
  }
}

// ---------------------------------------------------------------- estimator
struct MotionEstimator {
  const Camera* cam;
  const Params* prm;
  std::vector<FeatureMatch> matches;
  int num_inliers = 0;
  double mean_reprojection_error = 0;
  int num_reprojection_failures = 0;
  Iso motion;
  double cov[6][6];
  int status = NO_DATA;
  Frame* ref = nullptr;
  Frame* tgt = nullptr;
  Margins mg;
  int rot_iters = 0, rot_best_iter = 0;  // last EstimateInitialRotation: iterations run, iteration whose H was returned
  int num_tracks = 0;  // motion_estimate.cpp:116: one counter per MotionEstimator (per RGBDVO / per FrameMatcher)

  MotionEstimator(const Camera* c, const Params* p) : cam(c), prm(p) {
    motion.setIdentity();
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) cov[i][j] = (i == j);
  }

  // motion_estimate.cpp:757-900
  void refineFeatureMatch(const Level& rl, const Level& tl, const double ref_uv[2], const double init_uv[2],
                          double final_uv[2]) {
    const uint8_t* ref_gray = rl.gray.data();
    int rs = rl.stride;
    int len = rl.dx.len, dstride = rl.dx.stride;
    std::vector<uint8_t> ref_desc(dstride), tgt_desc(dstride);
    rl.dx.ExtractInterp(ref_gray, (float)ref_uv[0], (float)ref_uv[1], ref_desc.data());
    int nel = round_up(len * 2, 16) / 2;
    tl.dx.ExtractInterp(tl.gray.data(), (float)init_uv[0], (float)init_uv[1], tgt_desc.data());
    float tx = (float)init_uv[0], ty = (float)init_uv[1];
    const uint8_t* tg = tl.gray.data();
    int ts = tl.stride;
    std::vector<int16_t> err(nel, 0), rdx(nel, 0), rdy(nel, 0), Mx(nel, 0), My(nel, 0);
    for (int i = 0; i < len; i++) err[i] = tgt_desc[i] - ref_desc[i];
    int final_sse = dot_i16(err.data(), err.data(), nel);
    int rdesc_offset = (int)(ref_uv[1] * rs + ref_uv[0]);
    for (int i = 0; i < len; i++) {
      int k = rdesc_offset + rl.dx.offs[i];
      rdx[i] = ref_gray[k + 1] - ref_gray[k];
      rdy[i] = ref_gray[k + rs] - ref_gray[k];
    }
    for (int it = 0; it < 6; it++) {
      int tc = ((int)ty) * ts + (int)tx;
      for (int i = 0; i < len; i++) {
        int k = tc + tl.dx.offs[i];
        Mx[i] = rdx[i] + (tg[k + 1] - tg[k]);
        My[i] = rdy[i] + (tg[k + ts] - tg[k]);
      }
      double S00 = dot_i16(Mx.data(), Mx.data(), nel);
      double S01 = dot_i16(Mx.data(), My.data(), nel);
      double S11 = dot_i16(My.data(), My.data(), nel);
      double det = S00 * S11 - S01 * S01;
      if (det <= 1e-9) break;
      double Si00 = S11 / det, Si01 = -S01 / det, Si11 = S00 / det;
      double gx = dot_i16(Mx.data(), err.data(), nel);
      double gy = dot_i16(My.data(), err.data(), nel);
      double dxs = -2 * (Si00 * gx + Si01 * gy);
      double dys = -2 * (Si01 * gx + Si11 * gy);
      float ntx = tx + dxs, nty = ty + dys;
      if (!tl.isLegal(ntx, nty)) break;
      tl.dx.ExtractInterp(tg, ntx, nty, tgt_desc.data());
      for (int i = 0; i < len; i++) err[i] = tgt_desc[i] - ref_desc[i];
      int sse = dot_i16(err.data(), err.data(), nel);
      mg.note(&mg.sse_tie, fabs((double)sse - (double)final_sse));
      if (sse < final_sse) { tx = ntx; ty = nty; final_sse = sse; } else break;
      mg.note(&mg.esm_step, fabs(std::max(fabs(dxs), fabs(dys)) - 0.1));
      mg.n_esm++;
      if (fabs(dxs) < 0.1 && fabs(dys) < 0.1) break;
    }
    final_uv[0] = tx; final_uv[1] = ty;
  }

  // motion_estimate.cpp:307-393 + feature.cpp:6239-6335
  void matchFeatures(int li) {
    Level& rl = ref->levels[li];
    Level& tl = tgt->levels[li];
    const Intrinsics& rc = cam->rect;
    Iso r2t = iso_inverse(motion);
    double RP[3][4];
    proj_matrix(rc.fx, rc.fy, rc.cx, rc.cy, r2t, RP);
    int nr = (int)rl.kps.size(), nt = (int)tl.kps.size();
    std::vector<std::vector<int>> cand(nr);
    for (int r = 0; r < nr; r++) {
      const double* x = rl.kps[r].xyzw;
      double uvw[3] = {dot4(RP[0], x), dot4(RP[1], x), dot4(RP[2], x)};
      double inv = 1.0 / uvw[2];  // Eigen 3.2: v /= s  ==  v *= 1/s
      double pu = uvw[0] * inv, pv = uvw[1] * inv;
      if (g_pert.true_division) { pu = uvw[0] / uvw[2]; pv = uvw[1] / uvw[2]; }
      for (int t = 0; t < nt; t++) {
        double tu = (float)tl.kps[t].rect_base_uv[0];  // const float getters, pyramid.hpp:95-101
        double tv = (float)tl.kps[t].rect_base_uv[1];
        double ex = tu - pu, ey = tv - pv;
        double dist = sqrt(ex * ex + ey * ey);
        mg.note(&mg.window, fabs(dist - prm->feature_search_window));
        mg.n_window++;
        if (dist < prm->feature_search_window) cand[r].push_back(t);
      }
    }
    // FeatureMatcher::matchFeatures
    int len = rl.dx.len;
    int worst = len * 255;
    std::vector<int> r2t_s(nr, worst + 1), r2t_i(nr, -1), t2r_s(nt, worst + 1), t2r_i(nt, -1);
    for (int r = 0; r < nr; r++)
      for (int t : cand[r]) {
        int s = sad_score(rl.GetDescriptor(r), tl.GetDescriptor(t), len);
        if (s < r2t_s[r]) { r2t_s[r] = s; r2t_i[r] = t; }
        if (s < t2r_s[t]) { t2r_s[t] = s; t2r_i[t] = r; }
      }
    size_t old = matches.size();
    for (int r = 0; r < nr; r++) {
      int t = r2t_i[r];
      if (t >= 0 && t2r_i[t] == r) {
        FeatureMatch m;
        m.ref_idx = r; m.target_idx = t; m.level = li;
        m.ref_keypoint = &rl.kps[r];
        m.target_keypoint = &tl.kps[t];
        m.refined_target_keypoint = tl.kps[t];
        m.status = MATCH_OK;
        matches.push_back(m);
      }
    }
    // sub-pixel refinement, motion_estimate.cpp:347-378
    for (size_t n = old; n < matches.size(); n++) {
      FeatureMatch& m = matches[n];
      double ruv[2] = {m.ref_keypoint->kp.u, m.ref_keypoint->kp.v};
      double iuv[2] = {m.target_keypoint->kp.u, m.target_keypoint->kp.v};
      double fuv[2];
      refineFeatureMatch(rl, tl, ruv, iuv, fuv);
      double dx = iuv[0] - fuv[0], dy = iuv[1] - fuv[1];
      double ds = sqrt(dx * dx + dy * dy);
      if (ds > 0) mg.note(&mg.refine_gate, fabs(ds - 1.5));
      if (ds < 1e-9 || ds > 1.5) {
        m.refined_target_keypoint = *m.target_keypoint;
        m.status = MATCH_OK;
      } else {
        m.refined_target_keypoint.kp.u = (float)fuv[0];
        m.refined_target_keypoint.kp.v = (float)fuv[1];
        m.refined_target_keypoint.base_uv[0] = fuv[0] * (1 << m.target_keypoint->pyramid_level);
        m.refined_target_keypoint.base_uv[1] = fuv[1] * (1 << m.target_keypoint->pyramid_level);
        cam->BilinearLookup(m.refined_target_keypoint.base_uv, m.refined_target_keypoint.rect_base_uv);
        m.status = MATCH_NEEDS_DEPTH_REFINEMENT;
      }
    }
    // label matches with their track id, motion_estimate.cpp:380-391
    for (size_t n = old; n < matches.size(); n++) {
      FeatureMatch& m = matches[n];
      if (m.ref_keypoint->track_id < 0) m.ref_keypoint->track_id = num_tracks++;
      m.target_keypoint->track_id = m.ref_keypoint->track_id;
      m.track_id = m.ref_keypoint->track_id;
    }
  }

  // motion_estimate.cpp:442-570
  void computeMaximallyConsistentClique() {
    int N = (int)matches.size();
    if (!N) { num_inliers = 0; return; }  // reference leaves a stale count here (UB-adjacent); SURVEY.md H6
    for (auto& m : matches) m.consistency_vec.assign(N, 0);
    double thr = prm->clique_inlier_threshold;
    for (int i = 0; i < N; i++) {
      FeatureMatch& a = matches[i];
      const double* r1 = a.ref_keypoint->xyz;
      const double* t1 = a.refined_target_keypoint.xyz;
      bool ref_inf = a.ref_keypoint->xyzw[3] < 1e-9, tgt_inf = a.refined_target_keypoint.xyzw[3] < 1e-9;
      for (int j = i + 1; j < N; j++) {
        FeatureMatch& b = matches[j];
        const double* r2 = b.ref_keypoint->xyz;
        const double* t2 = b.refined_target_keypoint.xyz;
        bool consistent;
        if ((ref_inf && b.ref_keypoint->xyzw[3] < 1e-9) || (tgt_inf && b.refined_target_keypoint.xyzw[3] < 1e-9)) {
          consistent = true;
        } else {
          double dx = r2[0] - r1[0], dy = r2[1] - r1[1], dz = r2[2] - r1[2];
          double ref_dist = (g_pert.norm_order & 1) ? sqrt((dx * dx + dy * dy) + dz * dz) : sqrt(dx * dx + (dy * dy + dz * dz));
          dx = t2[0] - t1[0]; dy = t2[1] - t1[1]; dz = t2[2] - t1[2];
          double tgt_dist = (g_pert.norm_order & 1) ? sqrt((dx * dx + dy * dy) + dz * dz) : sqrt(dx * dx + (dy * dy + dz * dz));
          consistent = fabs(ref_dist - tgt_dist) < thr;
          mg.note(&mg.clique, fabs(fabs(ref_dist - tgt_dist) - thr));  // NaN distances (no depth) never lower the minimum
          mg.n_clique++;
        }
        if (consistent) {
          a.consistency_vec[b.id] = 1; a.compatibility_degree++;
          b.consistency_vec[a.id] = 1; b.compatibility_degree++;
        }
      }
    }
    std::sort(matches.begin(), matches.end(), [](const FeatureMatch& x, const FeatureMatch& y) {
      return x.compatibility_degree > y.compatibility_degree;
    });
    FeatureMatch& best = matches[0];
    best.in_maximal_clique = true; best.inlier = true;
    num_inliers = 1;
    std::vector<int> reject(N, 0);
    for (int m = 1; m < N; m++) { int o = matches[m].id; if (!best.consistency_vec[o]) reject[o] = 1; }
    for (int m = 1; m < N; m++) {
      FeatureMatch& c = matches[m];
      if (c.compatibility_degree < num_inliers) break;
      if (reject[c.id]) continue;
      c.in_maximal_clique = true; c.inlier = true;
      num_inliers++;
      for (int r = m + 1; r < N; r++) { int o = matches[r].id; if (!reject[o] && !c.consistency_vec[o]) reject[o] = 1; }
    }
    // std::sort is not stable: where two neighbours of the sorted order have equal degrees and different verdicts, the
    // verdict may hang on the order libstdc++ leaves them in (which the HIP path replays); report the smallest such gap
    for (int m = 1; m < N; m++)
      if (matches[m].inlier != matches[m - 1].inlier)
        mg.note(&mg.degree_gap, (double)(matches[m - 1].compatibility_degree - matches[m].compatibility_degree));
  }

  // motion_estimate.cpp:610-682 -> 1208-1331
  void refineMotionEstimate() {
    if (num_inliers < prm->min_features_for_estimate) { status = INSUFFICIENT_INLIERS; return; }
    std::vector<double> txyz, tproj, rxyz, rproj;
    for (auto& m : matches) {
      if (!m.inlier) continue;
      for (int k = 0; k < 4; k++) txyz.push_back(m.refined_target_keypoint.xyzw[k]);
      for (int k = 0; k < 2; k++) tproj.push_back(m.refined_target_keypoint.rect_base_uv[k]);
      for (int k = 0; k < 4; k++) rxyz.push_back(m.ref_keypoint->xyzw[k]);
      for (int k = 0; k < 2; k++) rproj.push_back(m.ref_keypoint->rect_base_uv[k]);
    }
    const Intrinsics& rp = cam->rect;
    double fx = rp.fx, px = rp.cx, py = rp.cy;
    size_t n = txyz.size() / 4;
    double est[6];
    iso_to_params(motion, est);
    std::vector<double> err(4 * n);
    reproj_error(txyz, rproj, fx, px, py, motion, &err[0]);
    reproj_error(rxyz, tproj, fx, px, py, iso_inverse(motion), &err[2 * n]);
    auto sse_of = [&](const std::vector<double>& e) { double s = 0; for (double v : e) s += v * v; return s; };
    double final_sse = sse_of(err);
    std::vector<double> M(4 * n * 6);
    auto build_M = [&](const double* p) {
      JacMats f, r;
      forward_jac_mats(p, fx, px, py, &f);
      reverse_jac_mats(p, fx, px, py, &r);
      for (size_t i = 0; i < n; i++) jac_rows(f, &txyz[4 * i], &M[(2 * i) * 6], &M[(2 * i + 1) * 6]);
      for (size_t i = 0; i < n; i++) jac_rows(r, &rxyz[4 * i], &M[(2 * n + 2 * i) * 6], &M[(2 * n + 2 * i + 1) * 6]);
    };
    auto MtM_of = [&](double A[6][6]) {
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
          double s = 0;
          for (size_t r = 0; r < 4 * n; r++) s += M[r * 6 + a] * M[r * 6 + b];
          A[a][b] = s;
        }
    };
    for (int it = 0; it < 6; it++) {
      build_M(est);
      double g[6], A[6][6], Ainv[6][6];
      for (int a = 0; a < 6; a++) { double s = 0; for (size_t r = 0; r < 4 * n; r++) s += M[r * 6 + a] * err[r]; g[a] = s; }
      MtM_of(A);
      inverse6(A, Ainv);
      double delta[6], next[6];
      for (int a = 0; a < 6; a++) {
        double s = 0;
        for (int b = 0; b < 6; b++) s += (-Ainv[a][b]) * g[b];
        delta[a] = s;
        next[a] = est[a] + delta[a];
      }
      Iso nest = iso_from_params(next);
      reproj_error(txyz, rproj, fx, px, py, nest, &err[0]);
      reproj_error(rxyz, tproj, fx, px, py, iso_inverse(nest), &err[2 * n]);
      double sse = sse_of(err);
      if (final_sse > 0) mg.note(&mg.gn_sse_rel, fabs(sse - final_sse) / final_sse);
      if (sse > final_sse) break;  // NaN sse falls through, as in the reference
      memcpy(est, next, sizeof(est));
      final_sse = sse;
      motion = nest;
      const double rt = 0.01 * M_PI / 180;
      if (fabs(delta[0]) < 0.0001 && fabs(delta[1]) < 0.0001 && fabs(delta[2]) < 0.0001 && fabs(delta[3]) < rt &&
          fabs(delta[4]) < rt && fabs(delta[5]) < rt)
        break;
    }
    build_M(est);
    double A[6][6];
    MtM_of(A);
    inverse6(A, cov);
  }

  // motion_estimate.cpp:684-720
  void computeReprojectionError() {
    if (status != SUCCESS) return;
    const Intrinsics& rp = cam->rect;
    double P[3][4];
    proj_matrix(rp.fx, rp.fy, rp.cx, rp.cy, motion, P);
    mean_reprojection_error = 0;
    for (auto& m : matches) {
      if (!m.inlier) continue;
      const double* x = m.refined_target_keypoint.xyzw;
      double h0 = dot4(P[0], x), h1 = dot4(P[1], x), h2 = dot4(P[2], x);
      double ex = m.ref_keypoint->rect_base_uv[0] - h0 / h2;
      double ey = m.ref_keypoint->rect_base_uv[1] - h1 / h2;
      m.reprojection_error = sqrt(ex * ex + ey * ey);
      mean_reprojection_error += m.reprojection_error;
    }
    mean_reprojection_error /= num_inliers;
  }

  // motion_estimate.cpp:149-243
  void EstimateMotion(Frame* r, Frame* t, const Iso& init) {
    ref = r; tgt = t;
    motion = init;
    // *_motion_estimate_covariance = init_motion_cov: both callers pass an empty (0x0) MatrixXd
    // (vo.cpp:115, fm.cpp:42-43); reported here as all zeros.  _mean_reprojection_error is never
    // initialised by the reference ctor and goes stale across calls: reset to 0.
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) cov[i][j] = 0;
    mean_reprojection_error = 0;
    matches.clear();
    mg = Margins();
    for (int l = 0; l < ref->num_levels; l++) matchFeatures(l);
    // Depth::RefineXYZ, depth.cpp:80-95 (depth source = the target frame's depth image)
    for (auto& m : matches)
      if (m.status == MATCH_NEEDS_DEPTH_REFINEMENT) {
        if (tgt->depth.GetXYZInterp(&m.refined_target_keypoint)) m.status = MATCH_OK;
        else { m.status = MATCH_REFINEMENT_FAILED; m.inlier = false; }
      }
    for (size_t i = 0; i < matches.size(); i++) matches[i].id = (int)i;
    computeMaximallyConsistentClique();
    status = SUCCESS;  // USE_ROBUST_STEREO_COMPATIBILITY branch, motion_estimate.cpp:190-192
    refineMotionEstimate();
    computeReprojectionError();
    num_reprojection_failures = 0;
    for (auto& m : matches) {
      if (!m.inlier) continue;
      if (status == SUCCESS) mg.note(&mg.reproj_gate, fabs(m.reprojection_error - prm->inlier_max_reprojection_error));
      if (m.reprojection_error > prm->inlier_max_reprojection_error) {
        m.inlier = false; num_inliers--; num_reprojection_failures++;
      }
    }
    // no track id travels through an outlier match, motion_estimate.cpp:216-222
    for (auto& m : matches) if (!m.inlier) m.target_keypoint->track_id = -1;
    refineMotionEstimate();
    computeReprojectionError();
    if (status == SUCCESS) mg.note(&mg.mean_reproj, fabs(mean_reprojection_error - prm->max_mean_reprojection_error));
    if (mean_reprojection_error > prm->max_mean_reprojection_error) status = REPROJECTION_ERROR;
  }

  // motion_estimate.cpp:268-304 + 1333-1571.  The float JacobiSVD pseudo-inverse
  // (1454-1470) is restated as fp64 normal equations and Matrix3f::exp() (1535) as
  // the closed form of exp([[0,a,x],[-a,0,y],[0,0,0]]): tolerance-class (H3).
  Quat EstimateInitialRotation(const Frame* prev, const Frame* cur) {
    const int irl = 4;
    int pyr = std::min(prev->num_levels - 1, irl);
    const Level& rl = prev->levels[pyr];
    const Level& tl = cur->levels[pyr];
    int ds = irl - pyr;
    int cols = rl.width >> ds, rows = rl.height >> ds;
    auto to_img = [&](const Level& L, std::vector<float>& im) {  // grayToEigen 1338-1367, row-major here
      im.assign((size_t)rows * cols, 0.f);
      if (ds > 0) {
        for (int y = 0; y < L.height; y++) {
          int ey = y >> ds;
          if (ey >= rows) continue;
          for (int x = 0; x < L.width; x++) {
            int ex = x >> ds;
            if (ex >= cols) continue;
            im[(size_t)ey * cols + ex] += L.gray[(size_t)y * L.stride + x];
          }
        }
        double pf = (1 << ds);
        for (auto& v : im) v = (float)(v / (pf * pf));
      } else {
        for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) im[(size_t)y * cols + x] = L.gray[(size_t)y * L.stride + x];
      }
    };
    std::vector<float> tmpl, test;
    to_img(rl, tmpl);
    to_img(tl, test);
    auto gradient = [&](const std::vector<float>& im, std::vector<float>& gx, std::vector<float>& gy) {  // 1494-1517
      gx.assign(im.size(), 0.f); gy.assign(im.size(), 0.f);
      for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
          float vx, vy;
          if (x == 0) vx = im[y * cols + 1] - im[y * cols];
          else if (x == cols - 1) vx = im[y * cols + x] - im[y * cols + x - 1];
          else vx = (float)((im[y * cols + x + 1] - im[y * cols + x - 1]) / 2.0);
          if (y == 0) vy = im[cols + x] - im[x];
          else if (y == rows - 1) vy = im[y * cols + x] - im[(y - 1) * cols + x];
          else vy = (float)((im[(y + 1) * cols + x] - im[(y - 1) * cols + x]) / 2.0);
          gx[y * cols + x] = vx; gy[y * cols + x] = vy;
        }
    };
    std::vector<float> tdx, tdy;
    gradient(tmpl, tdx, tdy);
    float H[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, bestH[3][3];
    memcpy(bestH, H, sizeof(H));
    double minError = INFINITY, lastRMS = INFINITY;
    int bestIter = 0, lastImproved = 0;
    std::vector<float> warped(tmpl.size()), wdx, wdy;
    size_t npx = tmpl.size();
    for (int iter = 0; iter < 8; iter++) {
      // constructWarpedImage, 1538-1571
      for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
          float hx = (H[0][0] * x + H[0][1] * y) + H[0][2] * 1.f;
          float hy = (H[1][0] * x + H[1][1] * y) + H[1][2] * 1.f;
          float hw = (H[2][0] * x + H[2][1] * y) + H[2][2] * 1.f;
          float px = hx / hw, py = hy / hw;
          int fx0 = (int)floorf(px), fy0 = (int)floorf(py), cx0 = (int)ceilf(px), cy0 = (int)ceilf(py);
          double val;
          if (0 <= px && px < cols - 1 && 0 <= py && py < rows - 1) {
            double x1 = px - fx0, y1 = py - fy0, x2 = 1 - x1, y2 = 1 - y1;
            val = x2 * y2 * test[fy0 * cols + fx0] + x1 * y2 * test[fy0 * cols + fx0 + 1] +
                  x2 * y1 * test[(fy0 + 1) * cols + fx0] + x1 * y1 * test[(fy0 + 1) * cols + fx0 + 1];
          } else if (0 <= fx0 && fx0 < cols && 0 <= fy0 && fy0 < rows) {
            val = test[fy0 * cols + fx0];
          } else if (0 <= cx0 && cx0 < cols && 0 <= cy0 && cy0 < rows) {
            val = test[cy0 * cols + cx0];
          } else {
            val = 128;
          }
          warped[y * cols + x] = (float)val;
        }
      // error image + RMS (computeError 1411-1414: float norm / sqrt(N))
      double ss = 0;
      std::vector<float> e(npx);
      for (size_t i = 0; i < npx; i++) { e[i] = warped[i] - tmpl[i]; ss += (double)e[i] * e[i]; }
      double rms = sqrt(ss) / sqrt((double)npx);
      if (rms < minError) { minError = rms; memcpy(bestH, H, sizeof(H)); bestIter = iter; }
      gradient(warped, wdx, wdy);
      // Jacobian columns [dx, dy, dx*(row+1) - dy*(col+1)] (1519-1526, xx_/yy_ 1396-1407)
      double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, b[3] = {0, 0, 0};
      // column-major pixel order as in the flattened Eigen arrays
      for (int x = 0; x < cols; x++)
        for (int y = 0; y < rows; y++) {
          size_t i = (size_t)y * cols + x;
          float jx = tdx[i] + wdx[i], jy = tdy[i] + wdy[i];
          float j2 = jx * (float)(y + 1) - jy * (float)(x + 1);
          double J[3] = {jx, jy, j2};
          for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) A[a][c] += J[a] * J[c]; b[a] += J[a] * e[i]; }
          if (g_pert.rot_float)  // every partial sum rounded to fp32
            for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) A[a][c] = (float)A[a][c]; b[a] = (float)b[a]; }
        }
      double lie[3] = {0, 0, 0};
      double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                   A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
      double scale = A[0][0] * A[1][1] * A[2][2];
      if (fabs(det) > 1e-12 * fabs(scale) && std::isfinite(det) && det != 0) {
        double inv[3][3];
        inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
        inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
        inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
        inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
        inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
        inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
        inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
        inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
        inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
        if (g_pert.rot_float) for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) inv[a][c] = (float)inv[a][c];
        for (int a = 0; a < 3; a++) lie[a] = (float)(-2 * ((inv[a][0] * b[0] + inv[a][1] * b[1]) + inv[a][2] * b[2]));
      }
      // lieToH 1528-1536: exp of [[0,l2,l0],[-l2,0,l1],[0,0,0]]
      double a = lie[2], sa, ca1;  // sa = sin(a)/a, ca1 = (1-cos a)/a
      if (fabs(a) < 1e-8) { sa = 1 - a * a / 6; ca1 = a / 2; } else { sa = sin(a) / a; ca1 = (1 - cos(a)) / a; }
      float E[3][3] = {{(float)cos(a), (float)sin(a), (float)(sa * lie[0] + ca1 * lie[1])},
                       {(float)-sin(a), (float)cos(a), (float)(-ca1 * lie[0] + sa * lie[1])},
                       {0, 0, 1}};
      float Hn[3][3];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Hn[r][c] = (H[r][0] * E[0][c] + H[r][1] * E[1][c]) + H[r][2] * E[2][c];
      memcpy(H, Hn, sizeof(H));
      if (rms < lastRMS) lastImproved = iter;
      double lnorm = sqrt((double)((float)lie[0] * (float)lie[0] + (float)lie[1] * (float)lie[1] + (float)lie[2] * (float)lie[2]));
      rot_iters = iter + 1; rot_best_iter = bestIter;  // (test instrumentation: orc_initial_rotation_info)
      if (lnorm < 1e-6 || (rms - minError > 3 && iter - bestIter > 2) || iter - bestIter > 4 || iter - lastImproved > 2) break;
      lastRMS = rms;
    }
    // motion_estimate.cpp:291-302: H = S*H*S^-1, S = diag(16,16,1)
    double sf = 1 << irl;
    double h12 = (double)(float)(bestH[1][2] * (float)sf), h02 = (double)(float)(bestH[0][2] * (float)sf);
    double fx = cam->in.fx;
    double roll = asin(h12 / fx), pitch = -asin(h02 / fx), yaw = -atan2((double)bestH[1][0], (double)bestH[0][0]);
    Quat q = rpy_to_quat(roll, pitch, yaw);
    if (g_pert.q_noise != 0) {  // stand-in for what a different solver (the reference's float JacobiSVD) may leave: H3
      unsigned st = g_pert.seed * 2654435761u + 12345u;
      double ax[3];
      for (int k = 0; k < 3; k++) { st = st * 1664525u + 1013904223u; ax[k] = (double)(st >> 8) / (1 << 24) - 0.5; }
      double nn = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      double h = 0.5 * g_pert.q_noise, sh = sin(h) / nn;
      Quat d{cos(h), ax[0] * sh, ax[1] * sh, ax[2] * sh};
      Quat r{q.w * d.w - q.x * d.x - q.y * d.y - q.z * d.z, q.w * d.x + q.x * d.w + q.y * d.z - q.z * d.y,
             q.w * d.y - q.x * d.z + q.y * d.w + q.z * d.x, q.w * d.z + q.x * d.y - q.y * d.x + q.z * d.w};
      q = r;
    }
    return q;
  }
};

// ---------------------------------------------------------------- context
struct Ctx {
  Intrinsics in;
  Params prm;
  Camera cam;
  std::vector<Frame*> slots;
  MotionEstimator est;
  Ctx(const Intrinsics& i, const Params& p) : in(i), prm(p), cam(i), est(&cam, &prm) {}
  ~Ctx() { for (auto f : slots) delete f; }
  Frame* slot(int s) {
    if (s >= (int)slots.size()) slots.resize(s + 1, nullptr);
    if (!slots[s]) slots[s] = new Frame(&cam, &prm);
    return slots[s];
  }
};

void fill_result(const MotionEstimator& e, PairResult* out) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) out->pose[i * 4 + j] = e.motion.R[i][j]; out->pose[i * 4 + 3] = e.motion.t[i]; }
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) out->cov[i * 6 + j] = e.cov[i][j];
  out->status = e.status;
  out->num_matches = (int)e.matches.size();
  out->num_inliers = e.num_inliers;
  out->num_reprojection_failures = e.num_reprojection_failures;
  out->mean_reprojection_error = e.mean_reprojection_error;
}

Iso iso_from_3x4(const double* p) {
  Iso m;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m.R[i][j] = p[i * 4 + j]; m.t[i] = p[i * 4 + 3]; }
  return m;
}
void iso_to_3x4(const Iso& m, double* p) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) p[i * 4 + j] = m.R[i][j]; p[i * 4 + 3] = m.t[i]; }
}

// vo.cpp:55-165 state machine (one stream)
struct VO {
  Ctx* c;
  int slot_ref, slot_prev, slot_curr;
  int init_cnt = 0, frame_cnt = 0;
  bool ref_changed_flag = false;
  int fast_thr, target_num_features;
  Iso pose, ref_to_prev, curr_to_prev, curr_to_ref;
  double curr_to_ref_cov[36];
  bool failed = false;
  int num_tracks = 0;          // this RGBDVO's MotionEstimator::_num_tracks (the oracle's estimator is shared by all streams)
  bool used_fallback = false;  // this Process() took its estimate from the retry against the previous frame (vo.cpp:131-151)
  VO(Ctx* ctx, int base_slot) : c(ctx) {
    slot_ref = base_slot; slot_prev = base_slot + 1; slot_curr = base_slot + 2;
    fast_thr = c->prm.fast_init_threshold;
    target_num_features = c->in.width * c->in.height / c->prm.target_pixels_per_feature;
    pose.setIdentity(); ref_to_prev.setIdentity(); curr_to_prev.setIdentity(); curr_to_ref.setIdentity();
    for (int i = 0; i < 36; i++) curr_to_ref_cov[i] = 0;
  }
  // returns 0 = warm-up skip, 1 = processed, -1 = the reference would exit(-1)
  int Process(const uint8_t* gray, const uint16_t* raw_depth) {
    if (init_cnt < c->prm.warmup_frames) {
      ++init_cnt;
      if (init_cnt != c->prm.warmup_frames) return 0;
    }
    if (ref_changed_flag) { std::swap(slot_ref, slot_curr); ref_to_prev.setIdentity(); }
    else std::swap(slot_prev, slot_curr);
    bool ref_changed = ref_changed_flag;
    ref_changed_flag = false;
    used_fallback = false;
    curr_to_prev.setIdentity();
    Frame* cur = c->slot(slot_curr);
    cur->Prepare(gray, raw_depth, fast_thr);
    {
      int32_t err = cur->NumDetected() - target_num_features;
      int32_t adjust = (int32_t)(err * c->prm.fast_adaptive_gain);
      fast_thr += adjust;
      fast_thr = std::min(fast_thr, c->prm.fast_threshold_max);
      fast_thr = std::max(fast_thr, c->prm.fast_threshold_min);
    }
    ++frame_cnt;
    if (frame_cnt < 2) { ref_changed_flag = true; return 1; }
    Frame* reff = c->slot(slot_ref);
    Frame* prev = c->slot(slot_prev);
    Quat q0 = ref_changed ? c->est.EstimateInitialRotation(reff, cur) : c->est.EstimateInitialRotation(prev, cur);
    Iso init = iso_inverse(ref_to_prev);
    iso_rotate(&init, q0);
    c->est.num_tracks = num_tracks;
    c->est.EstimateMotion(reff, cur, init);
    num_tracks = c->est.num_tracks;
    if (c->est.status == SUCCESS) {
      Iso to_ref = c->est.motion;
      curr_to_ref = to_ref;
      curr_to_prev = iso_mul(ref_to_prev, to_ref);
      memcpy(curr_to_ref_cov, c->est.cov, sizeof(curr_to_ref_cov));
      ref_to_prev = iso_inverse(to_ref);
      pose = iso_mul(pose, curr_to_prev);
    } else if (!ref_changed) {
      Iso init2; init2.setIdentity();
      iso_rotate(&init2, q0);
      c->est.num_tracks = num_tracks;
      c->est.EstimateMotion(prev, cur, init2);
      num_tracks = c->est.num_tracks;
      if (c->est.status == SUCCESS) {
        curr_to_prev = c->est.motion;
        pose = iso_mul(pose, curr_to_prev);
        ref_changed_flag = true;
        curr_to_ref = iso_mul(iso_inverse(ref_to_prev), curr_to_prev);
        used_fallback = true;
      }
    }
    if (c->est.status != SUCCESS) { failed = true; return -1; }  // reference: exit(-1), vo.cpp:154-158
    // vo.cpp:160-163: the reference compares against an uninitialised member; the
    // documented replacement is the option value.
    if (c->est.num_inliers < c->prm.reference_frame_change_threshold) ref_changed_flag = true;
    return 1;
  }
};

}  // namespace

// ================================================================ C exports
extern "C" {

void* orc_create(const Intrinsics* in, const Params* p) { return new Ctx(*in, *p); }
void orc_destroy(void* c) { delete (Ctx*)c; }

void orc_get_lut(void* cv, float* xmap, float* ymap) {
  Ctx* c = (Ctx*)cv;
  memcpy(xmap, c->cam.Xmap.data(), c->cam.Xmap.size() * 4);
  memcpy(ymap, c->cam.Ymap.data(), c->cam.Ymap.size() * 4);
}

void orc_prepare_frame(void* cv, int slot, const uint8_t* gray, const uint16_t* depth, int fast_thr) {
  ((Ctx*)cv)->slot(slot)->Prepare(gray, depth, fast_thr);
}

// level geometry: out = {width,height,stride,num_keypoints,num_detected,num_after_grid}
void orc_level_info(void* cv, int slot, int level, int32_t* out) {
  Level& l = ((Ctx*)cv)->slot(slot)->levels[level];
  out[0] = l.width; out[1] = l.height; out[2] = l.stride; out[3] = (int)l.kps.size(); out[4] = l.num_detected;
  out[5] = (int)l.initial.size();
}
void orc_level_image(void* cv, int slot, int level, uint8_t* out) {
  Level& l = ((Ctx*)cv)->slot(slot)->levels[level];
  memcpy(out, l.gray.data(), l.gray.size());
}
// grid-filtered candidate list (before border/LUT/depth filter): u,v,score floats
void orc_level_grid_keypoints(void* cv, int slot, int level, float* uvs) {
  Level& l = ((Ctx*)cv)->slot(slot)->levels[level];
  for (size_t i = 0; i < l.initial.size(); i++) { uvs[3 * i] = l.initial[i].u; uvs[3 * i + 1] = l.initial[i].v; uvs[3 * i + 2] = l.initial[i].score; }
}
// final keypoints: uvs[n][3] float, rect[n][2] double, xyz[n][3] double
void orc_level_keypoints(void* cv, int slot, int level, float* uvs, double* rect, double* xyz) {
  Level& l = ((Ctx*)cv)->slot(slot)->levels[level];
  for (size_t i = 0; i < l.kps.size(); i++) {
    const KeypointData& k = l.kps[i];
    uvs[3 * i] = k.kp.u; uvs[3 * i + 1] = k.kp.v; uvs[3 * i + 2] = k.kp.score;
    rect[2 * i] = k.rect_base_uv[0]; rect[2 * i + 1] = k.rect_base_uv[1];
    xyz[3 * i] = k.xyz[0]; xyz[3 * i + 1] = k.xyz[1]; xyz[3 * i + 2] = k.xyz[2];
  }
}
int orc_descriptor_stride(void* cv) { return ((Ctx*)cv)->slot(0)->levels[0].dx.stride; }
void orc_level_descriptors(void* cv, int slot, int level, uint8_t* out) {
  Level& l = ((Ctx*)cv)->slot(slot)->levels[level];
  memcpy(out, l.desc.data(), l.kps.size() * (size_t)l.dx.stride);
}

// FAST + NMS only (feature.cpp:6058-6077) on a caller image; returns count, fills up to cap
int orc_fast(const uint8_t* img, int w, int h, int stride, int thr, float* uvs, int cap) {
  std::vector<KeyPoint> k;
  FAST(img, w, h, stride, &k, thr);
  int n = (int)k.size();
  for (int i = 0; i < n && i < cap; i++) { uvs[3 * i] = k[i].u; uvs[3 * i + 1] = k[i].v; uvs[3 * i + 2] = k[i].score; }
  return n;
}
// raw segment test / score at one pixel (for the decision-tree check)
int orc_fast_pixel(const uint8_t* img, int stride, int x, int y, int thr, int* score) {
  init_has9();
  int off[16];
  for (int i = 0; i < 16; i++) off[i] = RING_DX[i] + stride * RING_DY[i];
  const uint8_t* p = &img[y * stride + x];
  int c = fast9_is_corner(p, off, thr);
  *score = c ? fast9_corner_score(p, off, thr) : -1;
  return c;
}
void orc_pyr_down(const uint8_t* src, int stride, int w, int h, uint8_t* dst, int dst_stride) {
  std::vector<uint8_t> buf((size_t)(w * h) / 2 + 64);
  BuildNextLevel(src, stride, w, h, dst, dst_stride, buf.data());
}

// ---- front-of-path conversions of the callers: /root/reference/helper.cpp (root directory,
// namespace rgbdslam), used by hand_held_demo.cpp:248 and pose_graph.cpp:101,128.  Parity unpinned (no fixtures).
// ConvertBGR2Gray, helper.cpp:55-64: double-weighted sum, left to right, truncated to u8
void orc_bgr_to_gray(const uint8_t* bgr, size_t npx, uint8_t* gray) {
  for (size_t i = 0, j = 0; i < npx; ++i, j += 3) gray[i] = 0.114 * bgr[j + 0] + 0.587 * bgr[j + 1] + 0.299 * bgr[j + 2];
}
// DepthToPointCloud::operator(), helper.cpp:30-53 (XtoZ = 2*tan(hfov/2), YtoZ = 2*tan(vfov/2), helper.hpp:32-36;
// width_/height_ are doubles there)
void orc_depth_to_cloud(const uint16_t* raw_depth, int w, int h, double xtoz, double ytoz, float* cloud) {
  const double width = w, height = h;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int idx = y * w + x;
      float z = raw_depth[idx] * 0.001;
      if (std::isnan(z)) {
        cloud[idx * 3 + 0] = 0; cloud[idx * 3 + 1] = 0; cloud[idx * 3 + 2] = 0;
      } else {
        cloud[idx * 3 + 0] = (static_cast<double>(x) / width - 0.5) * z * xtoz;
        cloud[idx * 3 + 1] = (0.5 - static_cast<double>(y) / height) * z * ytoz;
        cloud[idx * 3 + 2] = z;
      }
    }
}

// PoseGraphOptimizer::AddEdge, pose_graph.cpp:265: info = covariance.inverse() (Eigen fixed 6x6 -> partial-pivot LU)
int orc_information(const double* cov36, double* info36) {
  double A[6][6], inv[6][6];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) A[i][j] = cov36[i * 6 + j];
  const bool ok = inverse6(A, inv);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) info36[i * 6 + j] = inv[i][j];
  return ok ? 1 : 0;
}

void orc_initial_rotation(void* cv, int prev_slot, int cur_slot, double* q_wxyz) {
  Ctx* c = (Ctx*)cv;
  Quat q = c->est.EstimateInitialRotation(c->slot(prev_slot), c->slot(cur_slot));
  q_wxyz[0] = q.w; q_wxyz[1] = q.x; q_wxyz[2] = q.y; q_wxyz[3] = q.z;
}

// iterations the last EstimateInitialRotation ran and the one whose H it returned (tests/test_cpu_initial_rotation.py)
void orc_initial_rotation_info(void* cv, int32_t* out2) {
  Ctx* c = (Ctx*)cv;
  out2[0] = c->est.rot_iters; out2[1] = c->est.rot_best_iter;
}

// MotionEstimator::EstimateMotion with an explicit initial estimate (3x4 row-major)
void orc_estimate_motion(void* cv, int ref_slot, int tgt_slot, const double* init3x4, PairResult* out) {
  Ctx* c = (Ctx*)cv;
  c->est.EstimateMotion(c->slot(ref_slot), c->slot(tgt_slot), iso_from_3x4(init3x4));
  fill_result(c->est, out);
}

// FrameMatcher::Match semantics (fm.cpp:32-58) on two prepared slots
void orc_match_pair(void* cv, int ref_slot, int tgt_slot, PairResult* out) {
  Ctx* c = (Ctx*)cv;
  Quat q = c->est.EstimateInitialRotation(c->slot(ref_slot), c->slot(tgt_slot));
  Iso init; init.setIdentity();
  iso_rotate(&init, q);
  c->est.EstimateMotion(c->slot(ref_slot), c->slot(tgt_slot), init);
  fill_result(c->est, out);
}

// track ids (motion_estimate.cpp:380-391): of the last EstimateMotion's matches (post-sort order, the rows of
// orc_get_matches) and of a level's keypoints
void orc_get_match_tracks(void* cv, int cap, int32_t* out) {
  Ctx* c = (Ctx*)cv;
  for (size_t i = 0; i < c->est.matches.size() && (int)i < cap; i++) out[i] = c->est.matches[i].track_id;
}
void orc_level_tracks(void* cv, int slot, int level, int32_t* out) {
  const Level& lv = ((Ctx*)cv)->slot(slot)->levels[level];
  for (size_t i = 0; i < lv.kps.size(); i++) out[i] = lv.kps[i].track_id;
}
// FeatureMatch::consistency_vec of the last EstimateMotion's matches (rows as orc_get_matches, columns = ids): n x n bytes
void orc_get_match_consistency(void* cv, int n, uint8_t* out) {
  Ctx* c = (Ctx*)cv;
  for (size_t i = 0; i < c->est.matches.size() && (int)i < n; i++)
    for (int j = 0; j < n; j++) out[i * n + j] = (uint8_t)c->est.matches[i].consistency_vec[j];
}
int orc_num_tracks(void* cv) { return ((Ctx*)cv)->est.num_tracks; }
void orc_set_num_tracks(void* cv, int n) { ((Ctx*)cv)->est.num_tracks = n; }

// margins of the last EstimateMotion (struct Margins, 12 doubles in declaration order)
void orc_get_margins(void* cv, double* out12) {
  const Margins& m = ((Ctx*)cv)->est.mg;
  const double v[12] = {m.window, m.refine_gate, m.esm_step, m.sse_tie, m.clique, m.reproj_gate, m.mean_reproj, m.gn_sse_rel,
                        m.degree_gap, m.n_window, m.n_clique, m.n_esm};
  memcpy(out12, v, sizeof(v));
}
void orc_set_perturbation(int norm_order, int true_division, double q_noise_rad, unsigned seed, int rot_float) {
  g_pert.norm_order = norm_order; g_pert.true_division = true_division; g_pert.q_noise = q_noise_rad; g_pert.seed = seed;
  g_pert.rot_float = rot_float;
}

// dump of the last EstimateMotion's match array (post-sort order)
// ints[n][7] = ref_idx,tgt_idx,level,status,inlier,id,degree ; dbl[n][8] = refined u,v, rect u,v, xyz, reproj
void orc_get_matches(void* cv, int cap, int32_t* ints, double* dbl) {
  Ctx* c = (Ctx*)cv;
  for (size_t i = 0; i < c->est.matches.size() && (int)i < cap; i++) {
    const FeatureMatch& m = c->est.matches[i];
    int32_t* a = ints + 7 * i;
    a[0] = m.ref_idx; a[1] = m.target_idx; a[2] = m.level; a[3] = m.status; a[4] = m.inlier; a[5] = m.id; a[6] = m.compatibility_degree;
    double* d = dbl + 8 * i;
    d[0] = m.refined_target_keypoint.kp.u; d[1] = m.refined_target_keypoint.kp.v;
    d[2] = m.refined_target_keypoint.rect_base_uv[0]; d[3] = m.refined_target_keypoint.rect_base_uv[1];
    d[4] = m.refined_target_keypoint.xyz[0]; d[5] = m.refined_target_keypoint.xyz[1]; d[6] = m.refined_target_keypoint.xyz[2];
    d[7] = m.reprojection_error;
  }
}

// The real libstdc++ algorithms on (key, payload) records with a key-only comparator (descending), as used at
// pyramid.cpp:42-45 and motion_estimate.cpp:526; `keys[i] >> shift` is the key.  nth < 0: std::sort.
void orc_std_select_or_sort(uint32_t* keys, int n, int shift, int nth) {
  struct Rec { int key; uint32_t full; double pad[3]; };
  std::vector<Rec> v(n);
  for (int i = 0; i < n; i++) { v[i].key = (int)(keys[i] >> shift); v[i].full = keys[i]; }
  auto cmp = [](const Rec& a, const Rec& b) { return a.key > b.key; };
  if (nth < 0) std::sort(v.begin(), v.end(), cmp);
  else std::nth_element(v.begin(), v.begin() + nth, v.end(), cmp);
  for (int i = 0; i < n; i++) keys[i] = v[i].full;
}

// ---- test hooks into the Gauss-Newton pieces (tests/test_cpu_gn.py: analytic Jacobians against finite differences of an
// independent residual, the refined pose against scipy.optimize.least_squares).  n points each way; xyz arrays are n x 4
// (homogeneous), projections n x 2.
void orc_gn_residuals(const double* prm6, int n, const double* txyz, const double* rproj, const double* rxyz, const double* tproj,
                      double fx, double px, double py, double* err /* 4n: forward u,v per point, then reverse */) {
  std::vector<double> a(txyz, txyz + 4 * n), b(rproj, rproj + 2 * n), c2(rxyz, rxyz + 4 * n), d(tproj, tproj + 2 * n);
  Iso m = iso_from_params(prm6);
  reproj_error(a, b, fx, px, py, m, err);
  reproj_error(c2, d, fx, px, py, iso_inverse(m), err + 2 * n);
}
void orc_gn_jacobian(const double* prm6, int n, const double* txyz, const double* rxyz, double fx, double px, double py,
                     double* M /* 4n x 6, rows as orc_gn_residuals */) {
  JacMats f, r;
  forward_jac_mats(prm6, fx, px, py, &f);
  reverse_jac_mats(prm6, fx, px, py, &r);
  for (int i = 0; i < n; i++) jac_rows(f, txyz + 4 * i, M + (2 * i) * 6, M + (2 * i + 1) * 6);
  for (int i = 0; i < n; i++) jac_rows(r, rxyz + 4 * i, M + (2 * n + 2 * i) * 6, M + (2 * n + 2 * i + 1) * 6);
}
void orc_params_to_pose(const double* prm6, double* pose3x4) { iso_to_3x4(iso_from_params(prm6), pose3x4); }
void orc_pose_to_params(const double* pose3x4, double* prm6) { iso_to_params(iso_from_3x4(pose3x4), prm6); }
// camera the estimator works with (the rectified one: fx, fy, cx, cy)
void orc_rect_camera(void* cv, double* out4) {
  const Intrinsics& r = ((Ctx*)cv)->cam.rect;
  out4[0] = r.fx; out4[1] = r.fy; out4[2] = r.cx; out4[3] = r.cy;
}

// ---- streaming VO
void* orc_vo_create(void* cv, int base_slot) { return new VO((Ctx*)cv, base_slot); }
void orc_vo_destroy(void* v) { delete (VO*)v; }
// out_i[7] = rc, fast_thr(after), keyframe_changed, status, num_inliers, num_matches, used_fallback ; out_d[24] = pose 3x4, curr_to_ref 3x4
int orc_vo_process(void* vv, const uint8_t* gray, const uint16_t* depth, int32_t* out_i, double* out_d) {
  VO* v = (VO*)vv;
  int rc = v->Process(gray, depth);
  out_i[0] = rc; out_i[1] = v->fast_thr; out_i[2] = v->ref_changed_flag; out_i[3] = v->c->est.status;
  out_i[4] = v->c->est.num_inliers; out_i[5] = (int)v->c->est.matches.size(); out_i[6] = v->used_fallback;
  iso_to_3x4(v->pose, out_d);
  iso_to_3x4(v->curr_to_ref, out_d + 12);
  return rc;
}

}  // extern "C"
