// kernels_pair.h -- MotionEstimator on the GPU (rgbdvo/motion_estimate.cpp), one frame pair per workgroup
// (or per wave) with thousands of pairs in flight:
//   k_init_rotation  EstimateInitialRotation + InitialHomographyEstimator   :268-304, 1333-1571
//   k_match          candidate windows + SAD mutual matching                 :307-345, feature.cpp:6239-6335
//   k_refine         sub-pixel ESM, rectified lookup, depth refinement       :347-378, 757-900, depth.cpp:80-162
//   k_consistency    pairwise rigid-distance consistency bit matrix          :470-523
//   k_clique         degree sort (libstdc++ std::sort replay) + greedy clique :526-569
//   k_gn             bidirectional Gauss-Newton, reprojection pruning, status :198-232, 610-720, 1208-1331
// fp64 expressions keep the evaluation order of oracle/rvo_oracle.cpp (no FMA contraction); reductions over
// matches are tree-ordered here and sequential there, which is the only source of (1e-16-level) differences.
#pragma once
#include "kernels_frame.h"
#include "rvo_dev.h"
#include "stdalgo.h"
#include "wave_stdalgo.h"

#define NONE_KEY 0xFFFFFFFFu

// ------------------------------------------------------------------ small fp64 geometry
struct DIso { double R[9]; double t[3]; };
struct DQuat { double w, x, y, z; };

__device__ inline void iso_identity(DIso& m) {
  for (int i = 0; i < 9; i++) m.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  m.t[0] = m.t[1] = m.t[2] = 0.0;
}
__device__ inline void iso_load(DIso& m, const double* p) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m.R[i * 3 + j] = p[i * 4 + j]; m.t[i] = p[i * 4 + 3]; }
}
__device__ inline void iso_store(const DIso& m, double* p) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) p[i * 4 + j] = m.R[i * 3 + j]; p[i * 4 + 3] = m.t[i]; }
}
__device__ inline DIso iso_inverse(const DIso& a) {
  DIso r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.R[i * 3 + j] = a.R[j * 3 + i];
  for (int i = 0; i < 3; i++) r.t[i] = -((r.R[i * 3] * a.t[0] + r.R[i * 3 + 1] * a.t[1]) + r.R[i * 3 + 2] * a.t[2]);
  return r;
}
__device__ inline DIso iso_mul(const DIso& a, const DIso& b) {
  DIso r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) r.R[i * 3 + j] = (a.R[i * 3] * b.R[j] + a.R[i * 3 + 1] * b.R[3 + j]) + a.R[i * 3 + 2] * b.R[6 + j];
    r.t[i] = ((a.R[i * 3] * b.t[0] + a.R[i * 3 + 1] * b.t[1]) + a.R[i * 3 + 2] * b.t[2]) + a.t[i];
  }
  return r;
}
__device__ inline void quat_to_R(const DQuat& q, double* R) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ inline DQuat R_to_quat(const double* m) {
  DQuat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[7] - m[5]) * t;
    q.y = (m[2] - m[6]) * t;
    q.z = (m[3] - m[1]) * t;
  } else {  // Eigen's branch on the largest diagonal element, written out per case (static indexing: no scratch memory)
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 4]) i = 2;
    if (i == 0) {
      t = sqrt(m[0] - m[4] - m[8] + 1.0);
      q.x = 0.5 * t; t = 0.5 / t;
      q.w = (m[7] - m[5]) * t; q.y = (m[3] + m[1]) * t; q.z = (m[6] + m[2]) * t;
    } else if (i == 1) {
      t = sqrt(m[4] - m[8] - m[0] + 1.0);
      q.y = 0.5 * t; t = 0.5 / t;
      q.w = (m[2] - m[6]) * t; q.z = (m[7] + m[5]) * t; q.x = (m[1] + m[3]) * t;
    } else {
      t = sqrt(m[8] - m[0] - m[4] + 1.0);
      q.z = 0.5 * t; t = 0.5 / t;
      q.w = (m[3] - m[1]) * t; q.x = (m[2] + m[6]) * t; q.y = (m[5] + m[7]) * t;
    }
  }
  return q;
}
__device__ inline void iso_rotate(DIso& a, const DQuat& q) {
  double Rq[9], L[9];
  quat_to_R(q, Rq);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) L[i * 3 + j] = (a.R[i * 3] * Rq[j] + a.R[i * 3 + 1] * Rq[3 + j]) + a.R[i * 3 + 2] * Rq[6 + j];
  for (int i = 0; i < 9; i++) a.R[i] = L[i];
}
__device__ inline DQuat rpy_to_quat(double roll, double pitch, double yaw) {
  const double hr = roll / 2, hp = pitch / 2, hy = yaw / 2;
  const double sr = sin(hr), sp = sin(hp), sy = sin(hy), cr = cos(hr), cp = cos(hp), cy = cos(hy);
  DQuat q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  return q;
}
__device__ inline DIso iso_from_params(const double* p) {
  DIso r;
  iso_identity(r);
  r.t[0] = p[0]; r.t[1] = p[1]; r.t[2] = p[2];
  iso_rotate(r, rpy_to_quat(p[3], p[4], p[5]));
  return r;
}
__device__ inline void iso_to_params(const DIso& M, double* p) {
  const DQuat q = R_to_quat(M.R);
  const double roll_a = 2 * (q.w * q.x + q.y * q.z), roll_b = 1 - 2 * (q.x * q.x + q.y * q.y);
  const double pitch_sin = 2 * (q.w * q.y - q.z * q.x);
  const double yaw_a = 2 * (q.w * q.z + q.x * q.y), yaw_b = 1 - 2 * (q.y * q.y + q.z * q.z);
  p[0] = M.t[0]; p[1] = M.t[1]; p[2] = M.t[2];
  p[3] = atan2(roll_a, roll_b);
  p[4] = asin(pitch_sin);
  p[5] = atan2(yaw_a, yaw_b);
}
// K * motion.matrix(), K = [fx 0 px 0; 0 fy py 0; 0 0 1 0]; zeros kept so the rounding matches the host restatement
__device__ inline void proj_matrix(double fx, double fy, double px, double py, const DIso& m, double* P /*[12]*/) {
  const double K[12] = {fx, 0, px, 0, 0, fy, py, 0, 0, 0, 1, 0};
  const double M[16] = {m.R[0], m.R[1], m.R[2], m.t[0], m.R[3], m.R[4], m.R[5], m.t[1], m.R[6], m.R[7], m.R[8], m.t[2], 0, 0, 0, 1};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++)
      P[i * 4 + j] = ((K[i * 4] * M[j] + K[i * 4 + 1] * M[4 + j]) + K[i * 4 + 2] * M[8 + j]) + K[i * 4 + 3] * M[12 + j];
}
__device__ inline double dot4(const double* r, double x, double y, double z, double w) {
  return ((r[0] * x + r[1] * y) + r[2] * z) + r[3] * w;
}

// sum over the 64 lanes of a wave on the DPP network (no LDS crossbar); the total is valid in lane 63.  Fixed order.
template <int CTRL, int ROW_MASK>
__device__ inline double rot_dpp_add(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ inline double rot_wave_sum63(double v) {
  v = rot_dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = rot_dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = rot_dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = rot_dpp_add<0x140, 0xF>(v);  // row_mirror: every lane of a row holds the row's sum
  v = rot_dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = rot_dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// ------------------------------------------------------------------ initial rotation
// grid (n_pairs), 256 threads, dynamic LDS = 5 * npx floats + 64 doubles.
// mode 0: write q0 only; mode 1: also init_pose = Identity.rotate(q0) (fm.cpp:41-47).
#ifndef ROT_MIN_WAVES
#define ROT_MIN_WAVES 1
#endif
__global__ __launch_bounds__(256, ROT_MIN_WAVES) void k_init_rotation(FrameBufs fb, FrameGeom g, PairBufs pb, CamParams cam, int mode) {
  RVO_OCC_SCOPE(5, 1);
  extern __shared__ __align__(16) uint8_t lds[];
  const int pair = blockIdx.x + pb.pair0, tid = threadIdx.x;
  if (!pb.active[pair]) return;
  const int irl = 4;
  const int pyr = min(g.levels - 1, irl);
  const int ds = irl - pyr;
  const LevelGeom G = g.lv[pyr];
  const int cols = G.w >> ds, rows = G.h >> ds, npx = cols * rows;
  float* tmpl = reinterpret_cast<float*>(lds);
  float* test = tmpl + npx;
  float* tdx = test + npx;
  float* tdy = tdx + npx;
  float* warped = tdy + npx;
  double* s_red = reinterpret_cast<double*>(lds + (((size_t)5 * npx * 4 + 15) & ~(size_t)15));
  float* s_ctl = reinterpret_cast<float*>(s_red + 40);  // next homography [9] and the stop flag, written by wave 0 (s_red: 64 doubles)

  const uint8_t* rimg = level_img(fb, g, pb.rot_slot[pair], pyr);
  const uint8_t* timg = level_img(fb, g, pb.tgt_slot[pair], pyr);
  // grayToEigen, motion_estimate.cpp:1338-1367 (box sums are exact in float)
  const int f = 1 << ds;
  const double pf2 = (double)f * (double)f;
  for (int i = tid; i < npx; i += 256) {
    const int ey = div_u16(i, cols), ex = i - ey * cols;
    float a = 0.f, b = 0.f;
    for (int y = 0; y < f; y++)
      for (int x = 0; x < f; x++) {
        a += (float)gload<uint8_t>(rimg + (size_t)(ey * f + y) * G.stride + ex * f + x);
        b += (float)gload<uint8_t>(timg + (size_t)(ey * f + y) * G.stride + ex * f + x);
      }
    tmpl[i] = ds > 0 ? (float)((double)a / pf2) : a;
    test[i] = ds > 0 ? (float)((double)b / pf2) : b;
  }
  __syncthreads();
  auto grad = [&](const float* im, int x, int y, float& vx, float& vy) {  // computeGradient, :1494-1517
    if (x == 0) vx = im[y * cols + 1] - im[y * cols];
    else if (x == cols - 1) vx = im[y * cols + x] - im[y * cols + x - 1];
    else vx = (float)((double)(im[y * cols + x + 1] - im[y * cols + x - 1]) / 2.0);
    if (y == 0) vy = im[cols + x] - im[x];
    else if (y == rows - 1) vy = im[y * cols + x] - im[(y - 1) * cols + x];
    else vy = (float)((double)(im[(y + 1) * cols + x] - im[(y - 1) * cols + x]) / 2.0);
  };
  for (int i = tid; i < npx; i += 256) {
    const int y = div_u16(i, cols);
    grad(tmpl, i - y * cols, y, tdx[i], tdy[i]);
  }

  float H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, bestH[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double minError = INFINITY, lastRMS = INFINITY;
  int bestIter = 0, lastImproved = 0;
  for (int iter = 0; iter < 8; iter++) {
    // (no barrier here: the two at the end of an iteration separate its reads of `warped` from the writes below)
    // constructWarpedImage, :1538-1571
    for (int i = tid; i < npx; i += 256) {
      const int y = div_u16(i, cols), x = i - y * cols;
      const float hx = (H[0] * (float)x + H[1] * (float)y) + H[2] * 1.f;
      const float hy = (H[3] * (float)x + H[4] * (float)y) + H[5] * 1.f;
      const float hw = (H[6] * (float)x + H[7] * (float)y) + H[8] * 1.f;
      const float px = hx / hw, py = hy / hw;
      const int fx0 = (int)floorf(px), fy0 = (int)floorf(py), cx0 = (int)ceilf(px), cy0 = (int)ceilf(py);
      double val;
      if (0 <= px && px < (float)(cols - 1) && 0 <= py && py < (float)(rows - 1)) {
        const double x1 = (double)(px - (float)fx0), y1 = (double)(py - (float)fy0), x2 = 1 - x1, y2 = 1 - y1;
        val = ((x2 * y2 * (double)test[fy0 * cols + fx0] + x1 * y2 * (double)test[fy0 * cols + fx0 + 1]) +
               x2 * y1 * (double)test[(fy0 + 1) * cols + fx0]) + x1 * y1 * (double)test[(fy0 + 1) * cols + fx0 + 1];
      } else if (0 <= fx0 && fx0 < cols && 0 <= fy0 && fy0 < rows) {
        val = test[fy0 * cols + fx0];
      } else if (0 <= cx0 && cx0 < cols && 0 <= cy0 && cy0 < rows) {
        val = test[cy0 * cols + cx0];
      } else {
        val = 128;
      }
      warped[i] = (float)val;
    }
    __syncthreads();
    // error, gradients of the warped image, normal equations J'J, J'e
    double acc[10];
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = 0;
    for (int i = tid; i < npx; i += 256) {
      const int y = div_u16(i, cols), x = i - y * cols;
      const float e = warped[i] - tmpl[i];
      float wx, wy;
      grad(warped, x, y, wx, wy);
      const float jx = tdx[i] + wx, jy = tdy[i] + wy;
      const float j2 = jx * (float)(y + 1) - jy * (float)(x + 1);
      const double J0 = jx, J1 = jy, J2 = j2, E = e;
      acc[0] += J0 * J0; acc[1] += J0 * J1; acc[2] += J0 * J2; acc[3] += J1 * J1; acc[4] += J1 * J2; acc[5] += J2 * J2;
      acc[6] += J0 * E; acc[7] += J1 * E; acc[8] += J2 * E;
      acc[9] += E * E;
    }
    // the waves' sums on the DPP network (total in lane 63), then wave 0 alone adds the four partials, solves and steps:
    // the solve is a chain of fp64 divisions and sin / cos that every thread used to repeat (a fifth of the kernel's
    // instructions); the others pick the next homography and the stop flag up from LDS
#pragma unroll
    for (int k = 0; k < 10; k++) {
      const double sk = rot_wave_sum63(acc[k]);
      if ((tid & 63) == 63) s_red[k * 4 + (tid >> 6)] = sk;
    }
    __syncthreads();
    if (tid < 64) {  // computed identically by every lane of wave 0
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = ((s_red[k * 4] + s_red[k * 4 + 1]) + s_red[k * 4 + 2]) + s_red[k * 4 + 3];
    const double rms = sqrt(acc[9]) / sqrt((double)npx);
    if (rms < minError) { minError = rms; for (int k = 0; k < 9; k++) bestH[k] = H[k]; bestIter = iter; }
    const double A00 = acc[0], A01 = acc[1], A02 = acc[2], A11 = acc[3], A12 = acc[4], A22 = acc[5];
    const double A10 = A01, A20 = A02, A21 = A12;
    double lie[3] = {0, 0, 0};
    const double det = A00 * (A11 * A22 - A12 * A21) - A01 * (A10 * A22 - A12 * A20) + A02 * (A10 * A21 - A11 * A20);
    const double scale = A00 * A11 * A22;
    if (fabs(det) > 1e-12 * fabs(scale) && isfinite(det) && det != 0) {
      double inv[9];
      inv[0] = (A11 * A22 - A12 * A21) / det; inv[1] = (A02 * A21 - A01 * A22) / det; inv[2] = (A01 * A12 - A02 * A11) / det;
      inv[3] = (A12 * A20 - A10 * A22) / det; inv[4] = (A00 * A22 - A02 * A20) / det; inv[5] = (A02 * A10 - A00 * A12) / det;
      inv[6] = (A10 * A21 - A11 * A20) / det; inv[7] = (A01 * A20 - A00 * A21) / det; inv[8] = (A00 * A11 - A01 * A10) / det;
      for (int a = 0; a < 3; a++) lie[a] = (double)(float)(-2 * ((inv[a * 3] * acc[6] + inv[a * 3 + 1] * acc[7]) + inv[a * 3 + 2] * acc[8]));
    }
    const double a = lie[2];
    double sa, ca1;
    if (fabs(a) < 1e-8) { sa = 1 - a * a / 6; ca1 = a / 2; } else { sa = sin(a) / a; ca1 = (1 - cos(a)) / a; }
    const float E[9] = {(float)cos(a), (float)sin(a), (float)(sa * lie[0] + ca1 * lie[1]),
                        (float)-sin(a), (float)cos(a), (float)(-ca1 * lie[0] + sa * lie[1]), 0.f, 0.f, 1.f};
    float Hn[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) Hn[r * 3 + c] = (H[r * 3] * E[c] + H[r * 3 + 1] * E[3 + c]) + H[r * 3 + 2] * E[6 + c];
    if (rms < lastRMS) lastImproved = iter;
    const float l0 = (float)lie[0], l1 = (float)lie[1], l2 = (float)lie[2];
    const double lnorm = sqrt((double)(l0 * l0 + l1 * l1 + l2 * l2));
    const bool stop = lnorm < 1e-6 || (rms - minError > 3 && iter - bestIter > 2) || iter - bestIter > 4 || iter - lastImproved > 2;
    lastRMS = rms;
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 9; k++) s_ctl[k] = Hn[k];
      s_ctl[9] = stop ? 1.f : 0.f;
    }
    }
    __syncthreads();
    if (s_ctl[9] != 0.f) break;
#pragma unroll
    for (int k = 0; k < 9; k++) H[k] = s_ctl[k];
  }
  if (tid == 0) {
    const double sf = (double)(1 << irl);
    const double h12 = (double)(float)(bestH[5] * (float)sf), h02 = (double)(float)(bestH[2] * (float)sf);
    const double roll = asin(h12 / cam.in_fx), pitch = -asin(h02 / cam.in_fx), yaw = -atan2((double)bestH[3], (double)bestH[0]);
    const DQuat q = rpy_to_quat(roll, pitch, yaw);
    double* qo = pb.q0 + pair * 4;
    qo[0] = q.w; qo[1] = q.x; qo[2] = q.y; qo[3] = q.z;
    if (mode == 1) {
      DIso init;
      iso_identity(init);
      iso_rotate(init, q);
      iso_store(init, pb.init_pose + pair * 12);
    }
  }
}

// ------------------------------------------------------------------ matching
// 256 threads: a workgroup of 4 waves x 96 VGPRs + 24 KB of LDS fits into the room ONE retiring k_fast_grid workgroup
// leaves on a CU (the other batch's FAST stage floods every CU with 7 workgroups), so k_match is dispatched while FAST
// runs; with 1024 threads (faster alone: 0.12 vs 0.17 ms) it waited for FAST to drain: 0.34 ms in the pipeline.
#ifndef MATCH_THREADS
#define MATCH_THREADS 256
#endif
#define MATCH_WAVES (MATCH_THREADS / 64)
#ifndef MATCH_QCAP
#define MATCH_QCAP 256  // per-wave candidate queue entries; flushed when fewer than 64 slots are left
#endif

// SAD of two 80-byte descriptors (feature.hpp:364-381), both 16-byte aligned
__device__ inline uint32_t sad80(const uint8_t* a, const uint8_t* b) {
  const uint4* a4 = reinterpret_cast<const uint4*>(a);
  const uint4* b4 = reinterpret_cast<const uint4*>(b);
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint4 x = a4[k], y = b4[k];
    s = __builtin_amdgcn_sad_u8(x.x, y.x, s);
    s = __builtin_amdgcn_sad_u8(x.y, y.y, s);
    s = __builtin_amdgcn_sad_u8(x.z, y.z, s);
    s = __builtin_amdgcn_sad_u8(x.w, y.w, s);
  }
  return s;
}

// grid (levels, n_pairs), 1024 threads; dynamic LDS = match_lds_bytes(kp_cap(level 0)).
// Setup: every reference keypoint is projected once (fp64, motion_estimate.cpp:321-327) into LDS, and every run of
// 64 target keypoints (targets are stored bucket by bucket, so a run is spatially compact) gets a bounding box.
// Phase 1: a wave walks its reference keypoints; lanes test the run boxes against the search window, and only the
// runs that can hold a candidate are visited, 64 targets at a time: fp32 distance first, the reference's fp64
// err.norm() < window (motion_estimate.cpp:335) only inside the rounding band around the threshold.  In-window
// (ref,target) pairs are queued.  Phase 2 (whenever the wave's queue fills): one lane per queued pair computes the
// SAD and folds it into the per-ref and per-target bests with LDS atomicMin on (score<<16 | index) keys -- the
// lexicographic minimum is exactly the reference's "strictly smaller score wins, first in iteration order on ties"
// (feature.cpp:6301-6308).
__host__ __device__ inline int match_max_runs(int cap) { return (cap + 63) / 64 + 1; }
__host__ __device__ inline int match_lds_bytes(int cap) { return cap * 16 + match_max_runs(cap) * 16 + MATCH_WAVES * MATCH_QCAP * 4 + 128; }
#ifndef MATCH_MIN_WAVES
#define MATCH_MIN_WAVES 1
#endif
// A level whose keypoint capacity exceeds MATCH_PART_REFS is matched by several workgroups (parts), each taking every
// nparts-th group of MATCH_WAVES reference keypoints against ALL targets of the level: the one workgroup of level 0 (1200
// slots) was the longest serial piece of a pair's chain (0.18 ms alone).  A part's per-reference bests are complete; its
// per-target bests cover its references only, so the parts write both to global memory and k_match_finish takes the
// minimum over the parts (the keys are (score << 16 | index): the minimum is the reference's first-minimum rule whatever the
// order) and emits the mutual matches.  Levels with one part finish inside k_match as before.
#ifndef MATCH_PART_REFS
#define MATCH_PART_REFS 320
#endif
#define MATCH_PARTS_MAX 8
__host__ __device__ inline int match_parts(int kp_cap) { return min(MATCH_PARTS_MAX, max(1, (kp_cap + MATCH_PART_REFS - 1) / MATCH_PART_REFS)); }
__host__ __device__ inline int match_grid_x(const FrameGeom& g) {
  int n = 0;
  for (int l = 0; l < g.levels; l++) n += match_parts(g.lv[l].kp_cap);
  return n;
}
__global__ __launch_bounds__(MATCH_THREADS, MATCH_MIN_WAVES) void k_match(FrameBufs fb, FrameGeom g, PairBufs pb, CamParams cam, EstParams ep) {
  RVO_OCC_SCOPE(6, 1);
  extern __shared__ __align__(16) uint8_t lds[];
  const int pair = blockIdx.y + pb.pair0, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (!pb.active[pair]) return;
  int lvl = 0, part = blockIdx.x, nparts = match_parts(g.lv[0].kp_cap);  // blockIdx.x = the parts of level 0, then level 1, ...
  while (part >= nparts) { part -= nparts; lvl++; nparts = match_parts(g.lv[lvl].kp_cap); }
  const int gw = part * MATCH_WAVES + wid, TW = nparts * MATCH_WAVES;  // this wave among the level's waves
  const LevelGeom G = g.lv[lvl];
  const int cap = g.lv[0].kp_cap;
  float2* s_tuv = reinterpret_cast<float2*>(lds);
  uint32_t* s_bt = reinterpret_cast<uint32_t*>(lds + (size_t)cap * 8);
  uint32_t* s_br = s_bt + cap;
  float4* s_box = reinterpret_cast<float4*>(s_br + cap);                  // per run: min u, min v, max u, max v
  uint32_t* s_q = reinterpret_cast<uint32_t*>(s_box + match_max_runs(cap)) + (size_t)wid * MATCH_QCAP;
  int* s_wave = reinterpret_cast<int*>(reinterpret_cast<uint32_t*>(s_box + match_max_runs(cap)) + (size_t)MATCH_WAVES * MATCH_QCAP);
  const int rs = pb.ref_slot[pair], ts = pb.tgt_slot[pair];
  const int nr = fb.kp_cnt[rs * RVO_MAX_LEVELS + lvl], nt = fb.kp_cnt[ts * RVO_MAX_LEVELS + lvl];
  const size_t rbase = (size_t)rs * g.kp_total + G.kp_off, tbase = (size_t)ts * g.kp_total + G.kp_off;

  // reproj_mat = K_rect * (motion_estimate^-1), motion_estimate.cpp:309-313
  DIso init;
  iso_load(init, pb.init_pose + pair * 12);
  const DIso r2t = iso_inverse(init);
  double RP[12];
  proj_matrix(cam.fx, cam.fy, cam.cx, cam.cy, r2t, RP);

  const int nruns = (nt + 63) >> 6;
  for (int i0 = wid * 64; i0 < nt; i0 += MATCH_THREADS) {  // whole runs per wave
    const int i = i0 + lane;
    float2 uv = make_float2(0.f, 0.f);
    if (i < nt) { uv = fb.kp_rect[tbase + i]; s_tuv[i] = uv; s_bt[i] = NONE_KEY; }
    {
      float mnx = i < nt ? uv.x : INFINITY, mny = i < nt ? uv.y : INFINITY;
      float mxx = i < nt ? uv.x : -INFINITY, mxy = i < nt ? uv.y : -INFINITY;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mny = fminf(mny, __shfl_xor(mny, o, 64));
        mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
      }
      if (lane == 0) s_box[i0 >> 6] = make_float4(mnx, mny, mxx, mxy);
    }
  }
  for (int i = tid; i < nr; i += MATCH_THREADS) s_br[i] = NONE_KEY;
  __syncthreads();

  const double win = ep.feature_search_window;
  const double win2 = win * win, lo2 = win2 * (1.0 - 1e-9), hi2 = win2 * (1.0 + 1e-9);
  // fp32 band: keypoint coordinates are exact floats below 2048; a projection within window+1 of one is rounded by
  // < 1.3e-4, so the fp32 squared distance is off by < 0.002*(window+1) near the threshold; farther away it only grows.
  const float band = (float)(0.002 * (win + 1.0) + 1e-6 * win2);
  const float flo = (float)win2 - band, fhi = (float)win2 + band;
  int qn = 0;  // wave-uniform queue fill
  auto flush = [&]() {
    for (int e = lane; e < qn; e += 64) {
      const uint32_t rt = s_q[e];
      const uint32_t r = rt >> 16, t = rt & 0xFFFF;
      // a kept keypoint's descriptor stays where k_fast_grid wrote it (bucket order): row kp_src[] of the level's block
      const uint32_t s = sad80(fb.bk_desc + (rbase + fb.kp_src[rbase + r]) * RVO_DESC_LEN, fb.bk_desc + (tbase + fb.kp_src[tbase + t]) * RVO_DESC_LEN);
      atomicMin(&s_br[r], (s << 16) | t);
      atomicMin(&s_bt[t], (s << 16) | r);
    }
    qn = 0;
  };
  auto bcast = [](double v, int j) {  // v_readlane of a double (j is wave-uniform)
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), j);
    return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
  };
  // wave gw of the level's TW waves owns the reference keypoints gw, gw+TW, ...; 64 of them are projected at a time, one per lane.  Lanes
  // = references while the run boxes are tested (one broadcast LDS read per run), then every (reference, run) hit is
  // visited with lanes = the 64 targets of the run.
  for (int rb = gw; rb < nr; rb += TW * 64) {
    double lpu = 0, lpv = 0;
    const int rl = rb + TW * lane;
    if (rl < nr) {
      const double* X = fb.kp_xyz + (rbase + rl) * 3;
      const double u = dot4(RP, X[0], X[1], X[2], 1.0), v = dot4(RP + 4, X[0], X[1], X[2], 1.0), w = dot4(RP + 8, X[0], X[1], X[2], 1.0);
      const double inv = 1.0 / w;  // Eigen 3.2 `v /= s` multiplies by the reciprocal
      lpu = u * inv; lpv = v * inv;
    }
    const float lfpu = (float)lpu, lfpv = (float)lpv;
    for (int run = 0; run < nruns; run++) {
      bool hit = rl < nr;
      if (hit) {
        const float4 bx = s_box[run];
        // distance from the projection to the box, in fp64 on exact inputs; fmax drops a NaN projection to 0 (= visit)
        const double dx = fmax(fmax((double)bx.x - lpu, lpu - (double)bx.z), 0.0);
        const double dy = fmax(fmax((double)bx.y - lpv, lpv - (double)bx.w), 0.0);
        hit = !(dx * dx + dy * dy > hi2 * (1.0 + 1e-9));
      }
      unsigned long long hm = __ballot(hit);
      const int t = (run << 6) + lane;
      const float2 uv = t < nt ? s_tuv[t] : make_float2(0.f, 0.f);
      while (hm) {
        const int j = __ffsll((long long)hm) - 1;
        hm &= hm - 1ull;
        const int r = rb + TW * j;
        const float fpu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lfpu), j));
        const float fpv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lfpv), j));
        bool cand = false;
        if (t < nt) {
          const float fx = uv.x - fpu, fy = uv.y - fpv;
          const float fd2 = fx * fx + fy * fy;
          if (fd2 < flo) {
            cand = true;
          } else if (!(fd2 > fhi)) {  // rounding band or NaN: the reference's fp64 evaluation
            const double pu = bcast(lpu, j), pv = bcast(lpv, j);
            const double ex = (double)uv.x - pu, ey = (double)uv.y - pv;
            const double d2 = ex * ex + ey * ey;
            cand = d2 < lo2 ? true : (d2 > hi2 ? false : sqrt(d2) < win);
          }
        }
        const unsigned long long m = __ballot(cand);
        if (m) {
          if (cand) s_q[qn + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)r << 16) | (uint32_t)t;
          qn += __popcll(m);
          if (qn > MATCH_QCAP - 64) flush();
        }
      }
    }
  }
  flush();
  __syncthreads();
  if (nparts > 1) {  // hand the bests to k_match_finish
    uint32_t* gbr = pb.mt_br + (size_t)pair * g.kp_total + G.kp_off;
    uint32_t* gbt = pb.mt_bt + ((size_t)pair * MATCH_PARTS_MAX + part) * g.kp_total + G.kp_off;
    for (int r = tid; r < nr; r += MATCH_THREADS)
      if ((r % TW) / MATCH_WAVES == part) gbr[r] = s_br[r];
    for (int t = tid; t < nt; t += MATCH_THREADS) gbt[t] = s_bt[t];
    return;
  }
  // mutual best matches in ref-index order, feature.cpp:6313-6333
  int32_t* lm_ref = pb.lm_ref + (size_t)pair * g.kp_total + G.kp_off;
  int32_t* lm_tgt = pb.lm_tgt + (size_t)pair * g.kp_total + G.kp_off;
  int base = 0;
  for (int c0 = 0; c0 < nr; c0 += MATCH_THREADS) {
    const int r = c0 + tid;
    int flag = 0, t = 0;
    if (r < nr) {
      const uint32_t key = s_br[r];
      if (key != NONE_KEY) {
        t = key & 0xFFFF;
        flag = (int)(s_bt[t] & 0xFFFF) == r;
      }
    }
    int total;
    const int pos = base + block_excl_scan<MATCH_WAVES>(flag, s_wave, &total);
    if (flag) { lm_ref[pos] = r; lm_tgt[pos] = t; }
    base += total;
  }
  if (tid == 0) pb.lm_cnt[pair * RVO_MAX_LEVELS + lvl] = base;
}

// grid (levels, n_pairs), 256 threads: the mutual matches of the levels k_match worked on in parts, in ref-index order
// (feature.cpp:6313-6333); other levels were finished by k_match itself.
__global__ __launch_bounds__(256) void k_match_finish(FrameBufs fb, FrameGeom g, PairBufs pb) {
  __shared__ int s_wave[4];
  const int lvl = blockIdx.x, pair = blockIdx.y + pb.pair0, tid = threadIdx.x;
  if (!pb.active[pair]) return;
  const LevelGeom G = g.lv[lvl];
  const int nparts = match_parts(G.kp_cap);
  if (nparts == 1) return;
  const int rs = pb.ref_slot[pair];
  const int nr = fb.kp_cnt[rs * RVO_MAX_LEVELS + lvl];
  const uint32_t* gbr = pb.mt_br + (size_t)pair * g.kp_total + G.kp_off;
  const uint32_t* gbt = pb.mt_bt + (size_t)pair * MATCH_PARTS_MAX * g.kp_total + G.kp_off;
  int32_t* lm_ref = pb.lm_ref + (size_t)pair * g.kp_total + G.kp_off;
  int32_t* lm_tgt = pb.lm_tgt + (size_t)pair * g.kp_total + G.kp_off;
  int base = 0;
  for (int c0 = 0; c0 < nr; c0 += 256) {
    const int r = c0 + tid;
    int flag = 0, t = 0;
    if (r < nr) {
      const uint32_t key = gbr[r];
      if (key != NONE_KEY) {
        t = key & 0xFFFF;
        uint32_t best = NONE_KEY;
        for (int p = 0; p < nparts; p++) best = min(best, gbt[(size_t)p * g.kp_total + t]);
        flag = (int)(best & 0xFFFF) == r;
      }
    }
    int total;
    const int pos = base + block_excl_scan<4>(flag, s_wave, &total);
    if (flag) { lm_ref[pos] = r; lm_tgt[pos] = t; }
    base += total;
  }
  if (tid == 0) pb.lm_cnt[pair * RVO_MAX_LEVELS + lvl] = base;
}

// ------------------------------------------------------------------ sub-pixel refinement
__device__ inline int desc_off(int i, int stride) { return (i / 9 - 4) * stride + (i % 9 - 4); }

struct Wts { float wa, wb, wc, wd; int base, u, v; };
// bilinear weights of Descriptor::ExtractInterpDescriptor, feature.cpp:6116-6124
__device__ inline Wts make_wts(float x, float y, int stride) {
  const int u = (int)x, v = (int)y;
  const float wr = x - (float)u, wbm = y - (float)v;
  Wts w;
  w.wa = (float)((1. - (double)wr) * (1. - (double)wbm));
  w.wb = (float)((double)wr * (1. - (double)wbm));
  w.wc = (float)((1. - (double)wr) * (double)wbm);
  w.wd = wr * wbm;
  w.base = v * stride + u;
  w.u = u; w.v = v;
  return w;
}
__device__ inline int interp_raw(const uint8_t* __restrict__ gimg, int stride, const Wts& w, int i) {
  const int k = w.base + desc_off(i, stride);
  const float f = ((w.wa * (float)gimg[k] + w.wb * (float)gimg[k + 1]) + w.wc * (float)gimg[k + stride]) + w.wd * (float)gimg[k + stride + 1];
  return (int)f;  // (uint8_t) truncation; f is in [0, 256)
}
// Descriptor::Normalized, feature.cpp:6158-6178: mean < 128 adds 128 - mean and saturates at 255, mean > 128 subtracts
// mean - 128 and saturates at 0 -- for a byte v both are clamp(v + (128 - mean), 0, 255) (the bound that cannot be
// reached never binds), one add and one v_med3 instead of two compares and selects
__device__ inline int norm_px(int v, int mean) { return min(max(v + (128 - mean), 0), 255); }

#ifndef REFINE_GROUP
#define REFINE_GROUP 16  // lanes per match; 8 (10 taps per lane) needs 172 VGPRs and measured 28 % slower
#endif
// sum over the 16 lanes of a match group (= one DPP row): xor-1 and xor-2 butterflies as quad permutes, then the
// half-row and row mirrors; every lane ends with the total.  No LDS-crossbar shuffles on the iteration's critical path.
__device__ inline int group16_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
#if !defined(REFINE_GROUP) || REFINE_GROUP == 16
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
#endif
  return v;
}

#define REFINE_TAPS (RVO_DESC_LEN / REFINE_GROUP)  // 5 taps per lane
#define REFINE_WIN_ROWS 16
#define REFINE_WIN_PITCH 20
// grid (ceil(kp_total/8), n_pairs), REFINE_THREADS threads: 16 lanes per match, 5 descriptor taps per lane; all sums are exact
// int32, so the lane split does not change any value.  Branches below are uniform within a 16-lane group.
#ifndef REFINE_MIN_WAVES
#define REFINE_MIN_WAVES 1
#endif
// 8 matches per workgroup: alone the size does not matter (matches are independent), inside the pipeline a two-wave
// workgroup finds room beside k_fast_grid sooner than a four-wave one (311 k against 305 k pairs/s; 64 threads: 309 k)
#ifndef REFINE_THREADS
#define REFINE_THREADS 128
#endif
#ifndef REFINE_GRID_DIV
#define REFINE_GRID_DIV 2
#endif
__global__ __launch_bounds__(REFINE_THREADS, REFINE_MIN_WAVES) void k_refine(FrameBufs fb, FrameGeom g, PairBufs pb) {
  RVO_OCC_SCOPE(7, 16);
  int bx, py;  // XCD-aware: the workgroups of a pair run on one XCD, whose L2 then holds the pair's two images once
  xcd_map(gridDim.x, gridDim.y, blockIdx.x, blockIdx.y, &bx, &py);
  const int pair = py + pb.pair0;
  if (!pb.active[pair]) return;
  const int gl = threadIdx.x & (REFINE_GROUP - 1);
  const int i_grp = bx * (REFINE_THREADS / REFINE_GROUP) + (threadIdx.x / REFINE_GROUP);
  int off[RVO_MAX_LEVELS + 1];
  off[0] = 0;
  for (int l = 0; l < g.levels; l++) off[l + 1] = off[l] + pb.lm_cnt[pair * RVO_MAX_LEVELS + l];
  const int N = off[g.levels];
  if (i_grp == 0 && gl == 0) pb.n_matches[pair] = N;
  const int rs = pb.ref_slot[pair], ts = pb.tgt_slot[pair];
  // 16x16 target window around the matched keypoint (rows of 5 aligned dwords): positions within +-3 px of it, i.e.
  // practically every ESM step, read their 10x10 taps from LDS instead of paying an L2 round trip per iteration
  __shared__ __align__(4) uint8_t s_win[REFINE_THREADS / REFINE_GROUP][REFINE_WIN_ROWS * REFINE_WIN_PITCH];
  // a workgroup takes every gridDim.x-th set of 8 matches: the grid is sized for a fraction of the slots a pair could
  // fill (REFINE_GRID_DIV), so few workgroups start only to find nothing to do
  for (int i = i_grp; i < N; i += (int)gridDim.x * (REFINE_THREADS / REFINE_GROUP)) {
  int lvl = 0;
  for (int l = 1; l < g.levels; l++) if (i >= off[l]) lvl = l;
  const int j = i - off[lvl];
  const LevelGeom G = g.lv[lvl];
  const int r = pb.lm_ref[(size_t)pair * g.kp_total + G.kp_off + j];
  const int t = pb.lm_tgt[(size_t)pair * g.kp_total + G.kp_off + j];
  const size_t rk = (size_t)rs * g.kp_total + G.kp_off + r, tk = (size_t)ts * g.kp_total + G.kp_off + t;
  const uint32_t rkp = fb.kp_uvs[rk], tkp = fb.kp_uvs[tk];
  const uint8_t* __restrict__ rg = level_img(fb, g, rs, lvl);
  const uint8_t* __restrict__ tg = level_img(fb, g, ts, lvl);
  const int S = G.stride;

  // ---- refineFeatureMatch, motion_estimate.cpp:757-900 (descriptors recomputed per pass instead of stored)
  int toff[REFINE_TAPS];
#pragma unroll
  for (int m = 0; m < REFINE_TAPS; m++) toff[m] = desc_off(gl + REFINE_GROUP * m, S);
  // the four bilinear neighbours of a tap; the first three are also the taps of the target gradient at the same
  // integer position (motion_estimate.cpp:800-815), so an accepted step needs no further loads for the next one: kept as
  // the two differences (p01 - p00, p10 - p00), an i16 pair in one register
  typedef s2_t Nb;
  uint8_t* win = s_win[threadIdx.x / REFINE_GROUP];
  int ox, oy, xa;  // window = columns ox..ox+15, rows oy..oy+15; xa = aligned column the stored rows start at
  auto stage = [&](int cx, int cy) {  // window centred on (cx, cy); uniform per 16-lane group
    ox = cx - 7; oy = cy - 7; xa = ox & ~3;
    wavealgo::wave_lds_sync();  // earlier reads of the old window are done (one wave, LDS in order)
#pragma unroll
    for (int rr = 0; rr < REFINE_WIN_ROWS / REFINE_GROUP; rr++) {
      const int wr = gl + REFINE_GROUP * rr;  // window row staged by this lane
      const int y = oy + wr;
      uint32_t dw[REFINE_WIN_PITCH / 4];
#pragma unroll
      for (int q = 0; q < REFINE_WIN_PITCH / 4; q++) {
        const int x = xa + 4 * q;
        dw[q] = (y >= 0 && y < G.h && x >= 0 && x + 3 < S) ? gload<uint32_t>(tg + y * S + x) : 0u;
      }
#pragma unroll
      for (int q = 0; q < REFINE_WIN_PITCH / 4; q++) reinterpret_cast<uint32_t*>(win + wr * REFINE_WIN_PITCH)[q] = dw[q];
    }
    wavealgo::wave_lds_sync();  // the 16 lanes of a group are in one wave
  };
  stage(kp_x(tkp), kp_y(tkp));
  int woff[REFINE_TAPS];
#pragma unroll
  for (int m = 0; m < REFINE_TAPS; m++) {
    const int t9 = gl + REFINE_GROUP * m;
    woff[m] = (t9 / 9 - 4) * REFINE_WIN_PITCH + (t9 % 9 - 4);
  }
  // interpolated tap m at integer position (w.u, w.v); the window must cover it (ensure_window first)
  auto ensure_window = [&](const Wts& w) {  // a step that leaves the +-3 px the window covers re-centres it (rare)
    if (!(w.u - 4 >= ox && w.u + 5 <= ox + 15 && w.v - 4 >= oy && w.v + 5 <= oy + 15)) stage(w.u, w.v);
  };
  auto interp = [&](const Wts& w, int m, Nb& nb) -> int {
    const uint8_t* q = win + (w.v - oy) * REFINE_WIN_PITCH + (w.u - xa) + woff[m];
    const int p00 = q[0], p01 = q[1], p10 = q[REFINE_WIN_PITCH];
    nb = s2_t{(short)(p01 - p00), (short)(p10 - p00)};
    const float f = ((w.wa * (float)p00 + w.wb * (float)p01) + w.wc * (float)p10) + w.wd * (float)q[REFINE_WIN_PITCH + 1];
    return (int)f;
  };
  const int rbase = kp_y(rkp) * S + kp_x(rkp);
  const float tu = (float)kp_x(tkp), tv = (float)kp_y(tkp);
  float tx = tu, ty = tv;
  Wts wt = make_wts(tx, ty, S);
  int rraw[REFINE_TAPS], traw[REFINE_TAPS];
  s2_t rd[REFINE_TAPS];  // reference gradient taps (dx, dy), |values| <= 255
  Nb tnb[REFINE_TAPS];
  int rsum = 0, tsum = 0;
#pragma unroll
  for (int m = 0; m < REFINE_TAPS; m++) {
    const int kr = rbase + toff[m];
    rraw[m] = gload<uint8_t>(rg + kr);
    rd[m] = s2_t{(short)(gload<uint8_t>(rg + kr + 1) - rraw[m]), (short)(gload<uint8_t>(rg + kr + S) - rraw[m])};
    traw[m] = interp(wt, m, tnb[m]);
    rsum += rraw[m];
    tsum += traw[m];
  }
  const int rmean = group16_sum(rsum) / RVO_DESC_LEN;
  const int tmean = group16_sum(tsum) / RVO_DESC_LEN;
  // The lane's five taps as i16 pairs (|values| <= 510): the sums of products below are v_dot2_i32_i16, two taps per
  // instruction, exact in int32 like the reference's dot_int16_aligned (motion_estimate.cpp:723-755)
  static_assert(REFINE_TAPS == 5, "pairs (0,1), (2,3) and tap 4 below");
  struct P5 { s2_t a, b; int c; };
  auto pack5 = [](const int* v) { P5 p; p.a = s2_t{(short)v[0], (short)v[1]}; p.b = s2_t{(short)v[2], (short)v[3]}; p.c = v[4]; return p; };
  auto dot5 = [](const P5& x, const P5& y) { return __builtin_amdgcn_sdot2(x.a, y.a, __builtin_amdgcn_sdot2(x.b, y.b, x.c * y.c, false), false); };
  P5 rn;  // the reference's normalised descriptor taps (0..255), in dot5's pair layout
  {
    int v[REFINE_TAPS];
#pragma unroll
    for (int m = 0; m < REFINE_TAPS; m++) v[m] = norm_px(rraw[m], rmean);
    rn = pack5(v);
  }
  // error of a target descriptor (raw taps, their mean) against the reference's: one packed subtraction per pair of taps
  auto err5 = [&](const int* raw, int mean) {
    int v[REFINE_TAPS];
#pragma unroll
    for (int m = 0; m < REFINE_TAPS; m++) v[m] = norm_px(raw[m], mean);
    P5 e = pack5(v);
    e.a -= rn.a; e.b -= rn.b; e.c -= rn.c;
    return e;
  };
  P5 te = err5(traw, tmean);  // te = error of the current target descriptor against the reference's
  int final_sse = group16_sum(dot5(te, te));
  const int min_xy = RVO_WIN, max_x = G.w - RVO_WIN - 2, max_y = G.h - RVO_WIN - 2;  // pyramid.cpp:123-126
  for (int it = 0; it < 6; it++) {
    // mean gradient taps (x, y) of the five taps, one packed add each; then the x halves and the y halves gathered into
    // the pair layout of dot5 (two v_perm per pair of taps)
    uint32_t mw[REFINE_TAPS];
#pragma unroll
    for (int m = 0; m < REFINE_TAPS; m++) mw[m] = __builtin_bit_cast(uint32_t, (s2_t)(rd[m] + tnb[m]));
    P5 mx, my;
    mx.a = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm(mw[1], mw[0], 0x05040100u));
    mx.b = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm(mw[3], mw[2], 0x05040100u));
    mx.c = (int)(short)(mw[4] & 0xFFFFu);
    my.a = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm(mw[1], mw[0], 0x07060302u));
    my.b = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm(mw[3], mw[2], 0x07060302u));
    my.c = (int)mw[4] >> 16;
    const int S00 = group16_sum(dot5(mx, mx)), S01 = group16_sum(dot5(mx, my)), S11 = group16_sum(dot5(my, my));
    const int gxs = group16_sum(dot5(mx, te)), gys = group16_sum(dot5(my, te));
    const double s00 = S00, s01 = S01, s11 = S11;
    const double det = s00 * s11 - s01 * s01;
    if (det <= 1e-9) break;
    // the three quotients share the denominator: one refined reciprocal, then the same last steps as the correctly
    // rounded fp64 division sequence (q = n*r; q += fma(-det, q, n) * r).  det is an integer in [1, 2^50) and the
    // numerators are integers below 2^25, so none of that sequence's scaling / fix-up cases can occur.
    double rc = __builtin_amdgcn_rcp(det);
    rc = __builtin_fma(__builtin_fma(-det, rc, 1.0), rc, rc);
    rc = __builtin_fma(__builtin_fma(-det, rc, 1.0), rc, rc);
    auto quot = [&](double nmr) { const double q = nmr * rc; return __builtin_fma(__builtin_fma(-det, q, nmr), rc, q); };
    const double si00 = quot(s11), si01 = quot(-s01), si11 = quot(s00);
    const double gx = gxs, gy = gys;
    const double dxs = -2 * (si00 * gx + si01 * gy);
    const double dys = -2 * (si01 * gx + si11 * gy);
    const float ntx = (float)((double)tx + dxs), nty = (float)((double)ty + dys);
    if (!(ntx >= (float)min_xy && ntx <= (float)max_x && nty >= (float)min_xy && nty <= (float)max_y)) break;
    const Wts wn = make_wts(ntx, nty, S);
    ensure_window(wn);
    int nraw[REFINE_TAPS];
    Nb nnb[REFINE_TAPS];
    int nsum = 0;
#pragma unroll
    for (int m = 0; m < REFINE_TAPS; m++) { nraw[m] = interp(wn, m, nnb[m]); nsum += nraw[m]; }
    const int nmean = group16_sum(nsum) / RVO_DESC_LEN;
    const P5 ne = err5(nraw, nmean);
    const int sse = group16_sum(dot5(ne, ne));
    if (sse < final_sse) {
      tx = ntx; ty = nty; final_sse = sse;
      te = ne;
#pragma unroll
      for (int m = 0; m < REFINE_TAPS; m++) tnb[m] = nnb[m];
    } else break;
    if (fabs(dxs) < 0.1 && fabs(dys) < 0.1) break;
  }
  if (gl == 0) {  // hand the refined position to k_refine_finish
    const size_t mi = (size_t)pair * pb.mcap + i;
    pb.m_ref[mi] = r; pb.m_tgt[mi] = t;
    pb.m_level[mi] = (uint8_t)lvl;
    pb.m_uv[mi] = make_float2(tx, ty);
  }
  }
}

// Gating + refined record, motion_estimate.cpp:360-376: one thread per match (a kernel of its own so that its fp64
// lookups do not add to the register budget of the ESM loop above).  grid (ceil(mcap/256), n_pairs), 256 threads.
#ifndef FINISH_THREADS
#define FINISH_THREADS 256
#endif
__global__ __launch_bounds__(FINISH_THREADS) void k_refine_finish(FrameBufs fb, FrameGeom g, PairBufs pb) {
  RVO_OCC_SCOPE(8, 4);
  const int pair = blockIdx.y + pb.pair0;
  if (!pb.active[pair]) return;
  const int N = pb.n_matches[pair];
  const int i = blockIdx.x * FINISH_THREADS + threadIdx.x;
  if (i >= N) return;
  const size_t mi0 = (size_t)pair * pb.mcap + i;
  const int lvl = pb.m_level[mi0];
  const LevelGeom G = g.lv[lvl];
  const int rs = pb.ref_slot[pair], ts = pb.tgt_slot[pair];
  const int r = pb.m_ref[mi0], t = pb.m_tgt[mi0];
  const float2 txy = pb.m_uv[mi0];
  const float tx = txy.x, ty = txy.y;
  const size_t rk = (size_t)rs * g.kp_total + G.kp_off + r, tk = (size_t)ts * g.kp_total + G.kp_off + t;
  const uint32_t tkp = fb.kp_uvs[tk];
  const float tu = (float)kp_x(tkp), tv = (float)kp_y(tkp);
  const double ddx = (double)tu - (double)tx, ddy = (double)tv - (double)ty;
  const double dsn = sqrt(ddx * ddx + ddy * ddy);
  float2 ouv = make_float2(tu, tv);
  const float2 trect = fb.kp_rect[tk];
  double rect0 = trect.x, rect1 = trect.y;
  double X = fb.kp_xyz[tk * 3], Y = fb.kp_xyz[tk * 3 + 1], Z = fb.kp_xyz[tk * 3 + 2];
  int status = RVO_MATCH_OK;
  if (!(dsn < 1e-9) && !(dsn > 1.5)) {
    ouv = make_float2(tx, ty);
    const double bu = (double)tx * (double)(1 << lvl), bv = (double)ty * (double)(1 << lvl);
    {  // Camera::BilinearLookup, camera.hpp:85-117
      const int u = (int)bu, v = (int)bv;
      const float wr = (float)(bu - (double)u), wb = (float)(bv - (double)v);
      const float w[4] = {(1 - wr) * (1 - wb), wr * (1 - wb), (1 - wr) * wb, wr * wb};
      const int ra = v * g.W + u;
      const int idx[4] = {ra, ra + 1, ra + g.W, ra + g.W + 1};
      rect0 = 0; rect1 = 0;
      for (int q = 0; q < 4; q++) {
        rect0 += (double)(w[q] * fb.xmap[idx[q]]);
        rect1 += (double)(w[q] * fb.ymap[idx[q]]);
      }
    }
    status = RVO_MATCH_REFINEMENT_FAILED;
    {  // Depth::GetXYZInterp on the target frame's depth, depth.cpp:97-162
      const uint16_t* __restrict__ depth = fb.depth_tab[ts];
      const float u_f = (float)rect0, v_f = (float)rect1;
      const float v_f_d = v_f * g.ys, u_f_d = u_f * g.xs;
      const int v = (int)v_f_d, u = (int)u_f_d;
      const float wr = u_f_d - (float)u, wb = v_f_d - (float)v;
      if (u >= 0 && v >= 0 && u < g.Wd - 1 && v < g.Hd - 1) {
        const float w[4] = {(1 - wr) * (1 - wb), wr * (1 - wb), (1 - wr) * wb, wr * wb};
        const int di = v * g.Wd + u;
        const uint16_t dr[4] = {depth[di], depth[di + 1], depth[di + g.Wd], depth[di + g.Wd + 1]};
        const int missing = (dr[0] == 0) + (dr[1] == 0) + (dr[2] == 0) + (dr[3] == 0);
        if (missing < 4) {
          double z;
          if (missing) {
            z = NAN;  // depth.cpp:137-147 selects the NaN neighbour (isnan instead of !isnan); kept
          } else {
            z = 0;
            for (int q = 0; q < 4; q++) z += (double)(float)((double)dr[q] * 1e-3) * (double)w[q];
          }
          X = z * ((double)u_f * g.fx_inv + g.neg_cx_div_fx);
          Y = z * ((double)v_f * g.fy_inv + g.neg_cy_div_fy);
          Z = z;
          status = RVO_MATCH_OK;
        }
      }
    }
  }
  const size_t mi = (size_t)pair * pb.mcap + i;
  pb.m_ref[mi] = r; pb.m_tgt[mi] = t;
  pb.m_level[mi] = (uint8_t)lvl; pb.m_status[mi] = (uint8_t)status;
  pb.m_uv[mi] = ouv;
  pb.m_rect[mi * 2] = rect0; pb.m_rect[mi * 2 + 1] = rect1;
  pb.m_xyz[mi * 3] = X; pb.m_xyz[mi * 3 + 1] = Y; pb.m_xyz[mi * 3 + 2] = Z;
  // reference-side fields gathered once for the consistency / GN stages
  const float2 rrect = fb.kp_rect[rk];
  pb.m_rrect[mi * 2] = rrect.x; pb.m_rrect[mi * 2 + 1] = rrect.y;
  pb.m_rxyz[mi * 3] = fb.kp_xyz[rk * 3]; pb.m_rxyz[mi * 3 + 1] = fb.kp_xyz[rk * 3 + 1]; pb.m_rxyz[mi * 3 + 2] = fb.kp_xyz[rk * 3 + 2];
  pb.m_inlier[mi] = 0;
  pb.m_reproj[mi] = 0;
}

// ------------------------------------------------------------------ consistency matrix
#define CONS_TILE 256
// grid (T(T+1)/2, n_pairs), T = ceil(mcap/256), 256 threads: one workgroup per 256x256 tile (I, J >= I) of the upper
// triangle; thread = row i of row tile I, the 256 columns of tile J are staged in LDS as fp32 (broadcast reads).  The
// matrix is symmetric: for J > I every computed 32-bit word is also transposed with ballots and stored into the rows
// of tile J (their words for columns of tile I come from here).
//
// The test |‖ref_j-ref_i‖ - ‖tgt_j-tgt_i‖| < thr (motion_estimate.cpp:505-507) is first decided without square
// roots in fp32: with a, b the squared distances, |sqrt a - sqrt b| < t  <=>  a+b < t^2  or  (a-b)^2 < t^2 (2(a+b) - t^2).
// It is evaluated at t = thr -/+ band, band = EstParams::cons_band >= the worst-case fp32 error of (sqrt a - sqrt b)
// for the coordinate range of the camera (host: 4e-6 * largest coordinate); whatever is not decided the same way at
// both ends is repeated in fp64 exactly as the reference does it (Eigen 3.2: sqrt(x*x + (y*y + z*z))).  A NaN
// coordinate (depth.cpp:137-147 quirk) makes the reference's comparison false; the fp32 test classifies it as false too.
typedef float f2_t __attribute__((ext_vector_type(2)));
#ifndef CONS_MIN_WAVES
#define CONS_MIN_WAVES 1
#endif
// CONS_THREADS rows of a 256 x 256 tile per workgroup (256 / CONS_THREADS workgroups per tile): the waves of a tile
// work independently once the columns are staged, and small workgroups are placed sooner beside k_fast_grid
#ifndef CONS_THREADS
#define CONS_THREADS 256
#endif
#define CONS_PARTS (256 / CONS_THREADS)
#ifndef CONS_GRID
// workgroups per pair and part: the ten tiles of up to 1024 matches, one each.  Not a multiple of 8: workgroups are dealt
// round-robin to the 8 XCDs, and with 8 / 16 per pair every XCD always gets the same tile numbers, some of them never
// live (measured: 5 / 6 / 8 / 10 / 12 / 16 / 28 per pair: 342 / 344 / 333 / 347 / 343 / 329 / 342 k pairs/s)
#define CONS_GRID 10
#endif
__global__ __launch_bounds__(CONS_THREADS, CONS_MIN_WAVES) void k_consistency(PairBufs pb, EstParams ep) {
  RVO_OCC_SCOPE(9, 4);
  __shared__ float s_c[(CONS_TILE + 32) * 6];
  const int pair = blockIdx.y + pb.pair0;
  if (!pb.active[pair]) return;
  const int N = pb.n_matches[pair];
  // The tiles (I, J >= I) that hold matches are numbered column tile by column tile (J = 0: (0,0); J = 1: (0,1), (1,1);
  // ...), so the live ones are the first Tn (Tn + 1) / 2, Tn = ceil(N / 256); the grid has CONS_GRID workgroups per
  // pair, each taking every CONS_GRID-th live tile (a grid over all tiles of mcap x mcap started mostly workgroups that
  // had nothing to do: 18 of 28 at 950 matches of 1600).
  const int Tn = (N + 255) / 256, live_tiles = Tn * (Tn + 1) / 2;
  const int part = blockIdx.x % CONS_PARTS;
  for (int tile = blockIdx.x / CONS_PARTS; tile < live_tiles; tile += gridDim.x / CONS_PARTS) {
  int tj = 0;
  while ((tj + 1) * (tj + 2) / 2 <= tile) tj++;
  const int ti = tile - tj * (tj + 1) / 2;
  const int jstart = ti * 256, jcol = tj * 256;
  __syncthreads();  // the previous tile's readers of s_c are done
  for (int k = threadIdx.x; k < (CONS_TILE + 32) * 6; k += CONS_THREADS) s_c[k] = 0.f;  // padding columns stay finite
  const int lane = threadIdx.x & 63;
  const int i = jstart + part * CONS_THREADS + threadIdx.x;
  const int iw0 = jstart + part * CONS_THREADS + (threadIdx.x & ~63);  // first row of this wave
  const bool live = i < N;
  const bool wave_live = iw0 < N;
  const size_t mb = (size_t)pair * pb.mcap;
  const int ii = live ? i : N - 1;
  const double r1x = pb.m_rxyz[(mb + ii) * 3], r1y = pb.m_rxyz[(mb + ii) * 3 + 1], r1z = pb.m_rxyz[(mb + ii) * 3 + 2];
  const double t1x = pb.m_xyz[(mb + ii) * 3], t1y = pb.m_xyz[(mb + ii) * 3 + 1], t1z = pb.m_xyz[(mb + ii) * 3 + 2];
  const float fr1x = (float)r1x, fr1y = (float)r1y, fr1z = (float)r1z, ft1x = (float)t1x, ft1y = (float)t1y, ft1z = (float)t1z;
  const double thr = ep.clique_inlier_threshold;
  const float tlo = fmaxf((float)(thr - ep.cons_band), 0.f), thi = (float)(thr + ep.cons_band);
  const float tlo2 = tlo * tlo, thi2 = thi * thi;
  const float lo_m = 2.f * tlo2, lo_c = -tlo2 * tlo2, hi_m = 2.f * thi2, hi_c = -thi2 * thi2;  // rhs = fma(m, a+b, c)
  uint32_t* row = pb.cbits + (mb + ii) * pb.cwords;
  {
    const int j0 = jcol;
    const int nt = min(CONS_TILE, N - j0);
    __syncthreads();
    // LDS layout: per pair of columns (2p, 2p+1) twelve floats  rx0 rx1 ry0 ry1 rz0 rz1 tx0 tx1 ty0 ty1 tz0 tz1,
    // so the packed-fp32 evaluation of two columns reads its operands with three 16-byte broadcasts
    for (int k = threadIdx.x; k < nt * 3; k += CONS_THREADS) {
      const int c = k / 3, ax = k % 3;
      float* d = s_c + (c >> 1) * 12 + ax * 2 + (c & 1);
      d[0] = (float)pb.m_rxyz[(mb + j0) * 3 + k];
      d[6] = (float)pb.m_xyz[(mb + j0) * 3 + k];
    }
    __syncthreads();
    if (wave_live) {
    uint32_t quad[4] = {0u, 0u, 0u, 0u};
    for (int w0 = 0; w0 < nt; w0 += 32) {
      uint32_t word = 0, amb = 0;
      // branch-free fp32 pass over 32 columns (columns past the tile end read stale-but-finite LDS and are masked)
#pragma unroll
      for (int p = 0; p < 16; p++) {
        const f2_t* cj = reinterpret_cast<const f2_t*>(s_c + ((w0 >> 1) + p) * 12);
        f2_t dx = cj[0] - fr1x, dy = cj[1] - fr1y, dz = cj[2] - fr1z;
        const f2_t a = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
        dx = cj[3] - ft1x; dy = cj[4] - ft1y; dz = cj[5] - ft1z;
        const f2_t bb = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
        const f2_t e = a - bb, apb = a + bb, lhs = e * e;
        const f2_t rlo = __builtin_elementwise_fma(f2_t{lo_m, lo_m}, apb, f2_t{lo_c, lo_c});
        const f2_t rhi = __builtin_elementwise_fma(f2_t{hi_m, hi_m}, apb, f2_t{hi_c, hi_c});
        const bool yes0 = (lhs.x < rlo.x) | (apb.x < tlo2), yes1 = (lhs.y < rlo.y) | (apb.y < tlo2);
        const bool no0 = !(lhs.x <= rhi.x) & !(apb.x <= thi2), no1 = !(lhs.y <= rhi.y) & !(apb.y <= thi2);  // also true for NaN
        word |= (yes0 ? 1u << (2 * p) : 0u) | (yes1 ? 2u << (2 * p) : 0u);
        amb |= ((!yes0 & !no0) ? 1u << (2 * p) : 0u) | ((!yes1 & !no1) ? 2u << (2 * p) : 0u);
      }
      const int jend = min(32, nt - w0);
      const uint32_t vmask = jend >= 32 ? 0xFFFFFFFFu : ((1u << jend) - 1u);
      word &= vmask;
      amb &= vmask;
      while (amb) {  // rare: exact fp64 evaluation
        const int b = __ffs(amb) - 1;
        amb &= amb - 1;
        const int j = j0 + w0 + b;
        const double* rj = pb.m_rxyz + (mb + j) * 3;
        const double* tj = pb.m_xyz + (mb + j) * 3;
        double ex = rj[0] - r1x, ey = rj[1] - r1y, ez = rj[2] - r1z;
        const double rdd = sqrt(ex * ex + (ey * ey + ez * ez));
        ex = tj[0] - t1x; ey = tj[1] - t1y; ez = tj[2] - t1z;
        const double tdd = sqrt(ex * ex + (ey * ey + ez * ez));
        if (fabs(rdd - tdd) < thr) word |= 1u << b;
      }
      const int self = i - (j0 + w0);
      if (self >= 0 && self < 32) word &= ~(1u << self);
      if (!live) word = 0;
      // rows are 16-byte aligned and tiles start on multiples of 128 columns: four words per store
      quad[(w0 >> 5) & 3] = word;
      if (((w0 >> 5) & 3) == 3 || w0 + 32 >= nt) {
        if (live) *reinterpret_cast<uint4*>(row + (((j0 + w0) >> 5) & ~3)) = make_uint4(quad[0], quad[1], quad[2], quad[3]);
        quad[0] = quad[1] = quad[2] = quad[3] = 0u;
      }
      if (j0 + w0 >= jstart + 256) {
        // transpose: lane b collects the 64-bit column (rows iw0..iw0+63) of column j0+w0+b
        // (the ballots are scalar values: v_writelane drops each into its lane, no per-lane select)
        uint32_t col_lo = 0, col_hi = 0;
#pragma unroll
        for (int b = 0; b < 32; b++) {
          const unsigned long long m = __ballot((word & (1u << b)) != 0u);
          col_lo = __llvm_amdgcn_writelane((uint32_t)m, b, col_lo);
          col_hi = __llvm_amdgcn_writelane((uint32_t)(m >> 32), b, col_hi);
        }
        const int jc = j0 + w0 + lane;
        if (lane < 32 && jc < N) {
          // iw0 is a multiple of 64 and cwords a multiple of 4: both words exist and the pair is 8-byte aligned
          *reinterpret_cast<uint2*>(pb.cbits + (mb + jc) * pb.cwords + (iw0 >> 5)) = make_uint2(col_lo, col_hi);
        }
      }
    }
    }  // wave_live
  }
  }  // tiles of this workgroup
}

// ------------------------------------------------------------------ sort + greedy clique
// grid (n_pairs), CLIQUE_WAVES*64 threads; dynamic LDS = clique_lds_bytes(mcap, cwords).  All waves sort; wave 0
// then walks the sorted candidates.  8 waves are fastest alone; inside the staged pipeline 4 are (305 k against 298 k
// pairs/s; 2 waves: 300 k): a 512-thread workgroup at 185 VGPRs needs two free wave slots on all four SIMDs of a CU at
// once and waits long for them beside k_fast_grid (0.107 ms alone, 0.42 ms in the pipeline).
#ifndef CLIQUE_WAVES
#define CLIQUE_WAVES 4
#endif
__host__ __device__ inline int clique_ctl_ints() { return 4 + SORT_QCAP * 2 + CLIQUE_WAVES * SORT_STACK * 2; }
// keys | accepted mask | sort control | scratch: the sort's 12*mcap bytes, reused as the 64 x (cwords+1) staged rows
__host__ __device__ inline int clique_lds_bytes(int mcap, int cwords) {
  const int scratch = mcap * 12 > 256 * (cwords + 1) ? mcap * 12 : 256 * (cwords + 1);
  return mcap * 4 + cwords * 4 + clique_ctl_ints() * 4 + scratch + 64;
}
#ifndef CLIQUE_MIN_WAVES
#define CLIQUE_MIN_WAVES 1
#endif
#define CLIQUE_WALK_MAXN_FWD 2048  // pairs up to this many matches are walked by k_clique_walk (below)
// WALK = false (the host's choice when every pair fits k_clique_walk, mcap <= CLIQUE_WALK_MAXN): degrees and sort only,
// a fraction of the registers.
template <bool WALK>
__global__ __launch_bounds__(CLIQUE_WAVES * 64, CLIQUE_MIN_WAVES) void k_clique(PairBufs pb, int walk_split) {
  RVO_OCC_SCOPE(10, 1);
  extern __shared__ __align__(16) uint8_t lds[];
  const int pair = blockIdx.x + pb.pair0, tid = threadIdx.x, lane = tid & 63;
  if (!pb.active[pair]) return;
  uint32_t* keys = reinterpret_cast<uint32_t*>(lds);
  uint32_t* acc = keys + pb.mcap;
  int* ctl = reinterpret_cast<int*>(acc + pb.cwords);
  uint32_t* tmp = reinterpret_cast<uint32_t*>(ctl + clique_ctl_ints());
  uint16_t* li = reinterpret_cast<uint16_t*>(tmp + pb.mcap);
  uint16_t* ri = li + pb.mcap;
  uint16_t* ls = ri + pb.mcap;
  uint16_t* le = ls + pb.mcap;
  const int N = pb.n_matches[pair];
  const size_t mb = (size_t)pair * pb.mcap;
  rvo_pair_result* res = pb.result + pair;
  if (N == 0) {  // computeMaximallyConsistentClique returns early (motion_estimate.cpp:444-445); 0 inliers
    if (tid == 0) res->num_inliers = 0;
    return;
  }
  const int nw = (N + 31) >> 5;
  for (int i = tid; i < N; i += CLIQUE_WAVES * 64) {  // degree = row popcount (motion_estimate.cpp:509-512)
    const uint4* r = reinterpret_cast<const uint4*>(pb.cbits + (mb + i) * pb.cwords);  // rows are 16-byte aligned and
    int deg = 0;                                                                         // zero beyond column N
#pragma unroll 4
    for (int w = 0; w < (nw + 3) >> 2; w++) {
      const uint4 q = r[w];
      deg += __popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w);
    }
    pb.m_degree[mb + i] = deg;
    keys[i] = ((uint32_t)deg << 16) | (uint32_t)i;
  }
  for (int w = tid; w < nw; w += CLIQUE_WAVES * 64) acc[w] = 0;
  __syncthreads();
  // std::sort(_matches, _matches+N, consistencyCompare): unstable, replayed exactly
  wavealgo::block_sort<16>(keys, N, li, ri, ls, le, tmp, ctl);
  for (int i = tid; i < N; i += CLIQUE_WAVES * 64) pb.order[mb + i] = keys[i];
  if (tid >= 64) return;  // the walk below is one wave; LDS accesses of one wave are ordered, so it needs no barrier
  if (!WALK || (walk_split && N <= CLIQUE_WALK_MAXN_FWD)) return;  // k_clique_walk takes it from here

  // greedy clique, motion_estimate.cpp:529-569, 64 candidates per round.  The 64 candidates' rows of the bit
  // matrix are first copied into LDS with coalesced loads (the sort scratch is dead by now and is reused); lane k
  // then owns row k: "consistent with every inlier so far" is a word-wise test against the accepted mask, and the
  // 64x64 block of mutual consistency bits resolves the order dependence inside the round without memory traffic.
  uint32_t* srow = tmp;                 // [64][pitch], aliases tmp/li/ri/ls/le (sized by clique_lds_bytes)
  const int pitch = ((nw + 3) & ~3) | 1;  // whole quads per row, odd pitch: lane-strided reads are conflict-free
  int num_inl = 0;
  bool stop = false;
  // When a row fits one word per lane (N <= 2048) the next round's 64 rows are fetched into registers while the
  // current round is resolved (the candidate order is already fixed by the sort), hiding the HBM/L2 round trip.
  const bool pipelined = nw <= 64;
  // a row is read with 16-byte loads by 8 lanes (N <= 1024) or 16 lanes (N <= 2048): 8 or 4 rows per load instruction
  const int lsh = nw <= 32 ? 3 : 4, lpr = 1 << lsh, rpi = 64 >> lsh;  // lanes per row, rows per instruction; lpr loads per round
  const int lq = lane & (lpr - 1), lrow = lane >> lsh;
  uint4 pre[16];
  const uint32_t* __restrict__ cb = pb.cbits + mb * pb.cwords;  // wave-uniform base: rows are addressed with 32-bit offsets
  const uint32_t cw = (uint32_t)pb.cwords;
  auto fetch = [&](int c0) {
    const int mm = c0 + lane;
    const int idl = (mm < N) ? (int)(keys[mm] & 0xFFFF) : 0;  // lanes past N carry id 0: a harmless in-range row
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k < lpr) {
        const uint32_t idk = (uint32_t)__shfl(idl, k * rpi + lrow, 64);
        pre[k] = (4 * lq < nw) ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(cb) + (idk * cw + 4u * (uint32_t)lq) * 4u)
                               : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  if (pipelined && !stop) fetch(0);
  for (int c0 = 0; c0 < N && !stop; c0 += 64) {
    const int m = c0 + lane;
    const bool valid = m < N;
    const uint32_t key = valid ? keys[m] : 0;
    const int id = key & 0xFFFF, deg = (int)(key >> 16);
    const int kmax = min(64, N - c0);
    if (pipelined) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        if (k < lpr && 4 * lq < nw) {  // words past nw inside the last quad are zero in memory; they land in the row's slack
          uint32_t* d = srow + (k * rpi + lrow) * pitch + 4 * lq;
          d[0] = pre[k].x; d[1] = pre[k].y; d[2] = pre[k].z; d[3] = pre[k].w;
        }
      }
      wavealgo::wave_lds_sync();
      if (c0 + 64 < N) fetch(c0 + 64);
    } else {
      for (int w = lane; w < nw + ((-nw) & 63); w += 64) {  // wave-uniform trip count
        for (int k0 = 0; k0 < 64; k0 += 16) {
          uint32_t v[16];
#pragma unroll
          for (int u = 0; u < 16; u++) {  // 16 independent coalesced row loads in flight
            const int idk = __shfl(id, k0 + u, 64);  // invalid lanes carry id 0: a harmless in-range row
            v[u] = (w < nw) ? pb.cbits[(mb + idk) * pb.cwords + w] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 16; u++)
            if (w < nw) srow[(k0 + u) * pitch + w] = v[u];
        }
      }
      wavealgo::wave_lds_sync();
    }
    const uint32_t* row = srow + lane * pitch;
    bool ok = valid;
    unsigned long long cm = 0;
    {
      uint32_t bad = 0;
#pragma unroll 8
      for (int w = 0; w < nw; w++) bad |= acc[w] & ~row[w];
      ok = valid && bad == 0;
    }
    // cm bit k of lane j = consistent(candidate j, candidate k).  Wave-uniform lane indices come from v_readlane;
    // the 64 LDS reads are issued in independent batches of 16 (lanes k >= kmax carry id 0: harmless, never alive).
    {
      uint32_t half[2] = {0u, 0u};
#pragma unroll
      for (int k0 = 0; k0 < 64; k0 += 16) {
        uint32_t wv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) wv[u] = row[(uint32_t)__builtin_amdgcn_readlane(id, k0 + u) >> 5];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const uint32_t sh = (uint32_t)__builtin_amdgcn_readlane(id, k0 + u) & 31u;
          half[k0 >> 5] |= ((wv[u] >> sh) & 1u) << ((k0 + u) & 31);
        }
      }
      cm = (unsigned long long)half[0] | ((unsigned long long)half[1] << 32);
    }
    // In-round order dependence without a serial walk over the 64 candidates: the lowest alive lane that is
    // inconsistent with an earlier alive lane is rejected (every alive lane before it is accepted); repeat until no
    // conflict is left.  Iterations = candidates rejected inside the round, typically a handful.
    unsigned long long alive = __ballot(ok);  // consistent with every inlier of the earlier rounds
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (;;) {
      const bool conf = ((alive >> lane) & 1ull) && ((~cm & alive & lt) != 0ull);
      const unsigned long long cmask = __ballot(conf);
      if (!cmask) break;
      alive &= ~(1ull << (__ffsll((long long)cmask) - 1));
    }
    // the walk stops at the first candidate (accepted or not) whose degree is below the clique size reached before it
    unsigned long long accepted = alive;
    {
      const int before = num_inl + __popcll(alive & lt);
      const unsigned long long sm = __ballot(lane < kmax && deg < before);
      if (sm) {
        accepted &= (1ull << (__ffsll((long long)sm) - 1)) - 1ull;
        stop = true;
      }
    }
    num_inl += __popcll(accepted);
    if ((accepted >> lane) & 1ull) {
      atomicOr(&acc[id >> 5], 1u << (id & 31));
      pb.m_inlier[mb + id] = 1;
    }
    wavealgo::wave_lds_sync();
  }
  if (lane == 0) res->num_inliers = num_inl;
}

// The greedy walk of k_clique for pairs of at most CLIQUE_WALK_MAXN matches, as a kernel of its own: ONE wave per pair
// with next to no LDS, so that it is placed in the wave slot and the registers seven k_fast_grid workgroups leave free on
// a CU instead of waiting for two of them to retire (k_clique holds 28 KB and 185 VGPRs for the whole walk).  Lane k
// owns candidate k of the round and keeps its row of the bit matrix in registers; "bit id of my row" for a wave-uniform
// id is a register selected by a uniform index.  Rows of up to 32 words (N <= 1024) are double-buffered: the next
// round's rows are loaded while this one is resolved; rows of up to 64 words are loaded at the start of their round.
// grid (n_pairs), 64 threads.  k_clique has sorted (pb.order) and returns before its own walk for these pairs.
#define CLIQUE_WALK_MAXN CLIQUE_WALK_MAXN_FWD
template <int WORDS, bool PREFETCH>
__device__ inline void clique_walk_rounds(const PairBufs& pb, int pair, int N, uint32_t* acc) {
  const int lane = threadIdx.x;
  const size_t mb = (size_t)pair * pb.mcap;
  const int nw = (N + 31) >> 5, nq = (nw + 3) >> 2;
  const uint32_t* __restrict__ order = pb.order + mb;
  const uint32_t* __restrict__ cb = pb.cbits + mb * pb.cwords;
  const uint32_t cw = (uint32_t)pb.cwords;
  typedef uint32_t v32_t __attribute__((ext_vector_type(32)));  // (a 64-element vector indexed by a register goes to scratch memory)
  struct row_t { v32_t h[WORDS / 32]; };
  auto fetch = [&](int c0, uint32_t* key, row_t* row) {
    const int m = c0 + lane;
    *key = m < N ? order[m] : 0u;  // lanes past N carry id 0: a harmless in-range row
    const uint4* r = reinterpret_cast<const uint4*>(cb + (size_t)(*key & 0xFFFFu) * cw);  // rows are 16-byte aligned, zero beyond column N
    row_t v;
#pragma unroll
    for (int q = 0; q < WORDS / 4; q++) {
      const uint4 t = q < nq ? r[q] : make_uint4(0u, 0u, 0u, 0u);
      v.h[q / 8][4 * (q % 8)] = t.x; v.h[q / 8][4 * (q % 8) + 1] = t.y; v.h[q / 8][4 * (q % 8) + 2] = t.z; v.h[q / 8][4 * (q % 8) + 3] = t.w;
    }
    *row = v;
  };
  uint32_t nkey = 0;
  row_t nrow;
  if (PREFETCH) fetch(0, &nkey, &nrow);
  int num_inl = 0;
  bool stop = false;
  for (int c0 = 0; c0 < N && !stop; c0 += 64) {
    uint32_t key;
    row_t row;
    if (PREFETCH) {
      key = nkey; row = nrow;
      if (c0 + 64 < N) fetch(c0 + 64, &nkey, &nrow);
    } else {
      fetch(c0, &key, &row);
    }
    const bool valid = c0 + lane < N;
    const int id = key & 0xFFFF, deg = (int)(key >> 16);
    const int kmax = min(64, N - c0);
    uint32_t bad = 0;
#pragma unroll
    for (int w = 0; w < WORDS; w++)
      if (w < nw) bad |= acc[w] & ~row.h[w / 32][w % 32];
    const bool ok = valid && bad == 0;
    // cm bit k of lane j = consistent(candidate j, candidate k): bit id_k of row j
    uint32_t half[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 64; k++) {
      const uint32_t idk = (uint32_t)__builtin_amdgcn_readlane(id, k);
      uint32_t wv = row.h[0][(idk >> 5) & 31u];  // wave-uniform index
      if (WORDS > 32) { const uint32_t hi = row.h[WORDS / 32 - 1][(idk >> 5) & 31u]; wv = (idk >> 10) ? hi : wv; }
      half[k >> 5] |= ((wv >> (idk & 31u)) & 1u) << (k & 31);
    }
    const unsigned long long cm = (unsigned long long)half[0] | ((unsigned long long)half[1] << 32);
    // (the rest is k_clique's round: in-round order dependence by ballots, stop rule, acceptance)
    unsigned long long alive = __ballot(ok);
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (;;) {
      const bool conf = ((alive >> lane) & 1ull) && ((~cm & alive & lt) != 0ull);
      const unsigned long long cmask = __ballot(conf);
      if (!cmask) break;
      alive &= ~(1ull << (__ffsll((long long)cmask) - 1));
    }
    unsigned long long accepted = alive;
    {
      const int before = num_inl + __popcll(alive & lt);
      const unsigned long long sm = __ballot(lane < kmax && deg < before);
      if (sm) {
        accepted &= (1ull << (__ffsll((long long)sm) - 1)) - 1ull;
        stop = true;
      }
    }
    num_inl += __popcll(accepted);
    if ((accepted >> lane) & 1ull) {
      atomicOr(&acc[id >> 5], 1u << (id & 31));
      pb.m_inlier[mb + id] = 1;
    }
    wavealgo::wave_lds_sync();
  }
  if (lane == 0) pb.result[pair].num_inliers = num_inl;
}
__global__ __launch_bounds__(64) void k_clique_walk(PairBufs pb) {
  RVO_OCC_SCOPE(12, 1);
  __shared__ uint32_t acc[CLIQUE_WALK_MAXN / 32];
  const int pair = blockIdx.x + pb.pair0, lane = threadIdx.x;
  if (!pb.active[pair]) return;
  const int N = pb.n_matches[pair];
  if (N == 0 || N > CLIQUE_WALK_MAXN) return;
  if (lane < CLIQUE_WALK_MAXN / 32) acc[lane] = 0u;
  wavealgo::wave_lds_sync();
  if (N <= 1024) clique_walk_rounds<32, true>(pb, pair, N, acc);
  else clique_walk_rounds<64, false>(pb, pair, N, acc);
}

// ------------------------------------------------------------------ Gauss-Newton + status
struct JacMats { double P[12], Ju[24], Jv[24], Jw[24]; };

// computeProjectionJacobian, motion_estimate.cpp:1061-1101
// sc = sin/cos of roll, pitch, yaw: {sr, cr, sp, cp, sy, cy}
__device__ inline void forward_jac_mats(const double* prm, const double* sc, double fx, double px, double py, JacMats& m) {
  const double tx = prm[0], ty = prm[1], tz = prm[2];
  const double sr = sc[0], cr = sc[1], sp = sc[2], cp = sc[3], sy = sc[4], cy = sc[5];
  const double P[12] = {fx * cp * cy - px * sp, px * cp * sr - fx * (cr * sy - cy * sp * sr), fx * (sr * sy + cr * cy * sp) + px * cp * cr, fx * tx + px * tz,
                        fx * cp * sy - py * sp, fx * (cr * cy + sp * sr * sy) + py * cp * sr, py * cp * cr - fx * (cy * sr - cr * sp * sy), fx * ty + py * tz,
                        -sp, cp * sr, cp * cr, tz};
  const double Ju[24] = {0, 0, 0, fx,
                         0, 0, 0, 0,
                         0, 0, 0, px,
                         0, fx * (sr * sy + cr * cy * sp) + px * cp * cr, fx * (cr * sy - cy * sp * sr) - px * cp * sr, 0,
                         -px * cp - fx * cy * sp, -sr * (px * sp - fx * cp * cy), -cr * (px * sp - fx * cp * cy), 0,
                         -fx * cp * sy, -fx * cr * cy - fx * sp * sr * sy, fx * cy * sr - fx * cr * sp * sy, 0};
  const double Jv[24] = {0, 0, 0, 0,
                         0, 0, 0, fx,
                         0, 0, 0, py,
                         0, py * cp * cr - fx * (cy * sr - cr * sp * sy), -fx * (cr * cy + sp * sr * sy) - py * cp * sr, 0,
                         -py * cp - fx * sp * sy, -sr * (py * sp - fx * cp * sy), -cr * (py * sp - fx * cp * sy), 0,
                         fx * cp * cy, fx * cy * sp * sr - fx * cr * sy, fx * sr * sy + fx * cr * cy * sp, 0};
  const double Jw[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, cp * cr, -cp * sr, 0, -cp, -sp * sr, -cr * sp, 0, 0, 0, 0, 0};
  for (int i = 0; i < 12; i++) m.P[i] = P[i];
  for (int i = 0; i < 24; i++) { m.Ju[i] = Ju[i]; m.Jv[i] = Jv[i]; m.Jw[i] = Jw[i]; }
}

// computeReverseProjectionJacobian, motion_estimate.cpp:1133-1184
__device__ inline void reverse_jac_mats(const double* prm, const double* sc, double fx, double px, double py, JacMats& m) {
  const double tx = prm[0], ty = prm[1], tz = prm[2];
  const double sr = sc[0], cr = sc[1], sp = sc[2], cp = sc[3], sy = sc[4], cy = sc[5];
  const double P[12] = {
      px * (sr * sy + cr * cy * sp) + fx * cp * cy, fx * cp * sy - px * (cy * sr - cr * sp * sy), px * cp * cr - fx * sp,
      -px * (tx * (sr * sy + cr * cy * sp) - ty * (cy * sr - cr * sp * sy) + tz * cp * cr) - fx * (tx * cp * cy - tz * sp + ty * cp * sy),
      py * (sr * sy + cr * cy * sp) - fx * (cr * sy - cy * sp * sr), fx * (cr * cy + sp * sr * sy) - py * (cy * sr - cr * sp * sy), cp * (py * cr + fx * sr),
      -py * (tx * (sr * sy + cr * cy * sp) - ty * (cy * sr - cr * sp * sy) + tz * cp * cr) - fx * (ty * (cr * cy + sp * sr * sy) - tx * (cr * sy - cy * sp * sr) + tz * cp * sr),
      sr * sy + cr * cy * sp, cr * sp * sy - cy * sr, cp * cr, ty * (cy * sr - cr * sp * sy) - tx * (sr * sy + cr * cy * sp) - tz * cp * cr};
  const double Ju[24] = {
      0, 0, 0, -px * (sr * sy + cr * cy * sp) - fx * cp * cy,
      0, 0, 0, px * (cy * sr - cr * sp * sy) - fx * cp * sy,
      0, 0, 0, fx * sp - px * cp * cr,
      px * cr * sy - px * cy * sp * sr, -px * cr * cy - px * sp * sr * sy, -px * cp * sr, px * ty * (cr * cy + sp * sr * sy) - px * tx * (cr * sy - cy * sp * sr) + px * tz * cp * sr,
      -cy * (fx * sp - px * cp * cr), -sy * (fx * sp - px * cp * cr), -fx * cp - px * cr * sp, fx * (tz * cp + tx * cy * sp + ty * sp * sy) - px * (tx * cp * cr * cy - tz * cr * sp + ty * cp * cr * sy),
      px * (cy * sr - cr * sp * sy) - fx * cp * sy, px * (sr * sy + cr * cy * sp) + fx * cp * cy, 0, -fx * (ty * cp * cy - tx * cp * sy) - px * (tx * (cy * sr - cr * sp * sy) + ty * (sr * sy + cr * cy * sp))};
  const double Jv[24] = {
      0, 0, 0, fx * (cr * sy - cy * sp * sr) - py * (sr * sy + cr * cy * sp),
      0, 0, 0, py * (cy * sr - cr * sp * sy) - fx * (cr * cy + sp * sr * sy),
      0, 0, 0, -cp * (py * cr + fx * sr),
      fx * (sr * sy + cr * cy * sp) + py * (cr * sy - cy * sp * sr), -fx * (cy * sr - cr * sp * sy) - py * (cr * cy + sp * sr * sy), cp * (fx * cr - py * sr),
      py * (ty * (cr * cy + sp * sr * sy) - tx * (cr * sy - cy * sp * sr) + tz * cp * sr) - fx * (tx * (sr * sy + cr * cy * sp) - ty * (cy * sr - cr * sp * sy) + tz * cp * cr),
      cp * cy * (py * cr + fx * sr), cp * sy * (py * cr + fx * sr), -sp * (py * cr + fx * sr), -(py * cr + fx * sr) * (tx * cp * cy - tz * sp + ty * cp * sy),
      py * (cy * sr - cr * sp * sy) - fx * (cr * cy + sp * sr * sy), py * (sr * sy + cr * cy * sp) - fx * (cr * sy - cy * sp * sr), 0,
      fx * (tx * (cr * cy + sp * sr * sy) + ty * (cr * sy - cy * sp * sr)) - py * (tx * (cy * sr - cr * sp * sy) + ty * (sr * sy + cr * cy * sp))};
  const double Jw[24] = {
      0, 0, 0, -sr * sy - cr * cy * sp,
      0, 0, 0, cy * sr - cr * sp * sy,
      0, 0, 0, -cp * cr,
      cr * sy - cy * sp * sr, -cr * cy - sp * sr * sy, -cp * sr, ty * (cr * cy + sp * sr * sy) - tx * (cr * sy - cy * sp * sr) + tz * cp * sr,
      cp * cr * cy, cp * cr * sy, -cr * sp, -cr * (tx * cp * cy - tz * sp + ty * cp * sy),
      cy * sr - cr * sp * sy, sr * sy + cr * cy * sp, 0, -tx * (cy * sr - cr * sp * sy) - ty * (sr * sy + cr * cy * sp)};
  for (int i = 0; i < 12; i++) m.P[i] = P[i];
  for (int i = 0; i < 24; i++) { m.Ju[i] = Ju[i]; m.Jv[i] = Jv[i]; m.Jw[i] = Jw[i]; }
}

// 6x6 inverse by Gaussian elimination with partial pivoting (stand-in for Eigen's PartialPivLU inverse).
// Same operation order as oracle/rvo_oracle.cpp:inverse6; fully unrolled with static indices so the 6x12 tableau
// stays in registers (a dynamically indexed row swap would send it to scratch memory).
__device__ inline void inverse6(const double* A, double* inv) {
  double a[6][12];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) { a[i][j] = A[i * 6 + j]; a[i][6 + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
  for (int c = 0; c < 6; c++) {
    int piv = c;
    double best = fabs(a[c][c]);
#pragma unroll
    for (int r = c + 1; r < 6; r++) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
#pragma unroll
    for (int r = c + 1; r < 6; r++)
      if (r == piv) {
#pragma unroll
        for (int j = 0; j < 12; j++) { const double t = a[c][j]; a[c][j] = a[r][j]; a[r][j] = t; }
      }
    const double d = a[c][c];
#pragma unroll
    for (int r = c + 1; r < 6; r++) {
      const double f = a[r][c] / d;
#pragma unroll
      for (int j = c; j < 12; j++) a[r][j] -= f * a[c][j];
    }
  }
#pragma unroll
  for (int c = 5; c >= 0; c--) {
    const double d = a[c][c];
#pragma unroll
    for (int j = 6; j < 12; j++) {
      double s = a[c][j];
#pragma unroll
      for (int k = c + 1; k < 6; k++) s -= a[c][k] * a[k][j];
      a[c][j] = s / d;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) inv[i * 6 + j] = a[i][6 + j];
}

#include "kernels_gn.h"

// ------------------------------------------------------------------ track ids
// MotionEstimator::matchFeatures labels every match with a track id (motion_estimate.cpp:380-391): a reference keypoint
// without one opens a new track (`_num_tracks++`, in match order), the target keypoint and the match take the reference
// keypoint's id; after the reprojection pruning the target keypoints of non-inlier matches lose theirs again
// (:216-222).  Nothing on the path reads the ids, so they are settled after the estimate in two small launches:
// k_track_count ranks the new tracks of a pair in match order, k_track_assign turns ranks into ids.
// mode 0: every pair is a MotionEstimator of its own (the streams: RGBDVO, vo.hpp:96), ctr[pair] is its _num_tracks;
// mode 1: the batch is one estimator working through the pairs in order (FrameMatcher, fm.hpp:41): ctr_in[0] is
//         _num_tracks before the batch, ctr_out[0] after it.
// grid n_pairs, 256 threads.
__global__ __launch_bounds__(256) void k_track_count(FrameBufs fb, FrameGeom g, PairBufs pb) {
  __shared__ int s_wave[8];
  const int pair = blockIdx.x + pb.pair0, tid = threadIdx.x;
  if (!pb.active[pair]) {
    if (tid == 0) pb.new_tracks[pair] = 0;
    return;
  }
  const int N = pb.n_matches[pair], rs = pb.ref_slot[pair];
  const size_t mb = (size_t)pair * pb.mcap;
  int base = 0;
  for (int c0 = 0; c0 < N; c0 += 256) {
    const int i = c0 + tid;
    int cur = 0, flag = 0;
    if (i < N) {
      cur = fb.kp_track[(size_t)rs * g.kp_total + g.lv[pb.m_level[mb + i]].kp_off + pb.m_ref[mb + i]];
      flag = cur < 0;
    }
    int total;
    const int pos = base + block_excl_scan_256(flag, s_wave, &total);
    if (i < N) pb.m_track[mb + i] = flag ? -2 - pos : cur;  // <= -2: rank of a new track
    base += total;
  }
  if (tid == 0) pb.new_tracks[pair] = base;
}

__global__ __launch_bounds__(256) void k_track_assign(FrameBufs fb, FrameGeom g, PairBufs pb, int n_pairs, int mode,
                                                      int32_t* ctr_in, int32_t* ctr_out) {
  __shared__ int s_wave[8];
  const int pair = blockIdx.x + pb.pair0, tid = threadIdx.x;  // (pair0 = 0 in mode 1)
  int base;
  if (mode == 1) {
    int before = 0, all = 0;
    for (int j = tid; j < n_pairs; j += 256) {
      const int v = pb.new_tracks[j];
      all += v;
      if (j < pair) before += v;
    }
    int tb, ta;
    block_excl_scan_256(before, s_wave, &tb);
    block_excl_scan_256(all, s_wave, &ta);
    base = ctr_in[0] + tb;
    if (pair == 0 && tid == 0) ctr_out[0] = ctr_in[0] + ta;
  } else {
    base = ctr_in[pair];
    __syncthreads();
    if (tid == 0) ctr_out[pair] = base + pb.new_tracks[pair];
  }
  if (!pb.active[pair]) return;
  const int N = pb.n_matches[pair], rs = pb.ref_slot[pair], ts = pb.tgt_slot[pair];
  const size_t mb = (size_t)pair * pb.mcap;
  for (int i = tid; i < N; i += 256) {
    const int t = pb.m_track[mb + i];
    const int id = t < -1 ? base + (-2 - t) : t;
    const int off = g.lv[pb.m_level[mb + i]].kp_off;
    pb.m_track[mb + i] = id;
    fb.kp_track[(size_t)rs * g.kp_total + off + pb.m_ref[mb + i]] = id;
    fb.kp_track[(size_t)ts * g.kp_total + off + pb.m_tgt[mb + i]] = pb.m_inlier[mb + i] ? id : -1;
  }
}

// mode 1 when two pairs of the batch share a frame slot (the batched loop closure: one reference keyframe against all
// candidates, FrameMatcher::SetRefFrame once + Match per candidate): the pairs are NOT independent then -- the first pair
// that matches a reference keypoint opens its track and the later ones reuse it (motion_estimate.cpp:385-390) -- so one
// workgroup walks the pairs in order, each pair in parallel over its matches (a keypoint occurs in one match per pair).
// grid 1, 256 threads.
__global__ __launch_bounds__(256) void k_track_serial(FrameBufs fb, FrameGeom g, PairBufs pb, int n_pairs, const int32_t* ctr_in,
                                                      int32_t* ctr_out) {
  __shared__ int s_wave[8];
  const int tid = threadIdx.x;
  int ctr = ctr_in[0];
  for (int pair = 0; pair < n_pairs; pair++) {
    if (!pb.active[pair]) {
      if (tid == 0) pb.new_tracks[pair] = 0;
      continue;
    }
    const int N = pb.n_matches[pair], rs = pb.ref_slot[pair], ts = pb.tgt_slot[pair];
    const size_t mb = (size_t)pair * pb.mcap;
    int base = 0;
    for (int c0 = 0; c0 < N; c0 += 256) {
      const int i = c0 + tid;
      int cur = 0, flag = 0, off = 0;
      if (i < N) {
        off = g.lv[pb.m_level[mb + i]].kp_off;
        cur = fb.kp_track[(size_t)rs * g.kp_total + off + pb.m_ref[mb + i]];
        flag = cur < 0;
      }
      int total;
      const int pos = base + block_excl_scan_256(flag, s_wave, &total);
      if (i < N) {
        const int id = flag ? ctr + pos : cur;
        pb.m_track[mb + i] = id;
        fb.kp_track[(size_t)rs * g.kp_total + off + pb.m_ref[mb + i]] = id;
        fb.kp_track[(size_t)ts * g.kp_total + off + pb.m_tgt[mb + i]] = pb.m_inlier[mb + i] ? id : -1;
      }
      base += total;
    }
    if (tid == 0) pb.new_tracks[pair] = base;
    ctr += base;
    __threadfence_block();
    __syncthreads();  // the next pair reads the ids this one wrote
  }
  if (tid == 0) ctr_out[0] = ctr;
}

// "These frames were just prepared" as far as tracks go (KeypointData::track_id = -1, frame.cpp:138), for frames that
// stay cached on the device where the reference would call SetRefFrame / Match on fresh Frame objects.
// grid (ceil(kp_total / 256), n), 256 threads.
__global__ __launch_bounds__(256) void k_reset_tracks(FrameBufs fb, FrameGeom g, const int32_t* __restrict__ slots) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < g.kp_total) fb.kp_track[(size_t)slots[blockIdx.y] * g.kp_total + i] = -1;
}

// ------------------------------------------------------------------ edge information for the pose graph
// PoseGraphOptimizer::AddEdge (pose_graph.cpp:254-270): info = edge.covariance.inverse(), one 6x6 partial-pivot
// inverse per accepted pair.  One thread per pair; pairs without a valid estimate get zeros.
__global__ void k_pair_information(PairBufs pb, int n, double* __restrict__ info) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const rvo_pair_result* r = pb.result + i;
  double* o = info + (size_t)i * 36;
  if (r->status != RVO_SUCCESS) {
    for (int k = 0; k < 36; k++) o[k] = 0;
    return;
  }
  double A[36], inv[36];
  for (int k = 0; k < 36; k++) A[k] = r->cov[k];
  inverse6(A, inv);
  for (int k = 0; k < 36; k++) o[k] = inv[k];
}

// ------------------------------------------------------------------ streaming state machine (vo.cpp:55-165)
struct StreamState {
  int init_cnt, frame_cnt, ref_changed_flag, fast_thr;
  int slot_ref, slot_prev, slot_curr;
  int ref_changed;  // this step's copy of the flag (vo.cpp:78)
  int processed;    // past warm-up this step
  int estimate;     // frame_cnt >= 2 this step
  int retry;        // fallback vs previous frame pending
  int pad_;
  double pose[12], ref_to_prev[12], curr_to_prev[12], curr_to_ref[12];
  double cov[36];
};

struct StreamParams {
  int warmup, thr_init, thr_min, thr_max, target_num_features, keyframe_thr;
  double gain;
};

__device__ inline void iso_set_identity12(double* p) { for (int k = 0; k < 12; k++) p[k] = (k % 5 == 0) ? 1.0 : 0.0; }

__global__ void k_stream_reset(StreamState* st, rvo_stream_result* out, int n, StreamParams sp) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  StreamState& S = st[s];
  S.init_cnt = 0; S.frame_cnt = 0; S.ref_changed_flag = 0; S.fast_thr = sp.thr_init;
  S.slot_ref = 3 * s; S.slot_prev = 3 * s + 1; S.slot_curr = 3 * s + 2;
  S.ref_changed = 0; S.processed = 0; S.estimate = 0; S.retry = 0; S.pad_ = 0;
  iso_set_identity12(S.pose); iso_set_identity12(S.ref_to_prev); iso_set_identity12(S.curr_to_prev); iso_set_identity12(S.curr_to_ref);
  for (int k = 0; k < 36; k++) S.cov[k] = 0;
  memset(&out[s], 0, sizeof(rvo_stream_result));
  iso_set_identity12(out[s].pose); iso_set_identity12(out[s].curr_to_ref);
  out[s].fast_threshold = sp.thr_init;
}

// vo.cpp:58-80: warm-up skip, buffer rotation; emits the slot/threshold the frame kernels use (-1 = skip)
// (all k_stream_* kernels: streams s0 .. s0+n-1; a stream's pair work area has the stream's index)
__global__ void k_stream_pre(StreamState* st, int s0, int n, StreamParams sp, int32_t* sslots /*[2n]: slot, thr of the n streams*/, PairBufs pb) {
  const int ls = blockIdx.x * blockDim.x + threadIdx.x, s = s0 + ls;
  if (ls >= n) return;
  StreamState& S = st[s];
  S.processed = 0; S.estimate = 0; S.retry = 0;
  pb.active[s] = 0;
  sslots[ls] = -1;
  sslots[n + ls] = S.fast_thr;
  if (S.init_cnt < sp.warmup) {
    ++S.init_cnt;
    if (S.init_cnt != sp.warmup) return;
  }
  S.processed = 1;
  if (S.ref_changed_flag) {
    const int t = S.slot_ref; S.slot_ref = S.slot_curr; S.slot_curr = t;
    iso_set_identity12(S.ref_to_prev);
  } else {
    const int t = S.slot_prev; S.slot_prev = S.slot_curr; S.slot_curr = t;
  }
  S.ref_changed = S.ref_changed_flag;
  S.ref_changed_flag = 0;
  iso_set_identity12(S.curr_to_prev);
  sslots[ls] = S.slot_curr;
}

// vo.cpp:92-110: adaptive FAST threshold, first-frame handling, choice of the initial-rotation source
__global__ void k_stream_mid(StreamState* st, int s0, int n, StreamParams sp, FrameBufs fb, FrameGeom g, PairBufs pb) {
  const int s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s0 + n) return;
  StreamState& S = st[s];
  if (!S.processed) return;
  int det = 0;
  for (int l = 0; l < g.levels; l++) det += fb.det_cnt[S.slot_curr * RVO_MAX_LEVELS + l];
  const int err = det - sp.target_num_features;
  const int adjust = (int)((double)err * sp.gain);
  S.fast_thr += adjust;
  S.fast_thr = min(S.fast_thr, sp.thr_max);
  S.fast_thr = max(S.fast_thr, sp.thr_min);
  ++S.frame_cnt;
  if (S.frame_cnt < 2) { S.ref_changed_flag = 1; return; }
  S.estimate = 1;
  pb.active[s] = 1;
  pb.ref_slot[s] = S.slot_ref;
  pb.tgt_slot[s] = S.slot_curr;
  pb.rot_slot[s] = S.ref_changed ? S.slot_ref : S.slot_prev;
}

// vo.cpp:113-115: initial estimate = ref_to_prev^-1 rotated by the initial rotation
__global__ void k_stream_init(StreamState* st, int s0, int n, PairBufs pb) {
  const int s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s0 + n) return;
  StreamState& S = st[s];
  if (!S.estimate) return;
  DIso r2p;
  iso_load(r2p, S.ref_to_prev);
  DIso init = iso_inverse(r2p);
  const double* q = pb.q0 + s * 4;
  iso_rotate(init, DQuat{q[0], q[1], q[2], q[3]});
  iso_store(init, pb.init_pose + s * 12);
}

// vo.cpp:124-143: accept the estimate vs the reference frame, or arm the retry vs the previous frame
__global__ void k_stream_post1(StreamState* st, int s0, int n, PairBufs pb) {
  const int s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s0 + n) return;
  StreamState& S = st[s];
  if (!S.estimate) return;
  pb.active[s] = 0;
  const rvo_pair_result& r = pb.result[s];
  if (r.status == RVO_SUCCESS) {
    DIso to_ref, r2p, pose;
    iso_load(to_ref, r.pose); iso_load(r2p, S.ref_to_prev); iso_load(pose, S.pose);
    iso_store(to_ref, S.curr_to_ref);
    const DIso c2p = iso_mul(r2p, to_ref);
    iso_store(c2p, S.curr_to_prev);
    for (int k = 0; k < 36; k++) S.cov[k] = r.cov[k];
    iso_store(iso_inverse(to_ref), S.ref_to_prev);
    iso_store(iso_mul(pose, c2p), S.pose);
  } else if (!S.ref_changed) {
    S.retry = 1;
    DIso init;
    iso_identity(init);
    const double* q = pb.q0 + s * 4;
    iso_rotate(init, DQuat{q[0], q[1], q[2], q[3]});
    iso_store(init, pb.init_pose + s * 12);
    pb.ref_slot[s] = S.slot_prev;
    pb.active[s] = 1;
  }
}

// vo.cpp:145-163: retry result, failure report, keyframe decision; fills the per-stream result record
// traj (may be null): this step's row of the device-side trajectory, [stream][traj_cap][RVO_TRAJ_DOUBLES] with the step
// offset already applied: pose 3x4, status, inliers, rc.
__device__ inline void traj_write(double* traj, int traj_cap, int s, const rvo_stream_result& o) {
  if (!traj) return;
  double* t = traj + (size_t)s * traj_cap * RVO_TRAJ_DOUBLES;
  for (int k = 0; k < 12; k++) t[k] = o.pose[k];
  t[12] = (double)o.status; t[13] = (double)o.num_inliers; t[14] = (double)o.rc; t[15] = (double)o.used_fallback;
}

__global__ void k_stream_post2(StreamState* st, int s0, int n, StreamParams sp, PairBufs pb, rvo_stream_result* out, double* traj, int traj_cap) {
  const int s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= s0 + n) return;
  StreamState& S = st[s];
  rvo_stream_result& o = out[s];
  o.used_fallback = 0;
  if (!S.processed) { o.rc = 0; o.fast_threshold = S.fast_thr; o.keyframe_changed = S.ref_changed_flag; traj_write(traj, traj_cap, s, o); return; }
  o.rc = 1;
  if (S.estimate) {
    const rvo_pair_result& r = pb.result[s];
    if (S.retry && r.status == RVO_SUCCESS) {
      DIso c2p, pose, r2p;
      iso_load(c2p, r.pose); iso_load(pose, S.pose); iso_load(r2p, S.ref_to_prev);
      iso_store(c2p, S.curr_to_prev);
      iso_store(iso_mul(pose, c2p), S.pose);
      S.ref_changed_flag = 1;
      iso_store(iso_mul(iso_inverse(r2p), c2p), S.curr_to_ref);
      o.used_fallback = 1;
    }
    if (r.status != RVO_SUCCESS) {
      o.rc = -1;  // the reference prints "Cannot get valid motion estimation" and exit(-1)s here
    } else if (r.num_inliers < sp.keyframe_thr) {
      S.ref_changed_flag = 1;  // vo.cpp:160-163 (threshold = option value; the reference member is uninitialised)
    }
    o.status = r.status; o.num_matches = r.num_matches; o.num_inliers = r.num_inliers;
  } else {
    o.status = RVO_NO_DATA; o.num_matches = 0; o.num_inliers = 0;
  }
  for (int k = 0; k < 12; k++) { o.pose[k] = S.pose[k]; o.curr_to_ref[k] = S.curr_to_ref[k]; }
  for (int k = 0; k < 36; k++) o.cov[k] = S.cov[k];
  o.fast_threshold = S.fast_thr;
  o.keyframe_changed = S.ref_changed_flag;
  traj_write(traj, traj_cap, s, o);
}

// ------------------------------------------------------------------ self-test of wave_stdalgo.h (one wave)
__global__ __launch_bounds__(CLIQUE_WAVES * 64) void k_selftest_stdalgo(uint32_t* keys, int n, int shift, int nth) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint32_t* a = reinterpret_cast<uint32_t*>(lds);
  uint32_t* tmp = a + n;
  uint16_t* li = reinterpret_cast<uint16_t*>(tmp + n);
  uint16_t* ri = li + n;
  uint16_t* ls = ri + n;
  uint16_t* le = ls + n;
  int* ctl = reinterpret_cast<int*>(le + n);  // 16*n bytes in: 4-byte aligned
  for (int i = threadIdx.x; i < n; i += blockDim.x) a[i] = keys[i];
  __syncthreads();
  if (nth < 0) {
    if (shift == 16) wavealgo::block_sort<16>(a, n, li, ri, ls, le, tmp, ctl);
    else wavealgo::block_sort<22>(a, n, li, ri, ls, le, tmp, ctl);
  } else if (threadIdx.x < 64) {
    if (shift == 16) wavealgo::wave_nth_element<16>(a, nth, n, li, ri);
    else wavealgo::wave_nth_element<22>(a, nth, n, li, ri);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) keys[i] = a[i];
}
