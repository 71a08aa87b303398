// kernels_gn.h -- k_gn: MotionEstimator::refineMotionEstimate() twice with the reprojection pruning in between and the
// final status (rgbdvo/motion_estimate.cpp:198-232, 610-720, 1208-1331).  Included by kernels_pair.h.
//
// One workgroup of 256 threads (one wave per SIMD) per pair; 168 VGPRs and 7.3 KB of LDS with the inlier list: what seven
// k_fast_grid workgroups (39 VGPRs, 22 KB each) leave free on a CU, so a workgroup of this latency-bound kernel is placed
// without waiting for FAST workgroups to retire.
//
// refineMotionEstimateBidirectional (:1208-1331) evaluates, per iteration, the Jacobian at the current parameters and
// then the error at the trial estimate.  The error at an ACCEPTED trial estimate is the error vector of the next
// iteration and its parameters are the next iteration's linearisation point, so one pass over the inliers per iteration
// does both: it returns sse(trial), J'J(trial) and J'err(trial).  A rejected trial's sums are dropped; J'J at the
// accepted estimate is also exactly what the covariance (:1319-1329) inverts, so that costs no pass of its own.
// Every value is computed by the same expressions in the same order as oracle/rvo_oracle.cpp; only the summation over
// inliers is tree-ordered (fixed tree: deterministic).
#pragma once

#ifndef GN_THREADS
#define GN_THREADS 256
#endif
#define GN_WAVES (GN_THREADS / 64)
#ifndef GN_MIN_WAVES
#define GN_MIN_WAVES 3
#endif
#define GN_SEGS (GN_MAX_SEG_PER_WAVE * GN_WAVES)  // 64-match segments per round of the inlier-list builder
#define GN_MAX_SEG_PER_WAVE 8
#define GN_NACC 28  // 21 sums of J'J (upper triangle, row-major), 6 of J'err, 1 of err'err

struct GnShared {
  JacMats jac[2];      // forward / reverse projection Jacobian matrices at the parameters under evaluation
  double P[2][12];     // K(fx,fx) * T and K(fx,fx) * T^-1 of the estimate under evaluation
  double T[6][12];     // elimination tableau of the 6x6 inverse
  double H[GN_NACC];   // sums of the last pass
  double Hacc[GN_NACC];  // sums at the accepted estimate
  double part[GN_WAVES][32];  // per wave: its sums of the 28 (padded to 32) values
  double est[6], next[6], delta[6], motion[12], nmotion[12];
  double sc[6], sch[6];  // sin/cos of the trial roll, pitch, yaw and of their halves
  double final_sse, mean;
  int stop, status, n_inl;
  uint16_t segcnt[GN_MAX_SEG_PER_WAVE * GN_WAVES];
};

// sum over the 64 lanes of a wave on the DPP network; the total is valid in lane 63 (fixed order)
template <int CTRL, int ROW_MASK>
__device__ inline double gn_dpp_add(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ inline double gn_wave_sum63(double v) {
  v = gn_dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = gn_dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = gn_dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = gn_dpp_add<0x140, 0xF>(v);  // row_mirror: every lane of a row holds the row's sum
  v = gn_dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = gn_dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// Reduce-scatter of 32 doubles per lane over each row of 16 lanes: in every step a lane keeps one half of its values,
// hands the other half to its partner and adds what the partner hands over, so 32 values cost 16 + 8 + 4 + 2 exchanges
// instead of 32 x 4.  The partners (row mirror, half-row mirror, reversed quad, swapped pair) always lie inside the group
// of lanes that kept the same half in the steps before.  Lane l ends with the row's sums of values 2*(l & 15) and
// 2*(l & 15) + 1 in v[0], v[1].  Fixed order: deterministic.
template <int CTRL>
__device__ inline double gn_dpp_mov(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
template <int CTRL, int HALF>
__device__ inline void gn_exchange(double* v, bool upper) {
#pragma unroll
  for (int k = 0; k < HALF; k++) {
    const double keep = upper ? v[HALF + k] : v[k];
    const double send = upper ? v[k] : v[HALF + k];
    v[k] = keep + gn_dpp_mov<CTRL>(send);
  }
}
__device__ inline void gn_row_reduce_scatter32(double* v, int lane) {
  gn_exchange<0x140, 16>(v, (lane & 8) != 0);  // row_mirror: i <-> 15 - i
  gn_exchange<0x141, 8>(v, (lane & 4) != 0);   // row_half_mirror: i <-> 7 - i
  gn_exchange<0x1B, 4>(v, (lane & 2) != 0);    // quad_perm [3,2,1,0]: i <-> 3 - i
  gn_exchange<0xB1, 2>(v, (lane & 1) != 0);    // quad_perm [1,0,3,2]
}

// 6x6 inverse by Gaussian elimination with partial pivoting, the operation order of inverse6 (oracle/rvo_oracle.cpp),
// spread over the lanes of ONE wave with the tableau in LDS: the row updates of a step run on 60 lanes (5 rows x 12
// columns), the back substitution on 6 lanes (one per column of the inverse).  A holds the packed upper triangle.
// On return T[r][6 + j] is the inverse.  LDS accesses of one wave are ordered; wave_lds_sync keeps the compiler in line.
__device__ inline void gn_wave_inverse6(double (*T)[12], const double* A, int lane) {
  for (int e = lane; e < 72; e += 64) {
    const int r = e / 12, j = e - r * 12;
    double v;
    if (j < 6) {
      const int a = min(r, j), b = max(r, j);
      v = A[a * 6 - a * (a - 1) / 2 + (b - a)];
    } else {
      v = (j - 6 == r) ? 1.0 : 0.0;
    }
    T[r][j] = v;
  }
  wavealgo::wave_lds_sync();
#pragma unroll 1
  for (int c = 0; c < 6; c++) {
    int piv = c;
    double best = fabs(T[c][c]);
    for (int r = c + 1; r < 6; r++) {
      const double v = fabs(T[r][c]);
      if (v > best) { best = v; piv = r; }
    }
    if (piv != c && lane < 12) {
      const double t = T[c][lane], u = T[piv][lane];
      T[c][lane] = u;
      T[piv][lane] = t;
    }
    wavealgo::wave_lds_sync();
    const int r = c + 1 + lane / 12, j = lane % 12;
    const bool act = r < 6 && j >= c;
    double nv = 0;
    if (act) {
      const double d = T[c][c];
      const double f = T[r][c] / d;
      nv = T[r][j] - f * T[c][j];
    }
    wavealgo::wave_lds_sync();  // every read of this step is done before the row is rewritten
    if (act) T[r][j] = nv;
    wavealgo::wave_lds_sync();
  }
#pragma unroll 1
  for (int c = 5; c >= 0; c--) {
    if (lane < 6) {
      const double d = T[c][c];
      double s = T[c][6 + lane];
      for (int k = c + 1; k < 6; k++) s -= T[c][k] * T[k][6 + lane];
      T[c][6 + lane] = s / d;
    }
    wavealgo::wave_lds_sync();
  }
}

// twelve matrix words (three rows of four) from LDS with 16-byte reads
typedef double gn_d2 __attribute__((ext_vector_type(2)));
__device__ inline void gn_ld12(double* o, const double* r0, const double* r1, const double* r2) {
  const double* r[3] = {r0, r1, r2};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const gn_d2 a = *reinterpret_cast<const gn_d2*>(r[i]), b = *reinterpret_cast<const gn_d2*>(r[i] + 2);
    o[4 * i] = a.x; o[4 * i + 1] = a.y; o[4 * i + 2] = b.x; o[4 * i + 3] = b.y;
  }
}

// isometryFromParams (motion_estimate.cpp:903-948) with the six half-angle sin/cos values already computed:
// sch = {sin(roll/2), cos(roll/2), sin(pitch/2), cos(pitch/2), sin(yaw/2), cos(yaw/2)}
__device__ inline DIso gn_iso_from_params(const double* p, const double* sch) {
  const double sr = sch[0], cr = sch[1], sp = sch[2], cp = sch[3], sy = sch[4], cy = sch[5];
  DQuat q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  DIso r;
  iso_identity(r);
  r.t[0] = p[0]; r.t[1] = p[1]; r.t[2] = p[2];
  iso_rotate(r, q);
  return r;
}

#ifdef RVO_GN_TIMING  // one-off builds only (tools/exp_gn_phases.py): thread 0 of pair 0 prints where its time went
#define GN_T(i) do { if (tid == 0) { const long long t_ = (long long)__builtin_readcyclecounter(); gn_t[i] += t_ - gn_last; gn_last = t_; } } while (0)
#else
#define GN_T(i) do { } while (0)
#endif

// grid (n_pairs), GN_THREADS threads, dynamic LDS = mcap * 2 bytes (the inlier list)
__global__ __launch_bounds__(GN_THREADS, GN_MIN_WAVES) void k_gn(PairBufs pb, CamParams cam, EstParams ep) {
  RVO_OCC_SCOPE(11, 1);
  __shared__ __align__(16) GnShared S;
  extern __shared__ __align__(16) uint8_t gn_dyn[];
  uint16_t* list = reinterpret_cast<uint16_t*>(gn_dyn);  // ids of the inliers, ascending
  const int pair = blockIdx.x + pb.pair0, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (!pb.active[pair]) return;
  const int N = pb.n_matches[pair];
  const size_t mb = (size_t)pair * pb.mcap;
  rvo_pair_result* res = pb.result + pair;
  const double fx = cam.fx, px = cam.cx, py = cam.cy;

  if (tid == 0) {
    S.status = RVO_SUCCESS;  // motion_estimate.cpp:190-192
    S.mean = 0;
    for (int k = 0; k < 12; k++) S.motion[k] = pb.init_pose[pair * 12 + k];
  }
  if (tid < 36) res->cov[tid] = 0;
  int n_fail = 0;  // (thread 0) inliers dropped by the reprojection test
#ifdef RVO_GN_TIMING
  long long gn_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, gn_last = (long long)__builtin_readcyclecounter();
  int gn_iters = 0;
#endif

#pragma unroll 1
  for (int round = 0; round < 2; round++) {
    // ---- inlier list (ids ascending): the order every per-inlier sum below runs in
    __syncthreads();
    // segment g = ids [64g, 64g+64) is handled by wave g % GN_WAVES: one ballot per segment, the segment counts go to
    // LDS, every lane then sums the counts before its segments (two barriers per GN_SEGS segments = 2048 matches)
    int n_inl = 0;
    {
      const int nseg = (N + 63) >> 6;
#pragma unroll 1
      for (int g0 = 0; g0 < nseg; g0 += GN_SEGS) {
        unsigned long long mybits[GN_MAX_SEG_PER_WAVE];
#pragma unroll
        for (int t = 0; t < GN_MAX_SEG_PER_WAVE; t++) {
          const int gl = wid + t * GN_WAVES, g = g0 + gl;
          const int i = g * 64 + lane;
          bool flag = false;
          if (i < N) {
            flag = pb.m_inlier[mb + i] != 0;
            // drop inliers with a high reprojection error (motion_estimate.cpp:203-214) before the second refinement
            if (round == 1 && flag && pb.m_reproj[mb + i] > ep.inlier_max_reprojection_error) { flag = false; pb.m_inlier[mb + i] = 0; }
          }
          mybits[t] = __ballot(flag);
          if (lane == 0) S.segcnt[gl] = (uint16_t)__popcll(mybits[t]);
        }
        __syncthreads();
        int before = n_inl, g_next = 0;
#pragma unroll
        for (int t = 0; t < GN_MAX_SEG_PER_WAVE; t++) {
          const int gl = wid + t * GN_WAVES;
          for (; g_next < gl; g_next++) before += S.segcnt[g_next];
          const unsigned long long m = mybits[t];
          if ((m >> lane) & 1ull) list[before + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((g0 + gl) * 64 + lane);
        }
        for (; g_next < GN_SEGS; g_next++) before += S.segcnt[g_next];
        n_inl = before;
        __syncthreads();
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (round == 1) n_fail = S.n_inl - n_inl;
      S.n_inl = n_inl;
    }
    GN_T(0);  // inlier list

    // ---- refineMotionEstimate(), motion_estimate.cpp:610-682 -> refineMotionEstimateBidirectional :1208-1331
    if (n_inl < ep.min_features) {
      if (tid == 0) S.status = RVO_INSUFFICIENT_INLIERS;
      __syncthreads();
    } else {
      if (tid == 0) {
        DIso m;
        iso_load(m, S.motion);
        iso_to_params(m, S.est);
        for (int a = 0; a < 6; a++) S.next[a] = S.est[a];
        S.stop = 0;
      }
      bool finishing = false;
      GN_T(1);  // parameters of the initial estimate
#pragma unroll 1
      for (int it = -1;; it++) {
        __syncthreads();
        GN_T(2);  // barrier
        if (wid == 0) {
          if (it >= 0) {
            gn_wave_inverse6(S.T, S.Hacc, lane);
            if (finishing) {  // covariance = (M'M)^-1 at the accepted parameters, :1319-1329
              if (lane < 36) res->cov[lane] = S.T[lane / 6][6 + lane % 6];
            } else if (lane < 6) {
              double s = 0;
              for (int b = 0; b < 6; b++) s += (-S.T[lane][6 + b]) * S.Hacc[21 + b];
              S.delta[lane] = s;
              S.next[lane] = S.est[lane] + s;
            }
            wavealgo::wave_lds_sync();
          }
          GN_T(3);  // 6x6 inverse + step
          if (!finishing && lane < 6) {  // lanes 0-2: the angles (Jacobians), lanes 3-5: their halves (isometryFromParams)
            const double ang = S.next[3 + (lane < 3 ? lane : lane - 3)];
            const double x = lane < 3 ? ang : ang / 2;
            double* o = lane < 3 ? S.sc + 2 * lane : S.sch + 2 * (lane - 3);
            double sn, cs;
            sincos(x, &sn, &cs);
            o[0] = sn;
            o[1] = cs;
          }
        }
        GN_T(4);  // sin / cos
        if (finishing) break;
        __syncthreads();
        if (tid == 0) {  // the estimate under evaluation and its two projection matrices (fx on both axes, :1221-1223)
          DIso nm;
          if (it < 0) iso_load(nm, S.motion);  // the initial estimate itself, not its parameters' round trip (:1233-1240)
          else nm = gn_iso_from_params(S.next, S.sch);
          iso_store(nm, S.nmotion);
          proj_matrix(fx, fx, px, py, nm, S.P[0]);
          proj_matrix(fx, fx, px, py, iso_inverse(nm), S.P[1]);
        }
        if (tid == 64) forward_jac_mats(S.next, S.sc, fx, px, py, S.jac[0]);
        if (tid == 128) reverse_jac_mats(S.next, S.sc, fx, px, py, S.jac[1]);
        GN_T(5);  // trial isometry + projection matrices (thread 0's share)
        __syncthreads();
        GN_T(6);  // waiting for the Jacobian matrices
        // ---- one pass over the inliers: error at the estimate, Jacobian at its parameters
        double acc[32];  // GN_NACC sums, padded for the reduce-scatter
#pragma unroll
        for (int k = 0; k < 32; k++) acc[k] = 0;
        // waves 0 and 2 take the forward direction (target points into the reference image), waves 1 and 3 the reverse
        // one: the matrices are then the same LDS words for a whole wave (broadcast reads).  The 96 matrix words stay in
        // LDS (hoisted out of the loop they would take 192 registers) and are fetched in groups of 12 that are issued one
        // group ahead of their use; the next inlier's point is fetched from memory while this one is processed.
        const int dir = wid & 1;
        const double* xyz_base = (dir ? pb.m_rxyz : pb.m_xyz) + mb * 3;
        const double* prj_base = (dir ? pb.m_rect : pb.m_rrect) + mb * 2;
        int k = (wid >> 1) * 64 + lane;
        double nx = 0, ny = 0, nz = 0, np0 = 0, np1 = 0;
        if (k < n_inl) {
          const int i = list[k];
          nx = xyz_base[i * 3]; ny = xyz_base[i * 3 + 1]; nz = xyz_base[i * 3 + 2];
          np0 = prj_base[i * 2]; np1 = prj_base[i * 2 + 1];
        }
#pragma unroll 1
        while (k < n_inl) {
          const double x = nx, y = ny, z = nz, pj0 = np0, pj1 = np1;
          k += GN_THREADS / 2;
          if (k < n_inl) {
            const int i = list[k];
            nx = xyz_base[i * 3]; ny = xyz_base[i * 3 + 1]; nz = xyz_base[i * 3 + 2];
            np0 = prj_base[i * 2]; np1 = prj_base[i * 2 + 1];
          }
          int joff = dir;
          asm volatile("" : "+v"(joff));  // opaque: keeps the matrix reads inside the loop
          const double* Jd = S.jac[joff].P;  // P[12] Ju[24] Jv[24] Jw[24], contiguous
          const double* Pe = S.P[joff];
          double A[12], B[12], R[12];
          gn_ld12(A, Jd, Jd + 4, Jd + 8);
          gn_ld12(B, Pe, Pe + 4, Pe + 8);
          gn_ld12(R, Jd + 12, Jd + 36, Jd + 60);
          __builtin_amdgcn_sched_barrier(0);
          const double u = dot4(A, x, y, z, 1.0), v = dot4(A + 4, x, y, z, 1.0), w = dot4(A + 8, x, y, z, 1.0);
          const double w_inv_sq = 1 / (w * w);
          const double eu = dot4(B, x, y, z, 1.0), ev = dot4(B + 4, x, y, z, 1.0), ew = dot4(B + 8, x, y, z, 1.0);
          const double e0 = eu / ew - pj0, e1 = ev / ew - pj1;
          double ru[6], rv[6];
#pragma unroll
          for (int q = 0; q < 6; q++) {
            double Rn[12];
            if (q < 5) gn_ld12(Rn, Jd + 12 + 4 * (q + 1), Jd + 36 + 4 * (q + 1), Jd + 60 + 4 * (q + 1));
            __builtin_amdgcn_sched_barrier(0);
            const double du = dot4(R, x, y, z, 1.0), dv = dot4(R + 4, x, y, z, 1.0), dw = dot4(R + 8, x, y, z, 1.0);
            ru[q] = (du * w - dw * u) * w_inv_sq;
            rv[q] = (dv * w - dw * v) * w_inv_sq;
            if (q < 5) {
#pragma unroll
              for (int t = 0; t < 12; t++) R[t] = Rn[t];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          int q = 0;
#pragma unroll
          for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = a; b < 6; b++) { acc[q] += ru[a] * ru[b]; acc[q] += rv[a] * rv[b]; q++; }
            acc[21 + a] += ru[a] * e0;
            acc[21 + a] += rv[a] * e1;
          }
          acc[27] += e0 * e0;
          acc[27] += e1 * e1;
        }
        GN_T(7);  // pass over the inliers
        gn_row_reduce_scatter32(acc, lane);
        // the four rows of the wave, lane by lane: (row 0 + row 1) + (row 2 + row 3), then the waves in order
#pragma unroll
        for (int q = 0; q < 2; q++) {
          acc[q] += __shfl_xor(acc[q], 16, 64);
          acc[q] += __shfl_xor(acc[q], 32, 64);
        }
        if (lane < 16) {
          double* o = &S.part[wid][2 * lane];
          o[0] = acc[0];
          o[1] = acc[1];
        }
        __syncthreads();
        if (tid < GN_NACC) {
          double s = S.part[0][tid];
#pragma unroll
          for (int r = 1; r < GN_WAVES; r++) s += S.part[r][tid];
          S.H[tid] = s;
        }
        __syncthreads();
        GN_T(8);  // reduction
        if (wid == 0) {  // every lane of wave 0 takes the same decision; the copies of an accepted estimate are spread over lanes
          const double sse = S.H[27];
          const bool reject = it >= 0 && sse > S.final_sse;  // stepping would increase the error (:1294); NaN falls through
          bool small = false;
          if (it >= 0) {
            const double rt = 0.01 * M_PI / 180;
            const double* d = S.delta;
            small = fabs(d[0]) < 0.0001 && fabs(d[1]) < 0.0001 && fabs(d[2]) < 0.0001 && fabs(d[3]) < rt && fabs(d[4]) < rt && fabs(d[5]) < rt;
          }
          wavealgo::wave_lds_sync();  // all lanes have read final_sse before lane 0 replaces it
          if (!reject) {
            if (lane < 6) S.est[lane] = S.next[lane];
            if (lane < 12) S.motion[lane] = S.nmotion[lane];
            if (lane < GN_NACC) S.Hacc[lane] = S.H[lane];
            if (lane == 0) S.final_sse = sse;
          }
          if (lane == 0 && (reject || small)) S.stop = 1;
        }
        __syncthreads();
        GN_T(9);  // decision
#ifdef RVO_GN_TIMING
        gn_iters++;
#endif
        if (S.stop || it == 5) finishing = true;
      }
      __syncthreads();
    }

    // ---- computeReprojectionError(), motion_estimate.cpp:684-720 (K uses fx, fy); skipped unless SUCCESS
    if (S.status == RVO_SUCCESS) {
      if (tid == 0) {
        DIso m;
        iso_load(m, S.motion);
        proj_matrix(cam.fx, cam.fy, cam.cx, cam.cy, m, S.P[0]);
      }
      __syncthreads();
      double sum = 0;
      for (int k = tid; k < n_inl; k += GN_THREADS) {
        const int i = list[k];
        const double* t = pb.m_xyz + (mb + i) * 3;
        const double h0 = dot4(S.P[0], t[0], t[1], t[2], 1.0), h1 = dot4(S.P[0] + 4, t[0], t[1], t[2], 1.0), h2 = dot4(S.P[0] + 8, t[0], t[1], t[2], 1.0);
        const double ex = pb.m_rrect[(mb + i) * 2] - h0 / h2, ey = pb.m_rrect[(mb + i) * 2 + 1] - h1 / h2;
        const double e = sqrt(ex * ex + ey * ey);
        pb.m_reproj[mb + i] = e;
        sum += e;
      }
      sum = gn_wave_sum63(sum);
      if (lane == 63) S.part[wid][0] = sum;
      __syncthreads();
      if (tid == 0) {
        double t = S.part[0][0];
        for (int w = 1; w < GN_WAVES; w++) t += S.part[w][0];
        S.mean = t / (double)n_inl;
      }
      __syncthreads();
      GN_T(10);  // reprojection errors
      // (the writes to m_reproj are read by the same threads' successors only after the barrier at the top of the loop)
    }
  }
  __syncthreads();
  if (tid == 0) {
    int status = S.status;
    if (S.mean > ep.max_mean_reprojection_error) status = RVO_REPROJECTION_ERROR;
    res->status = status;
    res->num_matches = N;
    res->num_inliers = S.n_inl;
    res->num_reprojection_failures = n_fail;
    res->mean_reprojection_error = S.mean;
    for (int k = 0; k < 12; k++) res->pose[k] = S.motion[k];
  }
#ifdef RVO_GN_TIMING
  if (tid == 0 && pair == 0) {
    long long tot = 0;
    for (int k = 0; k < 11; k++) tot += gn_t[k];
    printf("k_gn pair 0: N %d inliers %d passes %d cycles %lld | list %lld params %lld barrier %lld solve %lld sincos %lld trial %lld jacwait %lld pass %lld reduce %lld decide %lld reproj %lld\n",
           N, S.n_inl, gn_iters, tot, gn_t[0], gn_t[1], gn_t[2], gn_t[3], gn_t[4], gn_t[5], gn_t[6], gn_t[7], gn_t[8], gn_t[9], gn_t[10]);
  }
#endif
}
