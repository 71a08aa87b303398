// rvo_dev.h -- device-side layouts shared by the kernels and the host API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rvo.h"

#define RVO_DESC_LEN 80  // feature.cpp:6081-6091: window 9 -> 81 px -> (81-1)%16==0 -> 80, stride 80, pad 0
#define RVO_WIN 9

// One pyramid level of a frame slot.
struct LevelGeom {
  int w, h, stride;  // stride = round_up(w,16)  (pyramid.cpp:114)
  int img_off;       // byte offset of the level image inside the slot's image block
  int gcols, grows;  // GridKeyPointFilter buckets (pyramid.hpp:13-21)
  int nbuckets;
  int bucket_off;    // first bucket of this level in the slot's bucket arrays
  int kp_cap;        // nbuckets * max_keypoints_per_grid
  int kp_off;        // first keypoint of this level in the slot's keypoint arrays
};

struct FrameGeom {
  int W, H, Wd, Hd;
  int levels;
  int gw, gh, max_per;
  int img_bytes;       // per slot
  int total_buckets;   // per slot
  int kp_total;        // per slot (sum of kp_cap)
  float xs, ys;        // Depth::x_scale_/y_scale_ (depth.cpp:14-15)
  double fx_inv, fy_inv, neg_cx_div_fx, neg_cy_div_fy;  // depth.cpp:23-26
  LevelGeom lv[RVO_MAX_LEVELS];
};

// Packed keypoint: x | y<<11 | score<<22  (level dims <= 2047, FAST score <= 254)
__host__ __device__ inline uint32_t kp_pack(int x, int y, int s) { return (uint32_t)x | ((uint32_t)y << 11) | ((uint32_t)s << 22); }
__host__ __device__ inline int kp_x(uint32_t k) { return k & 0x7FF; }
__host__ __device__ inline int kp_y(uint32_t k) { return (k >> 11) & 0x7FF; }
__host__ __device__ inline int kp_s(uint32_t k) { return k >> 22; }

// All frame slots, structure-of-arrays; index = slot * per_slot_extent + ...
struct FrameBufs {
  uint8_t* img;       // [slots][img_bytes]
  uint16_t* depth;    // [slots][Wd*Hd] raw millimetres, 0 = invalid (vo.cpp:83-87 conversion is done lazily)
  const uint8_t** l0_tab;      // [slots] where the slot's level-0 image lives (stride = lv[0].stride): its block of `img`, or
                               // (rvo_match_frames_dev, W % 16 == 0) the caller's device buffer, read in place
  const uint16_t** depth_tab;  // [slots] where the slot's depth image lives: its block of `depth`, or (rvo_match_frames_dev)
                               // the caller's device buffer, read in place
  int32_t* bk_cnt;    // [slots][total_buckets] kept per bucket (<= max_per)
  int32_t* bk_det;    // [slots][total_buckets] NMS survivors per bucket
  uint32_t* bk_kp;    // [slots][total_buckets][max_per] packed, post-nth_element order
  uint32_t* kp_uvs;   // [slots][kp_total] packed keypoints after the border/LUT/depth filter
  float2* kp_rect;    // [slots][kp_total] rect_base_uv (LUT values are floats, exact)
  double* kp_xyz;     // [slots][kp_total][3]
  uint8_t* bk_desc;   // [slots][total_buckets][max_per][80] descriptors of the grid-filtered keypoints, extracted by
                      // k_fast_grid from its LDS tile; kp_src names the row of every keypoint that survived k_build_keypoints
  int32_t* kp_track;  // [slots][kp_total] KeypointData::track_id: -1 after Frame::Prepare (frame.cpp:138), then what
                      // MotionEstimator::matchFeatures / EstimateMotion leave (motion_estimate.cpp:216-222, 380-391)
  uint16_t* kp_src;   // [slots][kp_total] level-local bucket slot (bucket * max_per + j) each kept keypoint came from
  int32_t* kp_cnt;    // [slots][RVO_MAX_LEVELS]
  int32_t* det_cnt;   // [slots][RVO_MAX_LEVELS] FAST corners after NMS (adaptive threshold input)
  int32_t* grid_cnt;  // [slots][RVO_MAX_LEVELS] candidates after the grid filter
  const float* xmap;  // [H][W] Camera::Xmap_ (camera.cpp:63)
  const float* ymap;
};

// match record of a pair (FeatureMatch, feature.hpp:210-308), structure-of-arrays per pair
struct PairBufs {
  int32_t* ref_slot;   // [pairs]
  int32_t* tgt_slot;   // [pairs]
  int32_t* rot_slot;   // [pairs] "prev" frame of EstimateInitialRotation (vo.cpp:107-110); = ref_slot for FrameMatcher
  int32_t* active;     // [pairs] 0 = skip every stage for this pair
  double* init_pose;   // [pairs][12]
  double* q0;          // [pairs][4]
  int32_t* lm_cnt;     // [pairs][RVO_MAX_LEVELS] matches per level
  int32_t* lm_ref;     // [pairs][kp_total] level segments at lv.kp_off
  int32_t* lm_tgt;
  uint32_t* mt_br;     // [pairs][kp_total] k_match in parts: best (score << 16 | target) of every reference keypoint
  uint32_t* mt_bt;     // [pairs][MATCH_PARTS_MAX][kp_total] per part: best (score << 16 | reference) of every target keypoint
  int32_t* n_matches;  // [pairs]
  int32_t* m_ref;      // [pairs][mcap] level-local keypoint index
  int32_t* m_tgt;
  uint8_t* m_level;    // [pairs][mcap]
  uint8_t* m_status;
  float2* m_uv;        // refined kp.u, kp.v
  double* m_rect;      // [pairs][mcap][2] refined rect_base_uv
  double* m_xyz;       // [pairs][mcap][3] refined xyz
  double* m_rrect;     // [pairs][mcap][2] reference keypoint rect_base_uv (gathered copy)
  double* m_rxyz;      // [pairs][mcap][3] reference keypoint xyz (gathered copy)
  int32_t* m_degree;   // [pairs][mcap]
  uint8_t* m_inlier;   // [pairs][mcap] indexed by id
  double* m_reproj;    // [pairs][mcap] indexed by id
  int32_t* m_track;    // [pairs][mcap] FeatureMatch::track_id, indexed by id
  int32_t* new_tracks; // [pairs] tracks the pair's EstimateMotion opened (reference keypoints that had none)
  uint32_t* order;     // [pairs][mcap] sorted keys degree<<16|id
  uint32_t* cbits;     // [pairs][mcap][cwords] consistency bit matrix
  rvo_pair_result* result;  // [pairs]
  int mcap, cwords;
  int pair0;           // first pair work area of this launch (chunks of a batch run on different HIP streams)
};

// image of pyramid level `lvl` of a slot
// v_writelane_b32 (clang has no builtin for it; the LLVM intrinsic is reached through its assembler name, and the
// compiler schedules it and its hazards like any other instruction): lane `lane` of `old` := the scalar `val`.
extern "C" __device__ unsigned __llvm_amdgcn_writelane(unsigned val, unsigned lane, unsigned old) __asm("llvm.amdgcn.writelane.i32");

// XCD-aware workgroup mapping.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in dispatch order,
// so of a grid of nx work items for each of ny units (frames, pairs) the items of one unit would sit behind 8 different
// L2s although they read the same images.  Within every group of 8 units the linear ids are re-dealt so that all items
// of a unit are congruent mod 8 (same XCD); a tail of fewer than 8 units keeps the natural mapping.
__device__ inline void xcd_map(int nx, int ny, int bx, int by, int* item, int* unit) {
  const int grp = by >> 3;  // = (by * nx + bx) / (8 * nx), without the division
  *item = bx; *unit = by;
  if (grp * 8 + 8 <= ny) {
    const int r = (by & 7) * nx + bx;
    *unit = grp * 8 + (r & 7);
    *item = r >> 3;
  }
}

// ceil(65536 / n) for n = 1..256: x / n == (x * kInv16[n]) >> 16 whenever x * n < 65536 (the error of the product is
// below x / 65536, which can only matter when x mod n == n - 1 and 1/n <= x / 65536).  The divisors in
// k_fast_grid are wave-uniform but not compile-time constants; without the table each costs a dozen VALU instructions
// of float reciprocal + fix-up.
struct Inv16Table { uint32_t v[257]; };
static constexpr Inv16Table make_inv16() {
  Inv16Table t{};
  t.v[0] = 0;
  for (int n = 1; n <= 256; n++) t.v[n] = (65536u + n - 1) / n;
  return t;
}
__constant__ Inv16Table kInv16 = make_inv16();
// floor((2^32 - 1) / n) + 1: idx / n == umulhi(idx, kInv32[n]) for idx < 2^16, 2 <= n <= 2^8.  n = 1 has no entry (2^32 does
// not fit a word; the slot holds 0): every use goes through div_inv32 / div_u16, which take x / 1 = x themselves -- a level
// one bucket wide, or one keypoint per bucket, once divided everything to 0 here (found by tools/fuzz_parity.py).
static constexpr Inv16Table make_inv32() {
  Inv16Table t{};
  t.v[0] = 0;
  t.v[1] = 0;
  for (int n = 2; n <= 256; n++) t.v[n] = 0xFFFFFFFFu / (uint32_t)n + 1u;
  return t;
}
__constant__ Inv16Table kInv32 = make_inv32();
// x / n with inv = kInv32.v[n] fetched once (x < 2^16, 1 <= n <= 256)
__device__ inline uint32_t div_inv32(uint32_t x, int n, uint32_t inv) { return n == 1 ? x : __umulhi(x, inv); }
// x / n for a divisor that is uniform but only known at run time (x < 2^16)
__device__ inline int div_u16(int x, int n) { return n == 1 ? x : n <= 256 ? (int)__umulhi((uint32_t)x, kInv32.v[n]) : x / n; }


// A pointer that was loaded from memory (the per-slot image / depth tables) is "generic" to the compiler, and accesses
// through it become FLAT instructions, which also occupy the LDS queue and its wait counter.  Everything these tables
// name is device global memory: gload reads through an address-space-1 pointer, i.e. with a GLOBAL load.
template <class T>
__device__ inline T gload(const void* p) {  // scalar types
  return *(const __attribute__((address_space(1))) T*)p;
}
__device__ inline uint4 gload_u4(const void* p) {  // 16 bytes, 16-byte aligned
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = *(const __attribute__((address_space(1))) u32x4_t*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline const uint8_t* level_img(const FrameBufs& fb, const FrameGeom& g, int slot, int lvl) {
  return lvl == 0 ? fb.l0_tab[slot] : fb.img + (size_t)slot * g.img_bytes + g.lv[lvl].img_off;
}

// ---- occupancy timeline (one-off builds with -DRVO_OCC only, tools/exp_occupancy.py): every workgroup adds +1 / -1
// to a histogram of 10 us buckets of the constant 100 MHz clock when it starts / ends, per kernel kind; the running sum
// over the buckets is the number of resident workgroups of that kernel over time, without any profiler in the way.
#ifdef RVO_OCC
#define RVO_OCC_KINDS 16
#define RVO_OCC_BUCKETS 65536
__device__ int g_occ[RVO_OCC_KINDS][RVO_OCC_BUCKETS];
struct OccScope {  // big grids are sampled (one workgroup in `sample`, counted `sample` times): the atomics must not become the bottleneck
  int kind, sample;
  bool on;
  __device__ OccScope(int k, int smp) : kind(k), sample(smp) {
    on = threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0 && (blockIdx.x + blockIdx.y * 7 + blockIdx.z * 13) % smp == 0;
    mark(smp);
  }
  __device__ ~OccScope() { mark(-sample); }
  __device__ void mark(int d) const {
    if (on) atomicAdd(&g_occ[kind][(wall_clock64() / 1000) & (RVO_OCC_BUCKETS - 1)], d);
  }
};
#define RVO_OCC_SCOPE(k, smp) OccScope occ_scope_(k, smp)
#else
#define RVO_OCC_SCOPE(k, smp) do { } while (0)
#endif

struct CamParams {
  double fx, fy, cx, cy;  // rectified camera (camera.cpp:7-12)
  double in_fx;           // input camera fx (motion_estimate.cpp:297-299)
};

struct EstParams {
  double inlier_max_reprojection_error, clique_inlier_threshold, max_mean_reprojection_error, feature_search_window;
  int min_features;
  double cons_band;  // k_consistency: half-width of the fp32 decision band around clique_inlier_threshold (metres)
};
