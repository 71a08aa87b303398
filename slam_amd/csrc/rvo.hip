// rvo.hip -- host side of librgbdvo_hip.so: context, launch sequencing and the C ABI of include/rvo.h.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels_frame.h"
#include "kernels_pair.h"
#include "kernels_io.h"
#include "rvo_dev.h"

static thread_local std::string g_err;
struct rvo_ctx;
static int pair_alloc(rvo_ctx* c);
const char* rvo_last_error(void) { return g_err.c_str(); }

// Staging area of rvo_get_pair_information_noalloc (the entry point that allocates nothing, include/rvo.h): one 6x6
// information matrix per pair, RVO_NOALLOC_MAX_PAIRS of them = 1 179 648 bytes that every GPU the library runs on holds
// until the process ends.  In host code the NAME `scratch` is only a stand-in for the symbol: the memory is reached
// through the per-device address scratch_for_device() resolves, never through the name.
static __device__ __align__(256) double scratch[(size_t)RVO_NOALLOC_MAX_PAIRS * 36];
static const int kMaxDevices = 64;
static double* g_scratch_ptr[kMaxDevices];  // [device] address of that device's copy of `scratch`; nullptr = not resolved yet
static std::mutex g_scratch_mu;

#define HIPCHK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      g_err = std::string(#x) + ": " + hipGetErrorString(e_);                            \
      return RVO_E_HIP;                                                                  \
    }                                                                                    \
  } while (0)

static inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

struct ProfRec { int name; hipEvent_t a, b; };

struct rvo_ctx {
  int device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  rvo_intrinsics intr;
  rvo_params prm;
  FrameGeom g;
  FrameBufs fb;
  PairBufs pb;
  CamParams cam;
  EstParams ep;
  FastLds fl;
  int max_frames = 0, max_pairs = 0;
  std::vector<void*> allocs;
  std::vector<float> xmap, ymap;
  std::vector<uint8_t> seen;  // check_slots scratch
  int32_t* d_fargs = nullptr;  // [2*max_frames] slots, thr
  uint8_t* d_gray_stage = nullptr;
  uint16_t* d_depth_stage = nullptr;
  size_t stage_frames = 0;
  // rvo_stream_step_host_async: two staging sets filled by a copy stream while the previous step computes
  uint8_t* d_gray_db[2] = {nullptr, nullptr};
  uint16_t* d_depth_db[2] = {nullptr, nullptr};
  size_t db_frames = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copy[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
  bool ev_used_rec[2] = {false, false};
  int db_flip = 0;
  // streams
  StreamState* d_streams = nullptr;
  rvo_stream_result* d_stream_res = nullptr;
  int n_streams = 0;
  int32_t* d_sslots = nullptr;  // [2n]: current slot (-1 = warm-up skip), FAST threshold (device-written)
  int32_t* d_src_index = nullptr;  // rvo_streams_set_source_index: [stream] image of the step's arrays the stream reads
  bool step_from_host = false;     // inside rvo_stream_step / rvo_stream_step_host_async
  bool depth_in_place = false;     // rvo_streams_depth_in_place: rvo_stream_step_dev reads the caller's depth images where they lie
  int src_sources = 0;             // images those arrays hold then (0: no index, stream s reads image s)
  double* d_traj = nullptr;     // rvo_streams_record: [n_streams][traj_cap][RVO_TRAJ_DOUBLES]
  int traj_cap = 0, traj_steps = 0, traj_streams = 0;
  double* d_pinit = nullptr;
  double* d_info = nullptr;  // [max_pairs][36] edge information matrices
  int lds_rot = 0, lds_match = 0, lds_clique = 0, lds_gn = 0;
  // pinned ring for small host->device argument arrays (slot lists, thresholds, initial poses)
  static const int kRing = 32;
  uint8_t* h_ring = nullptr;
  size_t ring_bytes = 0;
  hipEvent_t ring_ev[kRing] = {};
  hipEvent_t ev_pyr[2] = {}, ev_rot[2] = {};  // rvo_match_frames_dev: initial rotation on a side stream
  int ring_pos = 0;
  // auxiliary streams: chunks of a pair batch are pipelined against each other (fork/join around the main stream)
  static const int kAux = 3;
  hipStream_t aux[kAux] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kAux] = {};
  bool rot_side = true;  // k_init_rotation on a side stream (experiment builds, env RVO_ROT_SIDE=0: on the batch's own stream)
  hipEvent_t ev_mid = nullptr;  // recorded after k_fast_grid of a batch (rvo_wait_peer)
  int pair_chunks = 2;
  // Staged pipeline of rvo_match_frames_dev: the stages of successive batches run on four streams
  // -- frames (the context's stream), initial rotation, pair stages, tail (clique + Gauss-Newton) -- so that the
  // latency-bound tail of batch k runs beside the frame stages of batch k+1 and the pair stages of ... without ever
  // holding up the next submit.  Pair work areas exist kPairSets times (the R, P and T stages work on three different
  // batches at a time, one more set is slack); `pb` is the set of the latest batch.
  // rvo_stream_step_dev: the streams of a step are cut into groups that run as chains of their own (main stream, aux[]);
  // a group's FAST stage starts when the other group's is done, so the pair stages of one group run beside the frame
  // stages of the other (what two contexts with rvo_wait_peer did in round 1)
  int stream_groups = 4;         // groups of streams a step is cut into, each a chain on a stream of its own (env RVO_STREAM_GROUPS = 1..kMaxGroups; 0 and 1: one chain per step)
  static constexpr int kMaxGroups = 4;
  bool clique_split = true;      // env RVO_CLIQUE_SPLIT=0: k_clique does its own greedy walk for every pair
  int stream_group_min = 32;     // fewer streams than this run as one chain (env RVO_STREAM_GROUP_MIN; tests use 2); 32 / 64 streams gain 1 / 3 % from the groups
  hipEvent_t ev_gfast[kMaxGroups] = {}, ev_gdone[kMaxGroups - 1] = {};
  bool ev_gfast_rec[kMaxGroups] = {};
  int group_pending = 0;         // groups of the latest step (beyond the first) not yet joined into the main stream
  // Work that entry points outside the two pipelines put on the main stream (conversions, prepared frames, pair batches,
  // a peer wait) must precede what the pipelines run on the OTHER streams: main_epoch counts such calls, and a pipeline
  // whose last fork saw an older epoch orders its side streams after the main stream again (one event) before it goes on.
  long long main_epoch = 0, groups_epoch = -1, staged_epoch = -1;
  bool staged = true;            // env RVO_STAGED=0 / rvo_set_pipeline(ctx, 0): the chunked single-chain form
  static const int kPairSets = 4;
  int pair_sets = kPairSets;     // experiment builds: env RVO_PAIR_SETS (2..kPairSets)
  PairBufs pbs[kPairSets];
  long long batch_id = 0;        // staged batches submitted so far
  long long staged_valid_from = 0;  // staged batches before this one had their work areas reused by a non-staged call
  int staged_pending = 0;        // bit s: set s has staged work the main stream has not been ordered after yet
  hipEvent_t ev_f[kPairSets] = {}, ev_p[kPairSets] = {}, ev_t[kPairSets] = {}, ev_pyr2[kPairSets] = {}, ev_rot2[kPairSets] = {};
  bool ev_p_rec[kPairSets] = {}, ev_t_rec[kPairSets] = {};
  static const int kSlotRing = 8;  // tail completion events of the last eight staged batches (frame slot reuse)
  hipEvent_t ev_pdone[kSlotRing] = {};
  int32_t* d_track_ctr = nullptr;  // [2 + max_pairs] _num_tracks: [0], [1] alternate for batches that are ONE estimator (mode 1 of
                                   // k_track_assign), [2 + i] belongs to stream i (mode 0)
  long long track_batches = 0;     // mode-1 batches so far: the current counter is d_track_ctr[track_batches & 1]
  std::vector<long long> slot_reader;  // [max_frames] last staged batch whose pair stages read the slot (-1: none)
  // profiling
  bool prof = false;
  std::string prof_only;  // rvo_profile_select: bracket only launches of this kernel ("" = all)
  std::vector<std::string> prof_names;
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool;
  std::map<int, std::pair<float, int>> prof_acc;
};

template <class T>
static int dalloc(rvo_ctx* c, T** p, size_t n) {
  void* v = nullptr;
  hipError_t e = hipMalloc(&v, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) { g_err = std::string("hipMalloc: ") + hipGetErrorString(e); return RVO_E_NOMEM; }
  c->allocs.push_back(v);
  *p = (T*)v;
  return 0;
}

// Stream-ordered H2D copy of a small argument array through pinned memory; never blocks on prior GPU work
// (a pageable hipMemcpyAsync may) and never lets the host buffer be reused while a copy is pending.
static int stage_args(rvo_ctx* c, void* dst_dev, const void* src, size_t bytes, hipStream_t st = nullptr) {
  if (bytes == 0) return 0;
  if (!st) st = c->stream;
  if (bytes > c->ring_bytes) { g_err = "stage_args: argument array larger than the staging region"; return RVO_E_INVALID; }
  const int r = c->ring_pos;
  c->ring_pos = (r + 1) % rvo_ctx::kRing;
  hipError_t e = hipEventSynchronize(c->ring_ev[r]);
  if (e != hipSuccess) { g_err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); return RVO_E_HIP; }
  uint8_t* h = c->h_ring + (size_t)r * c->ring_bytes;
  memcpy(h, src, bytes);
  e = hipMemcpyAsync(dst_dev, h, bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipEventRecord(c->ring_ev[r], st);
  if (e != hipSuccess) { g_err = std::string("stage_args: ") + hipGetErrorString(e); return RVO_E_HIP; }
  return 0;
}

static int prof_id(rvo_ctx* c, const char* name) {
  for (size_t i = 0; i < c->prof_names.size(); i++) if (c->prof_names[i] == name) return (int)i;
  c->prof_names.push_back(name);
  return (int)c->prof_names.size() - 1;
}
static hipEvent_t get_event(rvo_ctx* c) {
  if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
static void prof_drain(rvo_ctx* c) {
  for (auto& r : c->prof_recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    auto& acc = c->prof_acc[r.name];
    acc.first += ms;
    acc.second += 1;
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->prof_recs.clear();
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of a context: contexts of different sizes
// in one process must never lower it, so the largest request per kernel (and device) wins.
static hipError_t raise_dyn_lds(const void* func, int device, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> cur;
  std::lock_guard<std::mutex> lk(mu);
  int& have = cur[{func, device}];
  if (bytes <= have) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

// Resolves (once per device, with that device current) where `scratch` lives on `device`.
static int scratch_for_device(int device) {
  if (device < 0 || device >= kMaxDevices) return RVO_OK;  // such a device simply has no allocation-free entry point
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[device]) return RVO_OK;
  double* p = nullptr;
  HIPCHK(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
  g_scratch_ptr[device] = p;
  return RVO_OK;
}

#ifdef RVO_SKIP_EXPERIMENT  // one-off builds: RVO_SKIP=<kernel names> drops those launches, results go stale (what do they cost the pipeline?)
// whole names of a comma-separated list ("refine" does not name refine_finish); after every slot set has been filled once
#define RVO_SKIPPED(NAME) ((c)->batch_id > 12 && getenv("RVO_SKIP") && (std::string(",") + getenv("RVO_SKIP") + ",").find(std::string(",") + NAME + ",") != std::string::npos)
#else
#define RVO_SKIPPED(NAME) false
#endif
#define LAUNCH_ON(c, STREAM, NAME, KERNEL, GRID, BLOCK, SHMEM, ...)                       \
  do {                                                                                    \
    if (RVO_SKIPPED(NAME)) break;                                                         \
    ProfRec pr_;                                                                          \
    const bool prof_ = (c)->prof && ((c)->prof_only.empty() || (c)->prof_only == NAME);   \
    if (prof_) {                                                                          \
      pr_.name = prof_id((c), NAME);                                                      \
      pr_.a = get_event(c);                                                               \
      pr_.b = get_event(c);                                                               \
      (void)hipEventRecord(pr_.a, (STREAM));                                              \
    }                                                                                     \
    hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, (STREAM), __VA_ARGS__);                \
    if (prof_) {                                                                          \
      (void)hipEventRecord(pr_.b, (STREAM));                                              \
      (c)->prof_recs.push_back(pr_);                                                      \
    }                                                                                     \
  } while (0)
#define LAUNCH(c, NAME, KERNEL, GRID, BLOCK, SHMEM, ...) LAUNCH_ON(c, (c)->stream, NAME, KERNEL, GRID, BLOCK, SHMEM, __VA_ARGS__)

// ---------------------------------------------------------------- defaults (settings.hpp:11-52)
void rvo_default_vo_params(rvo_params* p) {
  *p = rvo_params{9, 3, 20, 5, 70, 0.005, 250, 80, 80, 25, 1.5, 0.1, 10, 10.0, 25.0, 250, 100};
}
void rvo_default_fm_params(rvo_params* p) {
  rvo_default_vo_params(p);
  p->reference_frame_change_threshold = 150;
}

// Undistortion look-up tables (what Camera::MakeXYMap, camera.cpp:16-67, leaves in Xmap_ / Ymap_): for every input pixel
// the position it has in the rectified image.  One thread per pixel, once per context.  The lens model is inverted by
// five rounds of fixed-point iteration from the normalised pixel; every round evaluates the radial polynomial in Horner
// form and the two tangential terms on the current estimate.  fp64 throughout, no contraction, the reference's operation
// order (the tables must come out bit-equal: tests/test_gpu_frame.py), then one rounding to float.
struct LensModel { double fx, fy, cx, cy, k1, k2, k3, p1, p2; };
__global__ __launch_bounds__(256) void k_undistort_lut(LensModel lens, int width, int height, float* __restrict__ lut_u,
                                                       float* __restrict__ lut_v) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
  if (col >= width || row >= height) return;
  const double nu = (col - lens.cx) / lens.fx, nv = (row - lens.cy) / lens.fy;  // distorted, normalised
  double u = nu, v = nv;                                                        // running undistorted estimate
  for (int round = 0; round < 5; round++) {
    const double rr = u * u + v * v;
    const double radial_inv = 1. / (1. + ((lens.k3 * rr + lens.k2) * rr + lens.k1) * rr);
    const double tang_u = 2 * lens.p1 * u * v + lens.p2 * (rr + 2 * u * u);
    const double tang_v = lens.p1 * (rr + 2 * v * v) + 2 * lens.p2 * u * v;
    u = (nu - tang_u) * radial_inv;
    v = (nv - tang_v) * radial_inv;
  }
  // the rectified camera has the input camera's focal lengths and centre (camera.cpp:7-12)
  const size_t at = (size_t)row * width + col;
  lut_u[at] = (float)(u * lens.fx + lens.cx);
  lut_v[at] = (float)(v * lens.fy + lens.cy);
}

// Everything rvo_create does once the context object exists; on any failure the caller destroys the half-built context
// (rvo_destroy copes with members that were never created), so no path leaks streams, events, pinned or device memory.
static int ctx_init(rvo_ctx* c, const rvo_intrinsics* intr, const rvo_params* prm, int device, int max_frames, int max_pairs);

int rvo_create(const rvo_intrinsics* intr, const rvo_params* prm, int device, int max_frames, int max_pairs, rvo_ctx** out) {
  if (!intr || !prm || !out || max_frames < 1 || max_pairs < 0) { g_err = "rvo_create: bad argument"; return RVO_E_INVALID; }
  *out = nullptr;
  if (prm->feature_window_size != RVO_WIN) { g_err = "feature-window-size must be 9 in this build"; return RVO_E_UNSUPPORTED; }
  if (prm->num_pyramid_levels < 1 || prm->num_pyramid_levels > RVO_MAX_LEVELS) { g_err = "num-pyramid-levels out of range"; return RVO_E_UNSUPPORTED; }
  if (prm->grid_width < 8 || prm->grid_height < 8 || prm->max_keypoints_per_grid < 1) { g_err = "grid options out of range"; return RVO_E_UNSUPPORTED; }
  if (prm->fast_threshold_min < 0 || prm->fast_threshold_min > prm->fast_threshold_max || prm->fast_threshold_max > 255 ||
      prm->fast_init_threshold < 0 || prm->fast_init_threshold > 255) {
    g_err = "FAST thresholds must satisfy 0 <= min <= max <= 255 and 0 <= init <= 255"; return RVO_E_INVALID; }
  if (prm->target_pixels_per_feature < 1) { g_err = "target-pixels-per-feature must be positive"; return RVO_E_INVALID; }  // vo.cpp:42-43 divides by it
  if (intr->depth_width < 1 || intr->depth_height < 1) { g_err = "depth image size must be positive"; return RVO_E_INVALID; }
  if (!(std::isfinite(intr->fx) && std::isfinite(intr->fy) && intr->fx != 0.0 && intr->fy != 0.0 && std::isfinite(intr->cx) && std::isfinite(intr->cy))) {
    g_err = "fx, fy must be finite and non-zero, cx, cy finite"; return RVO_E_INVALID; }
  if (intr->width > 2047 || intr->height > 2047 || intr->width < 32 || intr->height < 32) { g_err = "image size out of range"; return RVO_E_UNSUPPORTED; }
  if ((intr->width >> (prm->num_pyramid_levels - 1)) < 24 || (intr->height >> (prm->num_pyramid_levels - 1)) < 24) {
    g_err = "coarsest pyramid level smaller than 24 px"; return RVO_E_UNSUPPORTED;
  }
  HIPCHK(hipSetDevice(device));
  { const int rc0 = scratch_for_device(device); if (rc0) return rc0; }
  rvo_ctx* c = new rvo_ctx();
  const int rc = ctx_init(c, intr, prm, device, max_frames, max_pairs);
  if (rc) { rvo_destroy(c); return rc; }
  *out = c;
  return RVO_OK;
}

static int ctx_init(rvo_ctx* c, const rvo_intrinsics* intr, const rvo_params* prm, int device, int max_frames, int max_pairs) {
  c->device = device;
  c->intr = *intr;
  c->prm = *prm;
  c->max_frames = max_frames;
  c->max_pairs = max_pairs;
  FrameGeom& g = c->g;
  memset(&g, 0, sizeof(g));
  g.W = intr->width; g.H = intr->height; g.Wd = intr->depth_width; g.Hd = intr->depth_height;
  g.levels = prm->num_pyramid_levels;
  g.gw = prm->grid_width; g.gh = prm->grid_height; g.max_per = prm->max_keypoints_per_grid;
  g.xs = (float)g.Wd / (float)g.W;
  g.ys = (float)g.Hd / (float)g.H;
  g.fx_inv = 1.0 / intr->fx;
  g.fy_inv = 1.0 / intr->fy;
  g.neg_cx_div_fx = -intr->cx * g.fx_inv;
  g.neg_cy_div_fy = -intr->cy * g.fy_inv;
  int img_off = 0, boff = 0, koff = 0;
  for (int l = 0; l < g.levels; l++) {
    LevelGeom& L = g.lv[l];
    L.w = g.W >> l; L.h = g.H >> l;
    L.stride = round_up(L.w, 16);
    L.img_off = img_off;
    img_off += round_up(L.stride * L.h, 256);
    L.grows = (int)std::ceil((float)L.h / g.gh);
    L.gcols = (int)std::ceil((float)L.w / g.gw);
    L.nbuckets = L.grows * L.gcols;
    L.bucket_off = boff; boff += L.nbuckets;
    L.kp_cap = L.nbuckets * g.max_per;
    L.kp_off = koff; koff += L.kp_cap;
  }
  g.img_bytes = img_off; g.total_buckets = boff; g.kp_total = koff;
  if (koff > 65535) { g_err = "more than 65535 keypoint slots per frame"; return RVO_E_UNSUPPORTED; }
  c->cam = CamParams{intr->fx, intr->fy, intr->cx, intr->cy, intr->fx};
  c->ep = EstParams{prm->inlier_max_reprojection_error, prm->clique_inlier_threshold, prm->max_mean_reprojection_error,
                    prm->feature_search_window, prm->min_features_for_estimate, 0.0};
  {
    // largest |coordinate| a keypoint can have: z <= 65.535 m (uint16 mm), x = z * (u/fx - cx/fx) with u in [0, W)
    const double rx = std::max(std::fabs(g.neg_cx_div_fx), std::fabs((g.W - 1) * g.fx_inv + g.neg_cx_div_fx));
    const double ry = std::max(std::fabs(g.neg_cy_div_fy), std::fabs((g.H - 1) * g.fy_inv + g.neg_cy_div_fy));
    const double R = 65.535 * std::max(1.0, std::max(rx, ry));
    // fp32 error of (sqrt a - sqrt b) is below 1.7e-6 * R (k_consistency); 4e-6 * R leaves a 2.3x margin
    c->ep.cons_band = std::max(1e-4, 4e-6 * R);
  }
  // FAST workgroup LDS carve-up
  FastLds& fl = c->fl;
  fl.tile_pitch = round_up(g.gw + 8 + (g.gw % 4 ? 3 : 0), 4);  // compact rows; +3 when cells do not start on multiples of 4
  fl.score_pitch = round_up(g.gw + 2, 4);
  fl.off_score = round_up(fl.tile_pitch * (g.gh + 8), 16);
  fl.off_list = fl.off_score + round_up(fl.score_pitch * (g.gh + 2), 16);
  fl.list_cap = ((g.gw + 1) / 2) * ((g.gh + 1) / 2) + 1;
  // the same region first holds the phase-A candidate queue (u16 per entry): room for half of the pixels of the
  // (gw+2)x(gh+2) score region (k_fast_grid falls back to a dense pass beyond that), at least the list's bytes
  fl.cand_cap = std::max(fl.list_cap * 2, ((g.gw + 2) * (g.gh + 2) + 1) / 2);
  fl.off_misc = fl.off_list + round_up(fl.cand_cap * 2, 16);
  fl.total = fl.off_misc + 64 + round_up((g.gw * g.gh + 31) / 32 * 4, 16);
#ifdef RVO_EXPERIMENT_KNOBS  // one-off builds only (tools/exp_variant.sh)
  if (const char* e = getenv("RVO_FAST_LDS_MIN")) {  // fewer k_fast_grid workgroups per CU
    fprintf(stderr, "k_fast_grid LDS %d -> at least %d bytes\n", fl.total, atoi(e));
    fl.total = std::max(fl.total, round_up(atoi(e), 16));
  }
#endif
  if (fl.total > 150 * 1024 || g.gw + 2 > 255 || g.gh + 2 > 255 || (g.gw + 2) * (g.gh + 2) > 64 * 256 ||
      ((g.gw + 7) / 4 + 1) * (g.gh + 2) > 16 * 256) {  // phase A of k_fast_grid: <= 16 four-pixel groups per thread
    g_err = "grid cell too large for LDS"; return RVO_E_UNSUPPORTED; }
  HIPCHK(raise_dyn_lds(reinterpret_cast<const void*>(k_fast_grid<0, 0>), device, fl.total));
  HIPCHK(raise_dyn_lds(reinterpret_cast<const void*>(k_fast_grid<88, 84>), device, fl.total));

  int rc = 0;
  c->ring_bytes = std::max<size_t>(4096, (size_t)std::max(max_frames, max_pairs) * 12 * 8);
  HIPCHK(hipHostMalloc((void**)&c->h_ring, c->ring_bytes * rvo_ctx::kRing, hipHostMallocDefault));
  for (int i = 0; i < rvo_ctx::kRing; i++) HIPCHK(hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming));
  // HIP multiplexes the streams of one priority onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and streams that
  // share a queue run in submission order (measured: tools/native/queue_probe.hip); every priority has a pool of its
  // own.  All four streams of a context are created at the greatest priority: the pool they share with nothing else in
  // a typical process (torch's streams are at the normal priority), so the staged pipeline's streams do not alias.
  // They must all have the SAME priority: with the frame stages at normal and the tail at high priority, a context
  // stopped making progress as soon as two more processes used the GPU (its high-priority queues, blocked on events
  // of the normal ones, kept those from being scheduled; tools/exp_three_procs.py).
  const char* prio = "hhhh";
#ifdef RVO_EXPERIMENT_KNOBS  // one-off builds only: mixed priorities can hang the context (see above), so the product has no such knob
  // RVO_STREAM_PRIORITY = one letter per stream (own, aux0, aux1, aux2; h greatest, n normal, l least)
  if (const char* prio_env = getenv("RVO_STREAM_PRIORITY")) if (strlen(prio_env) == 1 + rvo_ctx::kAux) prio = prio_env;
#endif
  int least = 0, greatest = 0;
  HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  auto prio_of = [&](char ch) { return ch == 'h' ? greatest : ch == 'l' ? least : 0; };
  HIPCHK(hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, prio_of(prio[0])));
  c->stream = c->own_stream;
  {
    for (int i = 0; i < rvo_ctx::kAux; i++) {
      HIPCHK(hipStreamCreateWithPriority(&c->aux[i], hipStreamNonBlocking, prio_of(prio[1 + i])));
      HIPCHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
  }
  for (int i = 0; i < rvo_ctx::kPairSets; i++) {
    HIPCHK(hipEventCreateWithFlags(&c->ev_f[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_p[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_t[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_pyr2[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_rot2[i], hipEventDisableTiming));
  }
  for (int i = 0; i < rvo_ctx::kSlotRing; i++) HIPCHK(hipEventCreateWithFlags(&c->ev_pdone[i], hipEventDisableTiming));
  if (const char* e = getenv("RVO_STAGED")) c->staged = atoi(e) != 0;
  if (const char* e = getenv("RVO_STREAM_GROUPS")) c->stream_groups = std::min(std::max(atoi(e), 1), (int)rvo_ctx::kMaxGroups);
  if (const char* e = getenv("RVO_CLIQUE_SPLIT")) c->clique_split = atoi(e) != 0;
  if (const char* e = getenv("RVO_STREAM_GROUP_MIN")) c->stream_group_min = std::max(2, atoi(e));
  for (int i = 0; i < rvo_ctx::kMaxGroups; i++) HIPCHK(hipEventCreateWithFlags(&c->ev_gfast[i], hipEventDisableTiming));
  for (int i = 0; i < rvo_ctx::kMaxGroups - 1; i++) HIPCHK(hipEventCreateWithFlags(&c->ev_gdone[i], hipEventDisableTiming));
  c->slot_reader.assign((size_t)max_frames, -1);
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_mid, hipEventDisableTiming));
#ifdef RVO_EXPERIMENT_KNOBS
  if (const char* e = getenv("RVO_ROT_SIDE")) c->rot_side = atoi(e) != 0;
#endif
  for (int i = 0; i < 2; i++) {
    HIPCHK(hipEventCreateWithFlags(&c->ev_pyr[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_rot[i], hipEventDisableTiming));
  }
  if (const char* e = getenv("RVO_PAIR_CHUNKS")) c->pair_chunks = std::max(1, std::min(rvo_ctx::kAux + 1, atoi(e)));
  FrameBufs& fb = c->fb;
  const size_t F = max_frames;
  rc |= dalloc(c, &fb.img, F * g.img_bytes);
  rc |= dalloc(c, &fb.depth, F * g.Wd * g.Hd);
  rc |= dalloc(c, &fb.depth_tab, F);
  rc |= dalloc(c, &fb.l0_tab, F);
  rc |= dalloc(c, &fb.bk_cnt, F * g.total_buckets);
  rc |= dalloc(c, &fb.bk_det, F * g.total_buckets);
  rc |= dalloc(c, &fb.bk_kp, F * g.total_buckets * g.max_per);
  rc |= dalloc(c, &fb.kp_uvs, F * g.kp_total);
  rc |= dalloc(c, &fb.kp_rect, F * g.kp_total);
  rc |= dalloc(c, &fb.kp_track, F * g.kp_total);
  rc |= dalloc(c, &fb.kp_xyz, F * g.kp_total * 3);
  rc |= dalloc(c, &fb.bk_desc, F * g.kp_total * RVO_DESC_LEN + 64);
  rc |= dalloc(c, &fb.kp_src, F * g.kp_total);
  rc |= dalloc(c, &fb.kp_cnt, F * RVO_MAX_LEVELS);
  rc |= dalloc(c, &fb.det_cnt, F * RVO_MAX_LEVELS);
  rc |= dalloc(c, &fb.grid_cnt, F * RVO_MAX_LEVELS);
  float *dx = nullptr, *dy = nullptr;
  rc |= dalloc(c, &dx, (size_t)g.W * g.H + g.W + 2);  // +1 row: BilinearLookup reads ra+W+1 (camera.hpp:103-109)
  rc |= dalloc(c, &dy, (size_t)g.W * g.H + g.W + 2);
  rc |= dalloc(c, &c->d_fargs, 2 * F);
  if (rc) return RVO_E_NOMEM;
  HIPCHK(hipMemset(fb.img, 0, F * g.img_bytes));  // PyramidLevel ctor memset, pyramid.cpp:121
  {
    std::vector<const uint16_t*> tab(F);
    for (size_t i = 0; i < F; i++) tab[i] = fb.depth + i * (size_t)g.Wd * g.Hd;
    HIPCHK(hipMemcpy(fb.depth_tab, tab.data(), F * sizeof(tab[0]), hipMemcpyHostToDevice));
    std::vector<const uint8_t*> itab(F);
    for (size_t i = 0; i < F; i++) itab[i] = fb.img + i * (size_t)g.img_bytes + g.lv[0].img_off;
    HIPCHK(hipMemcpy(fb.l0_tab, itab.data(), F * sizeof(itab[0]), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemset(fb.kp_track, 0xFF, F * g.kp_total * 4));  // -1: slots nobody prepared yet
  HIPCHK(hipMemset(fb.kp_cnt, 0, F * RVO_MAX_LEVELS * 4));
  HIPCHK(hipMemset(fb.det_cnt, 0, F * RVO_MAX_LEVELS * 4));
  HIPCHK(hipMemset(fb.grid_cnt, 0, F * RVO_MAX_LEVELS * 4));
  HIPCHK(hipMemsetAsync(dx, 0, ((size_t)g.W * g.H + g.W + 2) * 4, c->stream));  // same stream as the kernel below
  HIPCHK(hipMemsetAsync(dy, 0, ((size_t)g.W * g.H + g.W + 2) * 4, c->stream));
  {
    const LensModel lens{intr->fx, intr->fy, intr->cx, intr->cy, intr->k1, intr->k2, intr->k3, intr->p1, intr->p2};
    hipLaunchKernelGGL(k_undistort_lut, dim3((g.W + 255) / 256, g.H), dim3(256), 0, c->stream, lens, g.W, g.H, dx, dy);
    HIPCHK(hipGetLastError());
    c->xmap.resize((size_t)g.W * g.H);  // host copies for rvo_get_lut
    c->ymap.resize((size_t)g.W * g.H);
    HIPCHK(hipMemcpyAsync(c->xmap.data(), dx, c->xmap.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->ymap.data(), dy, c->ymap.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  fb.xmap = dx; fb.ymap = dy;

  rc = pair_alloc(c);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  return RVO_OK;
}

void rvo_destroy(rvo_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  prof_drain(c);
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->h_ring) (void)hipHostFree(c->h_ring);
  for (int i = 0; i < rvo_ctx::kRing; i++) if (c->ring_ev[i]) (void)hipEventDestroy(c->ring_ev[i]);
  if (c->d_gray_stage) (void)hipFree(c->d_gray_stage);
  if (c->d_depth_stage) (void)hipFree(c->d_depth_stage);
  for (int i = 0; i < 2; i++) {
    if (c->d_gray_db[i]) (void)hipFree(c->d_gray_db[i]);
    if (c->d_depth_db[i]) (void)hipFree(c->d_depth_db[i]);
    if (c->ev_copy[i]) (void)hipEventDestroy(c->ev_copy[i]);
    if (c->ev_used[i]) (void)hipEventDestroy(c->ev_used[i]);
  }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  for (int i = 0; i < rvo_ctx::kAux; i++) {
    if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
    if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
  }
  for (int i = 0; i < rvo_ctx::kPairSets; i++) {
    hipEvent_t* evs[5] = {&c->ev_f[i], &c->ev_p[i], &c->ev_t[i], &c->ev_pyr2[i], &c->ev_rot2[i]};
    for (auto e : evs) if (*e) (void)hipEventDestroy(*e);
  }
  for (int i = 0; i < rvo_ctx::kSlotRing; i++) if (c->ev_pdone[i]) (void)hipEventDestroy(c->ev_pdone[i]);
  for (int i = 0; i < rvo_ctx::kMaxGroups; i++) if (c->ev_gfast[i]) (void)hipEventDestroy(c->ev_gfast[i]);
  for (int i = 0; i < rvo_ctx::kMaxGroups - 1; i++) if (c->ev_gdone[i]) (void)hipEventDestroy(c->ev_gdone[i]);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
  for (int i = 0; i < 2; i++) {
    if (c->ev_pyr[i]) (void)hipEventDestroy(c->ev_pyr[i]);
    if (c->ev_rot[i]) (void)hipEventDestroy(c->ev_rot[i]);
  }
  (void)hipGetLastError();  // leave no stale error behind for the next HIP user in the process
  delete c;
}

// Orders the context's main stream after every staged batch still in flight (their pair and tail stages run on other
// streams): called by every entry point that touches pair work areas or frame slots outside the staged pipeline.
static int join_staged(rvo_ctx* c, bool keep_groups = false) {
  if (!keep_groups) c->main_epoch++;  // every caller but the grouped stream step goes on to use the main stream
  if (c->group_pending && !keep_groups) {  // the other groups of the latest stream step (rvo_stream_step_dev)
    for (int i = 0; i < c->group_pending; i++) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_gdone[i], 0));
    c->group_pending = 0;
  }
  for (int s = 0; s < rvo_ctx::kPairSets; s++) {
    if (!(c->staged_pending & (1 << s))) continue;
    if (c->ev_p_rec[s]) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_p[s], 0));
    if (c->ev_t_rec[s]) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_t[s], 0));
  }
  c->staged_pending = 0;
  return RVO_OK;
}
// waits for everything the context has enqueued on any of its streams
#define SYNC_MAIN(c)                                                  \
  do {                                                                \
    const int rcj_ = join_staged(c);                                  \
    if (rcj_) return rcj_;                                            \
    HIPCHK(hipStreamSynchronize((c)->stream));                        \
  } while (0)
// Frame slots about to be rewritten on stream st: wait for the tail of the staged batches that still use them.
static int wait_slots_free(rvo_ctx* c, hipStream_t st, int n, const int32_t* slots) {
  long long need = -1;
  for (int i = 0; i < n; i++) need = std::max(need, c->slot_reader[slots[i]]);
  if (need < 0) return RVO_OK;
  // events exist for the last kSlotRing batches; an older batch's tail precedes theirs on the tail stream
  const long long b = std::max(need, c->batch_id - rvo_ctx::kSlotRing);
  if (b >= 0 && b < c->batch_id) HIPCHK(hipStreamWaitEvent(st, c->ev_pdone[b % rvo_ctx::kSlotRing], 0));
  return RVO_OK;
}

int rvo_set_stream(rvo_ctx* c, void* s) {
  if (!c) return RVO_E_INVALID;
  { const int rc0 = join_staged(c); if (rc0) return rc0; }
  if (s) c->staged = false;  // work on a caller's stream runs as one chain per call (see rvo_set_pipeline)
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return RVO_OK;
}

int rvo_set_pair_chunks(rvo_ctx* c, int chunks) {
  if (!c || chunks < 1 || chunks > rvo_ctx::kAux + 1) { g_err = "rvo_set_pair_chunks: 1..4"; return RVO_E_INVALID; }
  c->pair_chunks = chunks;
  return RVO_OK;
}

int rvo_sync(rvo_ctx* c) {
  if (!c) return RVO_E_INVALID;
  SYNC_MAIN(c);
  return RVO_OK;
}

int rvo_set_pipeline(rvo_ctx* c, int staged) {
  if (!c) return RVO_E_INVALID;
  if (staged && c->stream != c->own_stream) {
    // the stages wait on each other across streams: a caller's stream of another priority would bring back the
    // priority inversion described in ctx_init
    g_err = "rvo_set_pipeline: the staged pipeline runs on the context's own streams (rvo_set_stream(ctx, NULL) first)";
    return RVO_E_UNSUPPORTED;
  }
  { const int rc0 = join_staged(c); if (rc0) return rc0; }
  c->staged = staged != 0;
  return RVO_OK;
}

static int ensure_stage(rvo_ctx* c, int n) {
  const FrameGeom& g = c->g;
  if (c->stage_frames >= (size_t)n) return RVO_OK;
  SYNC_MAIN(c);
  if (c->d_gray_stage) (void)hipFree(c->d_gray_stage);
  if (c->d_depth_stage) (void)hipFree(c->d_depth_stage);
  c->d_gray_stage = nullptr; c->d_depth_stage = nullptr; c->stage_frames = 0;
  HIPCHK(hipMalloc((void**)&c->d_gray_stage, (size_t)n * g.W * g.H));
  HIPCHK(hipMalloc((void**)&c->d_depth_stage, (size_t)n * g.Wd * g.Hd * 2));
  c->stage_frames = n;
  return RVO_OK;
}

// copy = 1: level-0 image and raw depth are copied into the slots.  copy = 0 (rvo_match_frames_dev): depth is always
// read in place; the gray image too when its rows already have the level-0 stride and 16-byte alignment, and then
// the launch only records the two pointers per slot.
static int launch_ingest_on(rvo_ctx* c, hipStream_t st, int n, const uint8_t* gray, const uint16_t* depth, const int32_t* d_slots,
                            int copy, const int32_t* d_src_index = nullptr, int copy_gray_always = 0) {
  const FrameGeom& g = c->g;
  const bool vec = (g.W % 16 == 0) && (((size_t)g.Wd * g.Hd) % 8 == 0) && (((uintptr_t)gray & 15) == 0) && (((uintptr_t)depth & 15) == 0);
  const int copy_gray = (copy || copy_gray_always || g.lv[0].stride != g.W || ((uintptr_t)gray & 15) != 0) ? 1 : 0;
  const dim3 grid(copy_gray ? 16 : 1, n);
  if (vec) LAUNCH_ON(c, st, "ingest", k_ingest<true>, grid, dim3(256), 0, c->fb, g, gray, depth, d_slots, copy, copy_gray, d_src_index);
  else LAUNCH_ON(c, st, "ingest", k_ingest<false>, grid, dim3(256), 0, c->fb, g, gray, depth, d_slots, copy, copy_gray, d_src_index);
  return RVO_OK;
}
static int launch_ingest(rvo_ctx* c, int n, const uint8_t* gray, const uint16_t* depth, const int32_t* d_slots) {
  return launch_ingest_on(c, c->stream, n, gray, depth, d_slots, 1);
}

// ---------------------------------------------------------------- PrepareFrame
// enqueue the frame kernels on stream `st` for n frames whose slots/thresholds are at d_slots/d_thr (device)
// gray / depth non-null: the images are read in place (see can_read_in_place) and the first launch records them as the
// slots' level-0 / depth images.
static int enqueue_pyramid_on(rvo_ctx* c, hipStream_t st, int n, const int32_t* d_slots, const uint8_t* gray = nullptr,
                              const uint16_t* depth = nullptr, const int32_t* d_src_index = nullptr, int copy_l0 = 0) {
  const FrameGeom& g = c->g;
  for (int l = 1; l < g.levels; l++) {
    const int dw = g.lv[l - 1].w / 2, dh = g.lv[l - 1].h / 2;
    dim3 grid((dw + PYR_TW - 1) / PYR_TW, (dh + PYR_TH - 1) / PYR_TH, n);
    LAUNCH_ON(c, st, "pyr_down", k_pyr_down, grid, dim3(256), 0, c->fb, g, d_slots, l, l == 1 ? gray : nullptr, l == 1 ? depth : nullptr,
              l == 1 ? d_src_index : nullptr, l == 1 ? copy_l0 : 0);
  }
  return RVO_OK;
}
// rvo_match_frames_dev: level 0 needs no copy when the caller's rows already have the level-0 stride and 16-byte
// alignment; with at least two pyramid levels the first k_pyr_down launch then does k_ingest's bookkeeping as well.
static bool can_read_in_place(const rvo_ctx* c, const uint8_t* gray) {
  return c->g.levels >= 2 && c->g.lv[0].stride == c->g.W && ((uintptr_t)gray & 15) == 0;
}
static int enqueue_features_on(rvo_ctx* c, hipStream_t st, int n, const int32_t* d_slots, const int32_t* d_thr,
                               hipEvent_t wait_before_fast = nullptr, hipEvent_t record_after_fast = nullptr) {
  const FrameGeom& g = c->g;
  if (wait_before_fast) HIPCHK(hipStreamWaitEvent(st, wait_before_fast, 0));
  if (c->fl.tile_pitch == 88 && c->fl.score_pitch == 84)  // the default 80x80 grid: pitches as compile-time constants
    LAUNCH_ON(c, st, "fast_grid", (k_fast_grid<88, 84>), dim3(g.total_buckets, n), dim3(256), c->fl.total, c->fb, g, c->fl, d_slots, d_thr);
  else
    LAUNCH_ON(c, st, "fast_grid", (k_fast_grid<0, 0>), dim3(g.total_buckets, n), dim3(256), c->fl.total, c->fb, g, c->fl, d_slots, d_thr);
  if (st == c->stream) HIPCHK(hipEventRecord(c->ev_mid, st));  // rvo_wait_peer
  if (record_after_fast) HIPCHK(hipEventRecord(record_after_fast, st));
  LAUNCH_ON(c, st, "build_keypoints", k_build_keypoints, dim3(g.levels, n), dim3(256), 0, c->fb, g, d_slots);
  HIPCHK(hipGetLastError());
  return RVO_OK;
}
static int enqueue_prepare_on(rvo_ctx* c, hipStream_t st, int n, const int32_t* d_slots, const int32_t* d_thr) {
  enqueue_pyramid_on(c, st, n, d_slots);
  return enqueue_features_on(c, st, n, d_slots, d_thr);
}
static int enqueue_prepare(rvo_ctx* c, int n, const int32_t* d_slots, const int32_t* d_thr) {
  return enqueue_prepare_on(c, c->stream, n, d_slots, d_thr);
}

// Slots of one batch must be distinct: every slot is written by its own workgroups, two frames in one slot would race.
static int check_slots(rvo_ctx* c, int n, const int32_t* slots, const char* who) {
  c->seen.assign((size_t)c->max_frames, 0);
  for (int i = 0; i < n; i++) {
    if (slots[i] < 0 || slots[i] >= c->max_frames) { g_err = std::string(who) + ": slot out of range"; return RVO_E_INVALID; }
    if (c->seen[slots[i]]) { g_err = std::string(who) + ": the same slot twice in one batch"; return RVO_E_INVALID; }
    c->seen[slots[i]] = 1;
  }
  return RVO_OK;
}

int rvo_prepare_frames_dev(rvo_ctx* c, int n, const uint8_t* gray, const uint16_t* depth, const int32_t* thr, const int32_t* slots) {
  if (!c || !gray || !depth || !thr || !slots || n < 0 || n > c->max_frames) { g_err = "rvo_prepare_frames: bad argument"; return RVO_E_INVALID; }
  if (n == 0) return RVO_OK;
  { const int rc0 = check_slots(c, n, slots, "rvo_prepare_frames"); if (rc0) return rc0; }
  for (int i = 0; i < n; i++)
    if (thr[i] < 0 || thr[i] > 255) { g_err = "rvo_prepare_frames: FAST threshold outside 0..255"; return RVO_E_INVALID; }
  HIPCHK(hipSetDevice(c->device));
  c->main_epoch++;
  { const int rc0 = wait_slots_free(c, c->stream, n, slots); if (rc0) return rc0; }
  std::vector<int32_t> args(2 * (size_t)n);
  memcpy(args.data(), slots, n * 4);
  memcpy(args.data() + n, thr, n * 4);
  { int rc0 = stage_args(c, c->d_fargs, args.data(), args.size() * 4); if (rc0) return rc0; }
  launch_ingest(c, n, gray, depth, c->d_fargs);  // level-0 copy (frame.cpp:47-50) + raw depth into the slots
  return enqueue_prepare(c, n, c->d_fargs, c->d_fargs + n);
}

int rvo_prepare_frames(rvo_ctx* c, int n, const uint8_t* gray, const uint16_t* depth, const int32_t* thr, const int32_t* slots) {
  if (!c || !gray || !depth || n < 0 || n > c->max_frames) { g_err = "rvo_prepare_frames: bad argument"; return RVO_E_INVALID; }
  if (n == 0) return RVO_OK;
  HIPCHK(hipSetDevice(c->device));
  const FrameGeom& g = c->g;
  { int rc0 = ensure_stage(c, n); if (rc0) return rc0; }
  { const int rc0 = join_staged(c); if (rc0) return rc0; }  // the staging buffers may still be read by a stream step's second group
  HIPCHK(hipMemcpyAsync(c->d_gray_stage, gray, (size_t)n * g.W * g.H, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_depth_stage, depth, (size_t)n * g.Wd * g.Hd * 2, hipMemcpyHostToDevice, c->stream));
  int rc = rvo_prepare_frames_dev(c, n, c->d_gray_stage, c->d_depth_stage, thr, slots);
  if (rc) return rc;
  SYNC_MAIN(c);
  return RVO_OK;
}

// ---------------------------------------------------------------- dump accessors
static int check_slot_level(rvo_ctx* c, int slot, int level) {
  if (!c || slot < 0 || slot >= c->max_frames || level < 0 || level >= c->g.levels) { g_err = "slot/level out of range"; return RVO_E_INVALID; }
  return 0;
}

int rvo_get_level_info(rvo_ctx* c, int slot, int level, int32_t* info) {
  if (check_slot_level(c, slot, level) || !info) return RVO_E_INVALID;
  SYNC_MAIN(c);
  const LevelGeom& L = c->g.lv[level];
  int32_t v[3];
  HIPCHK(hipMemcpy(&v[0], c->fb.kp_cnt + slot * RVO_MAX_LEVELS + level, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&v[1], c->fb.det_cnt + slot * RVO_MAX_LEVELS + level, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&v[2], c->fb.grid_cnt + slot * RVO_MAX_LEVELS + level, 4, hipMemcpyDeviceToHost));
  info[0] = L.w; info[1] = L.h; info[2] = L.stride; info[3] = v[0]; info[4] = v[1]; info[5] = v[2];
  return RVO_OK;
}

int rvo_get_level_image(rvo_ctx* c, int slot, int level, uint8_t* out) {
  if (check_slot_level(c, slot, level) || !out) return RVO_E_INVALID;
  SYNC_MAIN(c);
  const LevelGeom& L = c->g.lv[level];
  const uint8_t* src = c->fb.img + (size_t)slot * c->g.img_bytes + L.img_off;
  if (level == 0) HIPCHK(hipMemcpy(&src, c->fb.l0_tab + slot, sizeof(src), hipMemcpyDeviceToHost));  // may live in caller memory
  HIPCHK(hipMemcpy(out, src, (size_t)L.stride * L.h, hipMemcpyDeviceToHost));
  return RVO_OK;
}

int rvo_get_grid_keypoints(rvo_ctx* c, int slot, int level, float* uvs) {
  if (check_slot_level(c, slot, level) || !uvs) return RVO_E_INVALID;
  SYNC_MAIN(c);
  const FrameGeom& g = c->g;
  const LevelGeom& L = g.lv[level];
  std::vector<int32_t> cnt(L.nbuckets);
  std::vector<uint32_t> kp((size_t)L.nbuckets * g.max_per);
  HIPCHK(hipMemcpy(cnt.data(), c->fb.bk_cnt + (size_t)slot * g.total_buckets + L.bucket_off, cnt.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(kp.data(), c->fb.bk_kp + ((size_t)slot * g.total_buckets + L.bucket_off) * g.max_per, kp.size() * 4, hipMemcpyDeviceToHost));
  size_t o = 0;
  for (int b = 0; b < L.nbuckets; b++)
    for (int j = 0; j < cnt[b]; j++) {
      uint32_t k = kp[(size_t)b * g.max_per + j];
      uvs[o++] = (float)kp_x(k); uvs[o++] = (float)kp_y(k); uvs[o++] = (float)kp_s(k);
    }
  return RVO_OK;
}

int rvo_get_keypoints(rvo_ctx* c, int slot, int level, float* uvs, double* rect, double* xyz) {
  if (check_slot_level(c, slot, level)) return RVO_E_INVALID;
  SYNC_MAIN(c);
  const FrameGeom& g = c->g;
  const LevelGeom& L = g.lv[level];
  int32_t n = 0;
  HIPCHK(hipMemcpy(&n, c->fb.kp_cnt + slot * RVO_MAX_LEVELS + level, 4, hipMemcpyDeviceToHost));
  const size_t base = (size_t)slot * g.kp_total + L.kp_off;
  std::vector<uint32_t> k(n);
  std::vector<float2> r(n);
  HIPCHK(hipMemcpy(k.data(), c->fb.kp_uvs + base, (size_t)n * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(r.data(), c->fb.kp_rect + base, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (xyz) HIPCHK(hipMemcpy(xyz, c->fb.kp_xyz + base * 3, (size_t)n * 24, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    if (uvs) { uvs[3 * i] = (float)kp_x(k[i]); uvs[3 * i + 1] = (float)kp_y(k[i]); uvs[3 * i + 2] = (float)kp_s(k[i]); }
    if (rect) { rect[2 * i] = r[i].x; rect[2 * i + 1] = r[i].y; }
  }
  return RVO_OK;
}

int rvo_get_descriptors(rvo_ctx* c, int slot, int level, uint8_t* out) {
  if (check_slot_level(c, slot, level) || !out) return RVO_E_INVALID;
  SYNC_MAIN(c);
  const FrameGeom& g = c->g;
  int32_t n = 0;
  HIPCHK(hipMemcpy(&n, c->fb.kp_cnt + slot * RVO_MAX_LEVELS + level, 4, hipMemcpyDeviceToHost));
  if (n <= 0) return RVO_OK;
  // descriptors lie in bucket order (bk_desc); kp_src names each kept keypoint's row of the level's block
  const size_t kb = (size_t)slot * g.kp_total + g.lv[level].kp_off;
  std::vector<uint16_t> src((size_t)n);
  std::vector<uint8_t> rows((size_t)g.lv[level].kp_cap * RVO_DESC_LEN);
  HIPCHK(hipMemcpy(src.data(), c->fb.kp_src + kb, (size_t)n * 2, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(rows.data(), c->fb.bk_desc + kb * RVO_DESC_LEN, rows.size(), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) memcpy(out + (size_t)i * RVO_DESC_LEN, rows.data() + (size_t)src[i] * RVO_DESC_LEN, RVO_DESC_LEN);
  return RVO_OK;
}

int rvo_get_lut(rvo_ctx* c, float* xmap, float* ymap) {
  if (!c || !xmap || !ymap) return RVO_E_INVALID;
  memcpy(xmap, c->xmap.data(), c->xmap.size() * 4);
  memcpy(ymap, c->ymap.data(), c->ymap.size() * 4);
  return RVO_OK;
}

int rvo_profile_enable(rvo_ctx* c, int on) {
  if (!c) return RVO_E_INVALID;
  SYNC_MAIN(c);
  prof_drain(c);
  c->prof_acc.clear();
  c->prof = on != 0;
  return RVO_OK;
}

int rvo_profile_select(rvo_ctx* c, const char* kernel_name) {
  if (!c) return RVO_E_INVALID;
  c->prof_only = kernel_name ? kernel_name : "";
  return RVO_OK;
}

int rvo_profile_read(rvo_ctx* c, int cap, const char** names, float* total_ms, int32_t* calls) {
  if (!c) return RVO_E_INVALID;
  SYNC_MAIN(c);
  prof_drain(c);
  int n = 0;
  for (auto& kv : c->prof_acc) {
    if (n < cap) { names[n] = c->prof_names[kv.first].c_str(); total_ms[n] = kv.second.first; calls[n] = kv.second.second; }
    n++;
  }
  return n;
}

#include "rvo_pair_api.inc"
#include "rvo_comm.inc"

// ---------------------------------------------------------------- the entry point that allocates nothing
// rvo_get_pair_information with its staging area in the file-scope `scratch` array instead of the context's d_info:
// no hipMalloc / hipFree, no event or vector growth (the launch is not bracketed by profiling events), same kernel.
// The kernel writes all n*36 values it hands back, so nothing in the array has to start at zero.
extern "C" int rvo_get_pair_information_noalloc(rvo_ctx* c, int n, double* info) {
  if (!c || !info || n < 0 || n > c->max_pairs) { g_err = "rvo_get_pair_information_noalloc: bad argument"; return RVO_E_INVALID; }
  if (n > RVO_NOALLOC_MAX_PAIRS) { g_err = "rvo_get_pair_information_noalloc: at most 4096 pairs per call (RVO_NOALLOC_MAX_PAIRS)"; return RVO_E_INVALID; }
  if (n == 0) return RVO_OK;
  double* sp = (c->device >= 0 && c->device < kMaxDevices) ? g_scratch_ptr[c->device] : nullptr;  // resolved by rvo_create on this device
  if (!sp) { g_err = "rvo_get_pair_information_noalloc: no static staging area on this device"; return RVO_E_UNSUPPORTED; }
  HIPCHK(hipSetDevice(c->device));
  { const int rc0 = join_staged(c); if (rc0) return rc0; }
  hipLaunchKernelGGL(k_pair_information, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->pb, n, sp);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(info, sp, (size_t)n * 36 * 8, hipMemcpyDeviceToHost, c->stream));
  SYNC_MAIN(c);
  return RVO_OK;
}
