/* rvo.h -- C ABI of librgbdvo_hip.so: the MI355X (gfx950) implementation of the
 * rgbdvo per-frame odometry front-end of alanzplus/SLAM.
 *
 * The reference has no FFI layer; its boundary is the C++ class API of
 * librgbdvo.  This C ABI is what the C++ facade in
 * include/rgbdvo/ (RGBDVO, FrameMatcher, Camera, Options) sits on, and what
 * tests/bench bind through ctypes.  Each entry point cites the reference
 * interface it replaces (paths relative to /root/reference/).
 *
 * Conventions: every function returns 0 on success or a negative RVO_E_* code;
 * nothing throws or calls exit() across the ABI.  The context owns all device
 * memory.  One context per (device, host thread); a context is not thread-safe.
 * Work is enqueued on the context's HIP stream; functions that return results to
 * host memory synchronise that stream, the *_async / *_dev forms do not.
 * Poses are 3x4 row-major [R|t] doubles, covariances 6x6 row-major doubles
 * (symmetric, so equal to Eigen's column-major layout).
 */
#ifndef RVO_H_
#define RVO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVO_MAX_LEVELS 6

enum {
  RVO_OK = 0,
  RVO_E_INVALID = -1,     /* bad argument (null pointer, slot/pair index out of range, n > capacity) */
  RVO_E_UNSUPPORTED = -2, /* option value this build does not implement (e.g. feature-window-size != 9) */
  RVO_E_HIP = -3,         /* a HIP runtime call failed; rvo_last_error() has the text */
  RVO_E_NOMEM = -4
};

/* rgbdvo/camera.hpp:11-49 IntrinsicParameters (h_fov/v_fov are unused by the path) */
typedef struct rvo_intrinsics {
  int32_t width, height, depth_width, depth_height;
  double fx, fy, cx, cy, k1, k2, k3, p1, p2;
} rvo_intrinsics;

/* rgbdvo/settings.hpp:11-52 option keys, parsed (rgbdvo/helper.cpp:13-59).  warmup_frames is
 * the hard-coded 100 of rgbdvo/vo.cpp:58. */
typedef struct rvo_params {
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
  int32_t warmup_frames;
} rvo_params;

/* rgbdvo/motion_estimate.hpp:13-20 MotionEstimateStatusCode */
enum { RVO_NO_DATA = 0, RVO_SUCCESS = 1, RVO_INSUFFICIENT_INLIERS = 2, RVO_OPTIMIZATION_FAILURE = 3, RVO_REPROJECTION_ERROR = 4 };
/* rgbdvo/feature.hpp:188-201 MatchStatusCode */
enum { RVO_MATCH_NEEDS_DEPTH_REFINEMENT = 0, RVO_MATCH_REFINEMENT_FAILED = 1, RVO_MATCH_OK = 2 };

/* What MotionEstimator exposes after EstimateMotion (rgbdvo/motion_estimate.hpp:40-74). */
typedef struct rvo_pair_result {
  double pose[12];
  double cov[36];
  int32_t status;
  int32_t num_matches;
  int32_t num_inliers;
  int32_t num_reprojection_failures;
  double mean_reprojection_error;
} rvo_pair_result;

/* Per-stream output of one ProcessFrame step (rgbdvo/vo.hpp:22-45 getters). */
typedef struct rvo_stream_result {
  double pose[12];         /* GetPose() */
  double curr_to_ref[12];  /* GetCurrToRefMotionEstimate() */
  double cov[36];          /* GetCurrToRefCov() */
  int32_t rc;              /* 0 = warm-up frame skipped, 1 = processed, -1 = both estimates failed
                              (where the reference calls exit(-1), rgbdvo/vo.cpp:154-158) */
  int32_t fast_threshold;  /* adaptive FAST threshold after this frame (rgbdvo/vo.cpp:92-99) */
  int32_t keyframe_changed;/* IsKeyFrameChanged() */
  int32_t status;          /* last EstimateMotion status */
  int32_t num_matches, num_inliers;
  int32_t used_fallback;   /* 1 if the estimate came from the retry vs the previous frame (vo.cpp:131-152) */
  int32_t pad_;
} rvo_stream_result;

typedef struct rvo_ctx rvo_ctx;

/* settings.hpp:11-30 GetRGBDVODefaultOptions() / :33-52 GetFrameMatcherDefaultOptions() */
void rvo_default_vo_params(rvo_params* out);
void rvo_default_fm_params(rvo_params* out);

/* Replaces the constructors RGBDVO::RGBDVO (rgbdvo/vo.cpp:22-47) and FrameMatcher::FrameMatcher
 * (rgbdvo/fm.cpp:6-20): builds the undistortion LUT (Camera::MakeXYMap, rgbdvo/camera.cpp:16-67) with one device
 * kernel (a host copy is kept for rvo_get_lut) and allocates `max_frames` frame slots and `max_pairs` pair work areas on `device`.
 * RVO_E_UNSUPPORTED (with a message in rvo_last_error) for what this build does not hold: feature-window-size other than
 * 9, more than 6 pyramid levels, images beyond 2047 x 2047 or below 32 x 32, a coarsest level under 24 px, grid cells
 * below 8 px or too large for the LDS tile, more than 65535 keypoint slots per frame, and -- with max_pairs > 0 -- more
 * keypoint slots per frame than the LDS-resident pair kernels sort (about 10 000: 1600 x 1200 at the default grid). */
int rvo_create(const rvo_intrinsics* intr, const rvo_params* params, int device, int max_frames, int max_pairs,
               rvo_ctx** out);
void rvo_destroy(rvo_ctx* ctx);
const char* rvo_last_error(void);

/* Tuning knob without a counterpart in the reference: into how many chunks a batch is cut (each chunk runs on its own
 * HIP stream, see rvo_match_frames_dev).  Default 2 (env RVO_PAIR_CHUNKS overrides at rvo_create).  Use 1 when the
 * caller already keeps two batches in flight on two contexts (double buffering), which overlaps better. */
int rvo_set_pair_chunks(rvo_ctx* ctx, int chunks);
/* No counterpart in the reference either: for callers that keep batches in flight on two contexts of one device.
 * Everything enqueued on ctx after this call waits until the frames last submitted on peer (rvo_match_frames_dev /
 * rvo_prepare_frames*) have finished their FAST stage.  Called before every submit with the other context as peer it
 * keeps the two batches apart: their VALU-bound FAST launches never run against each other and the image stages of one
 * batch run beside the per-pair stages of the other (+2 % over free-running contexts, which drift into lockstep).
 * A peer that has not submitted anything yet does not block. */
int rvo_wait_peer(rvo_ctx* ctx, rvo_ctx* peer);
/* No counterpart in the reference: how rvo_match_frames_dev runs its batches.  staged = 1 (default; env RVO_STAGED=0
 * overrides at rvo_create): the stages of a batch run on four HIP streams (frames / initial rotation / pair stages /
 * clique + Gauss-Newton) on one of four sets of pair work areas, so that successive calls pipeline inside ONE context:
 * the latency-bound tail of batch k runs beside the frame stages of batch k+1.  Rules for the caller: the frame slots of
 * a batch are read by its pair stages, so give consecutive batches different slots (e.g. alternate two halves of the
 * slot range) or the next batch's frame stages wait for them; results of the last two batches stay available
 * (rvo_get_batch_results).  staged = 0: one chain per call, cut into rvo_set_pair_chunks chunks.
 * The staged form runs on the context's own streams only (all of one priority):
 * rvo_set_stream(ctx, s != NULL) switches it off, and rvo_set_pipeline(ctx, 1) then returns RVO_E_UNSUPPORTED. */
int rvo_set_pipeline(rvo_ctx* ctx, int staged);
/* Results of the staged batch submitted `batches_ago` calls ago (0 = the latest, 1 = the one before); waits for that
 * batch only, later batches keep running. */
int rvo_get_batch_results(rvo_ctx* ctx, int batches_ago, int n, rvo_pair_result* out);
/* Use an existing HIP stream (hipStream_t passed as void*; NULL = the context's own). */
int rvo_set_stream(rvo_ctx* ctx, void* hip_stream);
int rvo_sync(rvo_ctx* ctx);

/* Frame::PrepareFrame (rgbdvo/frame.cpp:37-158) incl. the depth conversion of rgbdvo/vo.cpp:83-87 for
 * n frames: gray is n tightly packed W*H u8 images, depth n tightly packed Wd*Hd u16 images
 * (millimetres, 0 = invalid), fast_thr[n] the FAST threshold per frame (0..255, else RVO_E_INVALID), slots[n] the destination
 * frame slots.  Host-buffer form (copies H2D, then runs, then synchronises). */
int rvo_prepare_frames(rvo_ctx* ctx, int n, const uint8_t* gray, const uint16_t* depth, const int32_t* fast_thr,
                       const int32_t* slots);
/* Same, gray/depth already in device memory; enqueues only (no synchronisation).  Like every entry point here it trusts its
 * pointers: the buffers must hold n images of the sizes the context's intrinsics name (Wd x Hd for depth), a shorter one is
 * read past its end by the GPU. */
int rvo_prepare_frames_dev(rvo_ctx* ctx, int n, const uint8_t* gray_dev, const uint16_t* depth_dev,
                           const int32_t* fast_thr, const int32_t* slots);

/* FrameMatcher::Match (rgbdvo/fm.cpp:32-58) on prepared slots, for n independent pairs:
 * EstimateInitialRotation (rgbdvo/motion_estimate.cpp:268-304) + EstimateMotion (:149-243).
 * init_pose = NULL -> initial estimate is identity rotated by the initial-rotation estimate (fm.cpp:41-47);
 * otherwise n 3x4 poses used as EstimateMotion's init_motion_est directly (no initial rotation).
 * out may be NULL for the _async form; fetch later with rvo_get_pair_results. */
int rvo_match_pairs(rvo_ctx* ctx, int n, const int32_t* ref_slots, const int32_t* tgt_slots, const double* init_pose,
                    rvo_pair_result* out);
int rvo_match_pairs_async(rvo_ctx* ctx, int n, const int32_t* ref_slots, const int32_t* tgt_slots,
                          const double* init_pose);

/* FrameMatcher::Match (rgbdvo/fm.cpp:32-58) end to end for n independent pairs whose images are already in device
 * memory: frames 2i (reference) and 2i+1 (target) of gray_dev/depth_dev go through Frame::Prepare
 * (rgbdvo/frame.cpp:38-149) into slots[2i], slots[2i+1] with FAST thresholds fast_thr[2i], fast_thr[2i+1], then pair
 * work area i estimates slots[2i] -> slots[2i+1] exactly as rvo_match_pairs_async does.  The batch is cut into
 * chunks that run the whole frame+pair pipeline on separate HIP streams, so the latency-bound pair stages of one
 * chunk overlap the throughput-bound frame stages of another.  Enqueues only; results via rvo_get_pair_results.
 * Nothing is copied when the images already have the slot layout (width % 16 == 0, 16-byte aligned buffers): the
 * level-0 image and the depth image (sampled at a few thousand pixels) are read in place, so gray_dev and depth_dev
 * must stay valid and unchanged for as long as these slots are used (until they are prepared again), e.g. by a
 * later rvo_match_pairs that names them.  rvo_prepare_frames[_dev] and the stream entry points always copy. */
int rvo_match_frames_dev(rvo_ctx* ctx, int n, const uint8_t* gray_dev, const uint16_t* depth_dev, const int32_t* fast_thr,
                         const int32_t* slots, const double* init_pose);

/* Environment read once by rvo_create (deployment switches, each covered by the GPU tests): RVO_STAGED=0 (start with
 * rvo_set_pipeline(ctx, 0)), RVO_PAIR_CHUNKS=1..4 (rvo_set_pair_chunks), RVO_STREAM_GROUPS=1..4 and
 * RVO_STREAM_GROUP_MIN (into how many interleaved groups a stream step is cut, and from how many streams on),
 * RVO_CLIQUE_SPLIT=0 (greedy clique walk inside the sort kernel), RVO_RCCL_LIB (the librccl to open).  Knobs that only
 * served experiments (stream priorities, LDS padding, work-area set counts) exist in -DRVO_EXPERIMENT_KNOBS builds only. */

/* Front-of-path conversions of the reference's callers, on device memory:
 * ConvertBGR2Gray (helper.cpp:55-64, called at hand_held_demo.cpp:248, pose_graph.cpp:101,128) for n frames of the
 * context's width x height: gray = (u8)(0.114*B + 0.587*G + 0.299*R) in fp64; and DepthToPointCloud::operator()
 * (helper.cpp:30-53) for n depth images of depth_width x depth_height: x_to_z = 2*tan(hfov/2), y_to_z = 2*tan(vfov/2)
 * (helper.hpp:32-36); cloud = n * Wd*Hd * 3 floats.  Both enqueue on the context's stream and return. */
int rvo_bgr_to_gray_dev(rvo_ctx* ctx, int n, const uint8_t* bgr_dev, uint8_t* gray_dev);
int rvo_depth_to_point_cloud_dev(rvo_ctx* ctx, int n, const uint16_t* depth_dev, double x_to_z, double y_to_z,
                                 float* cloud_dev);

/* PoseGraphOptimizer::AddEdge (pose_graph.cpp:254-270): information matrix of each of the first n pair results,
 * info = cov.inverse() (6x6, row-major = column-major: symmetric up to rounding), computed on the device in one
 * launch; zeros for pairs whose status is not RVO_SUCCESS.  Synchronises. */
int rvo_get_pair_information(rvo_ctx* ctx, int n, double* info36);
/* No counterpart in the reference: rvo_get_pair_information for callers that cannot allocate device memory (code
 * under HIP graph capture set-up, tight serving loops, harnesses that forbid hipMalloc).  Same kernel and same
 * results, but the device-side staging area is a fixed array inside the library (RVO_NOALLOC_MAX_PAIRS x 36 doubles
 * = 1 179 648 bytes per GPU, held until the process ends) instead of memory owned by the context; the call itself
 * allocates and frees nothing.  n > RVO_NOALLOC_MAX_PAIRS is rejected with RVO_E_INVALID.  Every call on a GPU shares
 * that GPU's array: calls that overlap on one GPU (from several contexts or threads) race; serialise them. */
#define RVO_NOALLOC_MAX_PAIRS 4096
int rvo_get_pair_information_noalloc(rvo_ctx* ctx, int n, double* info36);
int rvo_get_pair_results(rvo_ctx* ctx, int n, rvo_pair_result* out);
/* EstimateInitialRotation alone: q[n][4] = (w,x,y,z) */
int rvo_initial_rotation(rvo_ctx* ctx, int n, const int32_t* prev_slots, const int32_t* cur_slots, double* q_wxyz);

/* Streaming: n_streams independent RGBDVO instances (rgbdvo/vo.cpp:55-165 state machine each) living in
 * slots [3*s, 3*s+3).  rvo_streams_reset = constructor + SetInitPose(identity) (rgbdvo/vo.hpp:42-45). */
int rvo_streams_reset(rvo_ctx* ctx, int n_streams);
int rvo_stream_step(rvo_ctx* ctx, int n_streams, const uint8_t* gray, const uint16_t* depth, rvo_stream_result* out);
/* No counterpart in the reference (its ProcessFrame takes host pointers and blocks, rgbdvo/vo.cpp:55): the same step
 * for hosts that keep frames coming.  The images are uploaded by a copy stream into one of two device staging sets, the
 * step waits for that upload only, and the call returns when everything is enqueued; results with
 * rvo_get_stream_results (or rvo_streams_record).  With page-locked host buffers (rvo_host_alloc) the upload of step
 * k+1 runs beside the kernels of step k.  The host buffers may be overwritten after rvo_wait_upload returns. */
int rvo_stream_step_host_async(rvo_ctx* ctx, int n_streams, const uint8_t* gray, const uint16_t* depth);
int rvo_wait_upload(rvo_ctx* ctx);
/* Page-locked host memory for image buffers (hipHostMalloc / hipHostFree). */
int rvo_host_alloc(size_t bytes, void** out);
int rvo_host_free(void* p);
int rvo_stream_step_dev(rvo_ctx* ctx, int n_streams, const uint8_t* gray_dev, const uint16_t* depth_dev);
int rvo_get_stream_results(rvo_ctx* ctx, int n_streams, rvo_stream_result* out);
/* No counterpart in the reference: after this call stream s reads image index[s] (0 <= index[s] < n_sources) of the
 * arrays handed to rvo_stream_step_dev, which then hold n_sources images instead of n_streams -- many streams replaying
 * a few sequences (benchmarks at configs[2]'s size, replicated cameras) without materialising one image per stream.
 * The step copies the images into the streams' frame slots anyway, so the index costs nothing.  index = NULL or
 * n_sources <= 0 clears it; rvo_streams_reset clears it too.  The host-buffer step forms refuse to run with an index. */
int rvo_streams_set_source_index(rvo_ctx* ctx, int n_sources, const int32_t* index /* [n_streams] */);
/* No counterpart in the reference (RGBDVO::ProcessFrame copies every depth image, vo.cpp:83-87 + Depth::SetDepth
 * depth.cpp:42-46): on != 0 makes rvo_stream_step_dev read the depth images where the caller put them instead of copying
 * them into the frame slots (614 KB of the 921 KB a 640x480 frame brings; only the few thousand pixels under keypoints and
 * refined matches of the CURRENT frame are ever read, and never after the step).  The caller then keeps depth_dev untouched
 * until the step has completed (rvo_sync, or a synchronising read of its results), and a stream's frame slots are no longer
 * valid targets for rvo_match_pairs.  The gray images are always copied (a frame is reference / previous frame of later
 * steps).  Off after rvo_create and after every rvo_streams_reset; the host-buffer step forms always copy. */
int rvo_streams_depth_in_place(rvo_ctx* ctx, int on);

/* Frame slots stream `stream` currently holds: out[3] = reference, previous, current frame
 * (ref_frame_/prev_frame_/curr_frame_ of rgbdvo/vo.hpp:67-69) -- lets the facade resolve match indices to keypoints. */
int rvo_get_stream_slots(rvo_ctx* ctx, int stream, int32_t* out);
/* Device pointer to the stream-result array (for a device-side trajectory buffer / RCCL gather). */
int rvo_stream_results_dev(rvo_ctx* ctx, void** dev_ptr);

/* Device-side trajectories (no counterpart in the reference, whose caller reads GetPose() after every frame,
 * hand_held_demo.cpp:252): after rvo_streams_record(ctx, max_steps) every rvo_stream_step[_dev] also appends one row of
 * RVO_TRAJ_DOUBLES doubles per stream -- pose 3x4 row-major, status, num_inliers, rc, used_fallback -- to a device
 * buffer laid out [stream][max_steps][RVO_TRAJ_DOUBLES], so a run of many steps needs no host read-back (and no
 * synchronisation) until its end.  Steps beyond max_steps are not recorded.  rvo_streams_reset restarts at row 0. */
#define RVO_TRAJ_DOUBLES 16
int rvo_streams_record(rvo_ctx* ctx, int max_steps);
int rvo_trajectories_dev(rvo_ctx* ctx, void** dev_ptr, int32_t* steps_recorded, int32_t* max_steps);
/* out[n_streams][steps][RVO_TRAJ_DOUBLES] (16 doubles per row, see above); synchronises */
int rvo_get_trajectories(rvo_ctx* ctx, int n_streams, int steps, double* out);

/* ---- multi-GPU: the final trajectory gather.  No counterpart in the reference (one process,
 * rgbdvo/vo.cpp:55).  Streams are sharded over one process per GPU with no collective in the loop; at the end every
 * rank contributes one equal-sized device block and receives all of them, rank-major: ncclAllGather over RCCL / xGMI.
 * Rank 0 calls rvo_comm_get_unique_id and hands the RVO_COMM_ID_BYTES bytes to every rank by whatever channel the host
 * program has (MPI, a file, torch.distributed's store ...); all ranks then call rvo_comm_create (collective: returns
 * when every rank has joined).  librccl is opened with dlopen on first use; a world of one needs no RCCL at all.
 * rvo_gather_trajectories is collective, enqueues on the communicator's own stream and synchronises it; local_dev must
 * be complete before the call (rvo_sync the producing context first). */
typedef struct rvo_comm rvo_comm;
#define RVO_COMM_ID_BYTES 128
int rvo_comm_get_unique_id(uint8_t* id /* [RVO_COMM_ID_BYTES] */);
int rvo_comm_create(int device, int rank, int world, const uint8_t* id, rvo_comm** out);
int rvo_comm_destroy(rvo_comm* comm);
int rvo_gather_trajectories(rvo_comm* comm, const void* local_dev, size_t bytes_per_rank, void* all_dev /* world * bytes_per_rank */);

/* ---- per-stage dump accessors (parity tests; the inspection surface of rgbdvo/pyramid.hpp:62-151 and
 * rgbdvo/motion_estimate.hpp:56-74) ---- */
/* info[6] = width,height,stride,num_keypoints,num_detected(after NMS),num_after_grid */
int rvo_get_level_info(rvo_ctx* ctx, int slot, int level, int32_t* info);
int rvo_get_level_image(rvo_ctx* ctx, int slot, int level, uint8_t* out /* height*stride */);
int rvo_get_grid_keypoints(rvo_ctx* ctx, int slot, int level, float* uvs /* [n][3] u,v,score */);
int rvo_get_keypoints(rvo_ctx* ctx, int slot, int level, float* uvs /* [n][3] */, double* rect /* [n][2] */,
                      double* xyz /* [n][3] */);
int rvo_get_descriptors(rvo_ctx* ctx, int slot, int level, uint8_t* out /* [n][80] */);
/* matches of pair `pair` from the last rvo_match_pairs, in the reference's post-sort order (at most `cap` rows
 * are written; returns the number written, or a negative error):
 * ints[n][7] = ref_idx,tgt_idx,level,status,inlier,id,degree; dbl[n][8] = refined u,v, rect u,v, xyz, reproj error */
int rvo_get_matches(rvo_ctx* ctx, int pair, int cap, int32_t* ints, double* dbl);
/* FeatureMatch::track_id of the same rows (rgbdvo/feature.hpp:297-302, assigned by motion_estimate.cpp:380-391): a
 * reference keypoint without a track opens one, the match and the target keypoint take its id, and the target
 * keypoints of matches that end up as outliers lose theirs again (motion_estimate.cpp:216-222).  A batch of
 * rvo_match_pairs / rvo_match_frames_dev counts tracks like ONE MotionEstimator working through its pairs in order
 * (FrameMatcher, fm.hpp:41): rvo_match_pairs walks the pairs in order when a frame slot is named by two of them (one
 * reference keyframe against many candidates), so a reference keypoint's track is opened by the first pair that matches
 * it and reused by the later ones; rvo_match_frames_dev requires distinct slots.  Every stream of rvo_stream_step has a
 * counter of its own (RGBDVO's estimator, vo.hpp:96).  Returns the number of rows written. */
int rvo_get_match_tracks(rvo_ctx* ctx, int pair, int cap, int32_t* track_ids);
/* FeatureMatch::consistency_vec and compatibility_degree (rgbdvo/feature.hpp:262-270, filled by
 * computeMaximallyConsistentClique, motion_estimate.cpp:470-523): out[i * n + id] = 1 iff the match in row i (rows as
 * rvo_get_matches writes them) is rigidly consistent with the match whose id is `id` (0 on the diagonal); n = the
 * pair's number of matches, cap >= n * n bytes.  The library keeps the N x N relation as a bit matrix; this unpacks it into
 * the reference's layout.  Returns n, or a negative error. */
int rvo_get_match_consistency(rvo_ctx* ctx, int pair, int cap, uint8_t* out);
/* KeypointData::track_id of the keypoints of a slot's level (rgbdvo/feature.hpp:75; -1 = none, frame.cpp:138), rows as
 * rvo_get_keypoints writes them.  Returns the number of rows written. */
int rvo_get_keypoint_tracks(rvo_ctx* ctx, int slot, int level, int32_t* track_ids);
/* KeypointData::track_id = -1 for every keypoint of the n slots, as after Frame::PrepareFrame (rgbdvo/frame.cpp:138).
 * For frames that stay prepared on the device where the reference's caller would prepare them again: the batched loop
 * closure (rgbdvo::FrameMatcher::MatchCachedKeyFrames) calls it on the reference keyframe and the candidates, so its
 * track ids equal those of SetRefFrame + one Match per candidate (rgbdvo/fm.cpp:22-58, pose_graph.cpp:104,132).  Enqueues only. */
int rvo_reset_tracks(rvo_ctx* ctx, int n, const int32_t* slots);
int rvo_get_lut(rvo_ctx* ctx, float* xmap, float* ymap);

/* Self-test hook (no counterpart in the reference): runs the wave-parallel replay of libstdc++'s std::sort
 * (nth < 0) or std::nth_element(nth) that the kernels use (slam_amd/csrc/wave_stdalgo.h) on n <= 8192 host keys,
 * comparator = descending by (key >> shift), shift = 16 or 22.  out receives the permuted keys. */
int rvo_selftest_stdalgo(int device, const uint32_t* keys, int n, int shift, int nth, uint32_t* out);

/* ---- measurement support ---- */
/* When enabled, every kernel launch is bracketed by HIP events on the context's stream. */
int rvo_profile_enable(rvo_ctx* ctx, int on);
/* Restricts the bracketing to launches of one kernel ("fast_grid", "refine", ...; NULL = all again): two events per
 * batch instead of thirty, so that a timed region can measure its dominant kernel without slowing the pipeline down. */
int rvo_profile_select(rvo_ctx* ctx, const char* kernel_name);
/* names/ms/calls arrays of capacity cap; returns the number of kernel kinds (accumulated since enable). */
int rvo_profile_read(rvo_ctx* ctx, int cap, const char** names, float* total_ms, int32_t* calls);

#ifdef __cplusplus
}
#endif
#endif /* RVO_H_ */
