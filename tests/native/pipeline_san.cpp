// Host-sanitizer driver (tools/host_sanitized_run.sh, GPU box): the entry points whose host side keeps the most state -- the
// staged rvo_match_frames_dev pipeline (four streams, four work-area sets, slot reuse), rvo_match_pairs[_async] between
// staged batches, the grouped rvo_stream_step_dev with a source index, depth in place and recorded trajectories, the
// accessors -- straight through the C ABI.  Built with hipcc for hipMalloc / hipMemcpy only; prints a checksum per phase.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/rvo.h"

#define CHK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rvo_last_error()); return 1; } } while (0)
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s frames.raw W H N\n", argv[0]); return 2; }
  const int W = atoi(argv[2]), H = atoi(argv[3]), N = atoi(argv[4]);
  const size_t npx = (size_t)W * H;
  std::vector<uint8_t> gray(npx * N);
  std::vector<uint16_t> depth(npx * N);
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  for (int i = 0; i < N; i++) {
    if (fread(gray.data() + npx * i, 1, npx, f) != npx || fread(depth.data() + npx * i, 2, npx, f) != npx) return 2;
  }
  fclose(f);
  rvo_intrinsics in{};
  in.width = in.depth_width = W; in.height = in.depth_height = H;
  in.fx = in.fy = 525.0 * W / 640.0; in.cx = W / 2.0; in.cy = H / 2.0;
  rvo_params pf, pv;
  rvo_default_fm_params(&pf);
  rvo_default_vo_params(&pv);
  pv.warmup_frames = 1;

  // ---- staged pair batches: pairs (frame i, frame i+1), P per batch, five slot sets in rotation
  const int P = N - 1, SETS = 5;
  rvo_ctx* c = nullptr;
  CHK(rvo_create(&in, &pf, 0, 2 * P * SETS, P, &c));
  uint8_t* dg = nullptr; uint16_t* dd = nullptr;
  HCHK(hipMalloc((void**)&dg, npx * 2 * P));
  HCHK(hipMalloc((void**)&dd, npx * 2 * P * 2));
  for (int i = 0; i < P; i++) {
    HCHK(hipMemcpy(dg + npx * 2 * i, gray.data() + npx * i, npx * 2, hipMemcpyHostToDevice));  // frames i, i+1 are adjacent in the file
    HCHK(hipMemcpy(dd + npx * 2 * i, depth.data() + npx * i, npx * 2 * 2, hipMemcpyHostToDevice));
  }
  std::vector<int32_t> thr(2 * P, 15), slots(2 * P);
  std::vector<rvo_pair_result> res(P);
  long long sum = 0;
  for (int b = 0; b < 14; b++) {
    for (int i = 0; i < 2 * P; i++) slots[i] = 2 * P * (b % SETS) + i;
    CHK(rvo_match_frames_dev(c, P, dg, dd, thr.data(), slots.data(), nullptr));
    if (b % 3 == 2 && (b - 1) % 5 != 4) {  // the batch before this one, while this one runs (unless a non-staged call used its work areas since)
      CHK(rvo_get_batch_results(c, 1, P, res.data()));
      for (auto& r : res) sum += r.num_inliers;
    }
    if (b % 5 == 4) {  // a non-staged estimate on slots of an earlier batch in between (joins the pipeline)
      std::vector<int32_t> ref{slots[0], slots[2]}, tgt{slots[1], slots[3]};
      std::vector<rvo_pair_result> r2(2);
      CHK(rvo_match_pairs(c, 2, ref.data(), tgt.data(), nullptr, r2.data()));
      sum += r2[0].num_matches + r2[1].num_matches;
      CHK(rvo_match_pairs_async(c, 2, ref.data(), tgt.data(), nullptr));
    }
  }
  CHK(rvo_get_pair_results(c, P, res.data()));
  int ok = 0;
  for (auto& r : res) { ok += r.status == RVO_SUCCESS; sum += r.num_matches; }
  std::vector<int32_t> ints(7 * 4096); std::vector<double> dbl(8 * 4096);
  const int nm = rvo_get_matches(c, 0, 4096, ints.data(), dbl.data());
  std::vector<uint8_t> cons((size_t)(nm > 0 ? nm : 1) * (nm > 0 ? nm : 1));
  if (nm > 0) CHK(rvo_get_match_consistency(c, 0, nm * nm, cons.data()) < 0);  // cap counts bytes of the n x n matrix
  std::vector<double> info(36 * P);
  CHK(rvo_get_pair_information(c, P, info.data()));
  CHK(rvo_get_pair_information_noalloc(c, P, info.data()));
  int32_t li[6];
  CHK(rvo_get_level_info(c, slots[0], 0, li));
  std::vector<uint8_t> desc((size_t)li[3] * 80 + 80);
  CHK(rvo_get_descriptors(c, slots[0], 0, desc.data()));
  printf("staged: %d/%d pairs succeeded, %d matches in pair 0, checksum %lld\n", ok, P, nm, sum);
  rvo_destroy(c);

  // ---- grouped stream steps on device buffers: S streams replaying the one sequence through a source index
  const int S = 72;  // > 32: four groups
  CHK(rvo_create(&in, &pv, 0, 3 * S, S, &c));
  CHK(rvo_streams_reset(c, S));
  CHK(rvo_streams_depth_in_place(c, 1));
  CHK(rvo_streams_record(c, N));
  std::vector<int32_t> idx(S, 0);
  CHK(rvo_streams_set_source_index(c, 1, idx.data()));
  uint8_t* sg = nullptr; uint16_t* sd = nullptr;
  HCHK(hipMalloc((void**)&sg, npx * N));
  HCHK(hipMalloc((void**)&sd, npx * N * 2));
  HCHK(hipMemcpy(sg, gray.data(), npx * N, hipMemcpyHostToDevice));
  HCHK(hipMemcpy(sd, depth.data(), npx * N * 2, hipMemcpyHostToDevice));
  for (int t = 0; t < N; t++) CHK(rvo_stream_step_dev(c, S, sg + npx * t, sd + npx * t));
  std::vector<rvo_stream_result> sr(S);
  CHK(rvo_get_stream_results(c, S, sr.data()));
  std::vector<double> traj((size_t)S * N * 16);
  CHK(rvo_get_trajectories(c, S, N, traj.data()));
  int same = 0;
  for (int s = 1; s < S; s++) same += sr[s].num_inliers == sr[0].num_inliers && sr[s].pose[3] == sr[0].pose[3];
  printf("streams: rc %d, %d inliers in stream 0, %d of %d replicas identical\n", sr[0].rc, sr[0].num_inliers, same, S - 1);
  // and the host-buffer step on the same context afterwards
  CHK(rvo_streams_reset(c, S));
  std::vector<uint8_t> hg(npx * S); std::vector<uint16_t> hd(npx * S);
  for (int t = 0; t < 3; t++) {
    for (int s = 0; s < S; s++) { std::copy(gray.begin() + npx * t, gray.begin() + npx * (t + 1), hg.begin() + npx * s); std::copy(depth.begin() + npx * t, depth.begin() + npx * (t + 1), hd.begin() + npx * s); }
    CHK(rvo_stream_step(c, S, hg.data(), hd.data(), sr.data()));
  }
  printf("host steps: rc %d, %d inliers\n", sr[S - 1].rc, sr[S - 1].num_inliers);
  rvo_destroy(c);
  (void)hipFree(dg); (void)hipFree(dd); (void)hipFree(sg); (void)hipFree(sd);
  return ok == P && same == S - 1 ? 0 : 4;
}
