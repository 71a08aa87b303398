// kernels_frame.h -- Frame::PrepareFrame on the GPU (rgbdvo/frame.cpp:37-158):
//   k_pyr_down        GaussPyramid::BuildNextLevel            pyramid.cpp:53-99
//   k_fast_grid       FAST + NMS + GridKeyPointFilter         feature.cpp:6058-6077, pyramid.cpp:15-51
//   k_build_keypoints border/LUT/depth filter, descriptors,   frame.cpp:92-147, feature.cpp:6140-6207,
//                     back-projection                          depth.cpp:48-78
// All integer/byte stages are bit-exact restatements; see DESIGN.md for the layouts.
#pragma once
#include "rvo_dev.h"
#include "stdalgo.h"
#include "wave_stdalgo.h"

// ------------------------------------------------------------------ helpers
__device__ inline int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive scan over the 64 lanes of a wave (all lanes active) on the DPP network: Hillis-Steele inside each row of 16
// (row_shr 1,2,4,8), then the row totals are carried with row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3).
__device__ inline int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return v;
}

// exclusive scan of one int per thread over a 256-thread block; *total = block sum.
// s_wave must hold 4 ints.  Contains two barriers.
__device__ inline int block_excl_scan_256(int v, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int inc = wave_incl_scan(v);
  __syncthreads();  // protect s_wave from the previous call's readers
  if (lane == 63) s_wave[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    int t = s_wave[w];
    if (w < wid) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

// 64-bit variant (several small counters packed into one word): Hillis-Steele over the wave with shuffles
__device__ inline unsigned long long block_excl_scan_256_u64(unsigned long long v, unsigned long long* s_wave, unsigned long long* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) s_wave[wid] = inc;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const unsigned long long t = s_wave[w];
    if (w < wid) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

// same for any block size of NW waves (s_wave: NW ints)
template <int NW>
__device__ inline int block_excl_scan(int v, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int inc = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) s_wave[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    int t = s_wave[w];
    if (w < wid) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

// ------------------------------------------------------------------ ingest
// Level-0 copy (frame.cpp:47-50) and raw depth copy into the frame slots; grid (blocks, n_frames).
// VEC: 16-byte path (W % 16 == 0, Wd*Hd % 8 == 0 and 16-byte aligned sources).
// copy_depth = 0: the depth image is not copied; the slot's depth_tab entry points at the caller's buffer instead
// (depth is only sampled at keypoints and refined matches, ~5 KB of the 614 KB image).
template <bool VEC>
__global__ __launch_bounds__(256) void k_ingest(FrameBufs fb, FrameGeom g, const uint8_t* __restrict__ gray,
                                                const uint16_t* __restrict__ depth, const int32_t* __restrict__ slots, int copy_depth, int copy_gray,
                                                const int32_t* __restrict__ src_index /* null: frame f reads image f */) {
  const int f = blockIdx.y;
  const int slot = slots[f];
  if (slot < 0) return;
  const size_t npx = (size_t)g.W * g.H, ndp = (size_t)g.Wd * g.Hd;
  const int fsrc = src_index ? src_index[f] : f;  // rvo_streams_set_source_index: several streams replay one image
  const uint8_t* gs = gray + fsrc * npx;
  const uint16_t* dsrc = depth + fsrc * ndp;
  uint8_t* gd = fb.img + (size_t)slot * g.img_bytes + g.lv[0].img_off;
  uint16_t* dd = fb.depth + (size_t)slot * ndp;
  const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
  if (tid == 0) {
    fb.depth_tab[slot] = copy_depth ? dd : dsrc;
    fb.l0_tab[slot] = copy_gray ? gd : gs;
  }
  if (VEC) {
    const int cpr = g.W / 16, spr = g.lv[0].stride / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(gs);
    uint4* d4 = reinterpret_cast<uint4*>(gd);
    if (copy_gray)
      for (int i = tid; i < cpr * g.H; i += nth) d4[(i / cpr) * spr + i % cpr] = s4[i];
    const uint4* ds4 = reinterpret_cast<const uint4*>(dsrc);
    uint4* dd4 = reinterpret_cast<uint4*>(dd);
    if (copy_depth)
      for (int i = tid; i < (int)(ndp / 8); i += nth) dd4[i] = ds4[i];
  } else {
    if (copy_gray)
      for (int i = tid; i < (int)npx; i += nth) gd[(size_t)(i / g.W) * g.lv[0].stride + i % g.W] = gs[i];
    if (copy_depth)
      for (int i = tid; i < (int)ndp; i += nth) dd[i] = dsrc[i];
  }
}

// ------------------------------------------------------------------ pyramid
#define PYR_TW 64
#ifndef PYR_TH
#define PYR_TH 32
#endif
#define PYR_SROWS (2 * PYR_TH + 3)   // 67
#define PYR_SCOLS (2 * PYR_TW + 32)  // 160 bytes loaded per row as ten 16-byte words, from 2*ox0-16

// One tap set of HorizontalGaussFilter (pyramid.cpp:67-99): output index r of a line of n samples.
template <class F>
__device__ inline int gauss_tap(F at, int n, int r) {
  const int last = n / 2 - 1;
  if (r == 0) return (at(0) + 4 * at(1) + 3 * at(2)) >> 3;
  if (r < last) return (at(2 * r - 2) + 4 * at(2 * r - 1) + 6 * at(2 * r) + 4 * at(2 * r + 1) + at(2 * r + 2)) >> 4;
  const int o = 2 * (last - 1);
  if (n & 1) return (3 * at(o) + 4 * at(o + 1) + at(o + 2)) >> 3;
  return (at(o) + 4 * at(o + 1) + 7 * at(o + 2) + 4 * at(o + 3)) >> 4;
}

// grid (ceil(dw/64), ceil(dh/32), n_frames); both truncating passes fused through LDS.  Interior outputs are computed
// four at a time on u16 pairs packed in 32-bit registers (a tap sum is at most 16*255, so plain integer adds and
// multiplies never carry between the halves); the first/last output of a line takes the reference's edge formulas
// through gauss_tap.
// gray_inplace / depth_inplace (rvo_match_frames_dev, lvl == 1 only, else null): the batch's images, read where they
// lie; the launch then also records them as the slots' level-0 / depth images (what k_ingest would do in a launch of
// its own), for the kernels that follow on the stream.
// src_index (with gray_inplace, may be null): frame fi reads image src_index[fi] of the caller's arrays.
// copy_l0 (with gray_inplace; the stream step with depth read in place): the level-0 image must outlive the caller's buffer
// (a stream's frame is the reference / previous frame of later steps), so every tile also writes the 128 x 64 level-0
// pixels under it from its staged window into the slot's own image -- the copy k_ingest made in a launch and a pass over
// the image of its own (the tiles' interiors cover the image: W, H even).
__global__ __launch_bounds__(256) void k_pyr_down(FrameBufs fb, FrameGeom g, const int32_t* __restrict__ slots, int lvl,
                                                  const uint8_t* __restrict__ gray_inplace, const uint16_t* __restrict__ depth_inplace,
                                                  const int32_t* __restrict__ src_index, int copy_l0) {
  RVO_OCC_SCOPE(1, 16);
  const LevelGeom S = g.lv[lvl - 1], D = g.lv[lvl];
  int tile, fi;  // XCD-aware: the tiles of a frame share source lines
  xcd_map(gridDim.x * gridDim.y, gridDim.z, blockIdx.y * gridDim.x + blockIdx.x, blockIdx.z, &tile, &fi);
  const int tile_y = (tile * (int)kInv16.v[gridDim.x]) >> 16, tile_x = tile - tile_y * (int)gridDim.x;  // tile / gridDim.x (tile < 512)
  const int slot = slots[fi];
  if (slot < 0) return;  // stream still in warm-up (vo.cpp:58-69)
  const int fsrc = (gray_inplace && src_index) ? src_index[fi] : fi;
  const uint8_t* __restrict__ src = gray_inplace ? gray_inplace + (size_t)fsrc * g.W * g.H : level_img(fb, g, slot, lvl - 1);
  uint8_t* __restrict__ l0_own = fb.img + (size_t)slot * g.img_bytes + g.lv[0].img_off;
  if (gray_inplace && tile == 0 && threadIdx.x == 0) {
    fb.l0_tab[slot] = copy_l0 ? l0_own : src;
    fb.depth_tab[slot] = depth_inplace + (size_t)fsrc * g.Wd * g.Hd;
  }
  uint8_t* __restrict__ dst = fb.img + (size_t)slot * g.img_bytes + D.img_off;
  const int dw = S.w / 2, dh = S.h / 2;
  const int ox0 = tile_x * PYR_TW, oy0 = tile_y * PYR_TH;
  const int sx0 = 2 * ox0 - 16, sy0 = 2 * oy0 - 2;
  __shared__ __align__(16) uint8_t s_src[PYR_SROWS][PYR_SCOLS];
  __shared__ __align__(16) uint16_t s_mid[PYR_SROWS][PYR_TW];
  const int tid = threadIdx.x;
  const uint32_t M = 0x00FF00FFu;

  {
    // all of a thread's loads (three) are issued before the first store: the loop form waited for every load in turn, three
    // serial round trips to memory per workgroup.  Addresses are clamped into the image and the loads unpredicated; what
    // lies outside the image is staged as zeros as before.
    constexpr int NITEMS = PYR_SROWS * (PYR_SCOLS / 16), NLD = (NITEMS + 255) / 256;
    uint4 v[NLD];
    bool ok[NLD];
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      const int i = tid + 256 * u;
      const int r = i / (PYR_SCOLS / 16), c16 = i % (PYR_SCOLS / 16);
      const int gy = sy0 + r, gx = sx0 + 16 * c16;  // multiple of 16; level strides are multiples of 16
      ok[u] = gy >= 0 && gy < S.h && gx >= 0 && gx < S.stride;
      const int cy = min(max(gy, 0), S.h - 1), cx = min(max(gx, 0), S.stride - 16);
      v[u] = gload_u4(src + (uint32_t)(cy * S.stride + cx));
    }
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      const int i = tid + 256 * u;
      if (i < NITEMS) {
        const int r = i / (PYR_SCOLS / 16), c16 = i % (PYR_SCOLS / 16);
        *reinterpret_cast<uint4*>(&s_src[r][16 * c16]) = ok[u] ? v[u] : make_uint4(0u, 0u, 0u, 0u);
        // the tile's own 128 x 64 source pixels: staged rows 2..65, bytes 16..143
        if (copy_l0 && ok[u] && r >= 2 && r < 2 + 2 * PYR_TH && c16 >= 1 && c16 <= 2 * PYR_TW / 16)
          *reinterpret_cast<uint4*>(l0_own + (size_t)(sy0 + r) * S.stride + (sx0 + 16 * c16)) = v[u];
      }
    }
  }
  __syncthreads();
  // pass 1: along x (source rows -> s_mid), what the reference stores transposed in pyrbuf.  Every group of four takes
  // the packed interior formula (bytes outside the image were staged as zeros); the first and the last output of a
  // line are then replaced by the reference's edge formulas (pyramid.cpp:67-99), so edge tiles pay one short patch
  // instead of a scalar path for the whole wave.
  const int lastx = S.w / 2 - 1, lasty = S.h / 2 - 1;
  for (int i = tid; i < PYR_SROWS * (PYR_TW / 4); i += 256) {
    const int r = i / (PYR_TW / 4), x4 = (i % (PYR_TW / 4)) * 4;
    const int gy = sy0 + r, ox = ox0 + x4;
    uint2 out = make_uint2(0u, 0u);
    if (gy >= 0 && gy < S.h && ox < dw) {
      const uint8_t* row = s_src[r];
      {
        // bytes b0..b15 = 2*x4+12 .. 2*x4+27 of the staged row; output ox+j is the 1-4-6-4-1 sum of b[2+2j .. 6+2j].
        // v_dot4_u32_u8 with the taps scaled by 16 (16 64 96 64 16): byte 1 of the 16-bit result is the sum >> 4, so one
        // v_perm_b32 packs two outputs into the u16 pair the second pass reads.
        const uint32_t* rw = reinterpret_cast<const uint32_t*>(row + 2 * x4 + 12);
        const uint32_t d0 = rw[0], d1 = rw[1], d2 = rw[2], d3 = rw[3];
        const uint32_t o0 = __builtin_amdgcn_udot4(d1, 0x00104060u, __builtin_amdgcn_udot4(d0, 0x40100000u, 0u, false), false);  // b2 b3 | b4 b5 b6
        const uint32_t o1 = __builtin_amdgcn_udot4(d2, 0x00000010u, __builtin_amdgcn_udot4(d1, 0x40604010u, 0u, false), false);  // b4..b7 | b8
        const uint32_t o2 = __builtin_amdgcn_udot4(d2, 0x00104060u, __builtin_amdgcn_udot4(d1, 0x40100000u, 0u, false), false);  // b6 b7 | b8 b9 b10
        const uint32_t o3 = __builtin_amdgcn_udot4(d3, 0x00000010u, __builtin_amdgcn_udot4(d2, 0x40604010u, 0u, false), false);  // b8..b11 | b12
        out.x = __builtin_amdgcn_perm(o1, o0, 0x0C050C01u);
        out.y = __builtin_amdgcn_perm(o3, o2, 0x0C050C01u);
      }
      const int je = lastx - ox;  // element of this group that is the last output of the line, if in 0..3
      if (ox == 0 || (je >= 0 && je < 4)) {
        auto at = [&](int k) { return (int)row[k - sx0]; };
        if (ox == 0) out.x = (out.x & 0xFFFF0000u) | (uint32_t)gauss_tap(at, S.w, 0);
        if (je >= 0 && je < 4 && lastx > 0) {
          const uint32_t v = (uint32_t)gauss_tap(at, S.w, lastx);
          if (je == 0) out.x = (out.x & 0xFFFF0000u) | v;
          else if (je == 1) out.x = (out.x & 0x0000FFFFu) | (v << 16);
          else if (je == 2) out.y = (out.y & 0xFFFF0000u) | v;
          else out.y = (out.y & 0x0000FFFFu) | (v << 16);
        }
      }
    }
    *reinterpret_cast<uint2*>(&s_mid[r][x4]) = out;
  }
  __syncthreads();
  // pass 2: along y; thread = 4 adjacent outputs of one row.  Same scheme: packed interior formula (s_mid rows outside
  // the image are zero), then the first / last output row takes the edge formula, also packed.
  for (int y = tid >> 4; y < PYR_TH; y += 16) {
    const int x4 = (tid & 15) * 4;
    const int ox = ox0 + x4, oy = oy0 + y;
    if (ox < dw && oy < dh) {
      uint8_t* drow = dst + (size_t)oy * D.stride + ox;
      uint2 A[5];
#pragma unroll
      for (int k = 0; k < 5; k++) A[k] = *reinterpret_cast<const uint2*>(&s_mid[2 * y + k][x4]);  // rows 2*oy-2 .. 2*oy+2
      uint32_t r01 = ((A[0].x + A[4].x + 6u * A[2].x + 4u * (A[1].x + A[3].x)) >> 4) & M;
      uint32_t r23 = ((A[0].y + A[4].y + 6u * A[2].y + 4u * (A[1].y + A[3].y)) >> 4) & M;
      if (oy == 0) {  // rows 0,1,2 = A[2],A[3],A[4]: (s0 + 4 s1 + 3 s2) >> 3
        r01 = ((A[2].x + 4u * A[3].x + 3u * A[4].x) >> 3) & M;
        r23 = ((A[2].y + 4u * A[3].y + 3u * A[4].y) >> 3) & M;
      } else if (oy == lasty) {  // o = 2*(lasty-1) = row A[0]
        if (S.h & 1) {
          r01 = ((3u * A[0].x + 4u * A[1].x + A[2].x) >> 3) & M;
          r23 = ((3u * A[0].y + 4u * A[1].y + A[2].y) >> 3) & M;
        } else {
          r01 = ((A[0].x + 4u * A[1].x + 7u * A[2].x + 4u * A[3].x) >> 4) & M;
          r23 = ((A[0].y + 4u * A[1].y + 7u * A[2].y + 4u * A[3].y) >> 4) & M;
        }
      }
      if (ox + 3 < dw) {
        *reinterpret_cast<uint32_t*>(drow) = __builtin_amdgcn_perm(r23, r01, 0x06040200u);
      } else {
        const uint32_t v[4] = {r01 & 0xFFu, r01 >> 16, r23 & 0xFFu, r23 >> 16};
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (ox + j < dw) drow[j] = (uint8_t)v[j];
      }
    }
  }
}

// ------------------------------------------------------------------ FAST-9
// Segment test of fast9Detect (feature.cpp:3033-5953; equivalence with the generated tree is checked
// exhaustively by oracle/check_fast_tree.py) and fast9CornerScore (feature.cpp:67-2995): the binary
// search returns the largest b for which the test passes = A-1, A = max over the 16 nine-arcs of the
// arc's min |difference| with a consistent sign.
__device__ inline void fast9_ring_diffs(const uint8_t* t, int pitch, int* d) {
  const int p = t[0];
  d[0] = t[3 * pitch] - p;       d[1] = t[3 * pitch + 1] - p;   d[2] = t[2 * pitch + 2] - p;   d[3] = t[pitch + 3] - p;
  d[4] = t[3] - p;               d[5] = t[-pitch + 3] - p;      d[6] = t[-2 * pitch + 2] - p;  d[7] = t[-3 * pitch + 1] - p;
  d[8] = t[-3 * pitch] - p;      d[9] = t[-3 * pitch - 1] - p;  d[10] = t[-2 * pitch - 2] - p; d[11] = t[-pitch - 3] - p;
  d[12] = t[-3] - p;             d[13] = t[pitch - 3] - p;      d[14] = t[2 * pitch - 2] - p;  d[15] = t[3 * pitch - 1] - p;
}
// A = score + 1 (only meaningful for corners: A > b)
__device__ inline int fast9_A(const int* d) {
  int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { lo2[i] = min(d[i], d[(i + 1) & 15]); hi2[i] = max(d[i], d[(i + 1) & 15]); }
#pragma unroll
  for (int i = 0; i < 16; i++) { lo4[i] = min(lo2[i], lo2[(i + 2) & 15]); hi4[i] = max(hi2[i], hi2[(i + 2) & 15]); }
  int best_lo = -256, best_hi = 256;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    int lo9 = min(min(lo4[i], lo4[(i + 4) & 15]), d[(i + 8) & 15]);
    int hi9 = max(max(hi4[i], hi4[(i + 4) & 15]), d[(i + 8) & 15]);
    best_lo = max(best_lo, lo9);
    best_hi = min(best_hi, hi9);
  }
  return max(best_lo, -best_hi);
}

__device__ inline int sum16(int v) {  // 16 lanes = one DPP row: quad permutes, then half-row and row mirrors
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
  return v;
}

// compass test of phase A on two pixels held in the u16 halves of each operand: corner candidates need
// min(max(n,s), max(e,w)) > p + b (bright) or max(min(n,s), min(e,w)) < p - b (dark).  cd2 = 0x7FFF - b in both halves;
// bit 15 of a half of *tb / *td is the bright / dark verdict (x + 0x7FFF >= 0x8000 <=> x >= 1; all values stay in u16).
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
typedef short s2_t __attribute__((ext_vector_type(2)));
__device__ inline void compass2f(uint32_t p, uint32_t n, uint32_t s, uint32_t e, uint32_t w, uint32_t cd2, uint32_t* tb, uint32_t* td) {
  const us2_t P = __builtin_bit_cast(us2_t, p), N = __builtin_bit_cast(us2_t, n), S = __builtin_bit_cast(us2_t, s);
  const us2_t E = __builtin_bit_cast(us2_t, e), W = __builtin_bit_cast(us2_t, w), CD = __builtin_bit_cast(us2_t, cd2);
  const us2_t brv = __builtin_elementwise_min(__builtin_elementwise_max(N, S), __builtin_elementwise_max(E, W));
  const us2_t dkv = __builtin_elementwise_max(__builtin_elementwise_min(N, S), __builtin_elementwise_min(E, W));
  *tb = __builtin_bit_cast(uint32_t, (us2_t)(brv + (CD - P)));
  *td = __builtin_bit_cast(uint32_t, (us2_t)((P + CD) - dkv));
}

__device__ inline uint32_t pk_min_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s2_t, a), __builtin_bit_cast(s2_t, b)));
}
__device__ inline uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2_t, a), __builtin_bit_cast(s2_t, b)));
}
__device__ inline uint32_t swap16(uint32_t a) { return __builtin_amdgcn_alignbit(a, a, 16); }

// A of ONE polarity (sgn = +1: ring brighter than the centre, -1: darker) = max over the 16 nine-arcs of the arc's
// min of sgn*(ring - p).  A corner has exactly one polarity with A > b (two nine-arcs of a 16-ring overlap), and that
// A is fast9_A's.  Computed on i16 pairs: ring position i in the low half, i + 8 in the high half, so every sliding
// minimum is one packed op for two positions; "position + 8" is a swap of the halves.
__device__ inline int fast9_A_pol(const uint8_t* t, int pitch, int sgn) {
  const int p = t[0];
  const uint32_t r0 = t[3 * pitch], r1 = t[3 * pitch + 1], r2 = t[2 * pitch + 2], r3 = t[pitch + 3];
  const uint32_t r4 = t[3], r5 = t[-pitch + 3], r6 = t[-2 * pitch + 2], r7 = t[-3 * pitch + 1];
  const uint32_t r8 = t[-3 * pitch], r9 = t[-3 * pitch - 1], r10 = t[-2 * pitch - 2], r11 = t[-pitch - 3];
  const uint32_t r12 = t[-3], r13 = t[pitch - 3], r14 = t[2 * pitch - 2], r15 = t[3 * pitch - 1];
  const s2_t S = {(short)sgn, (short)sgn};
  const short c1 = (short)(-p * sgn);
  const s2_t Cc = {c1, c1};
  auto mk = [&](uint32_t lo, uint32_t hi) {
    const s2_t R = __builtin_bit_cast(s2_t, lo | (hi << 16));
    return __builtin_bit_cast(uint32_t, (s2_t)(R * S + Cc));
  };
  const uint32_t dp[8] = {mk(r0, r8), mk(r1, r9), mk(r2, r10), mk(r3, r11), mk(r4, r12), mk(r5, r13), mk(r6, r14), mk(r7, r15)};
  uint32_t a[8], c[8], e[8];
#pragma unroll
  for (int i = 0; i < 7; i++) a[i] = pk_min_i16(dp[i], dp[i + 1]);
  a[7] = pk_min_i16(dp[7], swap16(dp[0]));
#pragma unroll
  for (int i = 0; i < 6; i++) c[i] = pk_min_i16(a[i], a[i + 2]);
  c[6] = pk_min_i16(a[6], swap16(a[0]));
  c[7] = pk_min_i16(a[7], swap16(a[1]));
#pragma unroll
  for (int i = 0; i < 4; i++) e[i] = pk_min_i16(c[i], c[i + 4]);
#pragma unroll
  for (int i = 4; i < 8; i++) e[i] = pk_min_i16(c[i], swap16(c[i - 4]));
  uint32_t best = pk_min_i16(e[0], swap16(dp[0]));
#pragma unroll
  for (int i = 1; i < 8; i++) best = pk_max_i16(best, pk_min_i16(e[i], swap16(dp[i])));
  return max((int)(short)(best & 0xFFFFu), (int)best >> 16);
}

#ifdef RVO_PHASE_TIMING  // ablation builds only (never the product library; tools/exp_phases.py): stop after a phase
__device__ int g_fast_stop;  // ablation: return after phase i (99 = run everything)
#define FAST_PHASE(i) do { if (g_fast_stop_k == i) return; } while (0)
#else
#define FAST_PHASE(i) do { } while (0)
#endif

// descriptor tap m (0..9) of lane sub (0..7) of a keypoint's 8-lane group in k_fast_grid: descriptor dwords sub and
// 8 + sub (taps 4 sub.., 32 + 4 sub..) and the two bytes 64 + 2 sub, 65 + 2 sub
__host__ __device__ constexpr int desc_tap_of_lane(int sub, int m) {
  return m < 4 ? 4 * sub + m : m < 8 ? 32 + 4 * sub + (m - 4) : 64 + 2 * sub + (m - 8);
}
// the taps' byte offsets from the keypoint in a tile of pitch TP, two int16 per word
template <int TP>
struct DescTaps {
  uint32_t w[8][5];
  constexpr DescTaps() : w{} {
    for (int sub = 0; sub < 8; sub++)
      for (int q = 0; q < 5; q++) {
        const int t0 = desc_tap_of_lane(sub, 2 * q), t1 = desc_tap_of_lane(sub, 2 * q + 1);
        const int o0 = (t0 / 9 - 4) * TP + (t0 % 9 - 4), o1 = (t1 / 9 - 4) * TP + (t1 % 9 - 4);
        w[sub][q] = ((uint32_t)o0 & 0xFFFFu) | ((uint32_t)o1 << 16);
      }
  }
};
template <int TP>
__device__ const DescTaps<TP> kDescTaps{};

struct FastLds {  // byte offsets into dynamic LDS, computed on the host (all multiples of 4)
  int tile_pitch, score_pitch;
  int cand_cap;  // phase-A candidate queue entries (u16)
  int off_score, off_list, off_misc;
  int list_cap;
  int total;
};

// grid (total_buckets, n_frames), 256 threads: one workgroup per 80x80 grid cell.  The cell plus a
// 4-px halo is staged in LDS once; FAST scores are computed for the cell plus a 1-px halo; NMS
// (feature.cpp:5960-6056, non-strict: an equal neighbour suppresses) runs on the LDS score tile; the
// survivors are compacted in raster order (the order the reference pushes them into the bucket) and
// one lane replays libstdc++'s nth_element when the bucket exceeds max-keypoints-per-grid.
// TPC / SPC: tile and score pitch as compile-time constants (0 = take them from L): with the default 80x80 grid every
// row offset then folds into the 16-bit immediate of the LDS instructions instead of 64-bit multiply-adds.
template <int TPC, int SPC>
__global__ __launch_bounds__(256) void k_fast_grid(FrameBufs fb, FrameGeom g, FastLds L, const int32_t* __restrict__ slots,
                                                   const int32_t* __restrict__ thr) {
  RVO_OCC_SCOPE(2, 32);
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t* s_tile = lds;
  uint8_t* s_score = lds + L.off_score;
  uint32_t* s_list = reinterpret_cast<uint32_t*>(lds + L.off_list);
  int* s_misc = reinterpret_cast<int*>(lds + L.off_misc);  // [0..3] wave sums

  const int tid = threadIdx.x;
  int gb, fi;  // bucket index within the slot, frame of the batch (XCD-aware: a frame's tiles share halo lines)
  xcd_map(gridDim.x, gridDim.y, blockIdx.x, blockIdx.y, &gb, &fi);
  const int slot = slots[fi];
  if (slot < 0) return;
#ifdef RVO_PHASE_TIMING
  const int g_fast_stop_k = *(volatile int*)&g_fast_stop;
#endif
  const int b = thr[fi];
  int lvl = 0;
#pragma unroll
  for (int l = 1; l < RVO_MAX_LEVELS; l++)
    if (l < g.levels && gb >= g.lv[l].bucket_off) lvl = l;
  const LevelGeom G = g.lv[lvl];
  const int bi = gb - G.bucket_off;
  const int brow = div_u16(bi, G.gcols);  // bi / gcols (bi < 2^16)
  const int cx0 = (bi - brow * G.gcols) * g.gw, cy0 = brow * g.gh;
  const int cw = min(g.gw, G.w - cx0), ch = min(g.gh, G.h - cy0);
  const uint8_t* __restrict__ img = level_img(fb, g, slot, lvl);
  const int TP = TPC ? TPC : L.tile_pitch, SP = SPC ? SPC : L.score_pitch;

  // ---- stage cell + 4 px halo: 16-byte global loads from the 16-aligned column at or left of cx0-4, stored as dwords
  // into compact LDS rows of TP bytes whose column shl (= 0 when cells start on multiples of 4) is image column
  // cx0-4.  LDS per workgroup decides how many workgroups share a CU (measured: 4 -> 5 per CU is 13 % faster).
  const int trows = ch + 8;
  const int gx0 = (cx0 - 4) & ~15, gy0 = cy0 - 4;
  const int shx = cx0 - 4 - gx0;      // 0..15: columns between the first loaded byte and the tile's first column
  const int shl = shx & 3, sd = shx >> 2, tdw = TP / 4;
  {
    // thread = (column word c, row r0s) with c fixed: the column's bounds and store guards are computed once, rows advance
    // by 256 / t16 per pass; three loads are in flight before their stores
    const int t16 = (shx + cw + 8 + 15) >> 4;  // 16-byte words per row (<= 18: cells are at most 253 wide)
    const int inv_t16 = (int)kInv16.v[t16];
    const int rpp = (256 * inv_t16) >> 16;     // rows per pass = 256 / t16
    const int r0s = (tid * inv_t16) >> 16, c = tid - r0s * t16;  // tid / t16, tid % t16 (exact for tid <= 256)
    const int gx = gx0 + 16 * c;
    const int d0 = 4 * c - sd;
    // Loads are never predicated: row and column are clamped into the image instead (32-bit offsets from the uniform
    // image pointer).  Tile bytes outside the image then hold copies of edge pixels; every test that would read them is
    // masked by the image bounds (fast9Detect's loop bounds; keypoints whose 9x9 window leaves the image are dropped by
    // k_build_keypoints' border test, so their descriptor rows are never used).
    const int gxc = min(max(gx, 0), G.stride - 16);
    const bool w0 = d0 >= 0 && d0 < tdw, w1 = d0 + 1 >= 0 && d0 + 1 < tdw, w2 = d0 + 2 >= 0 && d0 + 2 < tdw, w3 = d0 + 3 >= 0 && d0 + 3 < tdw;
    if (r0s < rpp) {
      for (int r = r0s; r < trows; r += 3 * rpp) {
        uint4 v[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
          const int gy = min(max(gy0 + r + u * rpp, 0), G.h - 1);
          v[u] = gload_u4(img + (uint32_t)(gy * G.stride + gxc));
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
          if (r + u * rpp < trows) {
            uint32_t* drow = reinterpret_cast<uint32_t*>(s_tile + (r + u * rpp) * TP) + d0;
            if (w0) drow[0] = v[u].x;
            if (w1) drow[1] = v[u].y;
            if (w2) drow[2] = v[u].z;
            if (w3) drow[3] = v[u].w;
          }
        }
      }
    }
  }
  // ---- FAST, phase A: cheap necessary test on the 4 compass ring pixels for every pixel of the cell + 1 px halo.
  // A 9-arc of the 16-ring contains at least one pixel of each antipodal pair, so a corner needs
  // (p0|p8) & (p4|p12) all brighter than p+b, or all darker than p-b.  Survivors are queued in LDS as (sy<<8|sx).
  // Thread t owns four-pixel groups t, t+256, ... (<= 16 of them): pass bits are collected in a register mask first
  // (independent LDS reads, no per-iteration atomics), then each wave reserves queue space with ONE atomicAdd.
  const int sw = cw + 2, sh = ch + 2;
  uint16_t* s_cand = reinterpret_cast<uint16_t*>(s_list);  // aliases the survivor list (used after later barriers)
  int* s_ncand = s_misc + 8;  // [0] bright entries (from the front), [1] dark entries (from the back)
  int* s_over = s_misc + 10;  // the candidate queue (L.cand_cap entries, half of the pixels) overflowed: dense fallback
  uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_misc + 16);  // NMS survivor bitmap, one bit per cell pixel (raster)
  const int P = cw * ch;
  const int nwords = (P + 31) >> 5;
  if (tid == 0) { s_ncand[0] = 0; s_ncand[1] = 0; *s_over = 0; }
  for (int w = tid; w < nwords; w += 256) s_bits[w] = 0;
  for (int w = tid; w < (L.off_list - L.off_score) / 4; w += 256) reinterpret_cast<uint32_t*>(s_score)[w] = 0;
  __syncthreads();
  FAST_PHASE(0);  // staging
  const int lane = tid & 63;
  const int qcap = L.cand_cap;
  // four-pixel groups: group j of score row sy covers tile x = 4j..4j+3 (score column sx sits at tile x = sx + 3 + shl)
  const int jlo = (shl + 3) >> 2, ngr = ((shl + sw + 2) >> 2) - jlo + 1;
  const int inv_ngr = (int)kInv16.v[ngr];  // ngr <= 64; t / ngr == (t * inv_ngr) >> 16 for t <= 256
  const int nseg = (256 * inv_ngr) >> 16;  // 256 / ngr
  // Column strips (below) when the tile's columns are dword-aligned with the image's (shl == 0: cells that start on a
  // multiple of 4): four-pixel groups that start at score column 0 (tile x = 4j + 3), so the sw columns take
  // ceil(sw / 4) groups -- 21 for the default 82, which leaves 12 strips of 7 rows to the 256 threads (84 x 84 pixel tests for
  // 82 x 82; dword-aligned groups needed 22 x 11 strips of 8 rows = 88 x 88).  Other shapes: the generic walker.
  const int ngs = (sw + 3) >> 2;
  const int inv_ngs = (int)kInv16.v[ngs];
  const int nsegs = (256 * inv_ngs) >> 16, rows_per_s = ((sh + nsegs - 1) * (int)kInv16.v[nsegs]) >> 16;  // 256 / ngs, ceil(sh / nsegs)
  constexpr int NR = 7;  // rows the strip code is unrolled for (more would cost registers: 2 per row); row k of a strip sits at bit k + 8 - NR of its byte
  const bool strips = shl == 0 && rows_per_s <= NR;
  const int sylo = max(0, 4 - cy0);       // first score row with gy >= 3
  {
    // 4 horizontally adjacent pixels per iteration: aligned LDS dwords (centre, left, right, 3 rows up, 3 rows down)
    // feed two packed-u16 evaluations (even / odd bytes).  Group j of score row sy covers tile x = 4j..4j+3.
    const int q = nseg, rr = 256 - nseg * ngr;
    // the walker's first group (recomputed where it is used: nothing of it stays live across the strips)
    auto walker_start = [&](int* sy_, int* j_) { const int seg = (tid * inv_ngr) >> 16; *sy_ = seg; *j_ = jlo + tid - seg * ngr; };
    const int txlo = max(3, 7 - cx0) + shl, txhi = min(sw + 2, G.w - cx0) + shl;  // fast9Detect loop bounds, feature.cpp:3042-3043
    const uint32_t cd1 = 0x7FFFu - (uint32_t)min(b, 255);
    const uint32_t cd2 = cd1 | (cd1 << 16);
    // Column strips: when 256 threads can take (group column, strip of <= 8 score rows) each, a thread reads the centre
    // dwords of the 14 tile rows its strip touches once, unpacks them once, and reuses every row as the "3 up", centre
    // and "3 down" row of three score rows (static register indexing, no index walkers).  The verdicts of a strip are
    // shifted into two 32-bit masks (bright, dark): pixel p of row k at bit 8p + k.
    // Other shapes use the generic walker below (64-bit masks, 4 bits per visited group).
    uint32_t sb = 0, sdk = 0;             // strips: bright / dark masks
    unsigned long long gb64 = 0, gd64 = 0;  // generic walker: bright / dark masks
    const int segs = (tid * inv_ngs) >> 16, sjs = tid - segs * ngs;  // strips: tid / ngs, tid % ngs
    if (strips) {
      const int syhi = min(sh - 1, G.h - 3 - cy0);  // rows sylo..syhi have 3 <= gy <= h-4
      const int r0 = max(segs * rows_per_s, sylo);  // first score row of this thread's strip
      const int r1 = min(min(segs * rows_per_s + rows_per_s, sh) - 1, syhi);
      const int sxlo = max(0, 4 - cx0), sxhi = min(sw - 1, G.w - 3 - cx0);  // fast9Detect loop bounds in score columns
      const uint32_t vm = (0xFu << min(max(sxlo - 4 * sjs, 0), 4)) & (0xFu >> min(max(4 * sjs + 3 - sxhi, 0), 4)) & 0xFu;
      if (segs < nsegs && r0 <= r1 && vm) {
        const uint8_t* col = s_tile + r0 * TP + 4 * sjs;  // tile row r0 = "3 up" of score row r0; dword D0 of the group
        const int nrows = r1 - r0 + 1;
        // The four pixels of group j are tile bytes 4j+3 .. 4j+6 = D0.b3 D1.b0 D1.b1 D1.b2 of the row's dwords D0 D1 D2 at
        // 4j; their west neighbours (x-3) are D0.b0..b3, their east neighbours (x+3) D1.b2 D1.b3 D2.b0 D2.b1.  Rows past
        // the strip read other LDS rows of this workgroup and are masked off below.
        {
          uint32_t Ce[NR + 6], Co[NR + 6];
          // sched_barriers keep the compiler from issuing every LDS read of the strip first (one register per read: 100
          // VGPRs instead of 48, a third of the occupancy): four rows of centre dwords at a time, and in the loop below
          // the next row's three dwords while the current row is evaluated
#pragma unroll
          for (int k0 = 0; k0 < NR + 6; k0 += 4) {
#pragma unroll
            for (int k = k0; k < k0 + 4 && k < NR + 6; k++) {
              const uint32_t D0 = *reinterpret_cast<const uint32_t*>(col + k * TP);
              const uint32_t D1 = *reinterpret_cast<const uint32_t*>(col + k * TP + 4);
              Ce[k] = __builtin_amdgcn_perm(D1, D0, 0x0C050C03u);
              Co[k] = D1 & 0x00FF00FFu;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int k = 0; k < NR; k++) {
            const uint8_t* row = col + (k + 3) * TP;
            const uint32_t D0 = *reinterpret_cast<const uint32_t*>(row), D1 = *reinterpret_cast<const uint32_t*>(row + 4), D2 = *reinterpret_cast<const uint32_t*>(row + 8);
            const uint32_t We = D0 & 0x00FF00FFu, Wo = __builtin_amdgcn_perm(0u, D0, 0x0C030C01u);
            const uint32_t Ee = __builtin_amdgcn_perm(D2, D1, 0x0C040C02u), Eo = __builtin_amdgcn_perm(D2, D1, 0x0C050C03u);
            uint32_t tbe, tde, tbo, tdo;
            compass2f(Ce[k + 3], Ce[k + 6], Ce[k], Ee, We, cd2, &tbe, &tde);
            compass2f(Co[k + 3], Co[k + 6], Co[k], Eo, Wo, cd2, &tbo, &tdo);
            // the verdicts are bit 15 of the halves = bit 7 of bytes 1 and 3: one v_perm gathers the four bytes (pixel p ->
            // byte p), and shifting the mask right once per row leaves row k at bit k + 8 - NR of every byte
            sb = (sb >> 1) | (__builtin_amdgcn_perm(tbo, tbe, 0x07030501u) & 0x80808080u);
            sdk = (sdk >> 1) | (__builtin_amdgcn_perm(tdo, tde, 0x07030501u) & 0x80808080u);
            __builtin_amdgcn_sched_barrier(0);
          }
          const uint32_t keep = ((((1u << nrows) - 1u) << (8 - NR)) * 0x01010101u) &
                                (((vm & 1u) ? 0x000000FFu : 0u) | ((vm & 2u) ? 0x0000FF00u : 0u) | ((vm & 4u) ? 0x00FF0000u : 0u) |
                                 ((vm & 8u) ? 0xFF000000u : 0u));
          sb &= keep; sdk &= keep;
        }
      }
    }
    int sy = 0, j = 0;
    if (!strips) walker_start(&sy, &j);
    for (int k = 0; !strips && sy < sh; k++) {
      const int gy = cy0 + sy - 1;
      if (gy >= 3 && gy <= G.h - 4) {
        const uint8_t* row = s_tile + (sy + 3) * TP + 4 * j;
        const uint32_t C = *reinterpret_cast<const uint32_t*>(row);
        const uint32_t Lw = *reinterpret_cast<const uint32_t*>(row - 4);
        const uint32_t R = *reinterpret_cast<const uint32_t*>(row + 4);
        const uint32_t U = *reinterpret_cast<const uint32_t*>(row - 3 * TP);
        const uint32_t D = *reinterpret_cast<const uint32_t*>(row + 3 * TP);
        const uint32_t W = __builtin_amdgcn_alignbyte(C, Lw, 1);  // tile x-3 of the four pixels
        const uint32_t E = __builtin_amdgcn_alignbyte(R, C, 3);   // tile x+3
        const uint32_t M = 0x00FF00FFu;
        uint32_t tbe, tde, tbo, tdo;
        compass2f(C & M, D & M, U & M, E & M, W & M, cd2, &tbe, &tde);
        compass2f((C >> 8) & M, (D >> 8) & M, (U >> 8) & M, (E >> 8) & M, (W >> 8) & M, cd2, &tbo, &tdo);
        const uint32_t vm = (0xFu << min(max(txlo - 4 * j, 0), 4)) & (0xFu >> min(max(4 * j + 3 - txhi, 0), 4)) & 0xFu;
        const uint32_t bb = ((tbe >> 15) & 1u) | ((tbo >> 14) & 2u) | ((tbe >> 29) & 4u) | ((tbo >> 28) & 8u);
        const uint32_t bd = ((tde >> 15) & 1u) | ((tdo >> 14) & 2u) | ((tde >> 29) & 4u) | ((tdo >> 28) & 8u);
        gb64 |= (unsigned long long)(bb & vm) << (4 * k);
        gd64 |= (unsigned long long)(bd & vm) << (4 * k);
      }
      j += rr; sy += q;
      if (j >= jlo + ngr) { j -= ngr; sy++; }
    }
    // wave-level reservation: bright candidates fill the queue from the front, dark ones from the back (a pixel that
    // passes both tests is queued twice), so phase B evaluates ONE polarity per entry.
    const int mineb = strips ? __popc(sb) : __popcll(gb64), mined = strips ? __popc(sdk) : __popcll(gd64);
    const int mine = mineb | (mined << 16);
    const int inc = wave_incl_scan(mine);
    const int wtot = __builtin_amdgcn_readlane(inc, 63);
    const int wb = wtot & 0xFFFF, wd = wtot >> 16;
    int baseb = 0, based = 0;
    if (lane == 0 && wb) baseb = atomicAdd(s_ncand, wb);
    if (lane == 0 && wd) based = atomicAdd(s_ncand + 1, wd);
    baseb = __builtin_amdgcn_readfirstlane(baseb);
    based = __builtin_amdgcn_readfirstlane(based);
    const bool fits = baseb + wb <= qcap && based + wd <= qcap;  // wave-uniform; the two ends meeting is checked after the barrier
    if (!fits && lane == 0) *s_over = 1;
    int posb = baseb + ((inc - mine) & 0xFFFF);
    int posd = qcap - 1 - (based + ((inc - mine) >> 16));
    if (fits && strips) {
      // the loop runs as long as the fullest lane of the wave, so an entry is just (thread << 5 | mask bit); phase B,
      // where every lane has one entry, turns it into (sy << 8 | sx)
      auto drain = [&](uint32_t m, int pos, int step) {
        uint16_t* q = s_cand + pos;
        const uint32_t tag = (uint32_t)tid << 5;
        while (m) {
          *q = (uint16_t)(tag | (uint32_t)(__ffs(m) - 1));
          m &= m - 1;
          q += step;
        }
      };
      drain(sb, posb, 1);
      drain(sdk, posd, -1);
    }
    if (fits && !strips) {
      walker_start(&sy, &j);
      for (int k = 0; k < 16 && ((gb64 | gd64) >> (4 * k)); k++) {
        uint32_t bits = (uint32_t)(gb64 >> (4 * k)) & 0xFu;
        while (bits) {
          const int i = __ffs(bits) - 1;
          bits &= bits - 1;
          s_cand[posb++] = (uint16_t)((sy << 8) | (4 * j + i - 3 - shl));
        }
        bits = (uint32_t)(gd64 >> (4 * k)) & 0xFu;
        while (bits) {
          const int i = __ffs(bits) - 1;
          bits &= bits - 1;
          s_cand[posd--] = (uint16_t)((sy << 8) | (4 * j + i - 3 - shl));
        }
        j += rr; sy += q;
        if (j >= jlo + ngr) { j -= ngr; sy++; }
      }
    }
  }
  __syncthreads();
  FAST_PHASE(1);  // phase A
  // ---- phase B: score A = fast9CornerScore + 1 of the queued polarity for every queue entry; the pixel is a corner iff
  // A > b (the segment test passes at threshold b iff b < A).  Entry c < nb: bright, entry qcap-1-(c-nb): dark.
  const int nb = s_ncand[0], nd = s_ncand[1];
  const bool over = *s_over != 0 || nb + nd > qcap;
  const int ncand = over ? 0 : nb + nd;
  for (int c = tid; c < ncand; c += 256) {
    const bool bright = c < nb;
    const int qi = bright ? c : qcap - 1 - (c - nb);
    const int v = s_cand[qi];
    int sy = v >> 8, sx = v & 255;
    if (strips) {  // entry = (producer thread << 5 | bit of its strip mask): pixel p of row k at bit 8p + k
      const int t = v >> 5, qb = v & 31;
      const int seg = (t * inv_ngs) >> 16;
      sy = max(seg * rows_per_s, sylo) + (qb & 7) - (8 - NR);
      sx = 4 * (t - seg * ngs) + (qb >> 3);
      s_cand[qi] = (uint16_t)((sy << 8) | sx);  // for the NMS pass (same thread, same entry)
    }
    const int A = fast9_A_pol(s_tile + (sy + 3) * TP + (sx + 3 + shl), TP, bright ? 1 : -1);
    if (A > b) s_score[sy * SP + sx] = (uint8_t)A;
  }
  if (over) {  // more than half of the pixels passed the compass test (dense noise): score every pixel
    for (int i = tid; i < sw * sh; i += 256) {
      const int sy = i / sw, sx = i - sy * sw;
      const int gx = cx0 + sx - 1, gy = cy0 + sy - 1;
      if (gx < 3 || gx > G.w - 4 || gy < 3 || gy > G.h - 4) continue;  // fast9Detect loop bounds, feature.cpp:3042-3043
      // both polarities through the packed evaluation of phase B (fast9_A's sixteen-register form would set the
      // kernel's register count): A = the larger one, as in fast9_A
      const uint8_t* t = s_tile + (sy + 3) * TP + (sx + 3 + shl);
      const int A = max(fast9_A_pol(t, TP, 1), fast9_A_pol(t, TP, -1));
      if (A > b) s_score[sy * SP + sx] = (uint8_t)A;
    }
  }
  __syncthreads();
  FAST_PHASE(2);  // phase B

  // ---- NMS (feature.cpp:5960-6056: a corner survives iff no 8-neighbour corner has score >= its own) on the
  // queued pixels only (all cell pixels in the dense fallback); survivors set one bit of a raster-ordered bitmap.
  auto nms = [&](int sy, int sx) {
    const uint8_t* cc = s_score + sy * SP + sx;
    const int sc = cc[0];
    if (sc == 0) return;
    const bool sup = cc[-1] >= sc || cc[1] >= sc || cc[-SP - 1] >= sc || cc[-SP] >= sc || cc[-SP + 1] >= sc ||
                     cc[SP - 1] >= sc || cc[SP] >= sc || cc[SP + 1] >= sc;
    if (!sup) {
      const int idx = (sy - 1) * cw + (sx - 1);
      atomicOr(&s_bits[idx >> 5], 1u << (idx & 31));
    }
  };
  for (int c = tid; c < ncand; c += 256) {
    const uint16_t v = s_cand[c < nb ? c : qcap - 1 - (c - nb)];
    const int sy = v >> 8, sx = v & 255;
    if (sx < 1 || sx > cw || sy < 1 || sy > ch) continue;  // halo pixels belong to neighbouring buckets
    nms(sy, sx);
  }
  if (over)
    for (int i = tid; i < P; i += 256) nms(i / cw + 1, i % cw + 1);
  __syncthreads();
  FAST_PHASE(3);  // NMS
  // ---- raster-ordered compaction of the bitmap (the order the reference pushes keypoints into the bucket)
  int total = 0, base_run = 0;
  const uint32_t inv_cw = kInv32.v[cw];  // idx / cw == umulhi(idx, inv_cw) for idx < 2^16, cw < 2^8
  for (int w0 = 0; w0 < nwords; w0 += 256) {
    const int w = w0 + tid;
    uint32_t word = w < nwords ? s_bits[w] : 0u;
    int tot;
    int pos = base_run + block_excl_scan_256(__popc(word), s_misc, &tot);
    while (word) {
      const int bit = __ffs(word) - 1;
      word &= word - 1;
      const int idx = (w << 5) + bit;
      const int y = (int)div_inv32((uint32_t)idx, cw, inv_cw), x = idx - y * cw;
      if (pos < L.list_cap) s_list[pos] = kp_pack(cx0 + x, cy0 + y, s_score[(y + 1) * SP + (x + 1)] - 1);
      pos++;
    }
    base_run += tot;
  }
  total = base_run;
  __syncthreads();
  FAST_PHASE(4);  // compaction

  // the descriptor phase's tap table (below) is fetched now: its latency passes behind the nth_element replay instead of
  // stalling the workgroup's last phase
  uint32_t tapw[5] = {0u, 0u, 0u, 0u, 0u};
  if (TPC) {
#pragma unroll
    for (int q = 0; q < 5; q++) tapw[q] = kDescTaps<TPC ? TPC : 4>.w[tid & 7][q];
  }
  // ---- GridKeyPointFilter::Filter for this bucket (pyramid.cpp:38-50)
  const int n = min(total, L.list_cap);
  const int maxp = g.max_per;
  if (n > maxp && tid < 64) {  // wave 0 replays std::nth_element; the score tile is dead by now and serves as scratch
    uint16_t* li = reinterpret_cast<uint16_t*>(s_score);
    wavealgo::wave_nth_element<22>(s_list, maxp, n, li, li + L.list_cap);
  }
  __syncthreads();
  FAST_PHASE(5);  // nth_element
  const int keep = min(n, maxp);
  const size_t bslot = ((size_t)slot * g.total_buckets + gb) * maxp;
  uint32_t* out = fb.bk_kp + bslot;
#pragma unroll 1
  for (int i = tid; i < keep; i += 256) out[i] = s_list[i];
  if (tid == 0) {
    fb.bk_cnt[(size_t)slot * g.total_buckets + gb] = keep;
    fb.bk_det[(size_t)slot * g.total_buckets + gb] = total;
  }
  FAST_PHASE(6);  // bucket list written
  // ---- descriptors of the kept keypoints (ExtractAlignedDescriptor + Normalized twice, feature.cpp:6140-6156,
  // 6193-6207) straight from the LDS tile: the 9x9 window of a keypoint of this cell lies inside the cell + 4 px halo.
  // 8 lanes per keypoint, so the default 25 kept keypoints are one pass of the 32 groups: lane l owns descriptor dwords
  // l and 8 + l (taps 4l.., 32 + 4l..) and the two bytes 64 + 2l, 65 + 2l -- ten taps per lane.
  {
    static_assert(RVO_DESC_LEN == 80, "lane -> tap assignment below");
    const int sub = tid & 7, grp = tid >> 3;
    // tap offsets of this lane relative to the keypoint's tile byte: from a table in constant memory when the pitch is a
    // compile-time constant (two 16-bit offsets per word; computing t / 9, t % 9 for ten taps cost 70 instructions)
    int toff[10];
    if (TPC) {
#pragma unroll
      for (int q = 0; q < 5; q++) {
        const uint32_t w = tapw[q];
        toff[2 * q] = (int)(short)(w & 0xFFFFu);
        toff[2 * q + 1] = (int)w >> 16;
      }
    } else {
#pragma unroll
      for (int m = 0; m < 10; m++) {
        const int t = desc_tap_of_lane(sub, m);
        toff[m] = (t / 9 - 4) * TP + (t % 9 - 4);
      }
    }
    for (int i = grp; i < keep; i += 32) {
      const uint32_t k = s_list[i];
      const uint8_t* c = s_tile + (kp_y(k) - cy0 + 4) * TP + (kp_x(k) - cx0 + 4 + shl);
      // the ten taps as five u16 pairs, each value in the HIGH byte of its half (v << 8): the saturating packed adds
      // then clamp at 255 (0xFFFF >> 8) and at 0 exactly as Descriptor::Normalized's _mm_adds_epu8 / _mm_subs_epu8 do,
      // one instruction per two taps and direction
      us2_t P[5];
#pragma unroll
      for (int q = 0; q < 5; q++)
        P[q] = __builtin_bit_cast(us2_t, __builtin_amdgcn_perm((uint32_t)c[toff[2 * q + 1]], (uint32_t)c[toff[2 * q]], 0x040C000Cu));
#pragma unroll
      for (int pass = 0; pass < 2; pass++) {
        // the lane's tap sum: v_dot4 over the two value bytes of every pair (the low bytes are not always zero: a half that
        // saturated holds 0xFFFF), on two accumulators (a dependent dot chain pays wait states)
        auto hw = [](us2_t x) { return __builtin_bit_cast(uint32_t, x); };
        const uint32_t sa = __builtin_amdgcn_udot4(hw(P[4]), 0x01000100u, __builtin_amdgcn_udot4(hw(P[2]), 0x01000100u, __builtin_amdgcn_udot4(hw(P[0]), 0x01000100u, 0u, false), false), false);
        const uint32_t sb2 = __builtin_amdgcn_udot4(hw(P[3]), 0x01000100u, __builtin_amdgcn_udot4(hw(P[1]), 0x01000100u, 0u, false), false);
        uint32_t sum = sa + sb2;
        sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
        sum += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sum, 0x141, 0xF, 0xF, true);  // row_half_mirror: the 8 lanes of the group
        const int mean = (int)(sum / (uint32_t)RVO_DESC_LEN);
        const int o = 128 - mean;  // mean < 128: add and clamp at 255; mean > 128: subtract and clamp at 0
        const us2_t up = __builtin_bit_cast(us2_t, __builtin_amdgcn_perm(0u, (uint32_t)max(o, 0), 0x000C000Cu));
        const us2_t dn = __builtin_bit_cast(us2_t, __builtin_amdgcn_perm(0u, (uint32_t)max(-o, 0), 0x000C000Cu));
#pragma unroll
        for (int q = 0; q < 5; q++) P[q] = __builtin_elementwise_sub_sat(__builtin_elementwise_add_sat(P[q], up), dn);
      }
      uint8_t* dsc = fb.bk_desc + (bslot + i) * RVO_DESC_LEN;
      auto hb = [](us2_t x) { return __builtin_bit_cast(uint32_t, x); };
      reinterpret_cast<uint32_t*>(dsc)[sub] = __builtin_amdgcn_perm(hb(P[1]), hb(P[0]), 0x07050301u);
      reinterpret_cast<uint32_t*>(dsc)[8 + sub] = __builtin_amdgcn_perm(hb(P[3]), hb(P[2]), 0x07050301u);
      reinterpret_cast<uint16_t*>(dsc)[32 + sub] = (uint16_t)__builtin_amdgcn_perm(0u, hb(P[4]), 0x0C0C0301u);
    }
  }
}

// ------------------------------------------------------------------ keypoint records + XYZ
// grid (levels, n_frames), 256 threads.
#define BK_CP 4  // descriptor-copy trips in flight (8: slower)
#define BK_CH 3  // 256-slot chunks examined together (5, the whole default level 0 in one go: 0.081 ms alone against 0.054)
__global__ __launch_bounds__(256) void k_build_keypoints(FrameBufs fb, FrameGeom g, const int32_t* __restrict__ slots) {
  RVO_OCC_SCOPE(3, 1);
  __shared__ unsigned long long s_wave64[4];
  __shared__ int s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lvl = blockIdx.x;
  const int slot = slots[blockIdx.y];
  if (slot < 0) return;
  const LevelGeom G = g.lv[lvl];
  const uint16_t* __restrict__ depth = fb.depth_tab[slot];
  const int32_t* bk_cnt = fb.bk_cnt + (size_t)slot * g.total_buckets + G.bucket_off;
  const int32_t* bk_det = fb.bk_det + (size_t)slot * g.total_buckets + G.bucket_off;
  const uint32_t* bk_kp = fb.bk_kp + ((size_t)slot * g.total_buckets + G.bucket_off) * g.max_per;
  const size_t kbase = (size_t)slot * g.kp_total + G.kp_off;

  // level totals (num_detected_keypoints_, frame.cpp:76-77)
  int det = 0, kept = 0;
  for (int i = tid; i < G.nbuckets; i += 256) { det += bk_det[i]; kept += bk_cnt[i]; }
  det = wave_sum(det);
  kept = wave_sum(kept);
  if (lane == 0) { s_red[wid] = det; s_red[4 + wid] = kept; }
  __syncthreads();
  if (tid == 0) {
    fb.det_cnt[slot * RVO_MAX_LEVELS + lvl] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    fb.grid_cnt[slot * RVO_MAX_LEVELS + lvl] = s_red[4] + s_red[5] + s_red[6] + s_red[7];
  }

  // frame.cpp:92-142.  BK_CH chunks of 256 bucket slots are examined together: their gathers (bucket count -> keypoint ->
  // undistortion LUT -> depth) are all in flight at once, one block scan (the
  // chunks' flags packed 10 bits apiece) places them.
  const int mde = (RVO_WIN - 1) / 2 + 1;
  const int min_x = mde, min_y = mde, max_x = G.w - (mde + 1), max_y = G.h - (mde + 1);
  const int nslots = G.nbuckets * g.max_per;
  int base = 0;
  const int npx_rgb = g.W * g.H, npx_d = g.Wd * g.Hd;
  for (int c0 = 0; c0 < nslots; c0 += 256 * BK_CH) {
    // Every gather below is issued for every lane with a clamped (always valid) index and its result is masked
    // afterwards: loads under divergent conditions would each wait for their data before the next chunk's is issued.
    int flag[BK_CH];
    uint32_t k[BK_CH];
    float ru[BK_CH], rv[BK_CH];
    double X[BK_CH], Y[BK_CH], Z[BK_CH];
    bool have[BK_CH];
    int cnt[BK_CH], jj[BK_CH];
#pragma unroll
    for (int u = 0; u < BK_CH; u++) {
      const int s = min(c0 + u * 256 + tid, nslots - 1);
      const int bk = div_u16(s, g.max_per);
      jj[u] = s - bk * g.max_per;
      cnt[u] = bk_cnt[bk];
      k[u] = bk_kp[s];
    }
    int idx[BK_CH];
#pragma unroll
    for (int u = 0; u < BK_CH; u++) {
      have[u] = (c0 + u * 256 + tid < nslots) && jj[u] < cnt[u];
      const int x = kp_x(k[u]), y = kp_y(k[u]);
      have[u] = have[u] && !(x < min_x || x > max_x || y < min_y || y > max_y);
      idx[u] = have[u] ? (y << lvl) * g.W + (x << lvl) : 0;  // a slot beyond its bucket's count holds stale bits
    }
#pragma unroll
    for (int u = 0; u < BK_CH; u++) {
      ru[u] = fb.xmap[idx[u]];
      rv[u] = fb.ymap[idx[u]];
    }
    int du[BK_CH], dv[BK_CH];
    uint16_t dd[BK_CH];
#pragma unroll
    for (int u = 0; u < BK_CH; u++) {
      have[u] = have[u] && !(ru[u] < 0 || ru[u] >= (float)g.W || rv[u] < 0 || rv[u] >= (float)g.H);
      du[u] = (int)((double)ru[u] + 0.5); dv[u] = (int)((double)rv[u] + 0.5);
      // Depth::RGB2DepthIdx, depth.hpp:41-43
      const int di = (int)((double)dv[u] * (double)g.ys) * g.Wd + (int)((double)du[u] * (double)g.xs);
      // rect_u in [W-0.5, W) rounds to du = W (distorted cameras only): the reference then reads the depth pixel
      // and the ray (rays_->col(v*W+u), depth.cpp:72) of column 0 of the NEXT row; past the last row both reads
      // leave their buffers (undefined there: Eigen's assert aborts) -- such a keypoint is dropped here.
      const int ri = dv[u] * g.W + du[u];
      have[u] = have[u] && di < npx_d && ri < npx_rgb;
      dd[u] = depth[have[u] ? di : 0];
    }
#pragma unroll
    for (int u = 0; u < BK_CH; u++) {
      flag[u] = (have[u] && dd[u] != 0) ? 1 : 0;  // HaveXYZ: d ? d*1e-3 : NaN
      const double z = (double)(float)((double)dd[u] * 1e-3);
      const int rxu = du[u] < g.W ? du[u] : du[u] - g.W, ryv = du[u] < g.W ? dv[u] : dv[u] + 1;  // pixel the ray index ri names
      const double rx = (double)rxu * g.fx_inv + g.neg_cx_div_fx;
      const double ry = (double)ryv * g.fy_inv + g.neg_cy_div_fy;
      X[u] = z * rx; Y[u] = z * ry; Z[u] = z;
      if (!flag[u]) { ru[u] = 0.f; rv[u] = 0.f; }
    }
    // one block scan for all chunks: their flags travel in 12-bit fields of a 64-bit word (a chunk counts at most 256)
    unsigned long long packed = 0;
#pragma unroll
    for (int u = 0; u < BK_CH; u++) packed |= (unsigned long long)flag[u] << (12 * u);
    unsigned long long total;
    const unsigned long long pre = block_excl_scan_256_u64(packed, s_wave64, &total);
#pragma unroll
    for (int u = 0; u < BK_CH; u++) {
      const int pos = base + (int)((pre >> (12 * u)) & 4095);
      if (flag[u]) {
        fb.kp_uvs[kbase + pos] = k[u];
        fb.kp_src[kbase + pos] = (uint16_t)(c0 + u * 256 + tid);
        fb.kp_rect[kbase + pos] = make_float2(ru[u], rv[u]);
        fb.kp_track[kbase + pos] = -1;  // frame.cpp:138: not part of a track yet
        double* xyz = fb.kp_xyz + (kbase + pos) * 3;
        xyz[0] = X[u]; xyz[1] = Y[u]; xyz[2] = Z[u];
      }
      base += (int)((total >> (12 * u)) & 4095);
    }
  }
  if (tid == 0) fb.kp_cnt[slot * RVO_MAX_LEVELS + lvl] = base;
  // The survivors' descriptors stay where k_fast_grid wrote them (bk_desc, bucket order): kp_src names the row, k_match
  // reads through it.  (Rounds 1-2 copied the rows into keypoint order here: 160 B of traffic per keypoint and half of
  // this kernel's time.)
}

