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

// exclusive scan of one int per thread over a 256-thread block; *total = block sum.
// s_wave must hold 4 ints.  Contains two barriers.
__device__ inline int block_excl_scan_256(int v, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
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

// same for any block size of NW waves (s_wave: NW ints)
template <int NW>
__device__ inline int block_excl_scan(int v, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
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
template <bool VEC>
__global__ __launch_bounds__(256) void k_ingest(FrameBufs fb, FrameGeom g, const uint8_t* __restrict__ gray,
                                                const uint16_t* __restrict__ depth, const int32_t* __restrict__ slots) {
  const int f = blockIdx.y;
  const int slot = slots[f];
  if (slot < 0) return;
  const size_t npx = (size_t)g.W * g.H, ndp = (size_t)g.Wd * g.Hd;
  const uint8_t* gs = gray + f * npx;
  const uint16_t* dsrc = depth + f * ndp;
  uint8_t* gd = fb.img + (size_t)slot * g.img_bytes + g.lv[0].img_off;
  uint16_t* dd = fb.depth + (size_t)slot * ndp;
  const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
  if (VEC) {
    const int cpr = g.W / 16, spr = g.lv[0].stride / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(gs);
    uint4* d4 = reinterpret_cast<uint4*>(gd);
    for (int i = tid; i < cpr * g.H; i += nth) d4[(i / cpr) * spr + i % cpr] = s4[i];
    const uint4* ds4 = reinterpret_cast<const uint4*>(dsrc);
    uint4* dd4 = reinterpret_cast<uint4*>(dd);
    for (int i = tid; i < (int)(ndp / 8); i += nth) dd4[i] = ds4[i];
  } else {
    for (int i = tid; i < (int)npx; i += nth) gd[(size_t)(i / g.W) * g.lv[0].stride + i % g.W] = gs[i];
    for (int i = tid; i < (int)ndp; i += nth) dd[i] = dsrc[i];
  }
}

// ------------------------------------------------------------------ pyramid
#define PYR_TW 64
#define PYR_TH 16
#define PYR_SROWS (2 * PYR_TH + 3)   // 35
#define PYR_SCOLS (2 * PYR_TW + 8)   // 136 bytes loaded per row, from 2*ox0-4

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

// grid (ceil(dw/64), ceil(dh/16), n_frames); both truncating passes fused through LDS.
__global__ __launch_bounds__(256) void k_pyr_down(FrameBufs fb, FrameGeom g, const int32_t* __restrict__ slots, int lvl) {
  const LevelGeom S = g.lv[lvl - 1], D = g.lv[lvl];
  const int slot = slots[blockIdx.z];
  if (slot < 0) return;  // stream still in warm-up (vo.cpp:58-69)
  const uint8_t* __restrict__ src = fb.img + (size_t)slot * g.img_bytes + S.img_off;
  uint8_t* __restrict__ dst = fb.img + (size_t)slot * g.img_bytes + D.img_off;
  const int dw = S.w / 2, dh = S.h / 2;
  const int ox0 = blockIdx.x * PYR_TW, oy0 = blockIdx.y * PYR_TH;
  const int sx0 = 2 * ox0 - 4, sy0 = 2 * oy0 - 2;
  __shared__ __align__(16) uint8_t s_src[PYR_SROWS][PYR_SCOLS];
  __shared__ uint8_t s_mid[PYR_SROWS][PYR_TW];
  const int tid = threadIdx.x;

  for (int i = tid; i < PYR_SROWS * (PYR_SCOLS / 4); i += 256) {
    const int r = i / (PYR_SCOLS / 4), c4 = i % (PYR_SCOLS / 4);
    const int gy = sy0 + r, gx = sx0 + 4 * c4;
    uint32_t v = 0;
    if (gy >= 0 && gy < S.h) {
      const uint8_t* row = src + (size_t)gy * S.stride;
      if (gx >= 0 && gx + 3 < S.stride) {
        v = *reinterpret_cast<const uint32_t*>(row + gx);
      } else {
        for (int k = 0; k < 4; k++)
          if (gx + k >= 0 && gx + k < S.stride) v |= (uint32_t)row[gx + k] << (8 * k);
      }
    }
    *reinterpret_cast<uint32_t*>(&s_src[r][4 * c4]) = v;
  }
  __syncthreads();
  // pass 1: along x (source rows -> s_mid), what the reference stores transposed in pyrbuf
  for (int i = tid; i < PYR_SROWS * PYR_TW; i += 256) {
    const int r = i / PYR_TW, x = i % PYR_TW;
    const int gy = sy0 + r, ox = ox0 + x;
    int v = 0;
    if (gy >= 0 && gy < S.h && ox < dw) {
      const uint8_t* row = s_src[r];
      v = gauss_tap([&](int k) { return (int)row[k - sx0]; }, S.w, ox);
    }
    s_mid[r][x] = (uint8_t)v;
  }
  __syncthreads();
  // pass 2: along y
  const int x = tid & 63;
  for (int y = tid >> 6; y < PYR_TH; y += 4) {
    const int ox = ox0 + x, oy = oy0 + y;
    if (ox < dw && oy < dh) {
      int v = gauss_tap([&](int k) { return (int)s_mid[k - sy0][x]; }, S.h, oy);
      dst[(size_t)oy * D.stride + ox] = (uint8_t)v;
    }
  }
}

// ------------------------------------------------------------------ FAST-9
// 9 contiguous set bits in a cyclic 16-bit mask
__device__ inline bool has9(uint32_t m) {
  uint32_t d = m | (m << 16);
  uint32_t r = d & (d >> 1);
  r &= r >> 2;
  r &= r >> 4;
  r &= d >> 8;
  return (r & 0xFFFFu) != 0;
}

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
__device__ inline bool fast9_is_corner(const int* d, int b) {
  uint32_t br = 0, dk = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    br |= (uint32_t)(d[i] > b) << i;
    dk |= (uint32_t)(d[i] < -b) << i;
  }
  return has9(br) || has9(dk);
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

// append `val` for the lanes with pass == true to an LDS queue (order irrelevant); wave-aggregated counter bump
__device__ inline void lds_queue_push(bool pass, uint16_t val, uint16_t* q, int* count) {
  const unsigned long long m = __ballot(pass);
  if (m) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0, 64);
    if (pass) q[base + __popcll(m & ((1ull << lane) - 1ull))] = val;
  }
}

struct FastLds {  // byte offsets into dynamic LDS, computed on the host (all multiples of 4)
  int tile_pitch, score_pitch;
  int off_score, off_list, off_misc;
  int list_cap;
  int total;
};

// grid (total_buckets, n_frames), 256 threads: one workgroup per 80x80 grid cell.  The cell plus a
// 4-px halo is staged in LDS once; FAST scores are computed for the cell plus a 1-px halo; NMS
// (feature.cpp:5960-6056, non-strict: an equal neighbour suppresses) runs on the LDS score tile; the
// survivors are compacted in raster order (the order the reference pushes them into the bucket) and
// one lane replays libstdc++'s nth_element when the bucket exceeds max-keypoints-per-grid.
__global__ __launch_bounds__(256) void k_fast_grid(FrameBufs fb, FrameGeom g, FastLds L, const int32_t* __restrict__ slots,
                                                   const int32_t* __restrict__ thr) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint8_t* s_tile = lds;
  uint8_t* s_score = lds + L.off_score;
  uint32_t* s_list = reinterpret_cast<uint32_t*>(lds + L.off_list);
  int* s_misc = reinterpret_cast<int*>(lds + L.off_misc);  // [0..3] wave sums

  const int tid = threadIdx.x;
  const int slot = slots[blockIdx.y];
  if (slot < 0) return;
  const int b = thr[blockIdx.y];
  int lvl = 0;
  const int gb = blockIdx.x;  // bucket index within the slot
#pragma unroll
  for (int l = 1; l < RVO_MAX_LEVELS; l++)
    if (l < g.levels && gb >= g.lv[l].bucket_off) lvl = l;
  const LevelGeom G = g.lv[lvl];
  const int bi = gb - G.bucket_off;
  const int cx0 = (bi % G.gcols) * g.gw, cy0 = (bi / G.gcols) * g.gh;
  const int cw = min(g.gw, G.w - cx0), ch = min(g.gh, G.h - cy0);
  const uint8_t* __restrict__ img = fb.img + (size_t)slot * g.img_bytes + G.img_off;
  const int TP = L.tile_pitch, SP = L.score_pitch;

  // ---- stage cell + 4 px halo (index walkers avoid per-element integer division)
  const int trows = ch + 8, tdw = TP / 4;
  const int gx0 = cx0 - 4, gy0 = cy0 - 4;
  const bool aligned = (gx0 & 3) == 0;
  {
    const int q = 256 / tdw, rr = 256 % tdw;
    int r = tid / tdw, c4 = tid % tdw;
    while (r < trows) {
      const int gy = gy0 + r, gx = gx0 + 4 * c4;
      uint32_t v = 0;
      if (gy >= 0 && gy < G.h) {
        const uint8_t* row = img + (size_t)gy * G.stride;
        if (aligned && gx >= 0 && gx + 3 < G.stride) {
          v = *reinterpret_cast<const uint32_t*>(row + gx);
        } else {
          for (int k = 0; k < 4; k++)
            if (gx + k >= 0 && gx + k < G.stride) v |= (uint32_t)row[gx + k] << (8 * k);
        }
      }
      *reinterpret_cast<uint32_t*>(s_tile + r * TP + 4 * c4) = v;
      c4 += rr; r += q;
      if (c4 >= tdw) { c4 -= tdw; r++; }
    }
  }
  // ---- FAST, phase A: cheap necessary test on the 4 compass ring pixels for every pixel of the cell + 1 px halo.
  // A 9-arc of the 16-ring contains at least one pixel of each antipodal pair, so a corner needs
  // (p0|p8) & (p4|p12) all brighter than p+b, or all darker than p-b.  Survivors are queued in LDS as (sy<<8|sx).
  // Thread t owns pixels t, t+256, ... (<= 64 of them): pass bits are collected in a register mask first (independent
  // LDS reads, no per-iteration atomics), then each wave reserves queue space with ONE atomicAdd.
  const int sw = cw + 2, sh = ch + 2;
  const int npix = sw * sh;
  uint16_t* s_cand = reinterpret_cast<uint16_t*>(s_list);  // aliases the survivor list (used after later barriers)
  uint16_t* s_corner = s_cand + (npix + 1) / 2;             // second half: queue of true corners
  int* s_ncand = s_misc + 8;
  int* s_ncorner = s_misc + 9;
  uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_misc + 16);  // NMS survivor bitmap, one bit per cell pixel (raster)
  const int P = cw * ch;
  const int nwords = (P + 31) >> 5;
  if (tid == 0) { *s_ncand = 0; *s_ncorner = 0; }
  for (int w = tid; w < nwords; w += 256) s_bits[w] = 0;
  __syncthreads();
  const int lane = tid & 63;
  {
    const int q = 256 / sw, rr = 256 % sw;
    const int sy0 = tid / sw, sx0 = tid % sw;   // score-tile coords; cell pixel = (sx-1, sy-1)
    unsigned long long mask = 0;
    int sy = sy0, sx = sx0;
    for (int k = 0; sy < sh; k++) {
      const int gx = cx0 + sx - 1, gy = cy0 + sy - 1;
      s_score[sy * SP + sx] = 0;
      if (gx >= 3 && gx <= G.w - 4 && gy >= 3 && gy <= G.h - 4) {  // fast9Detect loop bounds, feature.cpp:3042-3043
        const uint8_t* t = s_tile + (sy + 3) * TP + (sx + 3);
        const int p = t[0];
        const int d0 = t[3 * TP] - p, d4 = t[3] - p, d8 = t[-3 * TP] - p, d12 = t[-3] - p;
        const bool br = ((d0 > b) | (d8 > b)) & ((d4 > b) | (d12 > b));
        const bool dk = ((d0 < -b) | (d8 < -b)) & ((d4 < -b) | (d12 < -b));
        mask |= (unsigned long long)(br | dk) << k;
      }
      sx += rr; sy += q;
      if (sx >= sw) { sx -= sw; sy++; }
    }
    // wave-level reservation
    const int mine = __popcll(mask);
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int n = __shfl_up(inc, o, 64); if (lane >= o) inc += n; }
    const int wtot = __shfl(inc, 63, 64);
    int base = 0;
    if (lane == 0 && wtot) base = atomicAdd(s_ncand, wtot);
    base = __shfl(base, 0, 64) + inc - mine;
    sy = sy0; sx = sx0;
    for (int k = 0; mask >> k; k++) {
      if ((mask >> k) & 1ull) s_cand[base++] = (uint16_t)((sy << 8) | sx);
      sx += rr; sy += q;
      if (sx >= sw) { sx -= sw; sy++; }
    }
  }
  __syncthreads();
  // ---- phase B: full segment test on the queued pixels; phase C: score for the true corners.  Both run with
  // all lanes busy.  If more than half of the pixels were queued (dense noise) the two queues do not fit and
  // the score is computed directly in phase B.
  const int ncand = *s_ncand;
  const bool two_queues = ncand <= npix / 2;
  for (int c0 = 0; c0 < ncand; c0 += 256) {
    const int c = c0 + tid;
    bool corner = false;
    uint16_t v = 0;
    if (c < ncand) {
      v = s_cand[c];
      const int sy = v >> 8, sx = v & 255;
      int d[16];
      fast9_ring_diffs(s_tile + (sy + 3) * TP + (sx + 3), TP, d);
      corner = fast9_is_corner(d, b);
      if (corner && !two_queues) s_score[sy * SP + sx] = (uint8_t)fast9_A(d);
    }
    if (two_queues) lds_queue_push(corner, v, s_corner, s_ncorner);
  }
  __syncthreads();
  const int nlist = two_queues ? *s_ncorner : ncand;
  const uint16_t* s_qlist = two_queues ? s_corner : s_cand;
  if (two_queues) {
    for (int c = tid; c < nlist; c += 256) {
      const uint16_t v = s_qlist[c];
      const int sy = v >> 8, sx = v & 255;
      int d[16];
      fast9_ring_diffs(s_tile + (sy + 3) * TP + (sx + 3), TP, d);
      s_score[sy * SP + sx] = (uint8_t)fast9_A(d);
    }
    __syncthreads();
  }

  // ---- NMS (feature.cpp:5960-6056: a corner survives iff no 8-neighbour corner has score >= its own) on the
  // queued pixels only; survivors set one bit of a raster-ordered bitmap of the cell.
  for (int c = tid; c < nlist; c += 256) {
    const uint16_t v = s_qlist[c];
    const int sy = v >> 8, sx = v & 255;
    if (sx < 1 || sx > cw || sy < 1 || sy > ch) continue;  // halo pixels belong to neighbouring buckets
    const uint8_t* cc = s_score + sy * SP + sx;
    const int sc = cc[0];
    if (sc == 0) continue;
    const bool sup = cc[-1] >= sc || cc[1] >= sc || cc[-SP - 1] >= sc || cc[-SP] >= sc || cc[-SP + 1] >= sc ||
                     cc[SP - 1] >= sc || cc[SP] >= sc || cc[SP + 1] >= sc;
    if (!sup) {
      const int idx = (sy - 1) * cw + (sx - 1);
      atomicOr(&s_bits[idx >> 5], 1u << (idx & 31));
    }
  }
  __syncthreads();
  // ---- raster-ordered compaction of the bitmap (the order the reference pushes keypoints into the bucket)
  int total = 0, base_run = 0;
  for (int w0 = 0; w0 < nwords; w0 += 256) {
    const int w = w0 + tid;
    uint32_t word = w < nwords ? s_bits[w] : 0u;
    int tot;
    int pos = base_run + block_excl_scan_256(__popc(word), s_misc, &tot);
    while (word) {
      const int bit = __ffs(word) - 1;
      word &= word - 1;
      const int idx = (w << 5) + bit;
      const int y = idx / cw, x = idx - y * cw;
      if (pos < L.list_cap) s_list[pos] = kp_pack(cx0 + x, cy0 + y, s_score[(y + 1) * SP + (x + 1)] - 1);
      pos++;
    }
    base_run += tot;
  }
  total = base_run;
  __syncthreads();

  // ---- GridKeyPointFilter::Filter for this bucket (pyramid.cpp:38-50)
  const int n = min(total, L.list_cap);
  const int maxp = g.max_per;
  if (n > maxp && tid < 64) {  // wave 0 replays std::nth_element; the image tile is dead by now and serves as scratch
    uint16_t* li = reinterpret_cast<uint16_t*>(s_tile);
    wavealgo::wave_nth_element<22>(s_list, maxp, n, li, li + L.list_cap);
  }
  __syncthreads();
  const int keep = min(n, maxp);
  uint32_t* out = fb.bk_kp + ((size_t)slot * g.total_buckets + gb) * maxp;
  for (int i = tid; i < keep; i += 256) out[i] = s_list[i];
  if (tid == 0) {
    fb.bk_cnt[(size_t)slot * g.total_buckets + gb] = keep;
    fb.bk_det[(size_t)slot * g.total_buckets + gb] = total;
  }
}

// ------------------------------------------------------------------ keypoint records + descriptors + XYZ
// grid (levels, n_frames), 256 threads.
__global__ __launch_bounds__(256) void k_build_keypoints(FrameBufs fb, FrameGeom g, const int32_t* __restrict__ slots) {
  __shared__ int s_wave[4];
  __shared__ int s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lvl = blockIdx.x;
  const int slot = slots[blockIdx.y];
  if (slot < 0) return;
  const LevelGeom G = g.lv[lvl];
  const uint8_t* __restrict__ img = fb.img + (size_t)slot * g.img_bytes + G.img_off;
  const uint16_t* __restrict__ depth = fb.depth + (size_t)slot * g.Wd * g.Hd;
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

  // frame.cpp:92-142
  const int mde = (RVO_WIN - 1) / 2 + 1;
  const int min_x = mde, min_y = mde, max_x = G.w - (mde + 1), max_y = G.h - (mde + 1);
  const int nslots = G.nbuckets * g.max_per;
  int base = 0;
  for (int c0 = 0; c0 < nslots; c0 += 256) {
    const int s = c0 + tid;
    int flag = 0;
    uint32_t k = 0;
    float ru = 0.f, rv = 0.f;
    double X = 0, Y = 0, Z = 0;
    if (s < nslots) {
      const int bk = s / g.max_per, j = s % g.max_per;
      if (j < bk_cnt[bk]) {
        k = bk_kp[s];
        const int x = kp_x(k), y = kp_y(k);
        if (!(x < min_x || x > max_x || y < min_y || y > max_y)) {
          const int idx = (y << lvl) * g.W + (x << lvl);
          ru = fb.xmap[idx];
          rv = fb.ymap[idx];
          if (!(ru < 0 || ru >= (float)g.W || rv < 0 || rv >= (float)g.H)) {
            const int du = (int)((double)ru + 0.5), dv = (int)((double)rv + 0.5);
            // Depth::RGB2DepthIdx, depth.hpp:41-43
            const int di = (int)((double)dv * (double)g.ys) * g.Wd + (int)((double)du * (double)g.xs);
            const uint16_t d = depth[di];
            if (d) {  // HaveXYZ: d ? d*1e-3 : NaN
              flag = 1;
              const double z = (double)(float)((double)d * 1e-3);
              const double rx = (double)du * g.fx_inv + g.neg_cx_div_fx;
              const double ry = (double)dv * g.fy_inv + g.neg_cy_div_fy;
              X = z * rx; Y = z * ry; Z = z;
            }
          }
        }
      }
    }
    int total;
    const int pos = base + block_excl_scan_256(flag, s_wave, &total);
    if (flag) {
      fb.kp_uvs[kbase + pos] = k;
      fb.kp_rect[kbase + pos] = make_float2(ru, rv);
      double* xyz = fb.kp_xyz + (kbase + pos) * 3;
      xyz[0] = X; xyz[1] = Y; xyz[2] = Z;
    }
    base += total;
  }
  if (tid == 0) fb.kp_cnt[slot * RVO_MAX_LEVELS + lvl] = base;
  __syncthreads();  // kp_uvs written by this block are re-read below (same workgroup: L1-coherent after barrier)
  __threadfence_block();

  // descriptors: one wave per keypoint; ExtractAlignedDescriptor + Normalized twice (feature.cpp:6140-6156, 6193-6207)
  for (int kp = wid; kp < base; kp += 4) {
    const uint32_t k = fb.kp_uvs[kbase + kp];
    const int x = kp_x(k), y = kp_y(k);
    int v0 = img[(size_t)(y + lane / 9 - 4) * G.stride + (x + lane % 9 - 4)];
    int v1 = 0;
    const int t1 = lane + 64;
    if (lane < RVO_DESC_LEN - 64) v1 = img[(size_t)(y + t1 / 9 - 4) * G.stride + (x + t1 % 9 - 4)];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int mean = wave_sum(v0 + v1) / RVO_DESC_LEN;
      if (mean < 128) {
        const int o = 128 - mean;
        v0 = min(255, v0 + o);
        if (lane < RVO_DESC_LEN - 64) v1 = min(255, v1 + o);
      } else if (mean > 128) {
        const int o = mean - 128;
        v0 = max(0, v0 - o);
        if (lane < RVO_DESC_LEN - 64) v1 = max(0, v1 - o);
      }
    }
    uint8_t* dsc = fb.kp_desc + (kbase + kp) * RVO_DESC_LEN;
    dsc[lane] = (uint8_t)v0;
    if (lane < RVO_DESC_LEN - 64) dsc[t1] = (uint8_t)v1;
  }
}
