// kernels_io.h -- front-of-path conversions of the reference's callers, streaming and
// HBM-bound: /root/reference/helper.cpp:30-64 (namespace rgbdslam), called at hand_held_demo.cpp:248 and
// pose_graph.cpp:101,128 before every ProcessFrame / Match.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ConvertBGR2Gray (helper.cpp:55-64): gray = (u8)(0.114*B + 0.587*G + 0.299*R), fp64, summed left to right,
// truncated.  Four pixels per thread: three dword loads, one dword store.  npx = pixels of all frames together.
__device__ inline uint32_t bgr_px_to_gray(uint32_t b, uint32_t g, uint32_t r) {
  return (uint32_t)(uint8_t)((0.114 * (double)b + 0.587 * (double)g) + 0.299 * (double)r);
}
__device__ inline uint32_t bgr4_to_gray4(uint32_t w0, uint32_t w1, uint32_t w2) {  // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
  const uint32_t g0 = bgr_px_to_gray(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
  const uint32_t g1 = bgr_px_to_gray(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
  const uint32_t g2 = bgr_px_to_gray((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
  const uint32_t g3 = bgr_px_to_gray((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
  return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}
// grid-stride over groups of 16 pixels: three 16-byte loads, one 16-byte store (buffers 16-byte aligned: VEC), or
// groups of 4 pixels with dwords (4-byte aligned).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <bool VEC>
__global__ __launch_bounds__(256) void k_bgr_to_gray(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, size_t npx) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t done;
  if (VEC) {
    const size_t ng = npx >> 4;
    for (size_t q = t0; q < ng; q += stride) {
      const u32x4_t* s = reinterpret_cast<const u32x4_t*>(bgr) + 3 * q;
      const u32x4_t a = __builtin_nontemporal_load(s), b = __builtin_nontemporal_load(s + 1), c = __builtin_nontemporal_load(s + 2);
      u32x4_t o;
      o.x = bgr4_to_gray4(a.x, a.y, a.z);
      o.y = bgr4_to_gray4(a.w, b.x, b.y);
      o.z = bgr4_to_gray4(b.z, b.w, c.x);
      o.w = bgr4_to_gray4(c.y, c.z, c.w);
      __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(gray) + q);
    }
    done = ng << 4;
  } else {
    const size_t nq = npx >> 2;
    for (size_t q = t0; q < nq; q += stride) {
      const uint32_t* s = reinterpret_cast<const uint32_t*>(bgr) + 3 * q;
      reinterpret_cast<uint32_t*>(gray)[q] = bgr4_to_gray4(s[0], s[1], s[2]);
    }
    done = nq << 2;
  }
  if (blockIdx.x == 0 && done + threadIdx.x < npx) {  // tail (< 16 pixels)
    const size_t i = done + threadIdx.x;
    gray[i] = (uint8_t)bgr_px_to_gray(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2]);
  }
}

// DepthToPointCloud::operator() (helper.cpp:30-53): z = (float)(raw * 0.001); x = (float)(((double)u / W - 0.5) * z * XtoZ),
// y = (float)((0.5 - (double)v / H) * z * YtoZ).  raw = 0 gives z = 0 and signed zeros in x, y, as in the reference.
// grid (ceil(W*H/256), n_frames); one pixel per thread, 12 contiguous bytes per lane, streamed past the caches
// (write-dominated: 2 B in, 12 B out per pixel).  A variant with LDS-tabulated factors and 16-byte stores measured
// slower (3.2 vs 3.7 TB/s): the kernel is bound by HBM writes, not by the two fp64 divisions.
__global__ __launch_bounds__(256) void k_depth_to_cloud(const uint16_t* __restrict__ depth, float* __restrict__ cloud, int W, int H,
                                                        double xtoz, double ytoz) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= W * H) return;
  const size_t f = blockIdx.y;
  const int y = idx / W, x = idx - y * W;
  const float z = (float)((double)__builtin_nontemporal_load(depth + f * W * H + idx) * 0.001);
  float* o = cloud + (f * W * H + idx) * 3;
  float ox = 0, oy = 0, oz = 0;
  if (!isnan(z)) {
    ox = (float)((((double)x / (double)W - 0.5) * (double)z) * xtoz);
    oy = (float)(((0.5 - (double)y / (double)H) * (double)z) * ytoz);
    oz = z;
  }
  __builtin_nontemporal_store(ox, o);
  __builtin_nontemporal_store(oy, o + 1);
  __builtin_nontemporal_store(oz, o + 2);
}
