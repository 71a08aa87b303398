// wave_stdalgo.h -- the libstdc++ 11.4 std::nth_element / std::sort permutations (see stdalgo.h for why they
// matter) computed by ONE 64-lane wavefront on an LDS array instead of one lane.
//
// Elements are uint32 keys; the comparator is `(a >> SHIFT) > (b >> SHIFT)` (descending by the high bits: FAST
// score for the grid filter, compatibility degree for the clique sort).  Everything is called by all 64 lanes of a
// wave with wave-uniform arguments.
//
// Exactness argument for the partition (stl_algo.h:1874-1894 __unguarded_partition(first+1, last, pivot=first)):
// the left pointer stops on elements with !comp(x, pivot), the right pointer on elements with !comp(pivot, x).  Let
// Lo be the ascending list of left-stops in [first+1, last) and Ro the descending list of right-stops in
// [first, last), both taken on the array BEFORE the partition.  While Lo[k] < Ro[k] the k-th iteration swaps exactly
// a[Lo[k]] and a[Ro[k]]: both pointers only ever walk over positions that no earlier swap has touched.  With K = the
// number of such k, the function returns min(Lo[K], Ro[K-1]) (the left pointer runs into either the next original
// left-stop or the element the last swap parked at Ro[K-1]).  The lists are built with ballots, the K swaps are
// done in parallel.  After __introsort_loop, __final_insertion_sort (stl_algo.h:1861-1867) never moves an element
// across the boundary of a <=16-element leaf or of a heap-sorted range (everything left of a cut is <= everything
// right of it), and insertion sort is stable, so it equals a stable sort of each leaf, done here by rank counting.
// tests/test_gpu_stdalgo.py checks both entry points against the real std:: functions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stdalgo.h"

namespace wavealgo {

template <int SHIFT>
struct KeyGreater {
  __device__ bool operator()(uint32_t a, uint32_t b) const { return (a >> SHIFT) > (b >> SHIFT); }
};

__device__ inline void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving accesses across.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// __unguarded_partition_pivot(first, last): returns the cut.  li/ri: scratch u16 arrays of >= (last-first) entries.
template <int SHIFT>
__device__ inline int wave_partition_pivot(uint32_t* a, int first, int last, uint16_t* li, uint16_t* ri) {
  const int lane = threadIdx.x & 63;
  KeyGreater<SHIFT> comp;
  // __move_median_to_first(first, first+1, mid, last-1) (stl_algo.h:79-98): every lane reads the three candidates and
  // a[first] at once and takes the decisions of that function; lane 0 performs its swap.  (The indices are distinct:
  // callers partition ranges of more than three elements.)
  const int mid = first + (last - first) / 2;
  const uint32_t vx = a[first + 1], vy = a[mid], vz = a[last - 1], vf = a[first];
  int mi;
  if (comp(vx, vy)) mi = comp(vy, vz) ? mid : comp(vx, vz) ? last - 1 : first + 1;
  else mi = comp(vx, vz) ? first + 1 : comp(vy, vz) ? last - 1 : mid;
  const uint32_t vm = mi == mid ? vy : mi == last - 1 ? vz : vx;
  wave_lds_sync();  // every lane has its four values before two of them are exchanged
  if (lane == 0) { a[first] = vm; a[mi] = vf; }
  wave_lds_sync();
  const uint32_t pv = vm >> SHIFT;
  // left-stops ascending over [first+1, last) and right-stops descending over [first, last), 64 positions of each per
  // trip (their reads are independent and issued together)
  int nl = 0, nr = 0;
  for (int o = 0; o < last - first; o += 64) {
    const int i = first + 1 + o + lane, j = last - 1 - o - lane;
    const uint32_t ai = i < last ? a[i] : 0u, aj = j >= first ? a[j] : 0u;
    const bool stopl = i < last && !((ai >> SHIFT) > pv);   // !comp(a[i], pivot)
    const bool stopr = j >= first && !(pv > (aj >> SHIFT));  // !comp(pivot, a[j])
    const unsigned long long ml = __ballot(stopl), mr = __ballot(stopr);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (stopl) li[nl + __popcll(ml & below)] = (uint16_t)i;
    if (stopr) ri[nr + __popcll(mr & below)] = (uint16_t)j;
    nl += __popcll(ml);
    nr += __popcll(mr);
  }
  wave_lds_sync();
  // K = #k with Lo[k] < Ro[k]  (Lo ascending, Ro descending: the predicate is monotone)
  const int nmin = min(nl, nr);
  int K = 0;
  for (int k0 = 0; k0 < nmin; k0 += 64) {
    const int k = k0 + lane;
    const bool sw = k < nmin && li[k] < ri[k];
    const unsigned long long m = __ballot(sw);
    if (sw) {
      const int x = li[k], y = ri[k];
      const uint32_t t = a[x];
      a[x] = a[y];
      a[y] = t;
    }
    const int c = __popcll(m);
    K += c;
    if (c < 64) break;
  }
  wave_lds_sync();
  int cut = 0x7fffffff;
  if (K < nl) cut = li[K];
  if (K > 0) cut = min(cut, (int)ri[K - 1]);
  return cut;
}

// std::nth_element(a, a+nth, a+n, comp)
template <int SHIFT>
__device__ inline void wave_nth_element(uint32_t* a, int nth, int n, uint16_t* li, uint16_t* ri) {
  const int lane = threadIdx.x & 63;
  KeyGreater<SHIFT> comp;
  int first = 0, last = n;
  if (first == last || nth == last) return;
  int depth_limit = stdalgo::lg_(n) * 2;
  while (last - first > 3) {
    if (depth_limit == 0) {
      if (lane == 0) {
        stdalgo::heap_select(a, first, nth + 1, last, comp);
        stdalgo::swap_(a[first], a[nth]);
      }
      wave_lds_sync();
      return;
    }
    --depth_limit;
    const int cut = wave_partition_pivot<SHIFT>(a, first, last, li, ri);
    if (cut <= nth) first = cut; else last = cut;
  }
  if (lane == 0) stdalgo::insertion_sort(a, first, last, comp);
  wave_lds_sync();
}

// std::sort(a, a+n, comp) by all waves of the workgroup (blockDim.x = 64 * NW, every thread calls).
// __introsort_loop (stl_algo.h:1940-1958) recurses on [cut, last) and loops on [first, cut): the two ranges are
// independent, so the recursion becomes a work queue: a wave partitions its range, publishes the right part in a
// shared LDS queue (or, when the queue is full, on its own small stack, whose depth is bounded by the depth limit
// 2*lg(n)) and continues with the left part.  The order in which ranges are processed does not change the result.
// Scratch: li, ri, leaf_s, leaf_e (u16[n] each; the partition of [first,last) only uses li/ri[first..last)), tmp
// (u32[n]), ctl (int[4 + SORT_QCAP*2 + NW*SORT_STACK*2]).
// Queue entries are used once (no wrap-around): a sort of n keys publishes at most n / 17 ranges, so 128 entries cover
// n <= 2176 (the default configuration's 1600 matches) without ever using the private stacks; beyond that a wave keeps
// the ranges the queue has no room for on its own stack and works through them itself (same result, less parallel).
#ifndef SORT_QCAP
#define SORT_QCAP 128
#endif
#define SORT_STACK 32
#ifndef SORT_LOCAL
#define SORT_LOCAL 64            // right parts up to this size are not published (128: the sort of 920 keys 0.065 ms against 0.057)
#endif
#define SORT_LOCAL_FLAG (1 << 30)  // in a private stack entry's depth word: not counted in `pending`
__device__ inline int sort_ctl_ints(int nwaves) { return 4 + SORT_QCAP * 2 + nwaves * SORT_STACK * 2; }

template <int SHIFT>
__device__ inline void block_sort(uint32_t* a, int n, uint16_t* li, uint16_t* ri, uint16_t* leaf_s, uint16_t* leaf_e,
                                  uint32_t* tmp, int* ctl) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthreads = blockDim.x;
  KeyGreater<SHIFT> comp;
  if (n == 0) return;
  int* head = ctl;          // next queue entry to claim
  int* tail = ctl + 1;      // reserved queue entries
  int* pending = ctl + 2;   // ranges published or being processed
  uint32_t* q_seg = reinterpret_cast<uint32_t*>(ctl + 4);  // first | last << 16; 0 = not yet written
  int* q_depth = ctl + 4 + SORT_QCAP;
  int* stk = ctl + 4 + SORT_QCAP * 2 + wid * SORT_STACK * 2;
  for (int i = tid; i < SORT_QCAP; i += nthreads) q_seg[i] = 0;
  __syncthreads();
  if (tid == 0) {
    *head = 0; *tail = 1; *pending = 1;
    q_depth[0] = stdalgo::lg_(n) * 2;
    q_seg[0] = (uint32_t)n << 16;
  }
  __syncthreads();
  auto mark_leaf = [&](int s, int e) {
    for (int i = s + lane; i < e; i += 64) { leaf_s[i] = (uint16_t)s; leaf_e[i] = (uint16_t)e; }
  };
  int sp = 0;  // entries on this wave's private stack (wave-uniform)
  for (;;) {
    int first, last, depth_limit;
    bool counted = true;  // the range is one of `pending` (published, or kept back because the queue was full)
    if (sp > 0) {
      sp--;
      const uint32_t seg = (uint32_t)stk[2 * sp];
      first = seg & 0xFFFF; last = seg >> 16; depth_limit = stk[2 * sp + 1];
      counted = (depth_limit & SORT_LOCAL_FLAG) == 0;
      depth_limit &= ~SORT_LOCAL_FLAG;
    } else {
      int h = -1;
      uint32_t seg = 0;
      if (lane == 0) {
        for (;;) {
          const int hd = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int tl = min(__hip_atomic_load(tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), SORT_QCAP);
          if (hd < tl) {
            if (atomicCAS(head, hd, hd + 1) == hd) { h = hd; break; }
          } else if (__hip_atomic_load(pending, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
            break;
          } else {
            __builtin_amdgcn_s_sleep(2);
          }
        }
        if (h >= 0)
          while ((seg = __hip_atomic_load(&q_seg[h], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) __builtin_amdgcn_s_sleep(1);
      }
      h = __builtin_amdgcn_readfirstlane(h);
      if (h < 0) break;
      seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg);
      first = seg & 0xFFFF; last = seg >> 16; depth_limit = q_depth[h];
    }
    while (last - first > 16) {
      if (depth_limit == 0) {
        if (lane == 0) {
          stdalgo::heap_select(a, first, last, last, comp);
          stdalgo::sort_heap_(a, first, last, comp);
        }
        wave_lds_sync();
        break;
      }
      --depth_limit;
      const int cut = wave_partition_pivot<SHIFT>(a, first, last, li + first, ri + first);
      if (last - cut > 16 && last - cut <= SORT_LOCAL && sp < SORT_STACK) {
        // a small right part stays with this wave (no queue traffic: four LDS atomics per range otherwise); nothing it
        // spawns is ever published, so the other waves need not know about it
        if (lane == 0) { stk[2 * sp] = (int)((uint32_t)cut | ((uint32_t)last << 16)); stk[2 * sp + 1] = depth_limit | SORT_LOCAL_FLAG; }
        sp++;
        wave_lds_sync();
      } else if (last - cut > 16) {
        int local = 0;
        if (lane == 0) {
          const uint32_t seg = (uint32_t)cut | ((uint32_t)last << 16);
          __hip_atomic_fetch_add(pending, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int idx = __hip_atomic_fetch_add(tail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (idx < SORT_QCAP) {
            q_depth[idx] = depth_limit;
            __hip_atomic_store(&q_seg[idx], seg, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // publishes a[] too
          } else {
            stk[2 * sp] = (int)seg; stk[2 * sp + 1] = depth_limit;
            local = 1;
          }
        }
        sp += __builtin_amdgcn_readfirstlane(local);
        wave_lds_sync();
      } else {
        mark_leaf(cut, last);  // std recurses into it and returns at once
      }
      last = cut;
    }
    mark_leaf(first, last);  // a <=16 leaf, or a heap-sorted range (already sorted: the stable sort is the identity)
    wave_lds_sync();
    if (lane == 0 && counted) __hip_atomic_fetch_sub(pending, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  // __final_insertion_sort == stable sort of every leaf
  for (int i = tid; i < n; i += nthreads) {
    const int s = leaf_s[i], e = leaf_e[i];
    const uint32_t me = a[i];
    const uint32_t mk = me >> SHIFT;
    int rank = s;
    for (int j = s; j < e; j++) {
      const uint32_t k = a[j] >> SHIFT;
      rank += (k > mk) || (k == mk && j < i);
    }
    tmp[rank] = me;
  }
  __syncthreads();
  for (int i = tid; i < n; i += nthreads) a[i] = tmp[i];
  __syncthreads();
}

}  // namespace wavealgo
