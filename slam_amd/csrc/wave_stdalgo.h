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
  if (lane == 0) {
    const int mid = first + (last - first) / 2;
    stdalgo::move_median_to_first(a, first, first + 1, mid, last - 1, comp);
  }
  wave_lds_sync();
  const uint32_t pv = a[first] >> SHIFT;
  // left-stops ascending over [first+1, last)
  int nl = 0;
  for (int i0 = first + 1; i0 < last; i0 += 64) {
    const int i = i0 + lane;
    const bool stop = i < last && !((a[i] >> SHIFT) > pv);  // !comp(a[i], pivot)
    const unsigned long long m = __ballot(stop);
    if (stop) li[nl + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
    nl += __popcll(m);
  }
  // right-stops descending over [first, last)
  int nr = 0;
  for (int j0 = last - 1; j0 >= first; j0 -= 64) {
    const int j = j0 - lane;
    const bool stop = j >= first && !(pv > (a[j] >> SHIFT));  // !comp(pivot, a[j])
    const unsigned long long m = __ballot(stop);
    if (stop) ri[nr + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)j;
    nr += __popcll(m);
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

// std::sort(a, a+n, comp).  Scratch: li, ri (u16[n]); leaf_s, leaf_e (u16[n]); tmp (u32[n]); stack (int[128]).
template <int SHIFT>
__device__ inline void wave_sort(uint32_t* a, int n, uint16_t* li, uint16_t* ri, uint16_t* leaf_s, uint16_t* leaf_e,
                                 uint32_t* tmp, int* stack) {
  const int lane = threadIdx.x & 63;
  KeyGreater<SHIFT> comp;
  if (n == 0) return;
  auto mark_leaf = [&](int s, int e) {
    for (int i = s + lane; i < e; i += 64) { leaf_s[i] = (uint16_t)s; leaf_e[i] = (uint16_t)e; }
  };
  // __introsort_loop with an explicit stack (stack accesses are wave-uniform; lane 0 writes, all lanes read)
  int sp = 0;
  if (lane == 0) { stack[0] = 0; stack[1] = n; stack[2] = stdalgo::lg_(n) * 2; }
  sp = 3;
  wave_lds_sync();
  while (sp > 0) {
    int depth_limit = stack[sp - 1], last = stack[sp - 2], first = stack[sp - 3];
    sp -= 3;
    bool heap_done = false;
    while (last - first > 16) {
      if (depth_limit == 0) {
        if (lane == 0) {
          stdalgo::heap_select(a, first, last, last, comp);
          stdalgo::sort_heap_(a, first, last, comp);
        }
        wave_lds_sync();
        heap_done = true;
        break;
      }
      --depth_limit;
      const int cut = wave_partition_pivot<SHIFT>(a, first, last, li, ri);
      if (lane == 0) { stack[sp] = cut; stack[sp + 1] = last; stack[sp + 2] = depth_limit; }
      sp += 3;
      wave_lds_sync();
      last = cut;
    }
    (void)heap_done;
    mark_leaf(first, last);  // a <=16 leaf, or a heap-sorted range (already sorted: the stable sort is the identity)
  }
  wave_lds_sync();
  // __final_insertion_sort == stable sort of every leaf
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    if (i < n) {
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
  }
  wave_lds_sync();
  for (int i = lane; i < n; i += 64) a[i] = tmp[i];
  wave_lds_sync();
}

}  // namespace wavealgo
