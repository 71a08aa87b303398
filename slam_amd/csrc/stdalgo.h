// stdalgo.h -- index-based re-implementation of libstdc++'s (g++ 11.4)
// std::nth_element and std::sort, usable from one GPU lane.
//
// Why: the reference's keypoint sets and inlier sets depend on the exact
// permutation these two algorithms leave behind:
//   * GridKeyPointFilter::Filter   /root/reference/rgbdvo/pyramid.cpp:42-45
//     keeps the first 25 elements *in post-nth_element order*;
//   * computeMaximallyConsistentClique /root/reference/rgbdvo/motion_estimate.cpp:526
//     walks matches in std::sort (unstable) order.
// The algorithm text mirrored here is /usr/include/c++/11/bits/stl_algo.h
// (__introselect :1964, __introsort_loop :1929, __unguarded_partition_pivot :1900,
// __move_median_to_first :79, __heap_select :1642, _S_threshold = 16 :1855) and
// bits/stl_heap.h (__push_heap :134, __adjust_heap :223, __pop_heap :253,
// __make_heap :339, __sort_heap :418).  tests/native/test_stdalgo.cpp checks it
// against the real std:: functions on the host.
#pragma once

#ifdef __HIPCC__
#define RVO_HD __host__ __device__ inline
#else
#define RVO_HD inline
#endif

namespace stdalgo {

template <class T>
RVO_HD void swap_(T& a, T& b) {
  T t = a;
  a = b;
  b = t;
}

RVO_HD int lg_(int n) {  // std::__lg
  int k = 0;
  while (n > 1) {
    n >>= 1;
    ++k;
  }
  return k;
}

template <class T, class C>
RVO_HD void move_median_to_first(T* a, int result, int x, int y, int z, C comp) {
  if (comp(a[x], a[y])) {
    if (comp(a[y], a[z]))
      swap_(a[result], a[y]);
    else if (comp(a[x], a[z]))
      swap_(a[result], a[z]);
    else
      swap_(a[result], a[x]);
  } else if (comp(a[x], a[z]))
    swap_(a[result], a[x]);
  else if (comp(a[y], a[z]))
    swap_(a[result], a[z]);
  else
    swap_(a[result], a[y]);
}

template <class T, class C>
RVO_HD int unguarded_partition(T* a, int first, int last, int pivot, C comp) {
  while (true) {
    while (comp(a[first], a[pivot])) ++first;
    --last;
    while (comp(a[pivot], a[last])) --last;
    if (!(first < last)) return first;
    swap_(a[first], a[last]);
    ++first;
  }
}

template <class T, class C>
RVO_HD int unguarded_partition_pivot(T* a, int first, int last, C comp) {
  int mid = first + (last - first) / 2;
  move_median_to_first(a, first, first + 1, mid, last - 1, comp);
  return unguarded_partition(a, first + 1, last, first, comp);
}

// heap helpers operate on base pointer b (= first) with 0-based indices
template <class T, class C>
RVO_HD void push_heap_(T* b, int hole, int top, T value, C comp) {
  int parent = (hole - 1) / 2;
  while (hole > top && comp(b[parent], value)) {
    b[hole] = b[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  b[hole] = value;
}

template <class T, class C>
RVO_HD void adjust_heap(T* b, int hole, int len, T value, C comp) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (comp(b[second], b[second - 1])) second--;
    b[hole] = b[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    b[hole] = b[second - 1];
    hole = second - 1;
  }
  push_heap_(b, hole, top, value, comp);
}

// __pop_heap(first, last, result): heap is b[0..len), result is an index relative to b
template <class T, class C>
RVO_HD void pop_heap_(T* b, int len, int result, C comp) {
  T value = b[result];
  b[result] = b[0];
  adjust_heap(b, 0, len, value, comp);
}

template <class T, class C>
RVO_HD void make_heap_(T* b, int len, C comp) {
  if (len < 2) return;
  int parent = (len - 2) / 2;
  while (true) {
    T value = b[parent];
    adjust_heap(b, parent, len, value, comp);
    if (parent == 0) return;
    parent--;
  }
}

template <class T, class C>
RVO_HD void heap_select(T* a, int first, int middle, int last, C comp) {
  T* b = a + first;
  make_heap_(b, middle - first, comp);
  for (int i = middle; i < last; ++i)
    if (comp(a[i], a[first])) pop_heap_(b, middle - first, i - first, comp);
}

template <class T, class C>
RVO_HD void sort_heap_(T* a, int first, int last, C comp) {
  T* b = a + first;
  while (last - first > 1) {
    --last;
    pop_heap_(b, last - first, last - first, comp);
  }
}

template <class T, class C>
RVO_HD void unguarded_linear_insert(T* a, int last, C comp) {
  T val = a[last];
  int next = last - 1;
  while (comp(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <class T, class C>
RVO_HD void insertion_sort(T* a, int first, int last, C comp) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (comp(a[i], a[first])) {
      T val = a[i];
      for (int k = i; k > first; --k) a[k] = a[k - 1];  // move_backward(first, i, i+1)
      a[first] = val;
    } else {
      unguarded_linear_insert(a, i, comp);
    }
  }
}

// std::nth_element(a, a+nth, a+n, comp)
template <class T, class C>
RVO_HD void nth_element(T* a, int nth, int n, C comp) {
  int first = 0, last = n;
  if (first == last || nth == last) return;
  int depth_limit = lg_(n) * 2;
  while (last - first > 3) {
    if (depth_limit == 0) {
      heap_select(a, first, nth + 1, last, comp);
      swap_(a[first], a[nth]);
      return;
    }
    --depth_limit;
    int cut = unguarded_partition_pivot(a, first, last, comp);
    if (cut <= nth)
      first = cut;
    else
      last = cut;
  }
  insertion_sort(a, first, last, comp);
}

// std::sort(a, a+n, comp).  __introsort_loop recurses on [cut,last) and loops on
// [first,cut); the sub-ranges are disjoint, so an explicit stack gives the same
// permutation.  `stack` needs 2*lg(n)+2 int3-like slots (passed in by the caller
// so the GPU side can keep it in registers/LDS).
template <class T, class C>
RVO_HD void sort(T* a, int n, C comp, int* stack /* >= 3*(2*lg(n)+2) ints */) {
  if (n == 0) return;
  int sp = 0;
  stack[sp++] = 0;
  stack[sp++] = n;
  stack[sp++] = lg_(n) * 2;
  while (sp > 0) {
    int depth_limit = stack[--sp];
    int last = stack[--sp];
    int first = stack[--sp];
    while (last - first > 16) {
      if (depth_limit == 0) {
        heap_select(a, first, last, last, comp);  // __partial_sort(first,last,last)
        sort_heap_(a, first, last, comp);
        break;
      }
      --depth_limit;
      int cut = unguarded_partition_pivot(a, first, last, comp);
      stack[sp++] = cut;
      stack[sp++] = last;
      stack[sp++] = depth_limit;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    insertion_sort(a, 0, 16, comp);
    for (int i = 16; i != n; ++i) unguarded_linear_insert(a, i, comp);
  } else {
    insertion_sort(a, 0, n, comp);
  }
}

}  // namespace stdalgo
