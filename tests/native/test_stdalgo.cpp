// Differential test: slam_amd/csrc/stdalgo.h vs the real libstdc++ algorithms,
// on the element types and comparators the reference uses
// (pyramid.cpp:42-45 KeyPoint by score desc; motion_estimate.cpp:396-399,526
// FeatureMatch by compatibility_degree desc).  Exit code 0 = identical permutations.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "../../slam_amd/csrc/stdalgo.h"

struct KeyPoint { float u, v, score; };
struct Match { int degree; int id; double payload[4]; };

int main() {
  std::mt19937 rng(12345);
  long checked = 0;
  // ---- nth_element on packed u32 keys vs KeyPoint structs
  for (int trial = 0; trial < 6000; trial++) {
    int n = 26 + (int)(rng() % (trial % 50 == 0 ? 6400 : 400));
    int nth = 25;
    if (trial % 7 == 0) nth = 1 + (int)(rng() % (n - 1));
    int score_range = (trial % 3 == 0) ? 4 : (trial % 3 == 1 ? 40 : 250);
    std::vector<KeyPoint> ref(n);
    std::vector<uint32_t> mine(n);
    for (int i = 0; i < n; i++) {
      int s = 5 + (int)(rng() % score_range);
      ref[i] = {(float)(i % 640), (float)(i / 640), (float)s};
      mine[i] = ((uint32_t)s << 22) | (uint32_t)i;  // payload = original index
    }
    if (trial % 11 == 0) std::sort(mine.begin(), mine.end()), std::sort(ref.begin(), ref.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.score < b.score || (a.score == b.score && a.v * 640 + a.u < b.v * 640 + b.u); });
    std::nth_element(ref.begin(), ref.begin() + nth, ref.end(), [](const KeyPoint& l, const KeyPoint& r) { return l.score > r.score; });
    stdalgo::nth_element(mine.data(), nth, n, [](uint32_t l, uint32_t r) { return (l >> 22) > (r >> 22); });
    for (int i = 0; i < n; i++) {
      uint32_t idx = mine[i] & 0x3FFFFF;
      if ((float)(idx % 640) != ref[i].u || (float)(idx / 640) != ref[i].v || (float)(mine[i] >> 22) != ref[i].score) {
        printf("nth_element mismatch trial %d n %d nth %d at %d\n", trial, n, nth, i);
        return 1;
      }
    }
    checked++;
  }
  // ---- sort on packed (degree<<16|id) vs Match structs
  for (int trial = 0; trial < 6000; trial++) {
    int n = 1 + (int)(rng() % (trial % 40 == 0 ? 6000 : 900));
    int deg_range = (trial % 4 == 0) ? 3 : (trial % 4 == 1 ? 30 : n);
    std::vector<Match> ref(n);
    std::vector<uint32_t> mine(n);
    for (int i = 0; i < n; i++) {
      int d = (int)(rng() % deg_range);
      if (trial % 13 == 0) d = i % deg_range;          // sawtooth
      if (trial % 17 == 0) d = (n - i) % deg_range;    // descending-ish (heapsort fallback pressure)
      ref[i].degree = d; ref[i].id = i;
      mine[i] = ((uint32_t)d << 16) | (uint32_t)i;
    }
    std::sort(ref.begin(), ref.end(), [](const Match& a, const Match& b) { return a.degree > b.degree; });
    int stack[128];
    stdalgo::sort(mine.data(), n, [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); }, stack);
    for (int i = 0; i < n; i++)
      if ((int)(mine[i] & 0xFFFF) != ref[i].id) { printf("sort mismatch trial %d n %d at %d\n", trial, n, i); return 1; }
    checked++;
  }
  // adversarial: organ-pipe / median-of-3 killers to force the heap fallbacks
  for (int n : {64, 100, 257, 1000, 4096}) {
    std::vector<Match> ref(n);
    std::vector<uint32_t> mine(n);
    for (int i = 0; i < n; i++) {
      int d = (i < n / 2) ? i : n - i;
      if (i % 2) d = n - d;
      ref[i].degree = d; ref[i].id = i;
      mine[i] = ((uint32_t)d << 16) | (uint32_t)i;
    }
    std::vector<Match> r2 = ref;
    std::vector<uint32_t> m2 = mine;
    std::sort(ref.begin(), ref.end(), [](const Match& a, const Match& b) { return a.degree > b.degree; });
    int stack[128];
    stdalgo::sort(mine.data(), n, [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); }, stack);
    for (int i = 0; i < n; i++) if ((int)(mine[i] & 0xFFFF) != ref[i].id) { printf("adv sort mismatch n %d\n", n); return 1; }
    std::nth_element(r2.begin(), r2.begin() + n / 3, r2.end(), [](const Match& a, const Match& b) { return a.degree > b.degree; });
    stdalgo::nth_element(m2.data(), n / 3, n, [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); });
    for (int i = 0; i < n; i++) if ((int)(m2[i] & 0xFFFF) != r2[i].id) { printf("adv nth mismatch n %d\n", n); return 1; }
    checked += 2;
  }
  printf("stdalgo OK: %ld cases identical to libstdc++\n", checked);
  return 0;
}
