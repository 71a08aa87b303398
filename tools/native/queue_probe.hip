// One-off probe: which HIP streams of a process share a hardware queue?  Launches a ~2 ms spin kernel (1 workgroup) on
// stream i and on stream j at the same time for every pair and reports the pairs that take twice as long (serialised).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 999) *sink = 1;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8;
  std::vector<hipStream_t> st(n);
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  printf("priority range: least %d greatest %d\n", lo, hi);
  // argv[2]: one letter per stream (h = greatest priority, l = least, anything else = normal); "half" = second half high
  const char* pr = argc > 2 ? argv[2] : "";
  const bool half = std::string(pr) == "half";
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 1000, nullptr);  // the null stream takes its queue first, as under torch
  for (int i = 0; i < n; i++) {
    const char ch = half ? (i < n / 2 ? 'n' : 'h') : (i < (int)strlen(pr) ? pr[i] : 'n');
    hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, ch == 'h' ? hi : ch == 'l' ? lo : 0);
  }
  const long long ticks = 200000;  // 2 ms at 100 MHz
  for (auto& s : st) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, nullptr); }
  hipDeviceSynchronize();
  auto run = [&](std::vector<int> ids) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i : ids) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], ticks, nullptr);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  printf("one stream: %.2f ms\n", run({0}));
  for (int i = 0; i < n; i++) {
    printf("stream %d:", i);
    for (int j = 0; j < n; j++) printf(" %s", i == j ? "-" : (run({i, j}) > 3.0 ? "S" : "."));
    printf("\n");
  }
  std::vector<int> all;
  for (int i = 0; i < n; i++) all.push_back(i);
  printf("all %d streams together: %.2f ms\n", n, run(all));
  return 0;
}
