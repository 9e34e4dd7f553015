// Do the host worker pools (scheduler-plugins_amd/host/parallel.hpp) actually run in parallel on this box?  40 back-to-back jobs of ~18 ms
// of single-thread work (20 000 rows).
//   g++ -O2 -std=c++17 -pthread -I scheduler-plugins_amd/host tools/micro/pool_scaling.cc -o /tmp/pool_scaling && /tmp/pool_scaling
// Round 4: the GPU box (256 hardware threads, 16 workers): 1.5-1.8 ms per job; the 8-vCPU build container: 18 ms — its kernel wakes the
// parked workers on the caller's CPU and jobs of milliseconds end before the load balancer moves them (with every worker pinned to a CPU
// of its own — tried, not kept: 3.5 ms there, 1.3 ms on the GPU box).  Host-side timings taken in the build container are single-thread timings.
#include "parallel.hpp"
#include <chrono>
#include <cstdio>
#include <cmath>
int main() {
  const int64_t n = 20000;
  std::vector<double> out(n);
  for (int rep = 0; rep < 40; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    spx_host::parallel_rows(n, [&](int64_t b, int64_t e) {
      for (int64_t i = b; i < e; ++i) { double s = 0; for (int k = 1; k < 600; ++k) s += std::sqrt(double(i + k)); out[i] = s; }
    }, 256);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep % 4 == 0) std::printf("rep %d: %.2f ms\n", rep, ms);
  }
}
