// parallel.hpp — chunked parallel-for over independent rows for the host flatteners (plain std::thread; the
// flatteners are called from cgo / ctypes threads and own no thread pool).  Small inputs run inline.
#pragma once

#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace spx_host {

template <typename Fn>
inline void parallel_rows(int64_t n, Fn&& fn, int64_t min_rows_per_thread = 8192, unsigned max_threads = 16) {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  const unsigned want = static_cast<unsigned>(std::min<int64_t>(std::min(hw, max_threads), n / min_rows_per_thread));
  if (want <= 1) {
    fn(static_cast<int64_t>(0), n);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(want);
  const int64_t chunk = (n + want - 1) / want;
  for (unsigned t = 0; t < want; ++t) {
    const int64_t b = static_cast<int64_t>(t) * chunk, e = std::min(n, b + chunk);
    if (b >= e) break;
    th.emplace_back([&fn, b, e] { fn(b, e); });
  }
  for (auto& x : th) x.join();
}

}  // namespace spx_host
