// harness.cc — drives libspx.so through the C++ host mirror with the call pattern upstream kube-scheduler uses:
// one pod at a time, Filter/Score fanned out over the nodes by 16 concurrent workers (the Parallelizer the
// reference's benchmarks copy, pkg/trimaran/targetloadpacking/targetloadpacking_test.go:386-405), then
// NormalizeScore once per plugin.  Checks that concurrent readers see exactly what a serial pass sees, that
// Allocatable's raw scores are negative (Least) and its normalised list spans [0,100], that TLP's
// NormalizeScore is a no-op.  Needs a GPU; exits 0 and prints "harness ok" on success.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../scheduler-plugins_amd/host/plugins.hpp"

using namespace spx::host;

int main() {
  const int64_t N = 777, P = 40;
  Engine e(0);
  e.n_nodes = N;
  // --- a small deterministic snapshot, straight in SoA form
  std::vector<int64_t> alloc(2 * N), cap(N), missing(N), acpu(N), amem(N), pod_milli(P), rcpu(P), rmem(P);
  std::vector<double> util(N), cavg(N), cstd(N), mavg(N), mstd(N);
  std::vector<uint8_t> valid(N), flags(N);
  for (int64_t i = 0; i < N; ++i) {
    cap[i] = 8000 + 1000 * (i % 57);
    acpu[i] = cap[i] - 500;
    amem[i] = (32ll + i % 100) << 30;
    alloc[i] = amem[i];       // resource order of the default params: memory, cpu
    alloc[N + i] = acpu[i];
    util[i] = (i * 37 % 1000) / 10.0 + 0.123;
    missing[i] = (i % 9 == 0) ? 750 : 0;
    valid[i] = i % 50 != 7;
    cavg[i] = util[i];
    cstd[i] = (i % 30) * 0.7;
    mavg[i] = (i * 13 % 100) + 0.5;
    mstd[i] = (i % 11) * 1.1;
    flags[i] = valid[i] ? 7 : 0;
  }
  for (int64_t p = 0; p < P; ++p) {
    pod_milli[p] = 100 + 371 * p;
    rcpu[p] = pod_milli[p];
    rmem[p] = (1ll + p) << 28;
  }
  spx_alloc_nodes_soa an{N, 2, alloc.data()};
  e.check(spx_upload_alloc_nodes(e.raw(), &an));
  spx_trimaran_nodes_soa tn{N, cap.data(), util.data(), missing.data(), valid.data(), acpu.data(), amem.data(),
                            cavg.data(), cstd.data(), mavg.data(), mstd.data(), flags.data()};
  e.check(spx_upload_trimaran_nodes(e.raw(), &tn));
  spx_trimaran_pods_soa tp{P, pod_milli.data(), rcpu.data(), rmem.data()};
  e.check(spx_upload_trimaran_pods(e.raw(), &tp));
  e.Eval((1u << SPX_PLUGIN_ALLOCATABLE) | (1u << SPX_PLUGIN_TLP) | (1u << SPX_PLUGIN_LVRB), 0, P);

  Allocatable alloc_pl;
  TargetLoadPacking tlp;
  LoadVariationRiskBalancing lvrb;
  constexpr int parallelism = 16;
  for (int64_t pod = 0; pod < P; ++pod) {
    CycleState serial(e, pod), state(e, pod);
    std::vector<int64_t> want_t(N), want_l(N), want_a(N), got_t(N), got_l(N), got_a(N);
    for (int32_t n = 0; n < N; ++n) {
      want_t[n] = tlp.Score(serial, n).first;
      want_l[n] = lvrb.Score(serial, n).first;
      want_a[n] = alloc_pl.Score(serial, n).first;
    }
    std::atomic<int> next{0};
    const int chunk = std::max<int>(1, static_cast<int>(N) / parallelism / 4);
    std::vector<std::thread> th;
    for (int w = 0; w < parallelism; ++w)
      th.emplace_back([&] {
        for (;;) {
          const int b = next.fetch_add(chunk);
          if (b >= N) return;
          for (int32_t n = b; n < std::min<int64_t>(N, b + chunk); ++n) {
            got_t[n] = tlp.Score(state, n).first;
            got_l[n] = lvrb.Score(state, n).first;
            got_a[n] = alloc_pl.Score(state, n).first;
          }
        }
      });
    for (auto& t : th) t.join();
    if (got_t != want_t || got_l != want_l || got_a != want_a) {
      std::fprintf(stderr, "pod %lld: concurrent readers disagree with the serial pass\n", static_cast<long long>(pod));
      return 1;
    }
    for (int32_t n = 0; n < N; ++n) {
      if (want_t[n] < MinNodeScore || want_t[n] > MaxNodeScore || want_a[n] >= 0) {
        std::fprintf(stderr, "pod %lld node %d: TLP %lld out of range or Allocatable raw %lld not negative\n",
                     static_cast<long long>(pod), n, static_cast<long long>(want_t[n]), static_cast<long long>(want_a[n]));
        return 1;
      }
      if (!valid[n] && want_t[n] != MinNodeScore) return 2;  // no metrics -> MinNodeScore (targetloadpacking.go:114-120)
    }
    NodeScoreList list;
    for (int32_t n = 0; n < N; ++n) list.push_back({n, want_a[n]});
    if (!alloc_pl.ScoreExtensions()->NormalizeScore(state, list).IsSuccess()) return 3;
    int64_t lo = 1000, hi = -1;
    for (auto& s : list) {
      lo = std::min(lo, s.score);
      hi = std::max(hi, s.score);
    }
    if (lo != MinNodeScore || hi != MaxNodeScore) return 4;
    // the node with the smallest weighted allocatable must be the Least-mode winner
    const auto raw_best = std::max_element(want_a.begin(), want_a.end()) - want_a.begin();
    if (list[static_cast<size_t>(raw_best)].score != MaxNodeScore) return 5;
    NodeScoreList tl;
    for (int32_t n = 0; n < N; ++n) tl.push_back({n, want_t[n]});
    tlp.ScoreExtensions()->NormalizeScore(state, tl);
    for (int32_t n = 0; n < N; ++n)
      if (tl[static_cast<size_t>(n)].score != want_t[n]) return 6;
  }
  std::printf("harness ok: %lld pods x %lld nodes, %d concurrent readers\n", static_cast<long long>(P), static_cast<long long>(N), parallelism);
  return 0;
}
