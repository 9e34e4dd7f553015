// harness.cc — drives libspx.so through the C++ host mirror with the call pattern upstream kube-scheduler uses:
// one pod at a time, Filter/Score fanned out over the nodes by 16 concurrent workers (the Parallelizer the
// reference's benchmarks copy, pkg/trimaran/targetloadpacking/targetloadpacking_test.go:386-405), then
// NormalizeScore once per plugin.  Checks that concurrent readers see exactly what a serial pass sees, that
// Allocatable's raw scores are negative (Least) and its normalised list spans [0,100], that TLP's
// NormalizeScore is a no-op; then the Filter side (filter_profile below): TopologyMatch, NetworkOverhead, CapacityScheduling and
// TopologicalSort with statuses, messages and scores known by construction.  Needs a GPU; exits 0 and prints "harness ok".
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <string>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "../../scheduler-plugins_amd/host/plugins.hpp"

using namespace spx::host;

// ---------------------------------------------------------------------------------------------------------------------
// The Filter side of the interface from C++: TopologyMatch.Filter / Score, NetworkOverhead.PreFilter / Filter / Score /
// NormalizeScore, CapacityScheduling.PreFilter and the TopologicalSort queue order, on a hand-built snapshot whose answers are
// known by construction, with the same 16-reader fan-out per pod.  Returns 0, or the number of the check that failed.
static int filter_profile() {
  const int64_t N = 203, P = 24;
  constexpr int R = 2, Z = SPX_NRT_MAX_ZONES, C = SPX_NRT_MAX_CTRS;
  const int64_t Gi = 1ll << 30;
  Engine e(0);
  e.n_nodes = N;
  // --- NodeResourceTopology: two NUMA zones per node; zone 0 holds 4, 6 or 8 cores, zone 1 one core; single-numa-node policy,
  //     every third node at pod scope
  const int32_t slot_res[R] = {SPX_RES_CPU, SPX_RES_MEMORY};
  const uint8_t slot_flags[R] = {SPX_NRT_SLOT_AFFINE | SPX_NRT_SLOT_CPU, SPX_NRT_SLOT_AFFINE};
  const int64_t slot_weight[R] = {1, 1};
  spx_nrt_slots slots{R, slot_res, slot_flags, slot_weight};
  e.check(spx_upload_nrt_slots(e.raw(), &slots));
  std::vector<uint8_t> nflags(N), nz(N, 2), zid(N * Z, 0), zpres(N * Z, 0), npres(N, 3);
  std::vector<int32_t> max_numa(N, 8), zcost(N * Z * Z, 255);
  std::vector<int64_t> zavail(N * Z * R, 0);
  std::vector<float> minavg(N * Z, 0.0f);
  for (int64_t i = 0; i < N; ++i) {
    nflags[i] = SPX_NRT_F_HAS_NRT | SPX_NRT_F_FRESH | SPX_NRT_F_SINGLE_NUMA | (i % 3 == 2 ? SPX_NRT_F_POD_SCOPE : 0);
    for (int z = 0; z < 2; ++z) {
      zid[i * Z + z] = static_cast<uint8_t>(z);
      zpres[i * Z + z] = 3;
      zavail[(i * Z + z) * R + 0] = z == 0 ? 4000 + 2000 * (i % 3) : 1000;
      zavail[(i * Z + z) * R + 1] = 8 * Gi;
      for (int o = 0; o < 2; ++o) zcost[(i * Z + z) * Z + o] = o == z ? 10 : 20;
    }
    minavg[i * Z + 0] = 10.0f, minavg[i * Z + 1] = 15.0f;
  }
  spx_nrt_nodes_soa nn{N, R, nflags.data(), max_numa.data(), nz.data(), zid.data(), zpres.data(), zavail.data(), zcost.data(), minavg.data(), npres.data()};
  e.check(spx_upload_nrt_nodes(e.raw(), &nn));
  // --- pods: Guaranteed, one container asking for 1 .. 8 cores and 1 GiB
  std::vector<uint8_t> qos(P, SPX_QOS_GUARANTEED), non_native(P, 0), n_ctr(P, 1), ckind(P * C, SPX_CTR_APP), cpres(P * C, 0), ppres(P, 3);
  std::vector<int64_t> creq(P * C * R, 0), preq(P * R, 0), cores(P);
  for (int64_t p = 0; p < P; ++p) {
    cores[p] = 1 + p % 8;
    cpres[p * C] = 3;
    creq[(p * C) * R + 0] = preq[p * R + 0] = 1000 * cores[p];
    creq[(p * C) * R + 1] = preq[p * R + 1] = Gi;
  }
  spx_nrt_pods_soa np{P, R, qos.data(), non_native.data(), n_ctr.data(), ckind.data(), cpres.data(), creq.data(), ppres.data(), preq.data()};
  e.check(spx_upload_nrt_pods(e.raw(), &np));
  // --- NetworkOverhead: region = i % 2, zone = i % 4 (zones 0, 2 in region 0; 1, 3 in region 1); zones of one region cost 5 / 7,
  //     regions 20.  Workload key 0: no AppGroup; key 1: depends (MaxNetworkCost 6) on a pod placed on node 0; key 2: that and a
  //     second dependency (max 30) placed on node 1
  std::vector<int32_t> region(N), zone(N);
  for (int64_t i = 0; i < N; ++i) region[i] = static_cast<int32_t>(i % 2), zone[i] = static_cast<int32_t>(i % 4);
  spx_net_nodes_soa netn{N, region.data(), zone.data()};
  e.check(spx_upload_net_nodes(e.raw(), &netn));
  const int32_t rcost[4] = {-1, 20, 20, -1};
  int32_t zc[16];
  for (int& v : zc) v = -1;
  zc[0 * 4 + 2] = zc[2 * 4 + 0] = 5;
  zc[1 * 4 + 3] = zc[3 * 4 + 1] = 7;
  spx_net_topo_soa topo{2, 4, rcost, zc};
  e.check(spx_upload_net_topo(e.raw(), &topo));
  std::vector<int32_t> pod_key(P), topo_order(P);
  for (int64_t p = 0; p < P; ++p) pod_key[p] = static_cast<int32_t>(p % 3), topo_order[p] = static_cast<int32_t>(p % 5);
  const uint8_t key_flag[3] = {1, 0, 0};
  const int32_t pair_ptr[4] = {0, 0, 1, 3}, pair_node[3] = {0, 0, 1};
  const int64_t pair_max[3] = {6, 6, 30};
  spx_net_pods_soa netp{P, 3, pod_key.data(), key_flag, pair_ptr, pair_node, pair_max, topo_order.data()};
  e.check(spx_upload_net_pods(e.raw(), &netp));
  // --- CapacityScheduling: namespace 0 has a quota (used 3 cores of max 4), namespace 1 has none
  std::vector<int32_t> pod_ns(P), pod_prio(P, 0);
  std::vector<int64_t> pod_req(P * 8, 0);
  std::vector<uint8_t> pod_reqp(P, 0);
  for (int64_t p = 0; p < P; ++p) pod_ns[p] = static_cast<int32_t>(p % 2), pod_req[p * 8] = 1000 * cores[p];
  const uint8_t has_quota[2] = {1, 0}, no_p[2] = {0, 0};
  int64_t used[16] = {0}, qmax[16], agg_used[8] = {3000}, agg_min[8], other[16] = {0};
  for (int64_t& v : qmax) v = INT64_MAX;
  for (int64_t& v : agg_min) v = INT64_MAX;
  used[0] = 3000, qmax[0] = 4000;
  const uint8_t agg_p = 0;
  const int32_t nom_ptr[3] = {0, 0, 0};
  spx_quota_soa q{};
  q.n_pods = P, q.n_namespaces = 2, q.pod_ns = pod_ns.data(), q.pod_priority = pod_prio.data(), q.pod_req = pod_req.data(), q.pod_req_present = pod_reqp.data();
  q.has_quota = has_quota, q.used = used, q.used_present = no_p, q.max = qmax, q.max_present = no_p, q.agg_used = agg_used, q.agg_used_present = &agg_p;
  q.agg_min = agg_min, q.agg_min_present = &agg_p, q.other_nominated = other, q.other_nominated_present = no_p, q.nom_ptr = nom_ptr;
  e.check(spx_upload_quota(e.raw(), &q));
  e.Eval((1u << SPX_PLUGIN_NRT) | (1u << SPX_PLUGIN_NETOVERHEAD) | (1u << SPX_PLUGIN_CAPACITY), 0, P);

  TopologyMatch tm;
  NetworkOverhead no;
  CapacityScheduling cs;
  constexpr int parallelism = 16;
  for (int64_t pod = 0; pod < P; ++pod) {
    // PreFilter (once per pod, as upstream)
    const Status pre = cs.PreFilter(e, pod, pod_ns[pod] ? "ns1" : "ns0", "p" + std::to_string(pod));
    const bool over_max = pod_ns[pod] == 0 && 3000 + 1000 * cores[pod] > 4000;
    if (pre.IsSuccess() == over_max) return 10;
    if (over_max && pre.message != "Pod ns0/p" + std::to_string(pod) + " is rejected in PreFilter because ElasticQuota ns0 is more than Max") return 11;
    CycleState serial(e, pod), state(e, pod);
    if (!no.PreFilter(state).IsSuccess()) return 12;
    std::vector<int> f_code(N), g_code(N), nf_code(N), ng_code(N);
    std::vector<std::string> f_msg(N), g_msg(N), nf_msg(N), ng_msg(N);
    std::vector<int64_t> f_score(N), g_score(N), nf_raw(N), ng_raw(N);
    auto probe = [&](CycleState& st, int32_t n, std::vector<int>& code, std::vector<std::string>& msg, std::vector<int64_t>& score, std::vector<int>& ncode,
                     std::vector<std::string>& nmsg, std::vector<int64_t>& nraw) {
      const Status f = tm.Filter(st, n);
      code[n] = f.code, msg[n] = f.message, score[n] = tm.Score(st, n).first;
      const Status nfs = no.Filter(st, n, "n" + std::to_string(n));
      ncode[n] = nfs.code, nmsg[n] = nfs.message, nraw[n] = no.Score(st, n).first;
    };
    for (int32_t n = 0; n < N; ++n) probe(serial, n, f_code, f_msg, f_score, nf_code, nf_msg, nf_raw);
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int w = 0; w < parallelism; ++w)
      th.emplace_back([&] {
        for (;;) {
          const int b = next.fetch_add(8);
          if (b >= N) return;
          for (int32_t n = b; n < std::min<int64_t>(N, b + 8); ++n) probe(state, n, g_code, g_msg, g_score, ng_code, ng_msg, ng_raw);
        }
      });
    for (auto& t : th) t.join();
    if (f_code != g_code || f_msg != g_msg || f_score != g_score || nf_code != ng_code || nf_msg != ng_msg || nf_raw != ng_raw) return 13;
    NodeScoreList list;
    for (int32_t n = 0; n < N; ++n)
      if (nf_code[n] == Success) list.push_back({n, nf_raw[n]});
    if (!no.ScoreExtensions()->NormalizeScore(state, list).IsSuccess()) return 14;
    for (int32_t n = 0; n < N; ++n) {
      // TopologyMatch: the container (or pod) fits a single NUMA zone iff zone 0 has the cores
      const bool fits = 1000 * cores[pod] <= 4000 + 2000 * (n % 3);
      const bool pod_scope = n % 3 == 2;
      if ((f_code[n] == Success) != fits) return 20;
      if (!fits && (f_code[n] != Unschedulable || f_msg[n] != (pod_scope ? "cannot align pod" : "cannot align container"))) return 21;
      if (f_score[n] < MinNodeScore || f_score[n] > MaxNodeScore) return 22;
      // NetworkOverhead, by construction (networkoverhead.go:536-633): same host 0, same zone 1, zones of one region their cost,
      // other region 20; a dependency is satisfied when the cost does not exceed its MaxNetworkCost
      const int key = pod_key[pod];
      if (key == 0) {
        if (nf_code[n] != Success || nf_raw[n] != 0) return 30;  // no AppGroup: every node passes, all scores equal
        continue;
      }
      auto cost_to = [&](int host) -> int64_t {
        if (n == host) return 0;
        if (region[n] != region[host]) return 20;
        return zone[n] == zone[host] ? SPX_NET_SAME_ZONE : (region[n] == 0 ? 5 : 7);
      };
      int64_t cost = cost_to(0), sat = cost_to(0) <= 6 ? 1 : 0, vio = 1 - sat;
      if (key == 2) cost += cost_to(1), sat += cost_to(1) <= 30 ? 1 : 0, vio += cost_to(1) <= 30 ? 0 : 1;
      if (nf_raw[n] != cost) return 31;
      if ((nf_code[n] == Success) != !(vio > sat)) return 32;
      if (vio > sat && nf_msg[n] != "Node n" + std::to_string(n) + " does not meet several network requirements from Workload dependencies: Satisfied: " +
                                       std::to_string(sat) + " Violated: " + std::to_string(vio))
        return 33;
    }
    // (the engine normalises over the nodes that passed EVERY Filter of the evaluation, like upstream's RunScorePlugins; with up to
    // 4 cores the pod fits every node's zone 0, so the feasible set is NetworkOverhead's own)
    if (pod_key[pod] == 1 && cores[pod] <= 4) {  // feasible costs {0, 1, 5}: NormalizeScore maps them to 100, 80, 0 (networkoverhead.go:389-418)
      for (const NodeScore& s : list) {
        const int64_t want = s.node == 0 ? 100 : (zone[s.node] == 0 ? 80 : 0);
        if (s.score != want) return 34;
      }
      if (list.size() != static_cast<size_t>((N + 1) / 2)) return 35;  // exactly the nodes of region 0 pass Filter
    }
  }
  // --- TopologicalSort: the queue order from the device sort, checked pair by pair with the plugin's Less
  std::vector<int32_t> prio(P), group(P);
  std::vector<int64_t> ts(P);
  for (int64_t p = 0; p < P; ++p) prio[p] = static_cast<int32_t>((p * 7) % 3) * 10, ts[p] = 1000 - 13 * ((p * 5) % P), group[p] = p % 4 == 3 ? -1 : static_cast<int32_t>(p % 2);
  spx_sort_keys_soa sk{P, prio.data(), ts.data(), group.data(), topo_order.data()};
  e.check(spx_upload_sort_keys(e.raw(), &sk));
  std::vector<int32_t> perm(P);
  e.check(spx_sort_keys(e.raw(), perm.data()));
  spx_pod_objects pobj{};
  pobj.n_pods = P, pobj.priority = prio.data(), pobj.queue_ts = ts.data(), pobj.appgroup = group.data();
  TopologicalSort sorter(&pobj, topo_order.data());
  std::vector<uint8_t> seen(P, 0);
  for (int64_t i = 0; i < P; ++i) {
    if (perm[i] < 0 || perm[i] >= P || seen[perm[i]]++) return 40;
    if (i + 1 < P) {
      const int64_t x = perm[i], y = perm[i + 1];
      const bool tie = (group[x] != group[y] || group[x] < 0) && prio[x] == prio[y] && ts[x] == ts[y];
      if (!sorter.Less(x, y) && !tie) return 41;
    }
  }
  std::printf("filter mirrors ok: TopologyMatch / NetworkOverhead / CapacityScheduling / TopologicalSort, %lld pods x %lld nodes\n",
              static_cast<long long>(P), static_cast<long long>(N));
  return 0;
}

// Boundary throughput (SURVEY 8b "threading"): what a per-pod Score() is bounded by once the sweep is done — rows per second
// through spx_fetch_scores / spx_fetch_score_rows under the reference's fan-out of 16 concurrent callers
// (targetloadpacking_test.go:386-405), on config #2's node count.
int boundary_throughput() {
  const int64_t N = 10000, P = 4096;
  Engine e(0);
  e.n_nodes = N;
  std::vector<int64_t> alloc(2 * N), cap(N), missing(N, 0), acpu(N), amem(N), pod_milli(P), rcpu(P), rmem(P);
  std::vector<double> util(N), z(N, 0.0);
  std::vector<uint8_t> valid(N, 1), flags(N, 7);
  for (int64_t i = 0; i < N; ++i) cap[i] = 8000 + 1000 * (i % 57), acpu[i] = cap[i], amem[i] = (32ll + i % 100) << 30, alloc[i] = amem[i], alloc[N + i] = acpu[i], util[i] = (i * 37 % 1000) / 10.0;
  for (int64_t p = 0; p < P; ++p) pod_milli[p] = 100 + (371 * p) % 7000, rcpu[p] = pod_milli[p], rmem[p] = (1ll + p % 64) << 28;
  spx_alloc_nodes_soa an{N, 2, alloc.data()};
  e.check(spx_upload_alloc_nodes(e.raw(), &an));
  spx_trimaran_nodes_soa tn{N, cap.data(), util.data(), missing.data(), valid.data(), acpu.data(), amem.data(), z.data(), z.data(), z.data(), z.data(), flags.data()};
  e.check(spx_upload_trimaran_nodes(e.raw(), &tn));
  spx_trimaran_pods_soa tp{P, pod_milli.data(), rcpu.data(), rmem.data()};
  e.check(spx_upload_trimaran_pods(e.raw(), &tp));
  e.Eval((1u << SPX_PLUGIN_ALLOCATABLE) | (1u << SPX_PLUGIN_TLP), 0, P);
  std::vector<uint8_t> want(static_cast<size_t>(N));
  e.check(spx_fetch_scores(e.raw(), SPX_PLUGIN_TLP, 17, want.data()));
  auto run = [&](int readers, int rows_per_call, double seconds) {
    std::atomic<int64_t> rows{0};
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int w = 0; w < readers; ++w)
      th.emplace_back([&, w] {
        std::vector<uint8_t> buf(static_cast<size_t>(N) * rows_per_call);
        int64_t pod = (w * 257) % P, mine = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
          const int64_t b = std::min<int64_t>(pod, P - rows_per_call);
          const int rc = rows_per_call == 1 ? spx_fetch_scores(e.raw(), SPX_PLUGIN_TLP, b, buf.data())
                                            : spx_fetch_score_rows(e.raw(), SPX_PLUGIN_TLP, b, b + rows_per_call, buf.data(), N);
          if (rc != SPX_OK) { bad++; return; }
          if (b <= 17 && 17 < b + rows_per_call && std::memcmp(buf.data() + (17 - b) * N, want.data(), static_cast<size_t>(N)) != 0) bad++;
          mine += rows_per_call;
          pod = (pod + 16 * rows_per_call + 1) % P;
        }
        rows += mine;
      });
    for (auto& t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (bad.load()) return -1.0;
    return static_cast<double>(rows.load()) / dt;
  };
  const double r1 = run(1, 1, 0.3), r16 = run(16, 1, 0.4), b1 = run(1, 64, 0.3), b16 = run(16, 64, 0.4);
  if (r1 < 0 || r16 < 0 || b1 < 0 || b16 < 0) return 60;
  std::printf("boundary throughput, %lld-node rows (uint8): 1 reader %.0f rows/s, 16 readers %.0f rows/s; 64-row bulk fetch: 1 reader %.0f rows/s, "
              "16 readers %.0f rows/s (%.2f GB/s)\n", static_cast<long long>(N), r1, r16, b1, b16, b16 * N / 1e9);
  return 0;
}

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
  std::printf("trimaran mirrors ok: %lld pods x %lld nodes, %d concurrent readers\n", static_cast<long long>(P), static_cast<long long>(N), parallelism);
  const int rc2 = filter_profile();
  if (rc2 != 0) {
    std::fprintf(stderr, "filter profile failed at check %d\n", rc2);
    return rc2;
  }
  const int rc3 = boundary_throughput();
  if (rc3 != 0) {
    std::fprintf(stderr, "boundary throughput failed at check %d\n", rc3);
    return rc3;
  }
  std::printf("harness ok: %lld pods x %lld nodes, %d concurrent readers\n", static_cast<long long>(P), static_cast<long long>(N), parallelism);
  return 0;
}
