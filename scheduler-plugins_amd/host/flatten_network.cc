// flatten_network.cc — object tables -> SoA for the network-aware plugins (host side, once per snapshot).
//
// Hoisted out of NetworkOverhead.PreFilter, which the reference runs once per pod and which itself loops
// over every node (networkoverhead.go:174-298):
//   CR fetch + sort + per-node costMap rebuild (sort.Sort + binary searches)     :438-497
//   GetDependencyList / GetScheduledList                                         util.go:194-232
//   the (scheduled pod x dependency) selector join                               :516-522, :590-594
// A pod's PreFilter state depends only on its (AppGroup, workload selector): the join is done once per such
// "workload key" and pods carry the key.  TopologicalSort's per-comparison CR Get + two binary searches
// (topologicalsort.go:118-127) become one FindPodOrder per pod.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstddef>
#include <utility>
#include <cstring>
#include <vector>

#include "../../include/spx.h"
#include "parallel.hpp"

namespace {

// util.FindPodOrder (util.go:138-153): binary search of Status.TopologyOrder by selector
int32_t find_pod_order(const spx_appgroup_objects* ag, int32_t g, int32_t selector) {
  const int32_t base = ag->topo_ptr[g];
  int low = 0, high = ag->topo_ptr[g + 1] - base - 1;
  while (low <= high) {
    const int mid = (low + high) / 2;
    const int32_t s = ag->topo_selector[base + mid];
    if (s == selector) return ag->topo_index[base + mid];
    if (s < selector) low = mid + 1;
    else high = mid - 1;
  }
  return -1;
}

}  // namespace

extern "C" int spx_flatten_net_topo(const spx_nettopo_objects* nt, int32_t* region_cost, int32_t* zone_cost) {
  if (!nt || !region_cost || !zone_cost) return SPX_ERR_ARG;
  const int64_t rg = nt->n_regions, zc = nt->n_zones;
  for (int64_t i = 0; i < rg * rg; ++i) region_cost[i] = -1;
  for (int64_t i = 0; i < zc * zc; ++i) zone_cost[i] = -1;
  for (int32_t o = 0; o < nt->n_regions; ++o)
    for (int32_t k = nt->rc_ptr[o]; k < nt->rc_ptr[o + 1]; ++k) {
      const int32_t d = nt->rc_dest[k];
      if (d < 0 || d >= nt->n_regions) continue;
      if (nt->rc_cost[k] < 0 || nt->rc_cost[k] > INT32_MAX) return SPX_ERR_ARG;  // costs are small non-negative ints
      region_cost[static_cast<int64_t>(o) * rg + d] = static_cast<int32_t>(nt->rc_cost[k]);
    }
  for (int32_t o = 0; o < nt->n_zones; ++o)
    for (int32_t k = nt->zc_ptr[o]; k < nt->zc_ptr[o + 1]; ++k) {
      const int32_t d = nt->zc_dest[k];
      if (d < 0 || d >= nt->n_zones) continue;
      if (nt->zc_cost[k] < 0 || nt->zc_cost[k] > INT32_MAX) return SPX_ERR_ARG;
      zone_cost[static_cast<int64_t>(o) * zc + d] = static_cast<int32_t>(nt->zc_cost[k]);
    }
  return SPX_OK;
}


// The sizing call leaves its result for the fill call that follows it on the same OS thread.  What identifies "the same tables" is
// their CONTENT, not their addresses: an ingest handle hands out the same table addresses cycle after cycle and batch sizes repeat,
// and under cgo the two calls of one cycle may land on different OS threads — so a result left behind by an abandoned sizing call
// could otherwise be served to a later cycle's fill call (round 3's advisor finding: wrong keys, and a heap overflow when the
// old cycle had more pairs).  A 64-bit fingerprint over every array the builders read costs one pass (~0.1 ms at 62.5k pods)
// against the 3-6 ms of the rebuild it saves.
uint64_t fnv(uint64_t h, const void* p, size_t bytes) {
  const uint64_t* w = static_cast<const uint64_t*>(p);
  size_t i = 0;
  for (; i + 8 <= bytes; i += 8) {  // the object tables' arrays are 8-byte aligned (malloc / numpy / std::vector)
    uint64_t v;
    std::memcpy(&v, reinterpret_cast<const char*>(p) + i, 8);
    h = (h ^ v) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  (void)w;
  for (; i < bytes; ++i) h = (h ^ reinterpret_cast<const unsigned char*>(p)[i]) * 0x100000001b3ull;
  return h;
}
uint64_t net_fingerprint(const spx_pod_objects* pods, const spx_appgroup_objects* ag) {
  uint64_t h = 0xcbf29ce484222325ull;
  const size_t P = static_cast<size_t>(pods->n_pods > 0 ? pods->n_pods : 0), G = static_cast<size_t>(ag->n_groups > 0 ? ag->n_groups : 0);
  h = fnv(h, &pods->n_pods, sizeof pods->n_pods);
  h = fnv(h, &ag->n_groups, sizeof ag->n_groups);
  if (P) h = fnv(h, pods->appgroup, P * 4), h = fnv(h, pods->selector, P * 4);
  if (G) {
    h = fnv(h, ag->wl_ptr, (G + 1) * 4), h = fnv(h, ag->topo_ptr, (G + 1) * 4), h = fnv(h, ag->placed_ptr, (G + 1) * 4);
    const size_t W = static_cast<size_t>(ag->wl_ptr[G]), T = static_cast<size_t>(ag->topo_ptr[G]), S = static_cast<size_t>(ag->placed_ptr[G]);
    if (W) {
      h = fnv(h, ag->wl_selector, W * 4), h = fnv(h, ag->dep_ptr, (W + 1) * 4);
      const size_t D = static_cast<size_t>(ag->dep_ptr[W]);
      if (D) h = fnv(h, ag->dep_selector, D * 4), h = fnv(h, ag->dep_max_cost, D * 8);
    }
    if (T) h = fnv(h, ag->topo_selector, T * 4), h = fnv(h, ag->topo_index, T * 4);
    if (S) h = fnv(h, ag->placed_selector, S * 4), h = fnv(h, ag->placed_node, S * 4);
  }
  return h ? h : 1;  // 0 = "nothing cached"
}

namespace {
// Everything spx_flatten_net_keys returns, computed in one pass.  The C entry point is called twice per batch (sizes, then the
// arrays): the sizing call leaves its result here, per calling thread, and the fill call that follows it with the same tables
// copies it out instead of repeating the pass.
struct NetKeys {
  uint64_t fp = 0;  // net_fingerprint of the tables the vectors were built from; 0 = nothing cached
  bool overflow = false;  // more pairs than the 32-bit offsets hold: the entry point refuses the batch
  std::vector<int32_t> pod_key, topo_order, pair_ptr, pair_node;
  std::vector<int64_t> pair_max_cost;
  std::vector<uint8_t> key_score_equally;
};
thread_local NetKeys tl_net_keys;

void build_net_keys(const spx_pod_objects* pods, const spx_appgroup_objects* ag, NetKeys& k) {
  const size_t P = static_cast<size_t>(pods->n_pods > 0 ? pods->n_pods : 0);
  k.fp = 0;  // set by the caller once the build is complete
  k.overflow = false;
  k.pod_key.assign(P, 0), k.topo_order.assign(P, -1);
  k.pair_ptr.clear(), k.pair_node.clear(), k.pair_max_cost.clear(), k.key_score_equally.clear();
  // (AppGroup, workload selector) -> key id, in order of first appearance.  A group has a handful of workloads: its keys sit in
  // a short list searched linearly (a hash map of packed pairs cost 50 ns per pod, an ordered map a tree walk)
  struct GroupKey {
    int32_t selector, key, topo;
  };
  std::vector<std::vector<GroupKey>> by_group(static_cast<size_t>(ag->n_groups > 0 ? ag->n_groups : 0));
  std::vector<std::pair<int32_t, int32_t>> order;
  order.push_back({-1, -1});  // key 0: "Pod does not belong to an AppGroup" -> scoreEqually (networkoverhead.go:187-190)
  for (size_t p = 0; p < P; ++p) {
    const int32_t g = pods->appgroup[p];
    if (g < 0 || g >= ag->n_groups) continue;
    const int32_t sel = pods->selector[p];
    auto& list = by_group[static_cast<size_t>(g)];
    const GroupKey* hit = nullptr;
    for (const GroupKey& e : list)
      if (e.selector == sel) {
        hit = &e;
        break;
      }
    if (!hit) {
      list.push_back(GroupKey{sel, static_cast<int32_t>(order.size()), find_pod_order(ag, g, sel)});
      order.push_back({g, sel});
      hit = &list.back();
    }
    k.pod_key[p] = hit->key;
    k.topo_order[p] = hit->topo;
  }
  // Per key: the flag and the (host, MaxNetworkCost) pairs of the pods already placed.  The keys are independent of each other:
  // counted on the host threads, offsets by a prefix sum, filled on the host threads (round 4; the key numbering above is the
  // serial part).  A key's workloads — those of its group whose selector matches — are listed once, not re-found per placed pod.
  const size_t K = order.size();
  k.key_score_equally.assign(K, 1);  // scoreEqually
  k.pair_ptr.assign(K + 1, 0);
  auto walk_keys = [&](bool fill) {
    spx_host::parallel_rows(static_cast<int64_t>(K), [&](int64_t k0, int64_t k1) {
      std::vector<int32_t> mine;  // the key's workloads, in workload order
      for (int64_t ki = k0; ki < k1; ++ki) {
        const int32_t g = order[static_cast<size_t>(ki)].first, sel = order[static_cast<size_t>(ki)].second;
        if (g < 0) continue;
        // dependencyList: Dependencies of every workload whose selector matches (util.go:203-209)
        mine.clear();
        bool any_dep = false;
        for (int32_t w = ag->wl_ptr[g]; w < ag->wl_ptr[g + 1]; ++w)
          if (ag->wl_selector[w] == sel) {
            mine.push_back(w);
            any_dep |= ag->dep_ptr[w + 1] > ag->dep_ptr[w];
          }
        const bool any_placed = ag->placed_ptr[g + 1] > ag->placed_ptr[g];
        if (!(any_dep && any_placed)) continue;
        uint8_t flag = 0;
        int32_t at = fill ? k.pair_ptr[static_cast<size_t>(ki)] : 0;
        for (int32_t s = ag->placed_ptr[g]; s < ag->placed_ptr[g + 1]; ++s)  // for each pod already allocated
          for (const int32_t w : mine)                                          //   for each dependency
            for (int32_t d = ag->dep_ptr[w]; d < ag->dep_ptr[w + 1]; ++d) {
              if (ag->placed_selector[s] != ag->dep_selector[d]) continue;
              if (ag->placed_node[s] < 0) flag = 2;  // host not in the snapshot: PreFilter returns Error (:258, :274)
              if (fill) k.pair_node[static_cast<size_t>(at)] = ag->placed_node[s], k.pair_max_cost[static_cast<size_t>(at)] = ag->dep_max_cost[d];
              ++at;
            }
        if (fill) k.key_score_equally[static_cast<size_t>(ki)] = flag;
        else k.pair_ptr[static_cast<size_t>(ki) + 1] = at;  // the key's count, turned into offsets below
      }
    }, 512);
  };
  walk_keys(false);
  int64_t total = 0;
  for (size_t ki = 0; ki < K; ++ki) {  // counts (stored one slot up) -> exclusive offsets
    const int64_t c = k.pair_ptr[ki + 1];
    k.pair_ptr[ki] = static_cast<int32_t>(total);
    total += c;
    if (total > INT32_MAX) {
      k.overflow = true;
      return;
    }
  }
  k.pair_ptr[K] = static_cast<int32_t>(total);
  k.pair_node.assign(static_cast<size_t>(total), 0), k.pair_max_cost.assign(static_cast<size_t>(total), 0);
  walk_keys(true);
}
}  // namespace

extern "C" int spx_flatten_net_keys(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int32_t* n_keys_out,
                                    int64_t* n_pairs_out, int32_t* pod_key, int32_t* topo_order, uint8_t* key_score_equally,
                                    int32_t* pair_ptr, int32_t* pair_node, int64_t* pair_max_cost) {
  if (!pods || !ag || !n_keys_out || !n_pairs_out) return SPX_ERR_ARG;
  const bool fill = pod_key && topo_order && key_score_equally && pair_ptr && pair_node && pair_max_cost;
  NetKeys& k = tl_net_keys;
  // a fill call after the sizing call for tables of the same CONTENT (the documented sequence) reuses that pass; a sizing call
  // always rebuilds
  const uint64_t fp = net_fingerprint(pods, ag);
  if (!(fill && k.fp == fp)) {
    build_net_keys(pods, ag, k);
    k.fp = fp;
  }
  *n_keys_out = static_cast<int32_t>(k.key_score_equally.size());
  *n_pairs_out = static_cast<int64_t>(k.pair_node.size());
  if (k.overflow) return SPX_ERR_ARG;
  if (fill) {
    std::copy(k.pod_key.begin(), k.pod_key.end(), pod_key);
    std::copy(k.topo_order.begin(), k.topo_order.end(), topo_order);
    std::copy(k.key_score_equally.begin(), k.key_score_equally.end(), key_score_equally);
    std::copy(k.pair_ptr.begin(), k.pair_ptr.end(), pair_ptr);
    std::copy(k.pair_node.begin(), k.pair_node.end(), pair_node);
    std::copy(k.pair_max_cost.begin(), k.pair_max_cost.end(), pair_max_cost);
    k = NetKeys{};  // one use: the tables may change before the next call
  }
  return SPX_OK;
}

// What binding pending pod p changes for NetworkOverhead's view of the pods scheduled after it (sequential commit, SURVEY 8f
// rank 1): p joins its AppGroup's scheduled list (util.GetScheduledList over the pod lister), so
//   * every workload key of the group that has dependencies stops "scoring equally" (the list is no longer empty:
//     networkoverhead.go:215-224) -> an entry (key, -1);
//   * every dependency of such a workload on p's workload selector gains a (host, MaxNetworkCost) pair -> (key, cost).
// Keys are numbered exactly as spx_flatten_net_keys numbers them.  Sizes first (NULL arrays), then the CSR.
namespace {
// computed in one pass; the sizing call leaves it for the fill call that follows (as spx_flatten_net_keys does)
struct NetCommit {
  uint64_t fp = 0;
  bool overflow = false;  // more entries than the 32-bit offsets hold: the entry point refuses the batch
  std::vector<int32_t> eff_ptr, eff_key;
  std::vector<int64_t> eff_cost;
};
thread_local NetCommit tl_net_commit;

void build_net_commit(const spx_pod_objects* pods, const spx_appgroup_objects* ag, NetCommit& out) {
  const size_t P = static_cast<size_t>(pods->n_pods > 0 ? pods->n_pods : 0);
  out.fp = 0;
  out.overflow = false;
  out.eff_ptr.clear();
  out.eff_key.clear(), out.eff_cost.clear();
  // key ids in order of first appearance, per group a short list (selector, key) — the numbering of spx_flatten_net_keys
  struct GroupKey {
    int32_t selector, key;
  };
  std::vector<std::vector<GroupKey>> by_group(static_cast<size_t>(ag->n_groups > 0 ? ag->n_groups : 0));
  std::vector<int32_t> key_of(P, 0);
  int32_t next = 1;
  for (size_t p = 0; p < P; ++p) {
    const int32_t g = pods->appgroup[p];
    if (g < 0 || g >= ag->n_groups) continue;
    const int32_t sel = pods->selector[p];
    auto& list = by_group[static_cast<size_t>(g)];
    int32_t key = -1;
    for (const GroupKey& e : list)
      if (e.selector == sel) {
        key = e.key;
        break;
      }
    if (key < 0) list.push_back(GroupKey{sel, key = next++});
    key_of[p] = key;
  }
  // The effects of binding a pod depend only on its (AppGroup, selector), i.e. on its key: computed once per key, copied per
  // pod (at 62.5k pods of 6.9k keys the per-pod evaluation was 21 ms per call, and the function used to run twice).  Round 4: the
  // keys' templates and the per-pod copies are independent of each other — both run on the host threads (the numbering above and
  // the prefix sum between them are the only serial passes): 18.7 -> see DESIGN.md 3.16
  const size_t K = static_cast<size_t>(next);
  // per key a template: the (affected key, cost or -1) entries binding a pod of that key adds — flat arrays, counted first and
  // filled second (a vector per key cost more in the allocator than the walk itself).  Per group, once: which of its keys have
  // dependencies at all and which workloads carry each key's selector (that walk of the group's workload list per (bound key,
  // affected key) pair was most of the function's time).
  std::vector<int32_t> t_off(K + 1, 0);
  std::vector<int32_t> t_key;
  std::vector<int64_t> t_cost;
  auto walk_groups = [&](bool fill) {
    spx_host::parallel_rows(static_cast<int64_t>(by_group.size()), [&](int64_t g0, int64_t g1) {
      std::vector<uint8_t> any_dep;
      std::vector<std::vector<int32_t>> wl_of;
      for (int64_t g = g0; g < g1; ++g) {
        const auto& list = by_group[static_cast<size_t>(g)];
        const size_t nk = list.size();
        if (nk == 0) continue;
        any_dep.assign(nk, 0);
        wl_of.resize(std::max(wl_of.size(), nk));
        for (size_t i = 0; i < nk; ++i) wl_of[i].clear();
        for (int32_t w = ag->wl_ptr[g]; w < ag->wl_ptr[g + 1]; ++w)
          for (size_t i = 0; i < nk; ++i)
            if (list[i].selector == ag->wl_selector[w]) {
              wl_of[i].push_back(w);
              if (ag->dep_ptr[w + 1] > ag->dep_ptr[w]) any_dep[i] = 1;
              break;  // (a selector sits once in the list)
            }
        for (const GroupKey& me : list) {  // the key of the pod being bound
          int32_t at = fill ? t_off[static_cast<size_t>(me.key)] : 0;
          for (size_t i = 0; i < nk; ++i) {
            if (!any_dep[i]) continue;
            if (fill) t_key[static_cast<size_t>(at)] = list[i].key, t_cost[static_cast<size_t>(at)] = -1;
            ++at;
            for (const int32_t w : wl_of[i])
              for (int32_t d = ag->dep_ptr[w]; d < ag->dep_ptr[w + 1]; ++d)
                if (ag->dep_selector[d] == me.selector) {
                  if (fill) t_key[static_cast<size_t>(at)] = list[i].key, t_cost[static_cast<size_t>(at)] = ag->dep_max_cost[d];
                  ++at;
                }
          }
          if (!fill) t_off[static_cast<size_t>(me.key) + 1] = at;  // the key's count, turned into offsets below
        }
      }
    }, 64);
  };
  walk_groups(false);
  int64_t t_total = 0;
  for (size_t k = 0; k < K; ++k) {  // counts (stored one slot up) -> exclusive offsets
    const int64_t c = t_off[k + 1];
    t_off[k] = static_cast<int32_t>(t_total);
    t_total += c;
    if (t_total > INT32_MAX) {
      out.overflow = true;
      return;
    }
  }
  t_off[K] = static_cast<int32_t>(t_total);
  t_key.resize(static_cast<size_t>(t_total)), t_cost.resize(static_cast<size_t>(t_total));
  walk_groups(true);
  out.eff_ptr.resize(P + 1);
  int64_t total = 0;
  for (size_t p = 0; p < P; ++p) {
    out.eff_ptr[p] = static_cast<int32_t>(total);
    const int32_t g = pods->appgroup[p];
    if (g >= 0 && g < ag->n_groups) total += t_off[static_cast<size_t>(key_of[p]) + 1] - t_off[static_cast<size_t>(key_of[p])];
    if (total > INT32_MAX) {
      out.overflow = true;
      return;
    }
  }
  out.eff_ptr[P] = static_cast<int32_t>(total);
  out.eff_key.resize(static_cast<size_t>(total)), out.eff_cost.resize(static_cast<size_t>(total));
  spx_host::parallel_rows(static_cast<int64_t>(P), [&](int64_t p0, int64_t p1) {
    for (int64_t p = p0; p < p1; ++p) {
      const int32_t g = pods->appgroup[p];
      if (g < 0 || g >= ag->n_groups) continue;
      const size_t k = static_cast<size_t>(key_of[static_cast<size_t>(p)]), n_e = static_cast<size_t>(t_off[k + 1] - t_off[k]);
      if (n_e == 0) continue;
      std::memcpy(&out.eff_key[static_cast<size_t>(out.eff_ptr[static_cast<size_t>(p)])], &t_key[static_cast<size_t>(t_off[k])], n_e * sizeof(int32_t));
      std::memcpy(&out.eff_cost[static_cast<size_t>(out.eff_ptr[static_cast<size_t>(p)])], &t_cost[static_cast<size_t>(t_off[k])], n_e * sizeof(int64_t));
    }
  }, 4096);
}
}  // namespace

extern "C" int spx_flatten_net_commit(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int64_t* n_entries_out, int32_t* eff_ptr,
                                      int32_t* eff_key, int64_t* eff_cost) {
  if (!pods || !ag || !n_entries_out) return SPX_ERR_ARG;
  const bool fill = eff_ptr && eff_key && eff_cost;
  NetCommit& k = tl_net_commit;
  const uint64_t fp = net_fingerprint(pods, ag);
  if (!(fill && k.fp == fp)) {
    build_net_commit(pods, ag, k);
    k.fp = fp;
  }
  if (k.overflow) return SPX_ERR_ARG;
  *n_entries_out = static_cast<int64_t>(k.eff_key.size());
  if (fill) {
    std::copy(k.eff_ptr.begin(), k.eff_ptr.end(), eff_ptr);
    std::copy(k.eff_key.begin(), k.eff_key.end(), eff_key);
    std::copy(k.eff_cost.begin(), k.eff_cost.end(), eff_cost);
    k = NetCommit{};  // one use: the tables may change before the next call
  }
  return SPX_OK;
}

// Pods that joined AppGroup scheduled lists since the tables were flattened (bound by an earlier cycle, by another scheduler):
// what each adds to the workload keys' pair lists, in the key numbering spx_flatten_net_keys gives the pending batch `pods` — the
// input of spx_update_net_placed.  Placed pod j = (group[j], selector[j], node[j]); per affected key of its group an entry
// (key, node, -1) "the scheduled list is no longer empty" and, per dependency of the key's workloads on the pod's selector, an
// entry (key, node, MaxNetworkCost) — exactly what a re-flatten with the pod appended to the group's placed list would add.
// Sizes first (NULL arrays), then the entries.
extern "C" int spx_flatten_net_placed(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int64_t n_placed, const int32_t* group,
                                      const int32_t* selector, const int32_t* node, int64_t* n_entries_out, int32_t* key_out, int32_t* node_out,
                                      int64_t* cost_out) {
  if (!pods || !ag || !n_entries_out || n_placed < 0 || (n_placed && (!group || !selector || !node))) return SPX_ERR_ARG;
  const bool fill = key_out && node_out && cost_out;
  const size_t P = static_cast<size_t>(pods->n_pods > 0 ? pods->n_pods : 0);
  struct GroupKey {
    int32_t selector, key;
  };
  std::vector<std::vector<GroupKey>> by_group(static_cast<size_t>(ag->n_groups > 0 ? ag->n_groups : 0));
  int32_t next = 1;
  for (size_t p = 0; p < P; ++p) {  // the numbering of spx_flatten_net_keys: first appearance in the pending batch
    const int32_t g = pods->appgroup[p];
    if (g < 0 || g >= ag->n_groups) continue;
    auto& list = by_group[static_cast<size_t>(g)];
    bool seen = false;
    for (const GroupKey& e : list) seen |= e.selector == pods->selector[p];
    if (!seen) list.push_back(GroupKey{pods->selector[p], next++});
  }
  int64_t n_out = 0;
  for (int64_t j = 0; j < n_placed; ++j) {
    const int32_t g = group[j];
    if (g < 0 || g >= ag->n_groups) continue;  // not an AppGroup member: no list changes
    for (const GroupKey& kk : by_group[static_cast<size_t>(g)]) {
      bool any_dep = false;
      for (int32_t w = ag->wl_ptr[g]; w < ag->wl_ptr[g + 1]; ++w)
        if (ag->wl_selector[w] == kk.selector && ag->dep_ptr[w + 1] > ag->dep_ptr[w]) any_dep = true;
      if (!any_dep) continue;
      if (fill) key_out[n_out] = kk.key, node_out[n_out] = node[j], cost_out[n_out] = -1;
      ++n_out;
      for (int32_t w = ag->wl_ptr[g]; w < ag->wl_ptr[g + 1]; ++w) {
        if (ag->wl_selector[w] != kk.selector) continue;
        for (int32_t d = ag->dep_ptr[w]; d < ag->dep_ptr[w + 1]; ++d) {
          if (ag->dep_selector[d] != selector[j]) continue;
          if (fill) key_out[n_out] = kk.key, node_out[n_out] = node[j], cost_out[n_out] = ag->dep_max_cost[d];
          ++n_out;
        }
      }
    }
  }
  *n_entries_out = n_out;
  return SPX_OK;
}

extern "C" int spx_toposort_less(const spx_pod_objects* pods, const int32_t* topo_order, int64_t n_pairs, const int64_t* a,
                                 const int64_t* b, uint8_t* less_out) {
  if (!pods || !topo_order || !a || !b || !less_out) return SPX_ERR_ARG;
  for (int64_t i = 0; i < n_pairs; ++i) {
    const int64_t p1 = a[i], p2 = b[i];
    if (p1 < 0 || p2 < 0 || p1 >= pods->n_pods || p2 >= pods->n_pods) return SPX_ERR_ARG;
    const int32_t g1 = pods->appgroup[p1], g2 = pods->appgroup[p2];
    bool less;
    if (g1 != g2 || g1 < 0) {  // queuesort.PrioritySort: priority desc, then queue timestamp asc (topologicalsort.go:109-113)
      less = pods->priority[p1] > pods->priority[p2] ||
             (pods->priority[p1] == pods->priority[p2] && pods->queue_ts[p1] < pods->queue_ts[p2]);
    } else {
      less = topo_order[p1] <= topo_order[p2];  // "Lower is better" :131
    }
    less_out[i] = less ? 1 : 0;
  }
  return SPX_OK;
}
