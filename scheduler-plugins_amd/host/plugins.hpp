// plugins.hpp — C++ host-side mirror of the reference's plugin interface over the C ABI (include/spx.h).
//
// The reference is Go; with no Go toolchain in this image the host side is written in C++ with the same
// names, argument meaning and status behaviour as the framework.FilterPlugin / ScorePlugin /
// PreFilterPlugin / QueueSortPlugin methods it stands in for:
//
//   Allocatable              pkg/noderesources/allocatable.go:45-168
//   TargetLoadPacking        pkg/trimaran/targetloadpacking/targetloadpacking.go:55-195
//   LoadVariationRiskBalancing  pkg/trimaran/loadvariationriskbalancing/loadvariationriskbalancing.go:45-136
//   TopologyMatch            pkg/noderesourcetopology/{plugin.go,filter.go:179,score.go:62}
//   NetworkOverhead          pkg/networkaware/networkoverhead/networkoverhead.go:66-435
//   TopologicalSort          pkg/networkaware/topologicalsort/topologicalsort.go:45-132
//   CapacityScheduling       pkg/capacityscheduling/capacity_scheduling.go:57-283 (PreFilter only)
//
// A pod is identified by its row in the evaluated batch, a node by its column.  Method bodies are row
// lookups: the first call for a pod fetches its N-byte row through spx_fetch_* into the CycleState, after
// which concurrent callers (upstream's 16-goroutine Parallelizer) only read memory.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/spx.h"

namespace spx::host {

// ---- fwk.Status (k8s.io/kube-scheduler/framework) -------------------------------------------------
enum Code { Success = 0, Error = 1, Unschedulable = 2 };
constexpr int64_t MaxNodeScore = 100, MinNodeScore = 0;

struct Status {
  Code code = Success;
  std::string message;
  bool IsSuccess() const { return code == Success; }  // a nil *fwk.Status is Success
  static Status Ok() { return {}; }
};

struct NodeScore {
  int32_t node;   // column of the node in the snapshot
  int64_t score;
};
using NodeScoreList = std::vector<NodeScore>;

// ---- RAII engine -----------------------------------------------------------------------------------
class Engine {
 public:
  explicit Engine(int device = 0) {
    if (spx_create(device, &e_) != SPX_OK) throw std::runtime_error(std::string("spx_create: ") + spx_last_error(nullptr));
  }
  ~Engine() { spx_destroy(e_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  spx_engine* raw() const { return e_; }
  void check(int rc) const {
    if (rc != SPX_OK) throw std::runtime_error(std::string("spx: ") + spx_last_error(e_));
  }
  void Eval(uint32_t mask, int64_t row_begin, int64_t row_end) {
    check(spx_eval(e_, mask, row_begin, row_end));
    check(spx_sync(e_));
  }
  int64_t n_nodes = 0;  // set by whoever uploaded the node tables

 private:
  spx_engine* e_ = nullptr;
};

// ---- fwk.CycleState: per-pod scratch shared by the plugins of one scheduling cycle -----------------
class CycleState {
 public:
  CycleState(Engine& e, int64_t pod_row) : e_(e), pod_(pod_row) {}
  int64_t pod() const { return pod_; }
  const std::vector<uint8_t>& ScoreRow(int plugin) {
    return Cached(score_, plugin, [&](std::vector<uint8_t>& r) { e_.check(spx_fetch_scores(e_.raw(), plugin, pod_, r.data())); });
  }
  const std::vector<uint8_t>& StatusRow(int plugin) {
    return Cached(status_, plugin, [&](std::vector<uint8_t>& r) { e_.check(spx_fetch_status(e_.raw(), plugin, pod_, r.data())); });
  }
  const std::vector<int64_t>& RawRow(int plugin, int which = 0) {
    std::lock_guard<std::mutex> g(mu_);
    auto& r = raw_[{plugin, which}];
    if (r.empty()) {
      r.resize(static_cast<size_t>(e_.n_nodes));
      e_.check(spx_fetch_raw(e_.raw(), plugin, which, pod_, r.data()));
    }
    return r;
  }

 private:
  template <typename F>
  const std::vector<uint8_t>& Cached(std::map<int, std::vector<uint8_t>>& m, int plugin, F fetch) {
    std::lock_guard<std::mutex> g(mu_);
    auto& r = m[plugin];
    if (r.empty()) {
      r.resize(static_cast<size_t>(e_.n_nodes));
      fetch(r);
    }
    return r;  // std::map nodes are stable: readers keep a valid reference after the lock is released
  }
  Engine& e_;
  int64_t pod_;
  std::mutex mu_;
  std::map<int, std::vector<uint8_t>> score_, status_;
  std::map<std::pair<int, int>, std::vector<int64_t>> raw_;
};

// ---- score plugins ---------------------------------------------------------------------------------
class Allocatable {
 public:
  static constexpr const char* AllocatableName = "NodeResourcesAllocatable";
  const char* Name() const { return AllocatableName; }
  // Score: the raw weighted allocatable sum (negative for Least) — allocatable.go:63-71
  std::pair<int64_t, Status> Score(CycleState& state, int32_t node) const {
    return {state.RawRow(SPX_PLUGIN_ALLOCATABLE)[static_cast<size_t>(node)], Status::Ok()};
  }
  const Allocatable* ScoreExtensions() const { return this; }
  // NormalizeScore: min-max onto [0,100] over the list it is handed — allocatable.go:143-168.  The engine
  // normalised over the feasible set it was evaluated with; the list must be that set.
  Status NormalizeScore(CycleState& state, NodeScoreList& scores) const {
    const auto& row = state.ScoreRow(SPX_PLUGIN_ALLOCATABLE);
    for (auto& s : scores) s.score = row[static_cast<size_t>(s.node)];
    return Status::Ok();
  }
};

class TargetLoadPacking {
 public:
  static constexpr const char* PluginName = "TargetLoadPacking";
  const char* Name() const { return PluginName; }
  std::pair<int64_t, Status> Score(CycleState& state, int32_t node) const {
    return {state.ScoreRow(SPX_PLUGIN_TLP)[static_cast<size_t>(node)], Status::Ok()};
  }
  const TargetLoadPacking* ScoreExtensions() const { return this; }
  Status NormalizeScore(CycleState&, NodeScoreList&) const { return Status::Ok(); }  // targetloadpacking.go:193-195
};

class LoadVariationRiskBalancing {
 public:
  static constexpr const char* PluginName = "LoadVariationRiskBalancing";
  const char* Name() const { return PluginName; }
  std::pair<int64_t, Status> Score(CycleState& state, int32_t node) const {
    return {state.ScoreRow(SPX_PLUGIN_LVRB)[static_cast<size_t>(node)], Status::Ok()};
  }
  const LoadVariationRiskBalancing* ScoreExtensions() const { return this; }
  Status NormalizeScore(CycleState&, NodeScoreList&) const { return Status::Ok(); }  // loadvariationriskbalancing.go:134-136
};

class TopologyMatch {
 public:
  static constexpr const char* PluginName = "NodeResourceTopologyMatch";
  const char* Name() const { return PluginName; }
  Status Filter(CycleState& state, int32_t node) const {  // filter.go:179-245
    static const char* const kMsg[] = {"", "invalid node topology data", "cannot align init container",
                                       "cannot align sidecar container", "cannot align container", "cannot align pod"};
    const uint8_t code = state.StatusRow(SPX_PLUGIN_NRT)[static_cast<size_t>(node)];
    if (code == 0) return Status::Ok();
    if (code > SPX_NRT_ST_POD) return {Error, "inconsistent resource accounting"};
    return {Unschedulable, kMsg[code]};
  }
  std::pair<int64_t, Status> Score(CycleState& state, int32_t node) const {  // score.go:62-102
    return {state.ScoreRow(SPX_PLUGIN_NRT)[static_cast<size_t>(node)], Status::Ok()};
  }
  const TopologyMatch* ScoreExtensions() const { return nullptr; }  // score.go:104-106
};

class NetworkOverhead {
 public:
  static constexpr const char* PluginName = "NetworkOverhead";
  const char* Name() const { return PluginName; }
  // PreFilter: the per-node maps of PreFilterState are the engine's tables; nothing to compute per pod here
  Status PreFilter(CycleState&) const { return Status::Ok(); }
  Status Filter(CycleState& state, int32_t node, const std::string& node_name) const {  // networkoverhead.go:326-359
    const uint8_t code = state.StatusRow(SPX_PLUGIN_NETOVERHEAD)[static_cast<size_t>(node)];
    if (code == 0) return Status::Ok();
    if (code == 255) return {Error, "pod hostname not found"};
    const int64_t sat = state.RawRow(SPX_PLUGIN_NETOVERHEAD, SPX_NET_RAW_SATISFIED)[static_cast<size_t>(node)];
    const int64_t vio = state.RawRow(SPX_PLUGIN_NETOVERHEAD, SPX_NET_RAW_VIOLATED)[static_cast<size_t>(node)];
    return {Unschedulable, "Node " + node_name + " does not meet several network requirements from Workload dependencies: Satisfied: " +
                               std::to_string(sat) + " Violated: " + std::to_string(vio)};
  }
  std::pair<int64_t, Status> Score(CycleState& state, int32_t node) const {  // accumulated cost, :362-386
    return {state.RawRow(SPX_PLUGIN_NETOVERHEAD, SPX_NET_RAW_COST)[static_cast<size_t>(node)], Status::Ok()};
  }
  const NetworkOverhead* ScoreExtensions() const { return this; }
  Status NormalizeScore(CycleState& state, NodeScoreList& scores) const {  // :389-418
    const auto& row = state.ScoreRow(SPX_PLUGIN_NETOVERHEAD);
    for (auto& s : scores) s.score = row[static_cast<size_t>(s.node)];
    return Status::Ok();
  }
};

class CapacityScheduling {
 public:
  static constexpr const char* PluginName = "CapacityScheduling";
  const char* Name() const { return PluginName; }
  Status PreFilter(Engine& e, int64_t pod_row, const std::string& pod_ns, const std::string& pod_name) const {  // :208-283
    uint8_t code = 0;
    e.check(spx_fetch_prefilter(e.raw(), SPX_PLUGIN_CAPACITY, pod_row, pod_row + 1, &code));
    if (code == SPX_QUOTA_ST_OVER_MAX)
      return {Unschedulable, "Pod " + pod_ns + "/" + pod_name + " is rejected in PreFilter because ElasticQuota " + pod_ns + " is more than Max"};
    if (code == SPX_QUOTA_ST_OVER_MIN)
      return {Unschedulable, "Pod " + pod_ns + "/" + pod_name + " is rejected in PreFilter because total ElasticQuota used is more than min"};
    return Status::Ok();
  }
};

// queue sort over the flattened keys (spx_flatten_net_keys -> topo_order), topologicalsort.go:102-132
class TopologicalSort {
 public:
  static constexpr const char* PluginName = "TopologicalSort";
  TopologicalSort(const spx_pod_objects* pods, const int32_t* topo_order) : pods_(pods), order_(topo_order) {}
  const char* Name() const { return PluginName; }
  bool Less(int64_t p1, int64_t p2) const {
    uint8_t out = 0;
    if (spx_toposort_less(pods_, order_, 1, &p1, &p2, &out) != SPX_OK) throw std::runtime_error("spx_toposort_less");
    return out != 0;
  }

 private:
  const spx_pod_objects* pods_;
  const int32_t* order_;
};

}  // namespace spx::host
