// spx_engine.h — the engine's state and the helpers its translation units share (round 6: csrc/spx_engine.hip, one file of 3 900 lines
// in round 5, is now spx_engine.hip (lifecycle, options, parameters, evaluation, fetch), spx_uploads.hip (tables and deltas into HBM, the
// one-call loaders) and spx_commit.hip (the one-pod-at-a-time loops)).  Not part of the ABI.  The helpers sit in an anonymous namespace:
// every translation unit gets its own copy of the small ones it uses; the three thread-local error slots exist once (spx_engine.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include <atomic>
#include <mutex>
#include <thread>

#include "spx_internal.h"
#include "../host/nrt_streams.hpp"
#include "../host/parallel.hpp"

extern thread_local std::string g_create_error;          // spx_create's failure (no engine to hold it)
extern thread_local std::string tl_err;                  // this thread's last failure ...
extern thread_local const spx_engine* tl_err_engine;     // ... and on which engine

namespace {


struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool external = false;
};

}  // namespace

struct spx_engine {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool hold_ev0 = false;  // spx_decide times its preparatory spx_eval together with its own sweep
  bool skip_alloc_masked = false;  // spx_decide folds Allocatable's masked normalisation into its argmax kernel
  bool alloc_compact = false;      // k_alloc_prepare found the raw scores spanning less than 2^32 (AllocPrepArgs.rel is valid)
  bool timed = false;
  // last error: the engine's own copy (whoever failed last) under a lock; every thread also keeps the text of ITS last failure
  // (spx_last_error returns thread-local storage: concurrent readers may fail concurrently)
  mutable std::string err;
  mutable std::mutex err_mu;
  std::mutex raw_mu;  // spx_fetch_raw launches on the engine stream into one scratch row: concurrent callers take turns

  int64_t n_nodes = -1;
  int64_t n_pods = -1;
  int64_t row_stride = 0;

  // spx_set_option state (per engine; nothing is read from the environment)
  int64_t option[SPX_NUM_OPTIONS] = {spx::kRowPad, 0, 0, 0, 0, 0, 44, 1, 1, 375, 1, 1, 0, 1, 1, 1, 1, 1, 1};

  // params
  int32_t alloc_mode = SPX_MODE_LEAST;
  std::vector<int32_t> alloc_res{SPX_RES_MEMORY, SPX_RES_CPU};
  std::vector<int64_t> alloc_weight{1, 1 << 20};  // defaultResourcesToWeightMap resource_allocation.go:36
  spx_tlp_params tlp{40, 1000, 1.5};             // apis/config/v1/defaults.go:51-55
  spx_lvrb_params lvrb{1.0, 1.0};                // defaults.go:65-67
  int64_t plugin_weight[SPX_NUM_PLUGINS] = {1, 1, 1, 1, 1, 1, 1, 1, 1};

  // device tables
  DevBuf d_alloc, d_alloc_w, d_alloc_raw, d_alloc_norm, d_alloc_rel;
  int32_t alloc_n_res = 0;
  bool alloc_ready = false;  // raw/norm computed for the current table + params
  DevBuf d_cap_cpu, d_tlp_util, d_tlp_missing, d_tlp_valid;
  DevBuf d_lv_acpu, d_lv_amem, d_lv_cavg, d_lv_cstd, d_lv_mavg, d_lv_mstd, d_lv_flags;
  bool tri_nodes = false;
  DevBuf d_tlp_pod, d_lv_rcpu, d_lv_rmem;
  bool tri_pods = false;
  DevBuf d_raw_row;  // int64 [n_nodes] staging for spx_fetch_raw
  DevBuf d_lv_exact; // double [n_nodes][8] scratch of the LVRB fast kernel
  DevBuf d_lv_fast, d_tlp_fast;  // float32 per-node constants of the fast sweeps (recomputed per launch)
  DevBuf d_tlp_amb;              // k_tlp_amb_build's table: per pod value, the node tiles holding a cell the float32 sweep cannot prove
  DevBuf d_lv_amb;               // k_lvrb_amb_build's table
  bool lv_amb_built = false;     // ... and whether d_lv_exact / d_lv_fast / d_lv_amb still describe the LVRB node columns and parameters
  int64_t tlp_amb_geom[3] = {0, 0, 0}, lv_amb_geom[3] = {0, 0, 0};  // the tiling / stride / target the tables were built for (tlp_prepare compares)
  bool tlp_amb_built = false;    // ... and whether it still describes d_cap_cpu / d_tlp_util / d_tlp_missing / d_tlp_valid and the target (cleared by every writer of those)
  DevBuf d_commit;               // scratch of spx_commit_sequential
  DevBuf d_decide;               // per-tile partial decisions of spx_decide
  DevBuf d_stats;                // uint64 [SPX_NUM_PLUGINS]: cells re-evaluated by the fast sweeps' exact fallback

  // LowRiskOverCommitment (reads the LVRB node columns above as well)
  spx_lroc_params lroc{5, 0.5, 0.5};  // apis/config/v1/defaults.go:72-80
  DevBuf d_lroc_nreq_c, d_lroc_nreq_m, d_lroc_nlim_c, d_lroc_nlim_m, d_lroc_preq_c, d_lroc_preq_m, d_lroc_plim_c, d_lroc_plim_m, d_lroc_tab, d_lroc_podf;
  bool lroc_nodes = false, lroc_pods = false, lroc_tab_ready = false;
  bool lroc_nodes_exact = false, lroc_pods_exact = false, lv_alloc_exact = false;  // all values in [0, 2^52)
  // the float32 sweep's preconditions (kernels_lroc.hip): all values in [0, 2^47), limits not below requests
  bool lroc_nodes_f32 = false, lroc_pods_f32 = false, lv_alloc_f32 = false;

  // Peaks
  DevBuf d_pk_cap, d_pk_util, d_pk_valid, d_pk_k1, d_pk_k2, d_pk_pod, d_pk_min, d_pk_max, d_pk_rowc, d_pk_tab, d_pk_seg, d_pk_segn;
  bool peaks_nodes = false, peaks_pods = false;

  // NodeResourceTopologyMatch
  spx_nrt_params nrt_params{SPX_NRT_LEAST_ALLOCATED, 0, nullptr, nullptr};  // defaults.go:87-90
  int32_t nrt_n_res = 0;
  uint8_t nrt_slot_flags[SPX_NRT_MAX_RES] = {0};
  int64_t nrt_slot_weight[SPX_NRT_MAX_RES] = {0};
  bool nrt_slots = false, nrt_nodes = false, nrt_pods = false;
  DevBuf d_nrt_flags, d_nrt_max_numa, d_nrt_nz, d_nrt_zid, d_nrt_zp, d_nrt_avail, d_nrt_cost, d_nrt_minavg, d_nrt_np;
  DevBuf d_nrt_qos, d_nrt_nn, d_nrt_nctr, d_nrt_ckind, d_nrt_cpres, d_nrt_creq, d_nrt_ppres, d_nrt_preq;
  // float64 formulation of the NRT sweep (kernels_nrt_fast.hip): derived tables + whether its preconditions hold
  DevBuf d_nrt_fav, d_nrt_frc, d_nrt_frcv, d_nrt_fcpu, d_nrt_fbraw, d_nrt_frep, d_nrt_items, d_nrt_perm, d_nrt_ln;
  std::vector<double> nrt_wtab;  // [2^n_res][2]: sum of the weights of a slot subset, its biased reciprocal
  bool nrt_fast_slots = false, nrt_fast_nodes = false, nrt_fast_pods = false;
  DevBuf d_nrt_lnrec;      // LeastNUMANodes: the nodes' tables as one record each (scratch of a batch launch)
  DevBuf d_nrt_redo;       // BalancedAllocation: list of the cells the float32 Score launch leaves to the float64 form
  uint32_t nrt_redo_cap = 0;
  uint32_t nrt_big_nodes = ~0u, nrt_big_pods = ~0u;  // slots with a capacity / a request (Value() form) that float32 does not hold exactly
  // per slot, Value() form: OR and maximum of the zone capacities / of the requests in place (the packed float32 LeastAllocated Score's
  // preconditions, nrt_packed_score; delta uploads only ever add to them)
  using NrtQty = spx_host::NrtQty;
  NrtQty nrt_qty_nodes, nrt_qty_pods;
  int32_t nrt_slot_res[SPX_NRT_MAX_RES] = {0};       // canonical resource id of each slot (the packed Score's table slot is memory's)
  DevBuf d_nrt_pk_tab;                               // k_nrt_pk_tab_build's table ...
  bool nrt_pk_tab_built = false;                     // ... and whether it describes the zone capacities in place
  // pod equivalence classes (spx_upload_nrt_pods): rows whose NRT records agree in everything the sweep reads
  bool nrt_creq_valid = false;   // d_nrt_creq (read by the reference-arithmetic kernel only) holds this batch's column
  void* h_stage = nullptr;       // pinned staging of the blob uploads (DeltaBlob: node tables and deltas) and spx_load_trimaran_pods
  size_t h_stage_bytes = 0;
  void* h_items = nullptr;       // pinned staging of the NRT pod record stream, built in place (its own buffer: spx_load_nrt's node and pod halves run side by side)
  size_t h_items_bytes = 0;
  DevBuf d_delta;                // staged rows of a node-table delta (spx_update_*_nodes)
  DevBuf d_nrt_uniq, d_nrt_dups;  // int32 [n_uniq] representative rows, ascending; int32 [n_dups][2] (row, its representative)
  int64_t nrt_n_uniq = 0, nrt_n_dups = 0, nrt_n_tasks = 0;  // (d_nrt_dups: the pairs sorted by representative, then the copy tasks — expand_tasks)
  DevBuf d_nrt_rk, d_nrt_rk_off;  // rank-space Filter: the chunk stream of the listed rows (nrt_build_rank_stream) and its chunk offsets
  uint32_t nrt_rk_max_dwords = 0;  // largest chunk block; 0 = no stream (the float64 Filter runs)
  // which rows the stream lists: 1 = the class representatives (d_nrt_uniq), 2 = every row in order (sweeps without pod classes:
  // built when such a sweep first asks for it, nrt_rank_stream_all); 0 = none, -1 = the batch has no finite stream (> 3 app containers)
  int nrt_rk_kind = 0;
  DevBuf d_nrt_rk_first;          // [chunks + 1] list position of each chunk's first row (a chunk holds up to 32)
  uint32_t nrt_rk_chunks = 0;
  bool nrt_rk_all_narrow = false;  // every chunk keeps four zones' counts per register (the only layout the fused sweep has)
  DevBuf d_nrt_wsort, d_nrt_wrank;  // the fused walk's per-window sorted cell quantities and the cells' ranks (k_nrt_window_sort) ...
  bool nrt_wsort_built = false;     // ... and whether they describe the zone tables in place (cleared with nrt_pk_tab_built: every writer of the zone quantities)
  DevBuf d_nrt_fz;  // fused Filter + Score sweep: the packed Score items of the listed rows (k_nrt_fused_pack)
  // what d_nrt_fz was packed from: generation of the pod records / slot table (bumped by their uploads), the row list's kind, the table
  // slot, the buffer — a sweep whose key matches skips the pack launch
  uint64_t nrt_items_gen = 1;
  struct FzKey {
    uint64_t gen = 0;
    int kind = 0, tab_slot = -2;
    const void* buf = nullptr;
    bool operator==(const FzKey& o) const { return gen == o.gen && kind == o.kind && tab_slot == o.tab_slot && buf == o.buf; }
  } nrt_fz_key;
  int last_nrt_filter = 0;         // spx_nrt_filter_path
  DevBuf d_pk_uniq, d_pk_dups;    // the same for Peaks: classes of pods with equal cpu requests
  int64_t pk_n_uniq = 0, pk_n_dups = 0, pk_n_tasks = 0;
  bool pk_negative = false;  // a Peaks pod row with a negative cpu request (never from a v1.Pod): the interval estimate's bounds assume >= 0
  bool nrt_ln_ok = false;  // LeastNUMANodes tables can be built: every zone cost within [0, 255]
  bool nrt_ln_built = false;
  std::vector<int32_t> h_nrt_cost;  // [N][Z][Z] host copy of the zone costs, what build_ln_tab works from
  std::vector<uint8_t> h_nrt_nz;
  int32_t nrt_cpu_slot = -1;
  DevBuf status[SPX_NUM_PLUGINS];

  // NetworkOverhead / TopologicalSort
  bool net_nodes = false, net_topo = false, net_pods = false;
  int32_t net_n_regions = 0, net_n_zones = 0, net_n_classes = 0;
  int64_t net_max_cost = SPX_NET_MAX_COST, net_max_pairs = 0;  // bound of a row's accumulated cost (the sweep adds in int32)
  DevBuf d_net_region, d_net_zone, d_net_class, d_net_class16, d_net_cls_size, d_net_cls_region, d_net_cls_zone, d_net_rcost, d_net_zcost;
  bool net_class16 = false;
  DevBuf d_net_pod_key, d_net_key_flag, d_net_pair_ptr, d_net_pair_node, d_net_pair_max;
  // TopologicalSort keys
  DevBuf d_sort_prio, d_sort_ts, d_sort_group, d_sort_topo, d_sort_scratch;
  int64_t sort_n = 0;
  unsigned* h_sort_hist = nullptr;  // pinned

  // profile-level state
  DevBuf d_ext_status;  // caller's feasibility mask, stored as a status table (0 = feasible)
  bool ext_mask = false;
  DevBuf d_best;              // [score int64 P | node int32 P | ties int32 P | feasible int32 P], one allocation
  void* h_best = nullptr;     // pinned staging of the same layout: one D2H per spx_fetch_best
  size_t h_best_bytes = 0;
  bool best_valid = false;

  // CapacityScheduling.PreFilter
  bool quota = false;
  int32_t q_n_namespaces = 0;
  int64_t q_agg_used[SPX_QUOTA_SLOTS] = {0}, q_agg_min[SPX_QUOTA_SLOTS] = {0};
  uint32_t q_agg_used_present = 0, q_agg_min_present = 0;
  DevBuf d_q_pod_ns, d_q_pod_prio, d_q_pod_req, d_q_pod_reqp, d_q_has, d_q_used, d_q_max, d_q_maxp, d_q_other, d_q_otherp;
  DevBuf d_q_nom_ptr, d_q_nom_prio, d_q_nom_idx, d_q_nom_req, d_q_nom_reqp, d_q_status;
  DevBuf d_q_usedp, d_q_min, d_q_minp, d_q_agg;  // commit loop: Used key presence, Min per namespace, [8 aggregate used | presence]
  bool q_has_min = false;
  size_t q_n_nominated = 0;
  const int64_t* q_agg_dyn = nullptr;  // set while the sequential commit loop runs: k_quota reads the aggregate from the device
  // NetworkOverhead in the commit loop: per-pod effects + the workload pair lists rebuilt with room to grow
  std::vector<int32_t> h_pair_ptr, h_eff_ptr, h_eff_key;
  std::vector<uint8_t> h_key_flag;            // host copy of key_score_equally (spx_update_net_placed edits it)
  DevBuf d_net_pair_node2, d_net_pair_max2;   // the other half of the pair lists' ping-pong (spx_update_net_placed)
  std::vector<int64_t> h_eff_cost;
  DevBuf d_net_eff_ptr, d_net_eff_key, d_net_eff_cost, d_net_dyn_ptr, d_net_dyn_end, d_net_dyn_node, d_net_dyn_max;
  bool net_commit = false, net_dyn_active = false;
  int32_t net_n_keys = 0;
  DevBuf d_commit_save;  // backup of every table the commit loop mutates
  DevBuf d_coop_sync, d_coop_node, d_coop_max;  // cooperative commit kernel: granules + error flag, the workgroups' private pair lists
  double load_nrt_ms[6] = {0};  // stages of the last spx_load_nrt (spx_last_load_nrt_ms)
  bool in_commit_loop = false;  // commit_with_filters' per-pod launches are running on mutated zone tables (fill_nrt)
  int coop_gave_up = 0;      // cooperative commit launches that ended with a workgroup giving up (served by the per-pod loop instead)
  int last_commit_path = 0;  // what the last spx_commit_sequential ran: 1 one-workgroup trimaran chain, 2 per-pod launches, 3 cooperative kernel
  DevBuf d_row_counter;  // int64: the row the replayed per-pod graph works on
  const int64_t* row_indirect = nullptr;  // non-NULL while that graph is captured: sweeps read their row from the device

  DevBuf score[SPX_NUM_PLUGINS];
  int64_t score_rows[SPX_NUM_PLUGINS] = {0};
  int64_t score_stride[SPX_NUM_PLUGINS] = {0};
  uint32_t evaluated = 0;  // plugins with valid rows
  // what each plugin's table currently holds: the row range evaluated, and under which feasibility context — the Filter
  // plugins of that spx_eval call and the caller's mask generation — NormalizeScore-type plugins ran (upstream normalises over
  // the nodes that passed every Filter of the cycle, so a table is only meaningful together with that set)
  struct EvalInfo {
    int64_t begin = 0, end = 0;
    uint32_t filters = 0;
    uint64_t ext_gen = 0;
  } eval_info[SPX_NUM_PLUGINS];
  uint64_t ext_gen = 0;
};


namespace {


int fail(const spx_engine* e, int code, const std::string& msg) {
  if (e) {
    {
      std::lock_guard<std::mutex> g(e->err_mu);
      e->err = msg;
    }
    tl_err = msg;
    tl_err_engine = e;
  } else {
    g_create_error = msg;
  }
  return code;
}

#define SPX_HIP(e, call)                                                                          \
  do {                                                                                            \
    hipError_t _st = (call);                                                                      \
    if (_st != hipSuccess)                                                                        \
      return fail((e), SPX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st));          \
  } while (0)

int ensure(spx_engine* e, DevBuf& b, size_t bytes) {
  if (b.external) return fail(e, SPX_ERR_STATE, "internal: resize of an externally bound buffer");
  if (bytes == 0) bytes = 16;
  if (b.bytes >= bytes) return SPX_OK;
  if (b.p) SPX_HIP(e, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  SPX_HIP(e, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return SPX_OK;
}

int upload(spx_engine* e, DevBuf& b, const void* src, size_t bytes) {
  if (!src && bytes) return fail(e, SPX_ERR_ARG, "NULL column in table");  // an empty column (e.g. no resource slots) may be NULL
  int rc = ensure(e, b, bytes);
  if (rc) return rc;
  if (bytes) SPX_HIP(e, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, e->stream));
  return SPX_OK;
}

// every value in [0, 2^52): sums and differences of two such values are exact in float64
bool all_below_2p52(const int64_t* v, size_t n) {
  uint64_t acc = 0;
  for (size_t i = 0; i < n; ++i) acc |= static_cast<uint64_t>(v[i]);
  return (acc >> 52) == 0;
}
// every value in [0, 2^47): such a value, and the difference of two, is the sum of two float32 exactly (k_lroc_fast)
bool all_below_2p47(const int64_t* v, size_t n) {
  uint64_t acc = 0;
  for (size_t i = 0; i < n; ++i) acc |= static_cast<uint64_t>(v[i]);
  return (acc >> 47) == 0;
}
bool none_below(const int64_t* hi, const int64_t* lo, size_t n) {
  bool ok = true;
  for (size_t i = 0; i < n; ++i) ok = ok && hi[i] >= lo[i];
  return ok;
}

int set_nodes(spx_engine* e, int64_t n) {
  if (n <= 0) return fail(e, SPX_ERR_ARG, "n_nodes must be positive");
  if (e->n_nodes != -1 && e->n_nodes != n)
    return fail(e, SPX_ERR_STATE, "n_nodes differs from tables already uploaded (one snapshot per engine; destroy and re-create to change shape)");
  e->n_nodes = n;
  e->row_stride = spx::round_up(n, e->option[SPX_OPT_ROW_ALIGN]);
  return SPX_OK;
}

int set_pods(spx_engine* e, int64_t p) {
  if (p <= 0) return fail(e, SPX_ERR_ARG, "n_pods must be positive");
  if (e->n_pods != -1 && e->n_pods != p)
    return fail(e, SPX_ERR_STATE, "n_pods differs from tables already uploaded");
  e->n_pods = p;
  return SPX_OK;
}

int ensure_score_table(spx_engine* e, int plugin) {
  DevBuf& b = e->score[plugin];
  if (b.external) {
    if (e->score_rows[plugin] < e->n_pods || e->score_stride[plugin] < e->row_stride)
      return fail(e, SPX_ERR_STATE, "bound score table is smaller than n_pods x row_stride");
    return SPX_OK;
  }
  int rc = ensure(e, b, static_cast<size_t>(e->n_pods) * static_cast<size_t>(e->row_stride));
  if (rc) return rc;
  e->score_rows[plugin] = e->n_pods;
  e->score_stride[plugin] = e->row_stride;
  return SPX_OK;
}

constexpr uint32_t kFilterPlugins = (1u << SPX_PLUGIN_NRT) | (1u << SPX_PLUGIN_NETOVERHEAD);
// plugins whose NormalizeScore depends on the feasible set of the cycle
constexpr uint32_t kNormalizingPlugins = (1u << SPX_PLUGIN_ALLOCATABLE) | (1u << SPX_PLUGIN_NETOVERHEAD) | (1u << SPX_PLUGIN_PEAKS);

// rows [b, e) of `plugin` hold results of an spx_eval
int rows_evaluated(const spx_engine* e, int plugin, int64_t b, int64_t en) {
  const spx_engine::EvalInfo& i = e->eval_info[plugin];
  if (!(e->evaluated & (1u << plugin)) || b < i.begin || en > i.end)
    return fail(e, SPX_ERR_STATE, "rows requested have not been evaluated for this plugin (spx_eval covers [" + std::to_string(i.begin) + ", " +
                                      std::to_string(i.end) + "))");
  return SPX_OK;
}

int ensure_status_table(spx_engine* e, int plugin) {
  DevBuf& b = e->status[plugin];
  const size_t need = static_cast<size_t>(e->n_pods) * static_cast<size_t>(e->row_stride);
  if (b.external) {
    if (b.bytes < need) return fail(e, SPX_ERR_STATE, "bound status table is smaller than n_pods x row_stride");
    return SPX_OK;
  }
  return ensure(e, b, need);
}

int prepare_alloc(spx_engine* e) {
  if (e->alloc_ready) return SPX_OK;
  if (!e->d_alloc.p) return fail(e, SPX_ERR_STATE, "Allocatable: spx_upload_alloc_nodes not called");
  if (e->alloc_n_res != static_cast<int32_t>(e->alloc_res.size()))
    return fail(e, SPX_ERR_STATE, "Allocatable: uploaded table has a different resource count than the params");
  int rc = upload(e, e->d_alloc_w, e->alloc_weight.data(), e->alloc_weight.size() * sizeof(int64_t));
  if (rc) return rc;
  if ((rc = ensure(e, e->d_alloc_raw, static_cast<size_t>(e->n_nodes) * sizeof(int64_t)))) return rc;
  if ((rc = ensure(e, e->d_alloc_rel, static_cast<size_t>(e->row_stride + 4) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(e, e->d_alloc_norm, static_cast<size_t>(e->row_stride)))) return rc;
  spx::AllocPrepArgs a{};
  a.n_nodes = e->n_nodes;
  a.row_stride = e->row_stride;
  a.n_res = e->alloc_n_res;
  a.mode = e->alloc_mode;
  a.alloc = static_cast<const int64_t*>(e->d_alloc.p);
  a.weight = static_cast<const int64_t*>(e->d_alloc_w.p);
  a.raw = static_cast<int64_t*>(e->d_alloc_raw.p);
  a.rel = static_cast<uint32_t*>(e->d_alloc_rel.p);
  a.norm = static_cast<uint8_t*>(e->d_alloc_norm.p);
  spx::launch_alloc_prepare(a, e->stream);
  SPX_HIP(e, hipGetLastError());
  uint32_t compact = 0;  // once per node table: the flag the kernel leaves behind the offsets
  SPX_HIP(e, hipMemcpyAsync(&compact, static_cast<const uint32_t*>(e->d_alloc_rel.p) + e->row_stride, sizeof compact, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->alloc_compact = compact != 0;
  e->alloc_ready = true;
  return SPX_OK;
}

bool forced_reference(const spx_engine* e, int plugin) { return (e->option[SPX_OPT_REFERENCE_KERNELS] >> plugin) & 1; }

// the engine's options as the launch-level switches the kernel translation units read
uint32_t launch_opts(const spx_engine* e) {
  uint32_t o = 0;
  if (forced_reference(e, SPX_PLUGIN_TLP) || forced_reference(e, SPX_PLUGIN_LVRB)) o |= spx::kOptTrimaranExact;
  if (forced_reference(e, SPX_PLUGIN_NRT)) o |= spx::kOptNrtGeneric;
  if (forced_reference(e, SPX_PLUGIN_NETOVERHEAD)) o |= spx::kOptNetGeneric;
  if (e->option[SPX_OPT_NRT_SINGLE_LAUNCH]) o |= spx::kOptNrtSingleLaunch;
  if (e->option[SPX_OPT_COMMIT_FROM_MEMORY]) o |= spx::kOptCommitFromMemory;
  if (e->option[SPX_OPT_PEAKS_TILE] / 10 == 8) o |= spx::kOptPeaksWideA;
  if (e->option[SPX_OPT_PEAKS_TILE] % 10 == 8) o |= spx::kOptPeaksWideB;
  if (!e->option[SPX_OPT_TLP_AMB_TABLE]) o |= spx::kOptTlpNoAmbTable;
  if (e->option[SPX_OPT_PEAKS_ESTIMATE]) o |= spx::kOptPeaksEstimate;
  if (e->option[SPX_OPT_PEAKS_ESTIMATE] == 8) o |= spx::kOptPeaksEst8;
  return o;
}

bool lroc_exact53(const spx_engine* e) {
  return e->lroc_nodes_exact && e->lroc_pods_exact && e->lv_alloc_exact && !forced_reference(e, SPX_PLUGIN_LROC);
}
// the float32 sweep (k_lroc_fast) may run: exact columns, below 2^47, every limit at least its request (SetMaxLimits, resourcestats.go:227-231)
bool lroc_f32_ok(const spx_engine* e) {
  return lroc_exact53(e) && e->lroc_nodes_f32 && e->lroc_pods_f32 && e->lv_alloc_f32 && !e->option[SPX_OPT_LROC_FLOAT64];
}

void fill_lroc(const spx_engine* e, spx::LrocArgs& a) {
  a.n_nodes = e->n_nodes;
  a.row_stride = e->row_stride;
  a.alloc_cpu_milli = static_cast<const int64_t*>(e->d_lv_acpu.p);
  a.alloc_mem = static_cast<const int64_t*>(e->d_lv_amem.p);
  a.cpu_avg = static_cast<const double*>(e->d_lv_cavg.p);
  a.cpu_std = static_cast<const double*>(e->d_lv_cstd.p);
  a.mem_avg = static_cast<const double*>(e->d_lv_mavg.p);
  a.mem_std = static_cast<const double*>(e->d_lv_mstd.p);
  a.flags = static_cast<const uint8_t*>(e->d_lv_flags.p);
  a.node_req_cpu = static_cast<const int64_t*>(e->d_lroc_nreq_c.p);
  a.node_req_mem = static_cast<const int64_t*>(e->d_lroc_nreq_m.p);
  a.node_lim_cpu = static_cast<const int64_t*>(e->d_lroc_nlim_c.p);
  a.node_lim_mem = static_cast<const int64_t*>(e->d_lroc_nlim_m.p);
  a.pod_req_cpu = static_cast<const int64_t*>(e->d_lroc_preq_c.p);
  a.pod_req_mem = static_cast<const int64_t*>(e->d_lroc_preq_m.p);
  a.pod_lim_cpu = static_cast<const int64_t*>(e->d_lroc_plim_c.p);
  a.pod_lim_mem = static_cast<const int64_t*>(e->d_lroc_plim_m.p);
  a.sqrt_window = std::sqrt(static_cast<double>(e->lroc.smoothing_window_size));  // math.Pow(x, 0.5) = Sqrt(x)
  a.w_cpu = e->lroc.risk_limit_weight_cpu;
  a.w_mem = e->lroc.risk_limit_weight_mem;
  a.node_tab = static_cast<double*>(e->d_lroc_tab.p);
  a.exact53 = lroc_exact53(e) ? 1 : 0;
  a.pod_f32 = lroc_f32_ok(e) ? static_cast<const float*>(e->d_lroc_podf.p) : nullptr;
  a.n_pods_total = e->n_pods;
  a.stats = static_cast<unsigned long long*>(e->d_stats.p);
}

void fill_peaks(const spx_engine* e, spx::PeaksArgs& a) {
  a.opts = launch_opts(e);
  a.n_nodes = e->n_nodes;
  a.row_stride = e->row_stride;
  a.cap_cpu_milli = static_cast<const int64_t*>(e->d_pk_cap.p);
  a.cpu_util = static_cast<const double*>(e->d_pk_util.p);
  a.valid = static_cast<const uint8_t*>(e->d_pk_valid.p);
  a.k1 = static_cast<const double*>(e->d_pk_k1.p);
  a.k2 = static_cast<const double*>(e->d_pk_k2.p);
  a.pod_cpu_milli = static_cast<const int64_t*>(e->d_pk_pod.p);
  a.row_min = static_cast<int64_t*>(e->d_pk_min.p);
  a.row_max = static_cast<int64_t*>(e->d_pk_max.p);
  a.row_c = static_cast<float*>(e->d_pk_rowc.p);
  a.node_tab = static_cast<double*>(e->d_pk_tab.p);
}

void fill_trimaran(const spx_engine* e, spx::TrimaranArgs& a) {
  a.opts = launch_opts(e);
  a.row_ptr = e->row_indirect;
  a.n_nodes = e->n_nodes;
  a.row_stride = e->row_stride;
  a.alloc_norm = static_cast<const uint8_t*>(e->d_alloc_norm.p);
  a.cap_cpu_milli = static_cast<const int64_t*>(e->d_cap_cpu.p);
  a.tlp_cpu_util = static_cast<const double*>(e->d_tlp_util.p);
  a.tlp_missing_milli = static_cast<const int64_t*>(e->d_tlp_missing.p);
  a.tlp_valid = static_cast<const uint8_t*>(e->d_tlp_valid.p);
  a.tlp_pod_milli = static_cast<const int64_t*>(e->d_tlp_pod.p);
  a.tlp_target = static_cast<double>(e->tlp.target_utilization);
  a.lv_alloc_cpu_milli = static_cast<const int64_t*>(e->d_lv_acpu.p);
  a.lv_alloc_mem = static_cast<const int64_t*>(e->d_lv_amem.p);
  a.lv_cpu_avg = static_cast<const double*>(e->d_lv_cavg.p);
  a.lv_cpu_std = static_cast<const double*>(e->d_lv_cstd.p);
  a.lv_mem_avg = static_cast<const double*>(e->d_lv_mavg.p);
  a.lv_mem_std = static_cast<const double*>(e->d_lv_mstd.p);
  a.lv_flags = static_cast<const uint8_t*>(e->d_lv_flags.p);
  a.lv_req_cpu_milli = static_cast<const int64_t*>(e->d_lv_rcpu.p);
  a.lv_req_mem = static_cast<const int64_t*>(e->d_lv_rmem.p);
  a.lv_margin = e->lvrb.safe_variance_margin;
  a.lv_sensitivity = e->lvrb.safe_variance_sensitivity;
  a.stats = static_cast<unsigned long long*>(e->d_stats.p);
}

// The packed float32 form of LeastAllocated's Score launch (nrt_fast_device.h, score_least_packed) needs every weighted slot to be
// "small" — with 2^s the largest power of two dividing all its capacities and requests, capacity / 2^s <= 32768 and request / 2^s < 2^24 —
// or, one slot at most and not cpu, to go through k_nrt_pk_tab_build's table indexed by request / unit, unit = the largest power of
// two dividing all its requests.  false = the float64 form.
struct NrtPacked {
  uint32_t small_slots = 0;
  int32_t tab_slot = -1;
  uint32_t tab_kmax = 0, tab_words = 0;
  double tab_inv_unit = 1.0;
};
constexpr int64_t kNrtSmallCap = 32768;
bool nrt_packed_score(const spx_engine* e, NrtPacked* out) {
  *out = NrtPacked{};
  // (MostAllocated: the same float32 products serve x = 100 v / c as serve 100 - x; only the fused walk consumes the answer for that strategy)
  if (!e->option[SPX_OPT_NRT_PACKED_SCORE] || !e->nrt_nodes || !e->nrt_pods || e->in_commit_loop ||
      (e->nrt_params.strategy != SPX_NRT_LEAST_ALLOCATED && !(e->nrt_params.strategy == SPX_NRT_MOST_ALLOCATED && e->option[SPX_OPT_NRT_FUSED])))
    return false;
  int64_t wsum = 0;
  for (int i = 0; i < e->nrt_n_res && i < SPX_NRT_MAX_RES; ++i) {
    if (e->nrt_slot_weight[i] < 0) return false;
    wsum += e->nrt_slot_weight[i];
  }
  if (wsum > spx::kNrtPkMaxWeightSum) return false;
  auto low_zeros = [](uint64_t bits) { return bits ? __builtin_ctzll(bits) : 63; };
  for (int i = 0; i < e->nrt_n_res && i < SPX_NRT_MAX_RES; ++i) {
    if (e->nrt_slot_weight[i] == 0) continue;  // contributes 0 whatever its resource score
    const uint64_t pod_bits = e->nrt_qty_pods.bits[i];
    if (pod_bits == 0) {  // no request but zeros: the resource score is 100 or 0 in both forms
      out->small_slots |= 1u << i;
      continue;
    }
    const int s = low_zeros(pod_bits | e->nrt_qty_nodes.bits[i]);
    if ((e->nrt_qty_nodes.most[i] >> s) <= kNrtSmallCap && (e->nrt_qty_pods.most[i] >> s) < (int64_t{1} << 24)) {
      out->small_slots |= 1u << i;
      continue;
    }
    const int su = low_zeros(pod_bits);
    const int64_t kmax = e->nrt_qty_pods.most[i] >> su;
    const size_t words = static_cast<size_t>((e->n_nodes + 255) / 256 + 31) / 32;
    if (out->tab_slot >= 0 || i == e->nrt_cpu_slot || kmax > spx::kNrtPkTabMaxK || (static_cast<size_t>(kmax) + 1) * words * 4 > spx::kNrtPkTabMaxBytes)
      return false;
    out->tab_slot = i, out->tab_kmax = static_cast<uint32_t>(kmax), out->tab_words = static_cast<uint32_t>(words);
    out->tab_inv_unit = std::ldexp(1.0, -su);
  }
  return true;
}

void fill_nrt(const spx_engine* e, spx::NrtArgs& na) {
  na.opts = launch_opts(e);
  na.row_ptr = e->row_indirect;
  na.n_nodes = e->n_nodes;
  na.n_pods = e->n_pods;
  na.row_stride = e->row_stride;
  na.n_res = e->nrt_n_res;
  na.strategy = e->nrt_params.strategy;
  std::memcpy(na.slot_flags, e->nrt_slot_flags, sizeof na.slot_flags);
  std::memcpy(na.slot_weight, e->nrt_slot_weight, sizeof na.slot_weight);
  na.flags = static_cast<const uint8_t*>(e->d_nrt_flags.p);
  na.max_numa = static_cast<const int32_t*>(e->d_nrt_max_numa.p);
  na.n_zones = static_cast<const uint8_t*>(e->d_nrt_nz.p);
  na.zone_id = static_cast<const uint8_t*>(e->d_nrt_zid.p);
  na.zone_present = static_cast<const uint8_t*>(e->d_nrt_zp.p);
  na.zone_avail = static_cast<const int64_t*>(e->d_nrt_avail.p);
  na.zone_cost = static_cast<const int32_t*>(e->d_nrt_cost.p);
  na.min_avg = static_cast<const float*>(e->d_nrt_minavg.p);
  na.node_present = static_cast<const uint8_t*>(e->d_nrt_np.p);
  na.qos = static_cast<const uint8_t*>(e->d_nrt_qos.p);
  na.non_native = static_cast<const uint8_t*>(e->d_nrt_nn.p);
  na.n_ctr = static_cast<const uint8_t*>(e->d_nrt_nctr.p);
  na.ctr_kind = static_cast<const uint8_t*>(e->d_nrt_ckind.p);
  na.ctr_present = static_cast<const uint8_t*>(e->d_nrt_cpres.p);
  na.ctr_req = static_cast<const int64_t*>(e->d_nrt_creq.p);
  na.pod_present = static_cast<const uint8_t*>(e->d_nrt_ppres.p);
  na.pod_req = static_cast<const int64_t*>(e->d_nrt_preq.p);
  na.fast = e->nrt_fast_slots && e->nrt_fast_nodes && e->nrt_fast_pods;
  na.cpu_slot = e->nrt_cpu_slot;
  for (int i = 0; i < SPX_NRT_MAX_RES; ++i) na.slot_weight_f[i] = static_cast<double>(e->nrt_slot_weight[i]);
  na.f_av = static_cast<const double*>(e->d_nrt_fav.p);
  na.f_rc = static_cast<const double*>(e->d_nrt_frc.p);
  na.f_rcv = static_cast<const double*>(e->d_nrt_frcv.p);
  na.f_cpu = static_cast<const double*>(e->d_nrt_fcpu.p);
  na.f_braw = static_cast<const double*>(e->d_nrt_fbraw.p);
  na.f_rep = static_cast<const uint8_t*>(e->d_nrt_frep.p);
  na.pod_items = static_cast<const uint32_t*>(e->d_nrt_items.p);
  na.perm = static_cast<const int32_t*>(e->d_nrt_perm.p);
  na.stats = static_cast<unsigned long long*>(e->d_stats.p);
  na.exact32_slots = ~(e->nrt_big_nodes | e->nrt_big_pods);
  na.pk_mode = 0, na.pk_tab_slot = -1;  // (spx_eval's NRT section turns the packed Score on)
  // inside the per-pod commit loop k_commit_apply subtracts requests from the zone table: "every quantity is a float32 value" is
  // not closed under subtraction (2^30 and 1 are, 2^30 - 1 is not) and the masks above describe the uploaded tables, so
  // BalancedAllocation's float32 "request > capacity" test gives way to the undecided -> float64 redo route there
  if (e->in_commit_loop) na.exact32_slots = 0;
  na.redo_list = static_cast<uint32_t*>(e->d_nrt_redo.p);
  na.redo_cap = e->nrt_redo_cap;
  na.ln_tab = (e->nrt_ln_ok && e->nrt_ln_built) ? static_cast<const uint32_t*>(e->d_nrt_ln.p) : nullptr;
  na.ln_const = na.ln_tab ? na.ln_tab + static_cast<size_t>(spx::make_ln_layout().rows) * static_cast<size_t>(e->n_nodes) : nullptr;
}

// the reference-arithmetic NRT kernel's request column, when the coming launch may take that kernel and the batch did not ship it
int ensure_nrt_creq(spx_engine* e) {
  const bool fast = e->nrt_fast_slots && e->nrt_fast_nodes && e->nrt_fast_pods && !forced_reference(e, SPX_PLUGIN_NRT) &&
                    !(e->nrt_params.strategy == SPX_NRT_LEAST_NUMA_NODES && !(e->nrt_ln_ok && e->nrt_ln_built));
  if (fast || e->nrt_creq_valid) return SPX_OK;
  const size_t bytes = static_cast<size_t>(e->n_pods) * SPX_NRT_MAX_CTRS * static_cast<size_t>(e->nrt_n_res) * sizeof(int64_t);
  int rc = ensure(e, e->d_nrt_creq, bytes);
  if (rc) return rc;
  spx::launch_nrt_creq_from_items(static_cast<const uint32_t*>(e->d_nrt_items.p), e->nrt_n_res, e->n_pods, static_cast<int64_t*>(e->d_nrt_creq.p), e->stream);
  SPX_HIP(e, hipGetLastError());
  e->nrt_creq_valid = true;
  return SPX_OK;
}

using spx_host::kNrtFastLimit;
using spx_host::kNrtWeightLimit;
using spx_host::nrt_biased_rcp;
using spx_host::nrt_exact_f32;
using spx_host::nrt_fast_qty;
using spx_host::nrt_value_of;
using spx_host::nrt_build_classes;
using spx_host::nrt_build_items;
using spx_host::nrt_build_rank_stream;

void fill_net(const spx_engine* e, spx::NetArgs& g) {
  g.opts = launch_opts(e);
  g.row_ptr = e->row_indirect;
  g.n_nodes = e->n_nodes;
  g.row_stride = e->row_stride;
  g.n_regions = e->net_n_regions;
  g.n_zones = e->net_n_zones;
  g.n_classes = e->net_n_classes;
  g.region = static_cast<const int32_t*>(e->d_net_region.p);
  g.zone = static_cast<const int32_t*>(e->d_net_zone.p);
  g.node_class = static_cast<const int32_t*>(e->d_net_class.p);
  g.node_class16 = e->net_class16 ? static_cast<const uint16_t*>(e->d_net_class16.p) : nullptr;
  g.cls_size = static_cast<const int32_t*>(e->d_net_cls_size.p);
  g.cls_region = static_cast<const int32_t*>(e->d_net_cls_region.p);
  g.cls_zone = static_cast<const int32_t*>(e->d_net_cls_zone.p);
  g.region_cost = static_cast<const int32_t*>(e->d_net_rcost.p);
  g.zone_cost = static_cast<const int32_t*>(e->d_net_zcost.p);
  g.pod_key = static_cast<const int32_t*>(e->d_net_pod_key.p);
  g.key_flag = static_cast<const uint8_t*>(e->d_net_key_flag.p);
  g.pair_ptr = static_cast<const int32_t*>(e->d_net_pair_ptr.p);
  g.pair_node = static_cast<const int32_t*>(e->d_net_pair_node.p);
  g.pair_max = static_cast<const int64_t*>(e->d_net_pair_max.p);
  if (e->net_dyn_active) {  // sequential commit: lists with slack that grow as pods are bound
    g.pair_ptr = static_cast<const int32_t*>(e->d_net_dyn_ptr.p);
    g.pair_end = static_cast<const int32_t*>(e->d_net_dyn_end.p);
    g.pair_node = static_cast<const int32_t*>(e->d_net_dyn_node.p);
    g.pair_max = static_cast<const int64_t*>(e->d_net_dyn_max.p);
  }
}

// host [N][inner] -> device [inner][N] so that lane = node reads coalesce
template <typename T>
int upload_transposed(spx_engine* e, DevBuf& b, const T* src, int64_t n, int64_t inner) {
  if (!src) return fail(e, SPX_ERR_ARG, "NULL column in table");
  std::vector<T> tmp(static_cast<size_t>(n) * static_cast<size_t>(inner));
  spx_host::parallel_rows(n, [&](int64_t row0, int64_t row1) {
    for (int64_t i = row0; i < row1; ++i)
      for (int64_t k = 0; k < inner; ++k) tmp[static_cast<size_t>(k) * n + i] = src[static_cast<size_t>(i) * inner + k];
  }, 2048);
  int rc = upload(e, b, tmp.data(), tmp.size() * sizeof(T));
  if (rc) return rc;
  SPX_HIP(e, hipStreamSynchronize(e->stream));  // tmp dies at scope exit
  return SPX_OK;
}

// The (row, representative) pairs of a pod batch's equivalence classes as launch_rows_expand reads them: sorted by representative (then row), and
// behind them the copy tasks — (first pair, count <= kRowsExpandFan) per run of pairs with one representative — so that a workgroup reads a
// representative's row once for up to eight copies.  Returns the task count; `dups` = [pairs | tasks].
inline int64_t expand_tasks(std::vector<int32_t>& dups, int64_t n_rows) {
  const size_t n = dups.size() / 2;
  // counting sort by representative (the pairs arrive in ascending order of the copied row, and stay so inside a representative's run)
  std::vector<int32_t> at(static_cast<size_t>(n_rows) + 1, 0), sorted(2 * n);
  for (size_t i = 0; i < n; ++i) ++at[static_cast<size_t>(dups[2 * i + 1]) + 1];
  for (int64_t r = 0; r < n_rows; ++r) at[static_cast<size_t>(r) + 1] += at[static_cast<size_t>(r)];
  for (size_t i = 0; i < n; ++i) {
    const size_t k = static_cast<size_t>(at[static_cast<size_t>(dups[2 * i + 1])]++);
    sorted[2 * k] = dups[2 * i], sorted[2 * k + 1] = dups[2 * i + 1];
  }
  dups.swap(sorted);
  int64_t tasks = 0;
  for (size_t i = 0; i < n;) {
    size_t j = i;
    while (j < n && j - i < static_cast<size_t>(spx::kRowsExpandFan) && dups[2 * j + 1] == dups[2 * i + 1]) ++j;
    dups.push_back(static_cast<int32_t>(i)), dups.push_back(static_cast<int32_t>(j - i));
    ++tasks;
    i = j;
  }
  return tasks;
}

}  // namespace

// defined in spx_uploads.hip, used by the evaluation and the commit loops (not exported)
extern "C" {
__attribute__((visibility("hidden"))) int build_ln_tab(spx_engine* e);
__attribute__((visibility("hidden"))) int nrt_rank_stream(spx_engine* e, int kind);
// defined with spx_decide (spx_engine.hip), also the commit loops' per-pod step
__attribute__((visibility("hidden"))) int decide_masked(spx_engine* e, uint32_t eval_mask, uint32_t score_mask, int64_t row_begin, int64_t row_end, bool* done);
}
