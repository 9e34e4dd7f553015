// spx_engine.hip — the C-ABI engine of libspx.so: device-resident SoA tables, result tables in
// HBM, kernel dispatch on one HIP stream, row-granular fetch.  See include/spx.h for the contract.
//
// There is deliberately no CPU fallback: spx_create() fails with SPX_ERR_NOGPU when no HIP device
// is usable, and every compute entry point needs an engine.
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
#include "../host/parallel.hpp"

namespace {

thread_local std::string g_create_error;

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
  struct NrtQty {
    uint64_t bits[SPX_NRT_MAX_RES] = {0};
    int64_t most[SPX_NRT_MAX_RES] = {0};
    void add(int r, int64_t v) { bits[r] |= static_cast<uint64_t>(v), most[r] = v > most[r] ? v : most[r]; }
    void merge(const NrtQty& o) {
      for (int r = 0; r < SPX_NRT_MAX_RES; ++r) bits[r] |= o.bits[r], most[r] = o.most[r] > most[r] ? o.most[r] : most[r];
    }
  };
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
  int64_t nrt_n_uniq = 0, nrt_n_dups = 0;
  DevBuf d_nrt_rk, d_nrt_rk_off;  // rank-space Filter: the chunk stream of the listed rows (nrt_build_rank_stream) and its chunk offsets
  uint32_t nrt_rk_max_dwords = 0;  // largest chunk block; 0 = no stream (the float64 Filter runs)
  // which rows the stream lists: 1 = the class representatives (d_nrt_uniq), 2 = every row in order (sweeps without pod classes:
  // built when such a sweep first asks for it, nrt_rank_stream_all); 0 = none, -1 = the batch has no finite stream (> 3 app containers)
  int nrt_rk_kind = 0;
  DevBuf d_nrt_rk_first;          // [chunks + 1] list position of each chunk's first row (a chunk holds up to 32)
  uint32_t nrt_rk_chunks = 0;
  bool nrt_rk_all_narrow = false;  // every chunk keeps four zones' counts per register (the only layout the fused sweep has)
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
  int64_t pk_n_uniq = 0, pk_n_dups = 0;
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
int decide_masked(spx_engine* e, uint32_t eval_mask, uint32_t score_mask, int64_t row_begin, int64_t row_end, bool* done);  // defined with spx_decide
}

namespace {

thread_local std::string tl_err;              // this thread's last failure ...
thread_local const spx_engine* tl_err_engine = nullptr;  // ... and on which engine

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
  a.pod_f64 = (a.exact53 && !e->option[SPX_OPT_LROC_FLOAT64]) ? static_cast<const double*>(e->d_lroc_podf.p) : nullptr;
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

// quantities the float64 NRT kernel may hold exactly, with room for x100 and the reciprocal trick
constexpr int64_t kNrtFastLimit = int64_t{1} << 42;
constexpr int64_t kNrtWeightLimit = int64_t{1} << 20;  // sum of the NRT scoring weights the float64 formulation accepts
inline bool nrt_fast_qty(int64_t v) { return v >= 0 && v < kNrtFastLimit; }
// RN(1/v) * (1 + 2^-49): floor(num * rc) == num / v for 0 <= num <= 101 * v, 0 < v < 2^42 (kernels_nrt_fast.hip)
inline double nrt_biased_rcp(double v) { return v > 0.0 ? (1.0 / v) * (1.0 + 0x1p-49) : 0.0; }
inline int64_t nrt_value_of(bool is_cpu, int64_t q) { return is_cpu ? (q + 999) / 1000 : q; }
// a quantity the float32 BalancedAllocation Score holds exactly: below 2^24, or any integer whose float32 image is itself (hugepage
// and device-memory quantities are small multiples of a power of two: 3 x 2^30 is as exact in float32 as 3).  Slots whose requests
// and capacities are all of that kind compare "request > capacity" exactly; the others (memory in bytes) are undecided near equality
inline bool nrt_exact_f32(double v) { return v >= 0.0 && v < 9.2e18 && static_cast<double>(static_cast<float>(v)) == v; }

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

}  // namespace

extern "C" {

int spx_abi_version(void) { return 1; }

const char* spx_last_error(const spx_engine* e) {
  if (!e) return g_create_error.c_str();
  if (tl_err_engine != e) {  // this thread has not failed on this engine: hand out a private copy of the engine's last message
    std::lock_guard<std::mutex> g(e->err_mu);
    tl_err = e->err;
    tl_err_engine = e;
  }
  return tl_err.c_str();
}

int spx_create(int device_id, spx_engine** out) {
  if (!out) return fail(nullptr, SPX_ERR_ARG, "out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t st = hipGetDeviceCount(&count);
  if (st != hipSuccess || count <= 0)
    return fail(nullptr, SPX_ERR_NOGPU, std::string("no HIP device available (") + hipGetErrorString(st) +
                                            "); libspx has no CPU fallback");
  if (device_id < 0 || device_id >= count) return fail(nullptr, SPX_ERR_ARG, "device_id out of range");
  spx_engine* e = new spx_engine();
  e->device = device_id;
  if ((st = hipSetDevice(device_id)) != hipSuccess || (st = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking)) != hipSuccess ||
      (st = hipEventCreate(&e->ev0)) != hipSuccess || (st = hipEventCreate(&e->ev1)) != hipSuccess) {
    std::string msg = std::string("engine init: ") + hipGetErrorString(st);
    delete e;
    return fail(nullptr, SPX_ERR_HIP, msg);
  }
  e->stream = e->own_stream;
  if ((st = hipMalloc(&e->d_stats.p, spx::kStatBytes)) != hipSuccess || (st = hipMemset(e->d_stats.p, 0, spx::kStatBytes)) != hipSuccess) {
    std::string msg = std::string("engine init: ") + hipGetErrorString(st);
    delete e;
    return fail(nullptr, SPX_ERR_HIP, msg);
  }
  e->d_stats.bytes = spx::kStatBytes;
  *out = e;
  return SPX_OK;
}

int spx_destroy(spx_engine* e) {
  if (!e) return SPX_OK;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  DevBuf* bufs[] = {&e->d_alloc,   &e->d_alloc_w,  &e->d_alloc_raw, &e->d_alloc_norm, &e->d_alloc_rel, &e->d_cap_cpu, &e->d_tlp_util,
                    &e->d_tlp_missing, &e->d_tlp_valid, &e->d_lv_acpu, &e->d_lv_amem, &e->d_lv_cavg, &e->d_lv_cstd,
                    &e->d_lv_mavg, &e->d_lv_mstd,  &e->d_lv_flags,  &e->d_tlp_pod,    &e->d_lv_rcpu, &e->d_lv_rmem,
                    &e->d_raw_row,   &e->d_lv_exact, &e->d_lv_fast, &e->d_tlp_fast, &e->d_tlp_amb, &e->d_lv_amb, &e->d_nrt_pk_tab, &e->d_commit, &e->d_nrt_flags, &e->d_nrt_max_numa, &e->d_nrt_nz, &e->d_nrt_zid, &e->d_nrt_zp,
                    &e->d_nrt_avail, &e->d_nrt_cost,  &e->d_nrt_minavg, &e->d_nrt_np,    &e->d_nrt_qos, &e->d_nrt_nn,
                    &e->d_nrt_nctr,  &e->d_nrt_ckind, &e->d_nrt_cpres,  &e->d_nrt_creq,  &e->d_nrt_ppres, &e->d_nrt_preq,
                    &e->d_nrt_frcv, &e->d_nrt_fav,   &e->d_nrt_frc,   &e->d_nrt_fcpu,   &e->d_nrt_frep,  &e->d_nrt_items, &e->d_nrt_perm, &e->d_nrt_ln, &e->d_nrt_fbraw, &e->d_nrt_redo,
                    &e->d_net_region, &e->d_net_zone, &e->d_net_class, &e->d_net_class16, &e->d_net_cls_size, &e->d_net_cls_region, &e->d_net_cls_zone,
                    &e->d_net_rcost, &e->d_net_zcost, &e->d_net_pod_key, &e->d_net_key_flag, &e->d_net_pair_ptr,
                    &e->d_net_pair_node, &e->d_net_pair_max, &e->d_q_pod_ns, &e->d_q_pod_prio, &e->d_q_pod_req, &e->d_q_pod_reqp,
                    &e->d_q_has, &e->d_q_used, &e->d_q_max, &e->d_q_maxp, &e->d_q_other, &e->d_q_otherp, &e->d_q_nom_ptr,
                    &e->d_q_nom_prio, &e->d_q_nom_idx, &e->d_q_nom_req, &e->d_q_nom_reqp, &e->d_q_status, &e->d_ext_status,
                    &e->d_q_usedp, &e->d_q_min, &e->d_q_minp, &e->d_q_agg, &e->d_net_eff_ptr, &e->d_net_eff_key, &e->d_net_eff_cost, &e->d_net_dyn_ptr,
                    &e->d_net_dyn_end, &e->d_net_dyn_node, &e->d_net_dyn_max, &e->d_commit_save, &e->d_row_counter, &e->d_coop_sync, &e->d_coop_node, &e->d_coop_max,
                    &e->d_sort_prio, &e->d_sort_ts, &e->d_sort_group, &e->d_sort_topo, &e->d_sort_scratch,
                    &e->d_best, &e->d_stats, &e->d_decide, &e->d_lroc_nreq_c, &e->d_lroc_nreq_m, &e->d_lroc_nlim_c, &e->d_lroc_nlim_m,
                    &e->d_lroc_preq_c, &e->d_lroc_preq_m, &e->d_lroc_plim_c, &e->d_lroc_plim_m, &e->d_lroc_tab, &e->d_lroc_podf,
                    &e->d_pk_cap, &e->d_pk_util, &e->d_pk_valid, &e->d_pk_k1, &e->d_pk_k2, &e->d_pk_pod, &e->d_pk_min, &e->d_pk_max, &e->d_pk_rowc, &e->d_pk_tab, &e->d_pk_seg, &e->d_pk_segn,
                    &e->d_nrt_uniq, &e->d_nrt_dups, &e->d_pk_uniq, &e->d_pk_dups, &e->d_delta, &e->d_nrt_lnrec, &e->d_net_pair_node2, &e->d_net_pair_max2, &e->d_nrt_rk, &e->d_nrt_rk_off, &e->d_nrt_rk_first, &e->d_nrt_fz};
  for (DevBuf* b : bufs)
    if (b->p && !b->external) (void)hipFree(b->p);
  for (int i = 0; i < SPX_NUM_PLUGINS; ++i) {
    if (e->score[i].p && !e->score[i].external) (void)hipFree(e->score[i].p);
    if (e->status[i].p && !e->status[i].external) (void)hipFree(e->status[i].p);
  }
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->h_best) (void)hipHostFree(e->h_best);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->h_items) (void)hipHostFree(e->h_items);
  if (e->h_sort_hist) (void)hipHostFree(e->h_sort_hist);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
  return SPX_OK;
}

int spx_set_stream(spx_engine* e, void* hip_stream) {
  if (!e) return SPX_ERR_ARG;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : e->own_stream;
  return SPX_OK;
}

int spx_get_stream(spx_engine* e, void** hip_stream) {
  if (!e || !hip_stream) return SPX_ERR_ARG;
  *hip_stream = static_cast<void*>(e->stream);
  return SPX_OK;
}

int spx_set_option(spx_engine* e, int option, int64_t value) {
  if (!e) return SPX_ERR_ARG;
  switch (option) {
    case SPX_OPT_ROW_ALIGN:
      if (value < spx::kRowAlign || value % spx::kRowAlign || value > 4096) return fail(e, SPX_ERR_ARG, "SPX_OPT_ROW_ALIGN: a multiple of 16 in [16, 4096]");
      if (e->n_nodes != -1) return fail(e, SPX_ERR_STATE, "SPX_OPT_ROW_ALIGN must be set before the first table upload");
      break;
    case SPX_OPT_REFERENCE_KERNELS:
      if (value < 0 || value >= (int64_t{1} << SPX_NUM_PLUGINS)) return fail(e, SPX_ERR_ARG, "SPX_OPT_REFERENCE_KERNELS: a mask of plugin ids");
      if (((value ^ e->option[option]) >> SPX_PLUGIN_LROC) & 1) e->lroc_tab_ready = false;
      break;
    case SPX_OPT_LROC_FLOAT64:
    case SPX_OPT_DECIDE_UNFUSED:
    case SPX_OPT_NRT_SINGLE_LAUNCH:
    case SPX_OPT_COMMIT_FROM_MEMORY:
    case SPX_OPT_NRT_POD_CLASSES:
    case SPX_OPT_PEAKS_POD_CLASSES:
    case SPX_OPT_COMMIT_COOP:
    case SPX_OPT_NRT_RANK_FILTER:
    case SPX_OPT_ROW_WORKGROUP:
    case SPX_OPT_TLP_AMB_TABLE:
    case SPX_OPT_NRT_PACKED_SCORE:
    case SPX_OPT_NET_ALLOC_FUSED:
    case SPX_OPT_NRT_RANK_NARROW:
    case SPX_OPT_NRT_FUSED:
      if (value != 0 && value != 1) return fail(e, SPX_ERR_ARG, "option takes 0 or 1");
      break;
    case SPX_OPT_NRT_LN_LIST_PERMILLE:
      if (value < 1 || value > 1000) return fail(e, SPX_ERR_ARG, "SPX_OPT_NRT_LN_LIST_PERMILLE: 1..1000");
      break;
    case SPX_OPT_PEAKS_ESTIMATE:
      if (value != 0 && value != 1 && value != 8) return fail(e, SPX_ERR_ARG, "SPX_OPT_PEAKS_ESTIMATE: 0, 1 or 8");
      break;
    case SPX_OPT_PEAKS_TILE:
      if (value != 44 && value != 84 && value != 48 && value != 88) return fail(e, SPX_ERR_ARG, "SPX_OPT_PEAKS_TILE: 44, 84, 48 or 88");
      break;
    default:
      return fail(e, SPX_ERR_ARG, "unknown option");
  }
  e->option[option] = value;
  return SPX_OK;
}

int spx_nrt_pod_classes(const spx_engine* e, int64_t* n_unique, int64_t* n_copies) {
  if (!e || !n_unique || !n_copies) return SPX_ERR_ARG;
  *n_copies = e->nrt_pods ? e->nrt_n_dups : 0;
  *n_unique = e->nrt_pods ? e->n_pods - *n_copies : 0;
  return SPX_OK;
}

int spx_peaks_pod_classes(const spx_engine* e, int64_t* n_unique, int64_t* n_copies) {
  if (!e || !n_unique || !n_copies) return SPX_ERR_ARG;
  *n_copies = e->peaks_pods ? e->pk_n_dups : 0;
  *n_unique = e->peaks_pods ? e->n_pods - *n_copies : 0;
  return SPX_OK;
}

int spx_get_option(const spx_engine* e, int option, int64_t* value) {
  if (!e || !value || option < 0 || option >= SPX_NUM_OPTIONS) return SPX_ERR_ARG;
  *value = e->option[option];
  return SPX_OK;
}

int spx_set_allocatable_params(spx_engine* e, const spx_allocatable_params* p) {
  if (!e || !p) return SPX_ERR_ARG;
  if (p->mode != SPX_MODE_LEAST && p->mode != SPX_MODE_MOST) return fail(e, SPX_ERR_ARG, "invalid mode");
  if (p->n_res <= 0) return fail(e, SPX_ERR_ARG, "n_res must be positive");
  for (int32_t r = 0; r < p->n_res; ++r) {
    if (p->weight[r] <= 0) {  // validateResources allocatable.go:53-61
      char buf[160];
      std::snprintf(buf, sizeof buf, "resource Weight of %d should be a positive value, got %lld", p->res[r],
                    static_cast<long long>(p->weight[r]));
      return fail(e, SPX_ERR_ARG, buf);
    }
  }
  e->alloc_mode = p->mode;
  e->alloc_res.assign(p->res, p->res + p->n_res);
  e->alloc_weight.assign(p->weight, p->weight + p->n_res);
  e->alloc_ready = false;
  return SPX_OK;
}

int spx_set_tlp_params(spx_engine* e, const spx_tlp_params* p) {
  if (!e || !p) return SPX_ERR_ARG;
  e->tlp = *p;
  e->tlp_amb_built = false;  // the table depends on the target utilisation
  return SPX_OK;
}

int spx_set_lvrb_params(spx_engine* e, const spx_lvrb_params* p) {
  if (!e || !p) return SPX_ERR_ARG;
  e->lvrb = *p;
  e->lv_amb_built = false;  // sigma (margin, sensitivity) is inside the per-node constants and the table
  return SPX_OK;
}

int spx_set_plugin_weights(spx_engine* e, const int64_t* weights) {
  if (!e || !weights) return SPX_ERR_ARG;
  std::memcpy(e->plugin_weight, weights, sizeof e->plugin_weight);
  return SPX_OK;
}

int spx_upload_alloc_nodes(spx_engine* e, const spx_alloc_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_nodes(e, t->n_nodes);
  if (rc) return rc;
  if (t->n_res <= 0) return fail(e, SPX_ERR_ARG, "n_res must be positive");
  rc = upload(e, e->d_alloc, t->alloc, static_cast<size_t>(t->n_res) * static_cast<size_t>(t->n_nodes) * sizeof(int64_t));
  if (rc) return rc;
  e->alloc_n_res = t->n_res;
  e->alloc_ready = false;
  SPX_HIP(e, hipStreamSynchronize(e->stream));  // host columns are only borrowed for the call
  return SPX_OK;
}

int spx_upload_trimaran_nodes(spx_engine* e, const spx_trimaran_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_nodes(e, t->n_nodes);
  if (rc) return rc;
  const size_t n = static_cast<size_t>(t->n_nodes);
  e->tlp_amb_built = e->lv_amb_built = false;  // (before the first column changes: a failed upload must not leave tables that describe the old ones)
  if ((rc = upload(e, e->d_cap_cpu, t->cap_cpu_milli, n * 8))) return rc;
  if ((rc = upload(e, e->d_tlp_util, t->tlp_cpu_util, n * 8))) return rc;
  if ((rc = upload(e, e->d_tlp_missing, t->tlp_missing_milli, n * 8))) return rc;
  if ((rc = upload(e, e->d_tlp_valid, t->tlp_valid, n))) return rc;
  if ((rc = upload(e, e->d_lv_acpu, t->lv_alloc_cpu_milli, n * 8))) return rc;
  if ((rc = upload(e, e->d_lv_amem, t->lv_alloc_mem, n * 8))) return rc;
  if ((rc = upload(e, e->d_lv_cavg, t->lv_cpu_avg, n * 8))) return rc;
  if ((rc = upload(e, e->d_lv_cstd, t->lv_cpu_std, n * 8))) return rc;
  if ((rc = upload(e, e->d_lv_mavg, t->lv_mem_avg, n * 8))) return rc;
  if ((rc = upload(e, e->d_lv_mstd, t->lv_mem_std, n * 8))) return rc;
  if ((rc = upload(e, e->d_lv_flags, t->lv_flags, n))) return rc;
  e->lv_alloc_exact = all_below_2p52(t->lv_alloc_cpu_milli, n) && all_below_2p52(t->lv_alloc_mem, n);
  e->lroc_tab_ready = false;
  e->tri_nodes = true;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}


namespace {
// one pinned blob for a delta's columns: [idx int32 n] then each column, 16-byte aligned; uploaded with one DMA
struct DeltaBlob {
  spx_engine* e;
  size_t bytes = 0;
  std::vector<std::pair<const void*, size_t>> parts;  // (source, bytes)
  std::vector<size_t> offset;
  size_t add(const void* src, size_t n) {
    const size_t at = bytes;
    parts.emplace_back(src, n);
    offset.push_back(at);
    bytes = (bytes + n + 15) & ~static_cast<size_t>(15);
    return at;
  }
  int ship() {
    if (e->h_stage_bytes < bytes) {
      if (e->h_stage) SPX_HIP(e, hipHostFree(e->h_stage));
      e->h_stage = nullptr, e->h_stage_bytes = 0;
      SPX_HIP(e, hipHostMalloc(&e->h_stage, bytes + 65536, hipHostMallocDefault));
      e->h_stage_bytes = bytes + 65536;
    }
    for (size_t k = 0; k < parts.size(); ++k) {
      char* dst = static_cast<char*>(e->h_stage) + offset[k];
      const char* src = static_cast<const char*>(parts[k].first);
      const int64_t blocks = static_cast<int64_t>((parts[k].second + 65535) / 65536);  // (a full node table: megabytes per column)
      const size_t len = parts[k].second;
      spx_host::parallel_rows(blocks, [&](int64_t b0, int64_t b1) {
        const size_t at = static_cast<size_t>(b0) * 65536, end = std::min(len, static_cast<size_t>(b1) * 65536);
        if (end > at) std::memcpy(dst + at, src + at, end - at);
      }, 16);
    }
    return upload(e, e->d_delta, e->h_stage, bytes);
  }
  const char* dev(size_t at) const { return static_cast<const char*>(e->d_delta.p) + at; }
};

int delta_indices(spx_engine* e, const int64_t* idx, int64_t n_rows, std::vector<int32_t>& out) {
  if (n_rows < 0 || (n_rows && !idx)) return fail(e, SPX_ERR_ARG, "delta: NULL index column");
  out.resize(static_cast<size_t>(n_rows));
  for (int64_t i = 0; i < n_rows; ++i) {
    if (idx[i] < 0 || idx[i] >= e->n_nodes) return fail(e, SPX_ERR_ARG, "delta: node index out of range");
    out[static_cast<size_t>(i)] = static_cast<int32_t>(idx[i]);
  }
  // a node listed twice would be scattered twice in no particular order — and the columns derived from the rows (the float64 images,
  // the host copies) could end up describing different rows of the delta: refused
  std::vector<int32_t> sorted(out);
  std::sort(sorted.begin(), sorted.end());
  if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(e, SPX_ERR_ARG, "delta: a node index is listed twice");
  return SPX_OK;
}
}  // namespace

int spx_update_trimaran_nodes(spx_engine* e, const int64_t* idx, const spx_trimaran_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->tri_nodes) return fail(e, SPX_ERR_STATE, "trimaran node delta: upload the full table first");
  const int64_t n = t->n_nodes;
  if (n == 0) return SPX_OK;
  if (!t->cap_cpu_milli || !t->tlp_cpu_util || !t->tlp_missing_milli || !t->tlp_valid || !t->lv_alloc_cpu_milli || !t->lv_alloc_mem ||
      !t->lv_cpu_avg || !t->lv_cpu_std || !t->lv_mem_avg || !t->lv_mem_std || !t->lv_flags)
    return fail(e, SPX_ERR_ARG, "NULL column in table");
  std::vector<int32_t> ix;
  int rc = delta_indices(e, idx, n, ix);
  if (rc) return rc;
  e->tlp_amb_built = e->lv_amb_built = false;  // rows of the columns k_tlp_amb_build / k_lvrb_amb_build read are about to change
  const size_t m = static_cast<size_t>(n);
  DeltaBlob b{e};
  const size_t o_idx = b.add(ix.data(), m * 4);
  struct Col { DevBuf* dst; const void* src; int bytes; } cols[] = {
      {&e->d_cap_cpu, t->cap_cpu_milli, 8}, {&e->d_tlp_util, t->tlp_cpu_util, 8}, {&e->d_tlp_missing, t->tlp_missing_milli, 8}, {&e->d_tlp_valid, t->tlp_valid, 1},
      {&e->d_lv_acpu, t->lv_alloc_cpu_milli, 8}, {&e->d_lv_amem, t->lv_alloc_mem, 8}, {&e->d_lv_cavg, t->lv_cpu_avg, 8}, {&e->d_lv_cstd, t->lv_cpu_std, 8},
      {&e->d_lv_mavg, t->lv_mem_avg, 8}, {&e->d_lv_mstd, t->lv_mem_std, 8}, {&e->d_lv_flags, t->lv_flags, 1}};
  size_t at[11];
  for (int k = 0; k < 11; ++k) at[k] = b.add(cols[k].src, m * static_cast<size_t>(cols[k].bytes));
  if ((rc = b.ship())) return rc;
  for (int k = 0; k < 11; ++k)
    spx::launch_scatter_rows(cols[k].dst->p, e->n_nodes, 1, reinterpret_cast<const int32_t*>(b.dev(o_idx)), b.dev(at[k]), n, cols[k].bytes, e->stream);
  SPX_HIP(e, hipGetLastError());
  // the aggregate property stays conservative: rows may only take it away (a full upload re-establishes it)
  e->lv_alloc_exact = e->lv_alloc_exact && all_below_2p52(t->lv_alloc_cpu_milli, m) && all_below_2p52(t->lv_alloc_mem, m);
  e->lroc_tab_ready = false;
  e->evaluated = 0;  // every table computed from the old rows is stale
  e->best_valid = false;
  SPX_HIP(e, hipStreamSynchronize(e->stream));  // host columns are only borrowed for the call
  return SPX_OK;
}

// AppGroup scheduled lists grow between cycles (networkoverhead.go:654-694 reads them from the pod lister): the new (key, host,
// MaxNetworkCost) pairs — spx_flatten_net_placed — are appended to the workload keys' lists on the device.  The host lays out the
// new CSR (key counts only), the old pairs move inside the device (k_spread_pairs), the new ones are scattered behind them.
int spx_update_net_placed(spx_engine* e, int64_t n, const int32_t* key, const int32_t* node, const int64_t* max_cost) {
  if (!e || n < 0 || (n && (!key || !node || !max_cost))) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->net_pods) return fail(e, SPX_ERR_STATE, "NetworkOverhead delta: upload the pod table first");
  if (n == 0) return SPX_OK;
  const size_t K = static_cast<size_t>(e->net_n_keys);
  std::vector<int32_t> add(K, 0);
  std::vector<uint8_t> flag = e->h_key_flag;
  for (int64_t i = 0; i < n; ++i) {
    if (key[i] < 0 || static_cast<size_t>(key[i]) >= K) return fail(e, SPX_ERR_ARG, "NetworkOverhead delta: key out of range");
    if (max_cost[i] < 0) {  // the group's scheduled list is no longer empty: the key stops scoring equally (networkoverhead.go:215-224)
      if (flag[static_cast<size_t>(key[i])] == 1) flag[static_cast<size_t>(key[i])] = 0;
      continue;
    }
    if (node[i] >= e->n_nodes) return fail(e, SPX_ERR_ARG, "NetworkOverhead delta: node index out of range");
    if (node[i] < 0) flag[static_cast<size_t>(key[i])] = 2;  // host not in the snapshot: PreFilter returns Error (:258, :274)
    else if (flag[static_cast<size_t>(key[i])] == 1) flag[static_cast<size_t>(key[i])] = 0;
    ++add[static_cast<size_t>(key[i])];
  }
  std::vector<int32_t> ptr(K + 1, 0), fill(K);
  for (size_t k = 0; k < K; ++k) {
    const int64_t next = static_cast<int64_t>(ptr[k]) + (e->h_pair_ptr[k + 1] - e->h_pair_ptr[k]) + add[k];
    if (next > INT32_MAX) return fail(e, SPX_ERR_ARG, "NetworkOverhead delta: more than 2^31 pairs");
    ptr[k + 1] = static_cast<int32_t>(next);
    fill[k] = ptr[k] + (e->h_pair_ptr[k + 1] - e->h_pair_ptr[k]);
  }
  std::vector<int32_t> pos, nd;
  std::vector<int64_t> cost;
  pos.reserve(static_cast<size_t>(n)), nd.reserve(static_cast<size_t>(n)), cost.reserve(static_cast<size_t>(n));
  for (int64_t i = 0; i < n; ++i)
    if (max_cost[i] >= 0) pos.push_back(fill[static_cast<size_t>(key[i])]++), nd.push_back(node[i]), cost.push_back(max_cost[i]);
  const size_t m = pos.size(), total = static_cast<size_t>(ptr[K]);
  int rc;
  if ((rc = ensure(e, e->d_net_pair_node2, (total ? total : 1) * 4)) || (rc = ensure(e, e->d_net_pair_max2, (total ? total : 1) * 8))) return rc;
  DeltaBlob b{e};
  const size_t o_ptr = b.add(ptr.data(), (K + 1) * 4), o_flag = b.add(flag.data(), K), o_pos = b.add(pos.data(), m * 4), o_node = b.add(nd.data(), m * 4),
               o_cost = b.add(cost.data(), m * 8);
  if ((rc = b.ship())) return rc;
  spx::launch_spread_pairs(static_cast<int32_t>(K), static_cast<const int32_t*>(e->d_net_pair_ptr.p), reinterpret_cast<const int32_t*>(b.dev(o_ptr)),
                           static_cast<const int32_t*>(e->d_net_pair_node.p), static_cast<const int64_t*>(e->d_net_pair_max.p),
                           static_cast<int32_t*>(e->d_net_pair_node2.p), static_cast<int64_t*>(e->d_net_pair_max2.p), e->stream);
  spx::launch_net_append(static_cast<int64_t>(m), reinterpret_cast<const int32_t*>(b.dev(o_pos)), reinterpret_cast<const int32_t*>(b.dev(o_node)),
                         reinterpret_cast<const int64_t*>(b.dev(o_cost)), static_cast<int32_t*>(e->d_net_pair_node2.p), static_cast<int64_t*>(e->d_net_pair_max2.p),
                         e->stream);
  SPX_HIP(e, hipGetLastError());
  SPX_HIP(e, hipMemcpyAsync(e->d_net_pair_ptr.p, b.dev(o_ptr), (K + 1) * 4, hipMemcpyDeviceToDevice, e->stream));
  SPX_HIP(e, hipMemcpyAsync(e->d_net_key_flag.p, b.dev(o_flag), K, hipMemcpyDeviceToDevice, e->stream));
  std::swap(e->d_net_pair_node, e->d_net_pair_node2);
  std::swap(e->d_net_pair_max, e->d_net_pair_max2);
  e->h_pair_ptr = std::move(ptr);
  e->h_key_flag = std::move(flag);
  e->net_max_pairs = 0;
  for (size_t k = 0; k < K; ++k) e->net_max_pairs = std::max<int64_t>(e->net_max_pairs, e->h_pair_ptr[k + 1] - e->h_pair_ptr[k]);
  e->evaluated &= ~(1u << SPX_PLUGIN_NETOVERHEAD);
  e->best_valid = false;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

// ElasticQuota Used moves with every pod added to or removed from a namespace (capacity_scheduling.go:679-803 -> elasticquota.go
// reserveResource / unreserveResource): the changed namespaces' rows replace the device rows, with the aggregate vector PreFilter
// compares against the aggregate Min (capacity_scheduling.go:260-262).
int spx_update_quota_used(spx_engine* e, int64_t n_rows, const int32_t* ns, const int64_t* used, const uint8_t* used_present, const int64_t* agg_used,
                          const uint8_t* agg_used_present) {
  if (!e || n_rows < 0 || !agg_used || !agg_used_present || (n_rows && (!ns || !used || !used_present))) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->quota) return fail(e, SPX_ERR_STATE, "quota delta: upload the quota table first");
  constexpr size_t S = SPX_QUOTA_SLOTS;
  for (int64_t i = 0; i < n_rows; ++i)
    if (ns[i] < 0 || ns[i] >= e->q_n_namespaces) return fail(e, SPX_ERR_ARG, "quota delta: namespace index out of range");
  {
    // two rows for one namespace would be scattered in unspecified order (d_q_used and d_q_usedp could end up from different rows)
    std::vector<int32_t> seen(ns, ns + n_rows);
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return fail(e, SPX_ERR_ARG, "quota delta: a namespace is listed twice");
  }
  const size_t m = static_cast<size_t>(n_rows);
  int64_t agg[SPX_QUOTA_SLOTS + 1];
  std::memcpy(agg, agg_used, sizeof e->q_agg_used);
  agg[SPX_QUOTA_SLOTS] = *agg_used_present;
  DeltaBlob b{e};
  const size_t o_idx = b.add(ns, m * 4), o_used = b.add(used, m * S * 8), o_p = b.add(used_present, m), o_agg = b.add(agg, sizeof agg);
  int rc;
  if ((rc = b.ship())) return rc;
  spx::launch_scatter_rows_rowmajor(e->d_q_used.p, static_cast<int>(S), reinterpret_cast<const int32_t*>(b.dev(o_idx)), b.dev(o_used), n_rows, 8, e->stream);
  spx::launch_scatter_rows_rowmajor(e->d_q_usedp.p, 1, reinterpret_cast<const int32_t*>(b.dev(o_idx)), b.dev(o_p), n_rows, 1, e->stream);
  SPX_HIP(e, hipGetLastError());
  SPX_HIP(e, hipMemcpyAsync(e->d_q_agg.p, b.dev(o_agg), sizeof agg, hipMemcpyDeviceToDevice, e->stream));
  std::memcpy(e->q_agg_used, agg_used, sizeof e->q_agg_used);
  e->q_agg_used_present = *agg_used_present;
  e->evaluated &= ~(1u << SPX_PLUGIN_CAPACITY);
  e->best_valid = false;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_update_nrt_nodes(spx_engine* e, const int64_t* idx, const spx_nrt_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->nrt_nodes || !e->nrt_slots) return fail(e, SPX_ERR_STATE, "NRT node delta: upload the slot and node tables first");
  if (t->n_res != e->nrt_n_res) return fail(e, SPX_ERR_ARG, "NRT node delta: n_res differs from the slot table");
  const int64_t n = t->n_nodes;
  if (n == 0) return SPX_OK;
  if (!t->flags || !t->max_numa || !t->n_zones || !t->zone_id || !t->zone_present || !t->zone_cost || !t->min_avg_dist || !t->node_present ||
      (!t->zone_avail && t->n_res))
    return fail(e, SPX_ERR_ARG, "NULL column in table");
  std::vector<int32_t> ix;
  int rc = delta_indices(e, idx, n, ix);
  if (rc) return rc;
  constexpr int64_t Zm = SPX_NRT_MAX_ZONES;
  const int64_t R = t->n_res, N = e->n_nodes;
  const size_t m = static_cast<size_t>(n);
  // the float64 formulation's preconditions for the new rows (the same tests as spx_upload_nrt_nodes); a row that breaks them
  // sends the whole table to the reference-arithmetic kernel until the next full upload
  bool ok = true, cost_changed = false, ln_ok = true;
  uint32_t big = 0;
  spx_engine::NrtQty qty;
  for (int64_t i = 0; i < n; ++i) {
    const int nz = t->n_zones[i];
    for (int z = 0; z < nz && z < Zm; ++z) {
      if (t->zone_id[i * Zm + z] != z) ok = false;
      for (int64_t r = 0; r < R; ++r) {
        if (!((t->zone_present[i * Zm + z] >> r) & 1u)) continue;
        const int64_t cap = t->zone_avail[(i * Zm + z) * R + r];
        if (!nrt_fast_qty(cap)) ok = false;
        if (!nrt_exact_f32(static_cast<double>(nrt_value_of(r == e->nrt_cpu_slot, cap)))) big |= 1u << r;
        if (cap >= 0) qty.add(static_cast<int>(r), nrt_value_of(r == e->nrt_cpu_slot, cap));
      }
    }
    const int32_t* hc = &e->h_nrt_cost[static_cast<size_t>(ix[static_cast<size_t>(i)]) * Zm * Zm];
    if (std::memcmp(hc, t->zone_cost + i * Zm * Zm, sizeof(int32_t) * Zm * Zm) != 0 || e->h_nrt_nz[static_cast<size_t>(ix[static_cast<size_t>(i)])] != t->n_zones[i]) {
      cost_changed = true;  // (the host copies follow once the rows have shipped: a failed delta leaves them describing the device)
      for (int za = 0; za < nz && za < Zm; ++za)
        for (int zb = 0; zb < nz && zb < Zm; ++zb) {
          const int64_t c = t->zone_cost[(i * Zm + za) * Zm + zb];
          if (c < 0 || c > 255) ln_ok = false;
        }
    }
  }
  DeltaBlob b{e};
  const size_t o_idx = b.add(ix.data(), m * 4);
  const size_t o_flags = b.add(t->flags, m), o_max = b.add(t->max_numa, m * 4), o_nz = b.add(t->n_zones, m), o_np = b.add(t->node_present, m);
  const size_t o_zid = b.add(t->zone_id, m * Zm), o_zp = b.add(t->zone_present, m * Zm);
  const size_t o_av = b.add(t->zone_avail, m * Zm * static_cast<size_t>(R) * 8), o_cost = b.add(t->zone_cost, m * Zm * Zm * 4);
  const size_t o_min = b.add(t->min_avg_dist, m * Zm * 4);
  if ((rc = b.ship())) return rc;
  const int32_t* d_idx = reinterpret_cast<const int32_t*>(b.dev(o_idx));
  hipStream_t s = e->stream;
  spx::launch_scatter_rows(e->d_nrt_flags.p, N, 1, d_idx, b.dev(o_flags), n, 1, s);
  spx::launch_scatter_rows(e->d_nrt_max_numa.p, N, 1, d_idx, b.dev(o_max), n, 4, s);
  spx::launch_scatter_rows(e->d_nrt_nz.p, N, 1, d_idx, b.dev(o_nz), n, 1, s);
  spx::launch_scatter_rows(e->d_nrt_np.p, N, 1, d_idx, b.dev(o_np), n, 1, s);
  spx::launch_scatter_rows(e->d_nrt_zid.p, N, static_cast<int>(Zm), d_idx, b.dev(o_zid), n, 1, s);
  spx::launch_scatter_rows(e->d_nrt_zp.p, N, static_cast<int>(Zm), d_idx, b.dev(o_zp), n, 1, s);
  if (R) spx::launch_scatter_rows(e->d_nrt_avail.p, N, static_cast<int>(Zm * R), d_idx, b.dev(o_av), n, 8, s);
  spx::launch_scatter_rows(e->d_nrt_cost.p, N, static_cast<int>(Zm * Zm), d_idx, b.dev(o_cost), n, 4, s);
  spx::launch_scatter_rows(e->d_nrt_minavg.p, N, static_cast<int>(Zm), d_idx, b.dev(o_min), n, 4, s);
  spx::NrtDeltaArgs da{};
  da.n_rows = n, da.n_nodes = N, da.n_res = static_cast<int32_t>(R), da.cpu_slot = e->nrt_cpu_slot;
  da.idx = d_idx, da.n_zones = reinterpret_cast<const uint8_t*>(b.dev(o_nz)), da.zone_present = reinterpret_cast<const uint8_t*>(b.dev(o_zp)), da.zone_avail = reinterpret_cast<const int64_t*>(b.dev(o_av));
  da.f_av = static_cast<double*>(e->d_nrt_fav.p), da.f_rc = static_cast<double*>(e->d_nrt_frc.p), da.f_rcv = static_cast<double*>(e->d_nrt_frcv.p);
  da.f_cpu = static_cast<double*>(e->d_nrt_fcpu.p), da.f_braw = static_cast<double*>(e->d_nrt_fbraw.p), da.f_rep = static_cast<uint8_t*>(e->d_nrt_frep.p);
  spx::launch_nrt_derive_rows(da, s);
  SPX_HIP(e, hipGetLastError());
  if (cost_changed)
    for (int64_t i = 0; i < n; ++i) {
      const size_t node = static_cast<size_t>(ix[static_cast<size_t>(i)]);
      std::memcpy(&e->h_nrt_cost[node * Zm * Zm], t->zone_cost + i * Zm * Zm, sizeof(int32_t) * Zm * Zm);
      e->h_nrt_nz[node] = t->n_zones[i];
    }
  e->nrt_fast_nodes = e->nrt_fast_nodes && ok;
  e->nrt_big_nodes |= big;
  e->nrt_qty_nodes.merge(qty);
  e->nrt_pk_tab_built = false;  // zone capacities changed
  if (cost_changed) {  // LeastNUMANodes' per-node tables are rebuilt when that strategy is next evaluated
    e->nrt_ln_built = false;
    e->nrt_ln_ok = e->nrt_ln_ok && ln_ok;
  }
  // (the window-local node order — perm — is a grouping hint for the sweep, not a correctness input: left as it is)
  e->evaluated = 0;  // NRT's tables, and every table normalised over the feasible nodes its status named (Allocatable, NetworkOverhead, Peaks)
  e->best_valid = false;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_set_lroc_params(spx_engine* e, const spx_lroc_params* p) {
  if (!e || !p) return SPX_ERR_ARG;
  // defaults.go:176-186 substitutes defaults for bad values before the plugin sees them; the engine takes the result
  if (p->smoothing_window_size <= 0) return fail(e, SPX_ERR_ARG, "LowRiskOverCommitment: SmoothingWindowSize must be positive");
  if (!(p->risk_limit_weight_cpu >= 0 && p->risk_limit_weight_cpu <= 1) || !(p->risk_limit_weight_mem >= 0 && p->risk_limit_weight_mem <= 1))
    return fail(e, SPX_ERR_ARG, "LowRiskOverCommitment: RiskLimitWeights must be in [0,1]");  // validation_pluginargs.go
  e->lroc = *p;
  e->lroc_tab_ready = false;
  return SPX_OK;
}

int spx_upload_lroc_nodes(spx_engine* e, const spx_lroc_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->tri_nodes) return fail(e, SPX_ERR_STATE, "LowRiskOverCommitment reads the trimaran node table: upload it first");
  int rc = set_nodes(e, t->n_nodes);
  if (rc) return rc;
  const size_t n = static_cast<size_t>(t->n_nodes);
  if ((rc = upload(e, e->d_lroc_nreq_c, t->req_cpu_milli, n * 8))) return rc;
  if ((rc = upload(e, e->d_lroc_nreq_m, t->req_mem, n * 8))) return rc;
  if ((rc = upload(e, e->d_lroc_nlim_c, t->lim_cpu_milli, n * 8))) return rc;
  if ((rc = upload(e, e->d_lroc_nlim_m, t->lim_mem, n * 8))) return rc;
  e->lroc_nodes_exact = all_below_2p52(t->req_cpu_milli, n) && all_below_2p52(t->req_mem, n) && all_below_2p52(t->lim_cpu_milli, n) &&
                        all_below_2p52(t->lim_mem, n);
  e->lroc_nodes = true;
  e->lroc_tab_ready = false;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_upload_lroc_pods(spx_engine* e, const spx_lroc_pods_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_pods(e, t->n_pods);
  if (rc) return rc;
  const size_t p = static_cast<size_t>(t->n_pods);
  if ((rc = upload(e, e->d_lroc_preq_c, t->req_cpu_milli, p * 8))) return rc;
  if ((rc = upload(e, e->d_lroc_preq_m, t->req_mem, p * 8))) return rc;
  if ((rc = upload(e, e->d_lroc_plim_c, t->lim_cpu_milli, p * 8))) return rc;
  if ((rc = upload(e, e->d_lroc_plim_m, t->lim_mem, p * 8))) return rc;
  e->lroc_pods_exact = all_below_2p52(t->req_cpu_milli, p) && all_below_2p52(t->req_mem, p) && all_below_2p52(t->lim_cpu_milli, p) &&
                       all_below_2p52(t->lim_mem, p);
  if (e->lroc_pods_exact) {  // float64 pod records of the fast sweep: limit and limit - request per resource (exact below 2^52)
    std::vector<double> f(4 * p);
    for (size_t i = 0; i < p; ++i) {
      const bool none = t->req_cpu_milli[i] == 0 && t->req_mem[i] == 0 && t->lim_cpu_milli[i] == 0 && t->lim_mem[i] == 0;
      f[i] = none ? std::nan("") : static_cast<double>(t->lim_cpu_milli[i]);
      f[p + i] = static_cast<double>(t->lim_cpu_milli[i] - t->req_cpu_milli[i]);
      f[2 * p + i] = static_cast<double>(t->lim_mem[i]);
      f[3 * p + i] = static_cast<double>(t->lim_mem[i] - t->req_mem[i]);
    }
    if ((rc = upload(e, e->d_lroc_podf, f.data(), f.size() * sizeof(double)))) return rc;
    SPX_HIP(e, hipStreamSynchronize(e->stream));  // f goes out of scope
  }
  e->lroc_pods = true;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_upload_peaks_nodes(spx_engine* e, const spx_peaks_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_nodes(e, t->n_nodes);
  if (rc) return rc;
  const size_t n = static_cast<size_t>(t->n_nodes);
  if ((rc = upload(e, e->d_pk_cap, t->cap_cpu_milli, n * 8))) return rc;
  if ((rc = upload(e, e->d_pk_util, t->cpu_util, n * 8))) return rc;
  if ((rc = upload(e, e->d_pk_valid, t->valid, n))) return rc;
  if ((rc = upload(e, e->d_pk_k1, t->k1, n * 8))) return rc;
  if ((rc = upload(e, e->d_pk_k2, t->k2, n * 8))) return rc;
  e->peaks_nodes = true;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_upload_peaks_pods(spx_engine* e, const spx_peaks_pods_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_pods(e, t->n_pods);
  if (rc) return rc;
  if ((rc = upload(e, e->d_pk_pod, t->cpu_milli, static_cast<size_t>(t->n_pods) * 8))) return rc;
  // Pod classes: the pod's cpu request is all Peaks.Score reads of it (peaks.go:134-138), so rows of equal requests are equal —
  // raw scores always, normalised scores when every pod's node list is the whole snapshot.  First row of each distinct value
  // (flat open-addressing table, rows in order), the others as (row, representative) pairs.
  e->pk_n_uniq = e->pk_n_dups = 0;
  e->pk_negative = false;
  for (int64_t i = 0; i < t->n_pods; ++i)
    if (t->cpu_milli[i] < 0) e->pk_negative = true;
  if (t->n_pods > 1) {
    const size_t p = static_cast<size_t>(t->n_pods);
    size_t cap = 64;
    while (cap < 2 * p) cap <<= 1;
    std::vector<int32_t> tab(cap, -1), uniq, dups;
    uniq.reserve(p), dups.reserve(2 * p);
    for (size_t i = 0; i < p; ++i) {
      const int64_t v = t->cpu_milli[i];
      size_t k = static_cast<size_t>((static_cast<uint64_t>(v) * 0x9e3779b97f4a7c15ull) >> 24) & (cap - 1);
      while (tab[k] >= 0 && t->cpu_milli[tab[k]] != v) k = (k + 1) & (cap - 1);
      if (tab[k] < 0) tab[k] = static_cast<int32_t>(i), uniq.push_back(static_cast<int32_t>(i));
      else dups.push_back(static_cast<int32_t>(i)), dups.push_back(tab[k]);
    }
    if (!dups.empty()) {
      if ((rc = upload(e, e->d_pk_uniq, uniq.data(), uniq.size() * sizeof(int32_t)))) return rc;
      if ((rc = upload(e, e->d_pk_dups, dups.data(), dups.size() * sizeof(int32_t)))) return rc;
      SPX_HIP(e, hipStreamSynchronize(e->stream));  // the vectors go out of scope
      e->pk_n_uniq = static_cast<int64_t>(uniq.size());
      e->pk_n_dups = static_cast<int64_t>(dups.size() / 2);
    }
  }
  e->peaks_pods = true;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_upload_trimaran_pods(spx_engine* e, const spx_trimaran_pods_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_pods(e, t->n_pods);
  if (rc) return rc;
  const size_t p = static_cast<size_t>(t->n_pods);
  if ((rc = upload(e, e->d_tlp_pod, t->tlp_pod_milli, p * 8))) return rc;
  if ((rc = upload(e, e->d_lv_rcpu, t->lv_req_cpu_milli, p * 8))) return rc;
  if ((rc = upload(e, e->d_lv_rmem, t->lv_req_mem, p * 8))) return rc;
  e->tri_pods = true;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_set_nrt_params(spx_engine* e, const spx_nrt_params* p) {
  if (!e || !p) return SPX_ERR_ARG;
  if (p->strategy < SPX_NRT_MOST_ALLOCATED || p->strategy > SPX_NRT_LEAST_NUMA_NODES)
    return fail(e, SPX_ERR_ARG, "illegal scoring strategy found");  // score.go:137-139
  if (e->nrt_params.strategy != p->strategy) {  // the packed Score's table of exceptions and the fused walk's items are per strategy
    e->nrt_pk_tab_built = false;
    ++e->nrt_items_gen;
  }
  e->nrt_params.strategy = p->strategy;  // weights travel through the slot table (spx_flatten_nrt_slots)
  return SPX_OK;
}

int spx_upload_nrt_slots(spx_engine* e, const spx_nrt_slots* t) {
  if (!e || !t) return SPX_ERR_ARG;
  if (t->n_res < 0 || t->n_res > SPX_NRT_MAX_RES) return fail(e, SPX_ERR_ARG, "NRT: more resource slots than this build supports");
  e->nrt_n_res = t->n_res;
  for (int i = 0; i < t->n_res; ++i) {
    e->nrt_slot_flags[i] = t->slot_flags[i];
    e->nrt_slot_weight[i] = t->slot_weight[i];
    e->nrt_slot_res[i] = t->slot_res ? t->slot_res[i] : -1;
  }
  e->nrt_slots = true;
  ++e->nrt_items_gen;
  e->nrt_nodes = e->nrt_pods = false;  // tables are laid out by slot count
  // float64 formulation: weight-subset table, cpu slot, weight range
  SPX_HIP(e, hipSetDevice(e->device));
  e->nrt_cpu_slot = -1;
  e->nrt_fast_slots = true;
  int64_t wtotal = 0;
  for (int i = 0; i < t->n_res; ++i) {
    if (t->slot_flags[i] & SPX_NRT_SLOT_CPU) e->nrt_cpu_slot = i;
    // the Least/MostAllocated Score accumulates integer zone totals (v_mad_u32_u24: weights below 2^24) whose high bit marks a
    // zero zone score: 100 * sum(weights) must stay below 2^31 — with room, sum(weights) < 2^20 (upstream weights are 1..100)
    if (t->slot_weight[i] < 0 || t->slot_weight[i] >= kNrtWeightLimit) e->nrt_fast_slots = false;
    else wtotal += t->slot_weight[i];
  }
  if (wtotal >= kNrtWeightLimit) e->nrt_fast_slots = false;
  std::vector<double> wtab(static_cast<size_t>(2) << t->n_res, 0.0);
  if (e->nrt_fast_slots)
    for (unsigned m = 0; m < (1u << t->n_res); ++m) {
      int64_t w = 0;
      for (int i = 0; i < t->n_res; ++i)
        if ((m >> i) & 1u) w += t->slot_weight[i];
      wtab[2 * m] = static_cast<double>(w);
      wtab[2 * m + 1] = nrt_biased_rcp(static_cast<double>(w));
    }
  e->nrt_wtab = std::move(wtab);
  return SPX_OK;
}

int spx_upload_nrt_nodes(spx_engine* e, const spx_nrt_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->nrt_slots || t->n_res != e->nrt_n_res) return fail(e, SPX_ERR_STATE, "NRT: upload the slot table first (n_res mismatch)");
  int rc = set_nodes(e, t->n_nodes);
  if (rc) return rc;
  e->nrt_nodes = false;  // (a call that fails half-way leaves "no NRT node table", not a mix of two)
  const int64_t n = t->n_nodes;
  constexpr int64_t Zm = SPX_NRT_MAX_ZONES;
  const int64_t R = t->n_res;
  if (!t->flags || !t->max_numa || !t->n_zones || !t->zone_id || !t->zone_present || !t->zone_cost || !t->min_avg_dist || !t->node_present ||
      (!t->zone_avail && R))
    return fail(e, SPX_ERR_ARG, "NULL column in table");
  // Round 4: the full upload takes the delta's road (spx_update_nrt_nodes) with every node listed — the rows as they are into ONE
  // pinned blob, one DMA, and the device turns them into the node-major columns (k_scatter_rows) and the float64 formulation's
  // derived columns (k_nrt_derive_rows: the expressions below used to run here, on the host, into five freshly allocated vectors
  // that were then copied from pageable memory: 12.6 of the 24 ms a full snapshot load took at 20 000 nodes).  What stays on the host:
  // the preconditions of the float64 formulation, the window-local node order, the host copy LeastNUMANodes' tables are built from.
  const size_t m = static_cast<size_t>(n), cells = static_cast<size_t>(Zm * R) * m;
  if ((rc = ensure(e, e->d_nrt_flags, m)) || (rc = ensure(e, e->d_nrt_max_numa, m * 4)) || (rc = ensure(e, e->d_nrt_nz, m)) || (rc = ensure(e, e->d_nrt_np, m)) ||
      (rc = ensure(e, e->d_nrt_zid, m * Zm)) || (rc = ensure(e, e->d_nrt_zp, m * Zm)) || (rc = ensure(e, e->d_nrt_avail, cells * 8)) ||
      (rc = ensure(e, e->d_nrt_cost, m * Zm * Zm * 4)) || (rc = ensure(e, e->d_nrt_minavg, m * Zm * 4)) || (rc = ensure(e, e->d_nrt_fav, cells * 8)) ||
      (rc = ensure(e, e->d_nrt_frc, cells * 8)) || (rc = ensure(e, e->d_nrt_frcv, cells * 8)) || (rc = ensure(e, e->d_nrt_fcpu, m * Zm * 8)) ||
      (rc = ensure(e, e->d_nrt_fbraw, m * Zm * 8)) || (rc = ensure(e, e->d_nrt_frep, static_cast<size_t>(R > 0 ? R : 1) * m)))
    return rc;
  {
    std::atomic<bool> ok{true}, ln_ok{true};
    std::atomic<uint32_t> big_nodes{0};
    spx_engine::NrtQty qty_all;
    std::mutex qty_mu;
    spx_host::parallel_rows(n, [&](int64_t row0, int64_t row1) {
    bool my_ok = true, my_ln = true;
    uint32_t my_big = 0;
    spx_engine::NrtQty my_qty;
    for (int64_t i = row0; i < row1; ++i) {
      const int nz = t->n_zones[i];
      for (int z = 0; z < nz && z < Zm; ++z) {
        if (t->zone_id[i * Zm + z] != z) my_ok = false;  // "lowest NUMA id" must be "lowest list position"
        for (int64_t r = 0; r < R; ++r) {
          if (!((t->zone_present[i * Zm + z] >> r) & 1u)) continue;
          const int64_t cap = t->zone_avail[(i * Zm + z) * R + r];
          if (!nrt_fast_qty(cap)) my_ok = false;
          if (!nrt_exact_f32(static_cast<double>(nrt_value_of(r == e->nrt_cpu_slot, cap)))) my_big |= 1u << r;
          if (cap >= 0) my_qty.add(static_cast<int>(r), nrt_value_of(r == e->nrt_cpu_slot, cap));
        }
        // LeastNUMANodes' tables can be built when every zone cost lies within [0, 255] (findSuitableCombination's 256 sentinel)
        for (int zb = 0; zb < nz && zb < Zm; ++zb) {
          const int64_t c = t->zone_cost[(i * Zm + z) * Zm + zb];
          if (c < 0 || c > 255) my_ln = false;
        }
      }
    }
    if (!my_ok) ok = false;
    if (!my_ln) ln_ok = false;
    if (my_big) big_nodes.fetch_or(my_big, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> g(qty_mu);
      qty_all.merge(my_qty);
    }
    }, 1024);
    e->nrt_fast_nodes = ok.load();
    e->nrt_big_nodes = big_nodes.load();
    e->nrt_qty_nodes = qty_all;
    e->nrt_pk_tab_built = false;
    e->nrt_ln_ok = ln_ok.load();
    e->nrt_ln_built = false;  // built when that strategy is first evaluated (build_ln_tab): more host time than everything else in this call
    // window-local node order: inside each run of 256 nodes, group the nodes by the code path their flags select
    // (not aligned / pod scope / container scope) so that wavefronts are mostly homogeneous; inside a group, by how tight the
    // node's two largest zones are (the smaller of its ranks, within the window, by the sum of the two largest zone quantities of
    // slot 0 and of slot 1 — cpu and memory): LeastNUMANodes' second pass runs for a wave when one of its lanes needs more than two
    // zones, and those lanes are the tight nodes — sorted, they share waves (config #3: 69 % -> 37 % of the waves)
    const int64_t n_slots = spx::round_up(n, 256);
    std::vector<int32_t> perm(static_cast<size_t>(n_slots), -1);
    spx_host::parallel_rows((n + 255) / 256, [&](int64_t win0, int64_t win1) {
    for (int64_t w0 = win0 * 256; w0 < std::min<int64_t>(win1 * 256, n); w0 += 256) {
      const int64_t w1 = std::min<int64_t>(w0 + 256, n);
      const int cnt = static_cast<int>(w1 - w0);
      int rank[2][256];
      for (int slot = 0; slot < 2; ++slot) {
        int64_t top2[256];
        int order[256];
        for (int k = 0; k < cnt; ++k) {
          const int64_t i = w0 + k;
          int64_t a = 0, b = 0;  // the two largest
          if (slot < R)
            for (int z = 0; z < t->n_zones[i] && z < Zm; ++z) {
              if (!((t->zone_present[i * Zm + z] >> slot) & 1u)) continue;
              const int64_t q = t->zone_avail[(i * Zm + z) * R + slot];
              if (q > a) b = a, a = q;
              else if (q > b) b = q;
            }
          top2[k] = a + b;
          order[k] = k;
        }
        std::stable_sort(order, order + cnt, [&](int x, int y) { return top2[x] < top2[y]; });
        for (int k = 0; k < cnt; ++k) rank[slot][order[k]] = k;
      }
      int order[256], cls_of[256], key[256];
      for (int k = 0; k < cnt; ++k) {
        const uint8_t f = t->flags[w0 + k];
        const bool aligned = (f & SPX_NRT_F_FRESH) && (f & SPX_NRT_F_HAS_NRT) && (f & SPX_NRT_F_SINGLE_NUMA);
        cls_of[k] = !aligned ? 0 : ((f & SPX_NRT_F_POD_SCOPE) ? 1 : 2);
        key[k] = std::min(rank[0][k], rank[1][k]);
        order[k] = k;
      }
      std::stable_sort(order, order + cnt, [&](int x, int y) { return cls_of[x] != cls_of[y] ? cls_of[x] < cls_of[y] : key[x] < key[y]; });
      for (int k = 0; k < cnt; ++k) perm[static_cast<size_t>(w0 + k)] = static_cast<int32_t>(w0 + order[k]);
    }
    }, 2);  // (three 256-key stable sorts per window, ~40 us: at 16 windows per thread config #5's 79 windows ran on 4 threads for 1 ms)
    std::vector<int32_t> all(m);
    for (size_t i = 0; i < m; ++i) all[i] = static_cast<int32_t>(i);
    DeltaBlob b{e};
    const size_t o_idx = b.add(all.data(), m * 4), o_perm = b.add(perm.data(), perm.size() * sizeof(int32_t));
    const size_t o_flags = b.add(t->flags, m), o_max = b.add(t->max_numa, m * 4), o_nz = b.add(t->n_zones, m), o_np = b.add(t->node_present, m);
    const size_t o_zid = b.add(t->zone_id, m * Zm), o_zp = b.add(t->zone_present, m * Zm);
    const size_t o_av = b.add(t->zone_avail, cells * 8), o_cost = b.add(t->zone_cost, m * Zm * Zm * 4);
    const size_t o_min = b.add(t->min_avg_dist, m * Zm * 4);
    if ((rc = ensure(e, e->d_nrt_perm, perm.size() * sizeof(int32_t)))) return rc;
    if ((rc = b.ship())) return rc;
    e->h_nrt_cost.assign(t->zone_cost, t->zone_cost + m * Zm * Zm);  // (the host copies follow the shipped rows)
    e->h_nrt_nz.assign(t->n_zones, t->n_zones + m);
    const int32_t* d_idx = reinterpret_cast<const int32_t*>(b.dev(o_idx));
    hipStream_t st = e->stream;
    SPX_HIP(e, hipMemcpyAsync(e->d_nrt_perm.p, b.dev(o_perm), perm.size() * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    SPX_HIP(e, hipMemcpyAsync(e->d_nrt_flags.p, b.dev(o_flags), m, hipMemcpyDeviceToDevice, st));
    SPX_HIP(e, hipMemcpyAsync(e->d_nrt_max_numa.p, b.dev(o_max), m * 4, hipMemcpyDeviceToDevice, st));
    SPX_HIP(e, hipMemcpyAsync(e->d_nrt_nz.p, b.dev(o_nz), m, hipMemcpyDeviceToDevice, st));
    SPX_HIP(e, hipMemcpyAsync(e->d_nrt_np.p, b.dev(o_np), m, hipMemcpyDeviceToDevice, st));
    spx::launch_scatter_rows(e->d_nrt_zid.p, n, static_cast<int>(Zm), d_idx, b.dev(o_zid), n, 1, st);
    spx::launch_scatter_rows(e->d_nrt_zp.p, n, static_cast<int>(Zm), d_idx, b.dev(o_zp), n, 1, st);
    if (R) spx::launch_scatter_rows(e->d_nrt_avail.p, n, static_cast<int>(Zm * R), d_idx, b.dev(o_av), n, 8, st);
    spx::launch_scatter_rows(e->d_nrt_cost.p, n, static_cast<int>(Zm * Zm), d_idx, b.dev(o_cost), n, 4, st);
    spx::launch_scatter_rows(e->d_nrt_minavg.p, n, static_cast<int>(Zm), d_idx, b.dev(o_min), n, 4, st);
    spx::NrtDeltaArgs da{};
    da.n_rows = n, da.n_nodes = n, da.n_res = static_cast<int32_t>(R), da.cpu_slot = e->nrt_cpu_slot;
    da.idx = d_idx, da.n_zones = reinterpret_cast<const uint8_t*>(b.dev(o_nz)), da.zone_present = reinterpret_cast<const uint8_t*>(b.dev(o_zp));
    da.zone_avail = reinterpret_cast<const int64_t*>(b.dev(o_av));
    da.f_av = static_cast<double*>(e->d_nrt_fav.p), da.f_rc = static_cast<double*>(e->d_nrt_frc.p), da.f_rcv = static_cast<double*>(e->d_nrt_frcv.p);
    da.f_cpu = static_cast<double*>(e->d_nrt_fcpu.p), da.f_braw = static_cast<double*>(e->d_nrt_fbraw.p), da.f_rep = static_cast<uint8_t*>(e->d_nrt_frep.p);
    spx::launch_nrt_derive_rows(da, st);
    SPX_HIP(e, hipGetLastError());
    SPX_HIP(e, hipStreamSynchronize(st));  // the blob is reused by the next staged call
  }
  e->nrt_nodes = true;
  return SPX_OK;
}

// LeastNUMANodes: per node the subsets of list positions at the node's minimum average distance for their size, and
// bit-planes of every subset's distance rank within its size (layout: LnLayout, spx_internal.h).  The average distance
// is nodesAvgDistance least_numa.go:115-138 — the sum over all ordered pairs, float32(sum) / float32(k*k); for one size
// the divisor is shared and sums below 2^14 stay distinct after the division, so ranking the integer sums ranks the
// reference's float32 values.  Only subsets of the node's own zones take part in the minimum (:102-113).
// Host-only; exported (not part of spx.h) so that tests/test_ln_tables.py can replay the kernel's selection against the
// reference's walk without a GPU.  zone_cost [n][Z][Z], n_zones [n], out [LnLayout.rows][n] zero-initialised by the callee.
int spx_internal_ln_tables(const int32_t* cost, const uint8_t* n_zones, int64_t n, uint32_t* tab) {
  if (!cost || !n_zones || !tab || n < 0) return SPX_ERR_ARG;
  constexpr int64_t Zm = SPX_NRT_MAX_ZONES;
  constexpr spx::LnLayout L = spx::make_ln_layout();
  std::fill(tab, tab + static_cast<size_t>(L.rows) * static_cast<size_t>(n), 0u);
  spx_host::parallel_rows(n, [&](int64_t row0, int64_t row1) {
    for (int64_t i = row0; i < row1; ++i) {
      const int nz = std::min<int>(n_zones[i], static_cast<int>(Zm));
      for (int k = 1; k <= 8; ++k) {
        int sums[70], order[70], cnt = 0;
        bool exists[70];
        for (int d = 0; d < L.nd[k]; ++d)
          for (int q = 0; q < L.cnt[L.first[k] + d]; ++q) {
            const unsigned m = L.subset[L.first[k] + d][q];
            int accu = 0;
            for (int za = 0; za < Zm; ++za)
              if (m >> za & 1u)
                for (int zb = 0; zb < Zm; ++zb)
                  if (m >> zb & 1u) accu += cost[(i * Zm + za) * Zm + zb];
            exists[cnt] = (m >> nz) == 0;
            sums[cnt] = accu;
            order[cnt] = cnt;
            ++cnt;
          }
        std::sort(order, order + cnt, [&](int x, int y) { return sums[x] < sums[y]; });
        int rank_of[70], level = -1, last = 0;
        for (int j = 0; j < cnt; ++j) rank_of[j] = (1 << L.bits[k]) - 1;  // subsets past the node's zones: never candidates
        for (int j = 0; j < cnt; ++j) {
          const int sidx = order[j];
          if (!exists[sidx]) continue;
          if (level < 0 || sums[sidx] != last) ++level, last = sums[sidx];
          rank_of[sidx] = level;
        }
        for (int pos = 0; pos < cnt; ++pos) {
          const size_t d = static_cast<size_t>(L.first[k] + pos / 32);
          const uint32_t bit = 1u << (pos % 32);
          if (exists[pos] && rank_of[pos] == 0) tab[d * static_cast<size_t>(n) + static_cast<size_t>(i)] |= bit;
          for (int b = 0; b < L.bits[k]; ++b)
            if ((rank_of[pos] >> b) & 1)
              tab[static_cast<size_t>(spx::kLnDwords + L.pbase[k] + b * L.nd[k] + pos / 32) * static_cast<size_t>(n) + static_cast<size_t>(i)] |= bit;
        }
      }
    }
  }, 512);
  return SPX_OK;
}

// the bit layout itself, for the same tests: subset[12][32] zone masks, then cnt[12], first[9], nd[9], bits[9], pbase[9], rows
int spx_internal_ln_layout(uint8_t* subset, uint8_t* cnt, uint8_t* first, uint8_t* nd, uint8_t* bits, uint8_t* pbase, int32_t* rows) {
  if (!subset || !cnt || !first || !nd || !bits || !pbase || !rows) return SPX_ERR_ARG;
  constexpr spx::LnLayout L = spx::make_ln_layout();
  std::memcpy(subset, L.subset, sizeof L.subset);
  std::memcpy(cnt, L.cnt, sizeof L.cnt);
  std::memcpy(first, L.first, sizeof L.first);
  std::memcpy(nd, L.nd, sizeof L.nd);
  std::memcpy(bits, L.bits, sizeof L.bits);
  std::memcpy(pbase, L.pbase, sizeof L.pbase);
  *rows = L.rows;
  return SPX_OK;
}

static int build_ln_tab(spx_engine* e) {
  if (e->nrt_ln_built || !e->nrt_ln_ok) return SPX_OK;
  constexpr spx::LnLayout L = spx::make_ln_layout();
  // [L.rows][N] per-node tables, then what every workgroup keeps in LDS (spx::LnConst: it used to be rebuilt by every block from
  // the constant-memory layout — 384 dependent byte loads per thread, ~30 us per block)
  const size_t per_node = static_cast<size_t>(L.rows) * static_cast<size_t>(e->n_nodes);
  std::vector<uint32_t> tab(per_node + spx::kLnConstWords);
  int rc = spx_internal_ln_tables(e->h_nrt_cost.data(), e->h_nrt_nz.data(), e->n_nodes, tab.data());
  if (rc) return fail(e, rc, "LeastNUMANodes tables");
  {
    uint32_t* allow = tab.data() + per_node;  // [256 zone sets V][kLnDwords]: the subsets inside V, in the bit layout
    for (uint32_t vset = 0; vset < 256; ++vset)
      for (int d = 0; d < spx::kLnDwords; ++d) {
        uint32_t bits = 0;
        for (int q = 0; q < 32; ++q) {
          const uint32_t sub = L.subset[d][q];
          if (sub != 0 && (sub & ~vset) == 0) bits |= 1u << q;
        }
        allow[vset * spx::kLnDwords + d] = bits;
      }
    std::memcpy(allow + 256 * spx::kLnDwords, L.subset, sizeof L.subset);  // [kLnDwords][32] bytes: bit position -> zone mask
  }
  if ((rc = upload(e, e->d_nrt_ln, tab.data(), tab.size() * sizeof(uint32_t)))) return rc;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->nrt_ln_built = true;
  return SPX_OK;
}

namespace {
// The canonical view of a pod record, on which the pod equivalence classes are built (nrt_build_classes).  Two pods get the same
// NRT rows on every node when their records agree in everything the sweep reads, and a queue is full of such pods: replicas of
// one Deployment, and every pod whose verdict does not depend on quantities — a pod that is not filtered (BestEffort without
// non-native resources, filter.go:186-190) passes and scores 100 whatever it asks for; a non-Guaranteed pod scores 100
// (score.go:72-76) and its NUMA-affine requests suit any reporting zone (numaresources.go:137-142), so only their presence
// counts.  The view is never materialised: a header pair and, per live item (pod level + the n_ctr containers), IW dwords
// produced on the stack; everything past the last container is out of it (equal headers = equal n_ctr).
struct NrtCanon {
  size_t RMs, IW;
  struct Head {
    uint32_t w0, w1;
    size_t n_items;  // 0: a pod nothing but whose class is read
    bool guaranteed;
  };
  Head head_of(const uint32_t* w) const {
    const uint32_t qos = w[0] & 0xffu, n_ctr = (w[0] >> 16) & 0xffu;
    const bool non_native = ((w[0] >> 8) & 0xffu) != 0;
    if (qos == SPX_QOS_BESTEFFORT && !non_native) return Head{qos, 0u, 0, false};
    const bool g = qos == SPX_QOS_GUARANTEED;
    return Head{w[0], g ? w[1] : 0u, 1 + static_cast<size_t>(n_ctr), g};  // the mean over containers (w1) belongs to the Score
  }
  void item(const uint32_t* it, bool guaranteed, uint32_t* c) const {
    std::memcpy(c, it, IW * sizeof(uint32_t));
    if (guaranteed) return;
    const uint32_t sets = it[2 * RMs], fit = (sets >> 8) & 0xffu;
    for (size_t r = 0; r < RMs; ++r)
      if (!((fit >> r) & 1u)) c[2 * r] = c[2 * r + 1] = 0;  // only compared quantities matter
    c[2 * RMs] = sets & 0xffffff00u;                        // "requested" steers the Score only
    for (size_t k = 2 * RMs + 1; k < IW; ++k) c[k] = 0;      // weight sums, Value() of the cpu request
  }
  static uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v;
    h *= 0xff51afd7ed558ccdull;
    return h ^ (h >> 29);
  }
  uint64_t hash(const uint32_t* w) const {
    uint32_t c[32];
    const Head hd = head_of(w);
    uint64_t h = mix(0x9e3779b97f4a7c15ull, (static_cast<uint64_t>(hd.w1) << 32) | hd.w0);
    for (size_t s = 1; s <= hd.n_items; ++s) {
      item(w + s * IW, hd.guaranteed, c);
      for (size_t k = 0; k < IW; k += 2) h = mix(h, (static_cast<uint64_t>(c[k + 1]) << 32) | c[k]);
    }
    return h;
  }
  bool equal(const uint32_t* wa, const uint32_t* wb) const {
    uint32_t ca[32], cb[32];
    const Head ha = head_of(wa), hb = head_of(wb);
    bool same = ha.w0 == hb.w0 && ha.w1 == hb.w1 && ha.n_items == hb.n_items;
    for (size_t s = 1; same && s <= ha.n_items; ++s) {
      item(wa + s * IW, ha.guaranteed, ca);
      item(wb + s * IW, hb.guaranteed, cb);
      same = std::memcmp(ca, cb, IW * sizeof(uint32_t)) == 0;
    }
    return same;
  }
};

// The pod record stream of the float64 NRT formulation, built on the host (no device involved: spx_internal_nrt_pod_classes lets
// the CPU tests see it).
// nrt_pod_items: per pod 10 items of IW dwords (IW = 16 for <= 4 resource slots, else 32), RM = 4 or 8 slots:
//   item 0   header: w0 = qos | non_native << 8 | n_ctr << 16 | last app container << 24 (0xff: none),
//                    w1 = ceil(2^16 / n_ctr)
//   item 1   the pod-level effective request;  items 2..9  the containers, in order
//   request item: doubles raw[RM] (dwords 0..2RM-1); dword 2RM = requested slots | compared slots << 8 |
//                 "any reporting zone suits" slots << 16 | kind << 24; dword 2RM+1 = sum of the weights of the requested
//                 slots as an integer; then what only the Score reads: Value() of the
//                 cpu request (2RM+2), sum of the weights of the requested slots (2RM+4), its biased reciprocal (2RM+6)
// hash_out (optional): the hash of each record's canonical view, taken while the record is still in cache
void nrt_build_items(const spx_nrt_pods_soa* t, const uint8_t* slot_flags, int cpu_slot, const std::vector<double>& wtab, uint32_t* items,
                     bool* ok_out, uint32_t* big_out, uint64_t* hash_out, spx_engine::NrtQty* qty_out = nullptr) {
  const size_t p = static_cast<size_t>(t->n_pods), R = static_cast<size_t>(t->n_res);
  constexpr size_t Cm = SPX_NRT_MAX_CTRS;
  const int RMs = R <= 4 ? 4 : 8;
  const size_t IW = R <= 4 ? 16 : 32;
  const uint32_t slot_mask = (1u << R) - 1u;
  std::atomic<bool> ok{wtab.size() == (static_cast<size_t>(2) << R)};
  std::atomic<uint32_t> big_pods{0};
  auto put_f64 = [](uint32_t* w, double v) { std::memcpy(w, &v, sizeof v); };
  const bool tab_ok = ok.load();
  const NrtCanon canon{static_cast<size_t>(RMs), IW};
  // bad / big: per calling thread, merged once per chunk (the shared flags would bounce between the cores otherwise)
  std::mutex qty_mu;
  auto fill = [&](uint32_t* w, uint32_t present, const int64_t* req, bool non_g, uint32_t kind, bool& bad, uint32_t& big, spx_engine::NrtQty& qty) {
    const uint32_t used = present & slot_mask;
    uint32_t fit = 0, always = 0;
    for (size_t r = 0; r < R; ++r) {
      if (!nrt_fast_qty(req[r])) bad = true;
      if (!nrt_exact_f32(static_cast<double>(nrt_value_of(static_cast<int>(r) == cpu_slot, req[r])))) big |= 1u << r;
      put_f64(w + 2 * r, static_cast<double>(req[r]));
      if (((used >> r) & 1u) && req[r] > 0) qty.add(static_cast<int>(r), nrt_value_of(static_cast<int>(r) == cpu_slot, req[r]));
      if (!((used >> r) & 1u) || req[r] == 0) continue;  // "ignoring zero-qty resource request" filter.go:103-106
      if (non_g && (slot_flags[r] & SPX_NRT_SLOT_AFFINE)) always |= 1u << r;
      else fit |= 1u << r;
    }
    const int64_t cpu_q = cpu_slot >= 0 ? req[cpu_slot] : 0;
    w[2 * RMs] = used | (fit << 8) | (always << 16) | (kind << 24);
    put_f64(w + 2 * RMs + 2, static_cast<double>(nrt_value_of(true, cpu_q)));
    if (tab_ok) {
      w[2 * RMs + 1] = static_cast<uint32_t>(wtab[2 * used]);  // the weight sum as an integer (< 2^20)
      put_f64(w + 2 * RMs + 4, wtab[2 * used]);
      put_f64(w + 2 * RMs + 6, wtab[2 * used + 1]);
    }
  };
  spx_host::parallel_rows(static_cast<int64_t>(p), [&](int64_t row0, int64_t row1) {
    bool bad = false;
    uint32_t big = 0;
    spx_engine::NrtQty qty;
    for (size_t i = static_cast<size_t>(row0); i < static_cast<size_t>(row1); ++i) {
      uint32_t* w = &items[i * 10 * IW];
      std::memset(w, 0, 10 * IW * sizeof(uint32_t));  // the record ends with the last container: zeros after it
      const bool non_g = t->qos[i] != SPX_QOS_GUARANTEED;
      const uint32_t n_ctr = t->n_ctr[i];
      uint32_t last_app = 0xffu;
      bool seen_app = false;
      for (size_t c = 0; c < Cm && c < n_ctr; ++c) {
        const uint32_t kind = t->ctr_kind[i * Cm + c];
        if (kind == SPX_CTR_APP) {
          last_app = static_cast<uint32_t>(c);
          seen_app = true;
        } else if (seen_app) {
          bad = true;  // the single-pass Filter needs init containers listed before app containers
        }
        fill(w + (2 + c) * IW, t->ctr_present[i * Cm + c], t->ctr_req + (i * Cm + c) * R, non_g, kind, bad, big, qty);
      }
      fill(w + IW, t->pod_present[i], t->pod_req + i * R, non_g, 0, bad, big, qty);
      w[0] = t->qos[i] | (static_cast<uint32_t>(t->non_native[i] != 0) << 8) | (n_ctr << 16) | (last_app << 24);
      w[1] = n_ctr ? (65536u + n_ctr - 1u) / n_ctr : 0u;
      if (hash_out) hash_out[i] = canon.hash(w);
    }
    if (bad) ok = false;
    if (big) big_pods.fetch_or(big, std::memory_order_relaxed);
    if (qty_out) {
      std::lock_guard<std::mutex> g(qty_mu);
      qty_out->merge(qty);
    }
  }, 4096);
  *ok_out = ok.load();
  *big_out = big_pods.load();
}

// Pod equivalence classes: rep[i] = the first row whose canonical record (NrtCanon) equals row i's (rep[i] == i: a
// representative).  hash[i] = NrtCanon::hash of row i (nrt_build_items); rows with equal hashes are verified word for word.
void nrt_build_classes(const uint32_t* items, const uint64_t* hash, size_t p, size_t R, int32_t* rep) {
  const NrtCanon canon{R <= 4 ? size_t{4} : size_t{8}, R <= 4 ? size_t{16} : size_t{32}};
  const size_t PW = 10 * canon.IW;
  // first row of each hash value: a flat open-addressing table, rows visited in order (serial: ~15 ns per row)
  {
    size_t cap = 64;
    while (cap < 2 * p) cap <<= 1;
    struct Slot {
      uint64_t h;
      int32_t row;
    };
    std::vector<Slot> tab(cap, Slot{0, -1});
    for (size_t i = 0; i < p; ++i) {
      size_t k = static_cast<size_t>(hash[i] >> 20) & (cap - 1);
      while (tab[k].row >= 0 && tab[k].h != hash[i]) k = (k + 1) & (cap - 1);
      if (tab[k].row < 0) tab[k] = Slot{hash[i], static_cast<int32_t>(i)};
      rep[i] = tab[k].row;
    }
  }
  spx_host::parallel_rows(static_cast<int64_t>(p), [&](int64_t row0, int64_t row1) {
    for (int64_t i = row0; i < row1; ++i) {
      const int32_t r0 = rep[static_cast<size_t>(i)];
      if (r0 != i && !canon.equal(items + static_cast<size_t>(i) * PW, items + static_cast<size_t>(r0) * PW))
        rep[static_cast<size_t>(i)] = static_cast<int32_t>(i);  // a hash collision: the row stands for itself
    }
  }, 2048);
}

// The rank-space Filter's input (kernels_nrt_rank.hip), built per chunk of 32 listed rows: what the chunk's pods ask for, as RANKS.
// For every resource slot the chunk's distinct compared quantities, sorted, behind a leading 0 ("any reporting zone"): a node's
// zone then needs one number per resource — how many of them its available quantity reaches — and "available >= request" becomes
// "that count >= the request's position + 1", an 11-bit integer comparison the kernel does with a subtract (two zones per dword).
// The container-scope handler charges an app container to the zone it chose before the next one is tested
// (filter.go:131-163 -> numaresources.go:145-182); instead of mutating the zone table, the later container is compared with the
// SUM of the requests a zone would have been charged — available - charged >= request  <=>  available >= charged + request, exact
// in integers — so the chunk's lists also hold those sums: per pod 13 comparison vectors (layout: spx::kRk*, spx_internal.h):
// the pod-level request, the eight containers, and for the second / third app container the sums with the earlier app
// containers a zone may carry.  Pods with more than three app containers have no such finite list: *ok_out = false and the
// batch keeps the float64 Filter.  A chunk = up to 32 consecutive listed rows (first_out[c] .. first_out[c + 1]); chunk block: 16 header
// dwords (per slot: search steps | list offset << 8; [8] rows; [9] narrow),
// the lists (2^steps - 1 doubles each, padded with +inf), then per pod kRkPodHead + 13 x RM dwords.
void nrt_build_rank_stream(const uint32_t* items, const int32_t* list, size_t n_list, size_t R, std::vector<uint32_t>& words, std::vector<uint32_t>& off,
                           std::vector<uint32_t>& first_out, uint32_t* max_dwords_out, bool* ok_out, bool* all_narrow_out, bool narrow_ok = true) {
  const size_t RM = R <= 4 ? 4 : 8, IW = R <= 4 ? 16 : 32, PW = 10 * IW, PWR = spx::kRkPodHead + spx::kRkVectors * RM;
  const size_t n_groups = (n_list + spx::kRkChunkRows - 1) / spx::kRkChunkRows;
  struct Block {
    uint32_t first, rows;
    std::vector<uint32_t> w;
  };
  std::vector<std::vector<Block>> groups(n_groups);  // a group = 32 consecutive listed rows = one chunk, or the chunks it was split into
  std::atomic<bool> ok{true}, all_narrow{true};
  auto f64 = [](const uint32_t* w) { double v; std::memcpy(&v, w, sizeof v); return v; };
  spx_host::parallel_rows(static_cast<int64_t>(n_groups), [&](int64_t c0, int64_t c1) {
    std::vector<double> vals[SPX_NRT_MAX_RES];
    for (int64_t c = c0; c < c1; ++c) {
      const size_t first = static_cast<size_t>(c) * spx::kRkChunkRows, rows = std::min<size_t>(spx::kRkChunkRows, n_list - first);
      // pass 1: every pod's 13 vectors (value per slot, NaN = not compared)
      std::vector<double> vec(rows * spx::kRkVectors * RM, std::numeric_limits<double>::quiet_NaN());
      std::vector<uint32_t> head(rows * spx::kRkPodHead, 0u);
      std::vector<uint8_t> any_always(rows * spx::kRkVectors, 0);  // per vector: the item's "any reporting zone suits" slots
      for (size_t i = 0; i < rows; ++i) {
        const uint32_t* w = items + static_cast<size_t>(list[first + i]) * PW;
        uint32_t* h = &head[i * spx::kRkPodHead];
        h[0] = w[0], h[1] = w[1];
        const uint32_t n_ctr = (w[0] >> 16) & 0xffu;
        uint32_t app[3] = {0xffu, 0xffu, 0xffu}, n_app = 0;
        for (size_t k = 1; k <= 9; ++k) h[1 + k] = w[k * IW + 2 * RM];  // the items' slot sets (absent items are zero)
        for (uint32_t ctr = 0; ctr < n_ctr && ctr < SPX_NRT_MAX_CTRS; ++ctr)
          if ((h[3 + ctr] >> 24) == SPX_CTR_APP) {
            if (n_app < 3) app[n_app] = ctr;
            ++n_app;
          }
        if (n_app > 3) ok = false;
        h[11] = app[0] | (app[1] << 8) | (app[2] << 16) | (std::min<uint32_t>(n_app, 255u) << 24);
        auto fit_of = [&](size_t item) { return (w[item * IW + 2 * RM] >> 8) & 0xffu; };
        auto raw_of = [&](size_t item, size_t r) { return f64(w + item * IW + 2 * r); };
        double* v = &vec[i * spx::kRkVectors * RM];
        auto put = [&](size_t vi, size_t item, std::initializer_list<uint32_t> charged) {
          const uint32_t fit = fit_of(item);
          any_always[i * spx::kRkVectors + vi] = static_cast<uint8_t>((w[item * IW + 2 * RM] >> 16) & 0xffu);
          for (size_t r = 0; r < R; ++r) {
            if (!((fit >> r) & 1u)) continue;
            double q = raw_of(item, r);
            for (uint32_t j : charged)
              if ((fit_of(2 + j) >> r) & 1u) q += raw_of(2 + j, r);
            v[vi * RM + r] = q;
          }
        };
        put(0, 1, {});
        for (uint32_t ctr = 0; ctr < n_ctr && ctr < SPX_NRT_MAX_CTRS; ++ctr) put(1 + ctr, 2 + ctr, {});
        if (n_app >= 2 && n_app <= 3) put(9, 2 + app[1], {app[0]});
        if (n_app == 3) put(10, 2 + app[2], {app[0]}), put(11, 2 + app[2], {app[1]}), put(12, 2 + app[2], {app[0], app[1]});
        // per container one byte of what the fused sweep (kernels_nrt_fused.hip) does with it, so that its loop tests bits instead of
        // deriving them (h[12]: containers 0-3, h[13]: 4-7): bits 0-2 the status a misfit sets, spx::kRkOp*
        const uint32_t last_app = w[0] >> 24;
        for (uint32_t ctr = 0; ctr < n_ctr && ctr < SPX_NRT_MAX_CTRS; ++ctr) {
          const uint32_t kind = h[3 + ctr] >> 24, fit = fit_of(2 + ctr);
          uint32_t op = kind == SPX_CTR_APP ? SPX_NRT_ST_CONTAINER : (kind == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER);
          if (kind == SPX_CTR_APP && fit != 0 && n_app <= 3) {
            if (ctr == app[1]) op |= spx::kRkOpMerge1;
            else if (ctr == app[2]) op |= spx::kRkOpMerge3;
            if (ctr != last_app) op |= ctr == app[0] ? spx::kRkOpCharge0 : spx::kRkOpCharge1;
          }
          h[12 + (ctr >> 2)] |= op << (8 * (ctr & 3));
        }
      }
      // pass 2: the chunk [lo, hi) of the group — its lists, then the thresholds.  A chunk whose lists all have at most 127 entries (leading 0
      // included) is "narrow": positions and counts fit 7 bits, the kernels pack four zones per dword (RkLayout<true>) and the thresholds are
      // replicated into four bytes instead of two halves.  With narrow_ok a chunk that is not is split in halves until it is (a single pod
      // compares at most 13 values per slot), so that every chunk of the stream is narrow — the fused sweep has no other layout.
      std::vector<Block>& out = groups[static_cast<size_t>(c)];
      auto emit = [&](auto&& self, size_t lo, size_t hi) -> void {
        for (size_t r = 0; r < R; ++r) {
          auto& a = vals[r];
          a.clear();
          a.push_back(0.0);
          for (size_t i = lo; i < hi; ++i)
            for (size_t vi = 0; vi < spx::kRkVectors; ++vi) {
              const double q = vec[(i * spx::kRkVectors + vi) * RM + r];
              if (q == q) a.push_back(q);
            }
          std::sort(a.begin(), a.end());
          a.erase(std::unique(a.begin(), a.end()), a.end());
        }
        bool narrow = true;
        for (size_t r = 0; r < R; ++r) narrow = narrow && vals[r].size() <= 127;
        if (narrow_ok && !narrow && hi - lo > 1) {
          const size_t mid = lo + (hi - lo) / 2;
          self(self, lo, mid);
          self(self, mid, hi);
          return;
        }
        narrow = narrow && narrow_ok;
        if (!narrow) all_narrow = false;
        uint32_t steps[SPX_NRT_MAX_RES] = {0}, loff[SPX_NRT_MAX_RES] = {0};
        size_t list_doubles = 0;
        for (size_t r = 0; r < R; ++r) {
          uint32_t k = 1;
          while ((size_t{1} << k) - 1 < vals[r].size()) ++k;
          steps[r] = k, loff[r] = static_cast<uint32_t>(list_doubles);
          list_doubles += (size_t{1} << k);  // 2^k - 1 entries and one pad: every list starts 16-byte aligned
        }
        out.emplace_back();
        Block& blk = out.back();
        blk.first = static_cast<uint32_t>(first + lo), blk.rows = static_cast<uint32_t>(hi - lo);
        std::vector<uint32_t>& b = blk.w;
        b.assign(16 + 2 * list_doubles + (hi - lo) * PWR, 0u);
        for (size_t r = 0; r < R; ++r) b[r] = steps[r] | (loff[r] << 8);
        b[8] = static_cast<uint32_t>(hi - lo);
        b[9] = narrow ? 1u : 0u;
        for (size_t r = 0; r < R; ++r) {
          double* dst = reinterpret_cast<double*>(&b[16]) + loff[r];
          const size_t n = size_t{1} << steps[r];
          for (size_t j = 0; j < n; ++j) dst[j] = j < vals[r].size() ? vals[r][j] : std::numeric_limits<double>::infinity();
        }
        for (size_t i = lo; i < hi; ++i) {
          uint32_t* dst = &b[16 + 2 * list_doubles + (i - lo) * PWR];
          std::memcpy(dst, &head[i * spx::kRkPodHead], spx::kRkPodHead * sizeof(uint32_t));
          for (size_t vi = 0; vi < spx::kRkVectors; ++vi)
            for (size_t r = 0; r < R; ++r) {
              const double q = vec[(i * spx::kRkVectors + vi) * RM + r];
              // a non-Guaranteed pod's NUMA-affine request: "count >= 1" (filter.go:120-129); k_nrt_filter_rank derives it from the slot
              // sets, the fused sweep reads it here; a slot the item does not compare keeps 0 ("count >= 0": every zone passes)
              if ((any_always[i * spx::kRkVectors + vi] >> r) & 1u) dst[spx::kRkPodHead + vi * RM + r] = narrow ? 0x01010101u : 0x00010001u;
              if (q != q) continue;
              const uint32_t t = static_cast<uint32_t>(std::lower_bound(vals[r].begin(), vals[r].end(), q) - vals[r].begin()) + 1u;
              dst[spx::kRkPodHead + vi * RM + r] = narrow ? t * 0x01010101u : (t | (t << 16));
            }
        }
      };
      emit(emit, 0, rows);
    }
  }, 8);
  size_t n_chunks = 0;
  for (const auto& g : groups) n_chunks += g.size();
  off.assign(n_chunks + 1, 0u);
  first_out.assign(n_chunks + 1, static_cast<uint32_t>(n_list));
  std::vector<const Block*> flat;
  flat.reserve(n_chunks);
  for (const auto& g : groups)
    for (const Block& blk : g) flat.push_back(&blk);
  uint32_t max_dwords = 0;
  for (size_t c = 0; c < n_chunks; ++c) {
    off[c + 1] = off[c] + static_cast<uint32_t>((flat[c]->w.size() + 3) & ~size_t{3});
    first_out[c] = flat[c]->first;
    max_dwords = std::max<uint32_t>(max_dwords, off[c + 1] - off[c]);
  }
  words.assign(off[n_chunks], 0u);
  spx_host::parallel_rows(static_cast<int64_t>(n_chunks), [&](int64_t c0, int64_t c1) {
    for (int64_t c = c0; c < c1; ++c) std::memcpy(&words[off[static_cast<size_t>(c)]], flat[static_cast<size_t>(c)]->w.data(), flat[static_cast<size_t>(c)]->w.size() * sizeof(uint32_t));
  }, 64);
  *max_dwords_out = max_dwords;
  *ok_out = ok.load();
  *all_narrow_out = all_narrow.load();
}
}  // namespace

// builds the stream of `list` and ships it; e->nrt_rk_kind = kind on success, 0 when it does not fit, -1 when the batch has none
int nrt_rank_stream_upload(spx_engine* e, const uint32_t* items, const int32_t* list, size_t n_list, int kind) {
  std::vector<uint32_t> rk, rk_off, rk_first;
  uint32_t rk_max = 0;
  bool rk_ok = false, all_narrow = false;
  nrt_build_rank_stream(items, list, n_list, static_cast<size_t>(e->nrt_n_res), rk, rk_off, rk_first, &rk_max, &rk_ok, &all_narrow, e->option[SPX_OPT_NRT_RANK_NARROW] != 0);
  e->nrt_rk_max_dwords = 0;
  e->nrt_rk_kind = rk_ok ? 0 : -1;
  if (rk_ok && rk_max * sizeof(uint32_t) <= spx::kRkMaxChunkBytes) {
    int rc;
    if ((rc = upload(e, e->d_nrt_rk, rk.data(), rk.size() * sizeof(uint32_t)))) return rc;
    if ((rc = upload(e, e->d_nrt_rk_off, rk_off.data(), rk_off.size() * sizeof(uint32_t)))) return rc;
    if ((rc = upload(e, e->d_nrt_rk_first, rk_first.data(), rk_first.size() * sizeof(uint32_t)))) return rc;
    SPX_HIP(e, hipStreamSynchronize(e->stream));
    e->nrt_rk_max_dwords = rk_max;
    e->nrt_rk_chunks = static_cast<uint32_t>(rk_first.size() - 1);
    e->nrt_rk_all_narrow = all_narrow;
    e->nrt_rk_kind = kind;
  }
  return SPX_OK;
}

// The rank stream over EVERY row of the uploaded batch, in order (a whole-batch sweep without pod classes: SPX_OPT_NRT_POD_CLASSES 0,
// or a queue with too few repeats for them): built the first time such a sweep runs after an upload — the record stream comes back
// from the device (the host copy was staging) — and kept until the next upload or until a sweep over the classes replaces it.
int nrt_rank_stream(spx_engine* e, int kind) {
  if (e->nrt_rk_kind == kind) return SPX_OK;
  if (e->nrt_rk_kind < 0 || !e->nrt_fast_pods || e->n_pods <= 0) return SPX_OK;  // no finite stream for this batch: the float64 Filter
  if (kind == 1 && e->nrt_n_dups == 0) return SPX_OK;
  const size_t p = static_cast<size_t>(e->n_pods), R = static_cast<size_t>(e->nrt_n_res), IW = R <= 4 ? 16 : 32;
  std::vector<uint32_t> items(p * 10 * IW);
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  SPX_HIP(e, hipMemcpy(items.data(), e->d_nrt_items.p, items.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<int32_t> list;
  if (kind == 2) {
    list.resize(p);
    for (size_t i = 0; i < p; ++i) list[i] = static_cast<int32_t>(i);
  } else {
    list.resize(static_cast<size_t>(e->nrt_n_uniq));
    SPX_HIP(e, hipMemcpy(list.data(), e->d_nrt_uniq.p, list.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return nrt_rank_stream_upload(e, items.data(), list.data(), list.size(), kind);
}

// test hook (host only, no device): the representative row of every pod of a batch, as spx_upload_nrt_pods computes it
// (rep_out[i] == i for a representative); *fast_ok_out = whether the batch satisfies the float64 formulation's preconditions
// (the classes are only built, and only used, when it does)
int spx_internal_nrt_pod_classes(const spx_nrt_slots* slots, const spx_nrt_pods_soa* t, int32_t* rep_out, int32_t* fast_ok_out) {
  if (!slots || !t || !rep_out || !fast_ok_out || t->n_res != slots->n_res || t->n_pods <= 0) return SPX_ERR_ARG;
  const int R = t->n_res;
  int cpu_slot = -1;
  int64_t wtotal = 0;
  bool slots_ok = true;
  for (int i = 0; i < R; ++i) {
    if (slots->slot_flags[i] & SPX_NRT_SLOT_CPU) cpu_slot = i;
    if (slots->slot_weight[i] < 0 || slots->slot_weight[i] >= kNrtWeightLimit) slots_ok = false;
    else wtotal += slots->slot_weight[i];
  }
  if (wtotal >= kNrtWeightLimit) slots_ok = false;
  std::vector<double> wtab(static_cast<size_t>(2) << R, 0.0);
  for (unsigned m = 0; m < (1u << R); ++m) {
    int64_t w = 0;
    for (int i = 0; i < R; ++i)
      if ((m >> i) & 1u) w += slots->slot_weight[i];
    wtab[2 * m] = static_cast<double>(w);
    wtab[2 * m + 1] = nrt_biased_rcp(static_cast<double>(w));
  }
  const size_t p = static_cast<size_t>(t->n_pods), IW = R <= 4 ? 16 : 32;
  std::vector<uint32_t> items(p * 10 * IW);
  std::vector<uint64_t> hash(p);
  bool ok = false;
  uint32_t big = 0;
  nrt_build_items(t, slots->slot_flags, cpu_slot, wtab, items.data(), &ok, &big, hash.data());
  *fast_ok_out = (ok && slots_ok) ? 1 : 0;
  for (size_t i = 0; i < p; ++i) rep_out[i] = static_cast<int32_t>(i);
  if (ok && slots_ok) nrt_build_classes(items.data(), hash.data(), p, static_cast<size_t>(R), rep_out);
  return SPX_OK;
}

int spx_upload_nrt_pods(spx_engine* e, const spx_nrt_pods_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->nrt_slots || t->n_res != e->nrt_n_res) return fail(e, SPX_ERR_STATE, "NRT: upload the slot table first (n_res mismatch)");
  int rc = set_pods(e, t->n_pods);
  if (rc) return rc;
  const size_t p = static_cast<size_t>(t->n_pods);
  const size_t R = static_cast<size_t>(t->n_res);
  constexpr size_t Cm = SPX_NRT_MAX_CTRS;
  if ((rc = upload(e, e->d_nrt_qos, t->qos, p))) return rc;
  if ((rc = upload(e, e->d_nrt_nn, t->non_native, p))) return rc;
  if ((rc = upload(e, e->d_nrt_nctr, t->n_ctr, p))) return rc;
  if ((rc = upload(e, e->d_nrt_ckind, t->ctr_kind, p * Cm))) return rc;
  if ((rc = upload(e, e->d_nrt_cpres, t->ctr_present, p * Cm))) return rc;
  if (!t->ctr_req && p * R) return fail(e, SPX_ERR_ARG, "NULL column in table");
  if ((rc = upload(e, e->d_nrt_ppres, t->pod_present, p))) return rc;
  if ((rc = upload(e, e->d_nrt_preq, t->pod_req, p * R * 8))) return rc;
  {  // float64 formulation: the pod record stream (nrt_build_items) + its preconditions, then the pod equivalence classes
    const size_t IW = R <= 4 ? 16 : 32;
    const size_t items_bytes = p * 10 * IW * sizeof(uint32_t);
    if (e->h_items_bytes < items_bytes) {
      if (e->h_items) SPX_HIP(e, hipHostFree(e->h_items));
      e->h_items = nullptr, e->h_items_bytes = 0;
      SPX_HIP(e, hipHostMalloc(&e->h_items, items_bytes + (items_bytes >> 3), hipHostMallocDefault));
      e->h_items_bytes = items_bytes + (items_bytes >> 3);
    }
    uint32_t* const items = static_cast<uint32_t*>(e->h_items);  // pinned: built in place (rows zeroed by the thread that fills them)
    bool ok = false;
    uint32_t big = 0;
    std::vector<uint64_t> hash(p);
    spx_engine::NrtQty qty;
    nrt_build_items(t, e->nrt_slot_flags, e->nrt_cpu_slot, e->nrt_wtab, items, &ok, &big, hash.data(), &qty);
    if ((rc = upload(e, e->d_nrt_items, items, items_bytes))) return rc;  // from pinned memory: one DMA at link speed, asynchronous
    e->nrt_fast_pods = ok;
    e->nrt_big_pods = big;
    e->nrt_qty_pods = qty;
    e->nrt_pk_tab_built = false;  // (the table's unit and length follow the batch)
    // the reference-arithmetic kernel's request column: shipped only when the record stream cannot stand in for it
    e->nrt_creq_valid = false;
    if (!e->nrt_fast_pods) {
      if ((rc = upload(e, e->d_nrt_creq, t->ctr_req, p * Cm * R * 8))) return rc;
      e->nrt_creq_valid = true;
    }
    e->nrt_n_uniq = e->nrt_n_dups = 0;
    e->nrt_rk_max_dwords = 0;
    e->nrt_rk_kind = 0;
    ++e->nrt_items_gen;
    if (e->nrt_fast_pods && p > 0) {
      std::vector<int32_t> rep(p);
      nrt_build_classes(items, hash.data(), p, R, rep.data());
      std::vector<int32_t> uniq, dups;
      uniq.reserve(p), dups.reserve(2 * p);
      for (size_t i = 0; i < p; ++i) {
        if (rep[i] == static_cast<int32_t>(i)) uniq.push_back(static_cast<int32_t>(i));
        else dups.push_back(static_cast<int32_t>(i)), dups.push_back(rep[i]);
      }
      if (!dups.empty()) {
        if ((rc = upload(e, e->d_nrt_uniq, uniq.data(), uniq.size() * sizeof(int32_t)))) return rc;
        if ((rc = upload(e, e->d_nrt_dups, dups.data(), dups.size() * sizeof(int32_t)))) return rc;
        SPX_HIP(e, hipStreamSynchronize(e->stream));
        e->nrt_n_uniq = static_cast<int64_t>(uniq.size());
        e->nrt_n_dups = static_cast<int64_t>(dups.size() / 2);
        // the representatives' requests as ranks, per chunk of up to 32 (kernels_nrt_rank.hip, kernels_nrt_fused.hip)
        if ((rc = nrt_rank_stream_upload(e, items, uniq.data(), uniq.size(), 1))) return rc;
      }
    }
  }
  e->nrt_pods = true;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_upload_net_nodes(spx_engine* e, const spx_net_nodes_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_nodes(e, t->n_nodes);
  if (rc) return rc;
  if (!t->region || !t->zone) return fail(e, SPX_ERR_ARG, "NULL column in table");
  const int64_t n = t->n_nodes;
  // topology classes: nodes with identical (region, zone) labels are interchangeable for every pair that is
  // not hosted on them
  std::vector<int32_t> cls(static_cast<size_t>(n)), cr, cz;
  {
    std::vector<std::pair<int64_t, int32_t>> seen;  // sorted (packed label pair -> class)
    std::vector<int64_t> keys(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) keys[i] = (static_cast<int64_t>(t->region[i]) << 32) ^ static_cast<uint32_t>(t->zone[i]);
    std::vector<int64_t> uniq(keys);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    for (int64_t i = 0; i < n; ++i)
      cls[i] = static_cast<int32_t>(std::lower_bound(uniq.begin(), uniq.end(), keys[i]) - uniq.begin());
    cr.resize(uniq.size());
    cz.resize(uniq.size());
    for (int64_t i = 0; i < n; ++i) {
      cr[cls[i]] = t->region[i];
      cz[cls[i]] = t->zone[i];
    }
  }
  int32_t n_classes = static_cast<int32_t>(cr.size());
  if (spx::net_lds_bytes(n_classes, n) > 52 * 1024) n_classes = 0;  // too many label pairs for LDS (64 KB with a single-row launch's staged pairs): exact path only
  e->net_n_classes = n_classes;
  if ((rc = upload(e, e->d_net_region, t->region, static_cast<size_t>(n) * 4))) return rc;
  if ((rc = upload(e, e->d_net_zone, t->zone, static_cast<size_t>(n) * 4))) return rc;
  if ((rc = upload(e, e->d_net_class, cls.data(), static_cast<size_t>(n) * 4))) return rc;
  {
    std::vector<uint16_t> c16(static_cast<size_t>(spx::round_up(n, 16)), 0);  // (k_net_cls reads groups of 16)
    std::vector<int32_t> size(cr.size() ? cr.size() : 1, 0);
    e->net_class16 = cr.size() <= 65535;
    for (int64_t i = 0; i < n; ++i) {
      c16[static_cast<size_t>(i)] = static_cast<uint16_t>(cls[static_cast<size_t>(i)]);
      ++size[static_cast<size_t>(cls[static_cast<size_t>(i)])];
    }
    if ((rc = upload(e, e->d_net_class16, c16.data(), c16.size() * 2))) return rc;
    if ((rc = upload(e, e->d_net_cls_size, size.data(), size.size() * 4))) return rc;
    SPX_HIP(e, hipStreamSynchronize(e->stream));
  }
  if ((rc = upload(e, e->d_net_cls_region, cr.data(), cr.size() * 4))) return rc;
  if ((rc = upload(e, e->d_net_cls_zone, cz.data(), cz.size() * 4))) return rc;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->net_nodes = true;
  return SPX_OK;
}

int spx_upload_net_topo(spx_engine* e, const spx_net_topo_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (t->n_regions < 0 || t->n_zones < 0) return fail(e, SPX_ERR_ARG, "negative topology size");
  int rc;
  if ((rc = upload(e, e->d_net_rcost, t->region_cost ? static_cast<const void*>(t->region_cost) : static_cast<const void*>(&rc),
                   static_cast<size_t>(t->n_regions) * t->n_regions * 4)))
    return rc;
  if ((rc = upload(e, e->d_net_zcost, t->zone_cost ? static_cast<const void*>(t->zone_cost) : static_cast<const void*>(&rc),
                   static_cast<size_t>(t->n_zones) * t->n_zones * 4)))
    return rc;
  e->net_n_regions = t->n_regions;
  e->net_n_zones = t->n_zones;
  e->net_max_cost = SPX_NET_MAX_COST;
  for (int64_t i = 0; t->region_cost && i < static_cast<int64_t>(t->n_regions) * t->n_regions; ++i) e->net_max_cost = std::max<int64_t>(e->net_max_cost, t->region_cost[i]);
  for (int64_t i = 0; t->zone_cost && i < static_cast<int64_t>(t->n_zones) * t->n_zones; ++i) e->net_max_cost = std::max<int64_t>(e->net_max_cost, t->zone_cost[i]);
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->net_topo = true;
  return SPX_OK;
}

int spx_upload_net_pods(spx_engine* e, const spx_net_pods_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_pods(e, t->n_pods);
  if (rc) return rc;
  if (t->n_keys <= 0 || !t->pair_ptr) return fail(e, SPX_ERR_ARG, "net pods: key table missing");
  const size_t pairs = static_cast<size_t>(t->pair_ptr[t->n_keys]);
  e->net_max_pairs = 0;
  for (int32_t k = 0; k < t->n_keys; ++k) e->net_max_pairs = std::max<int64_t>(e->net_max_pairs, t->pair_ptr[k + 1] - t->pair_ptr[k]);
  e->h_pair_ptr.assign(t->pair_ptr, t->pair_ptr + t->n_keys + 1);
  e->h_key_flag.assign(t->key_score_equally, t->key_score_equally + t->n_keys);
  e->net_n_keys = t->n_keys;
  e->net_commit = false;  // the commit effects refer to the previous key numbering
  if ((rc = upload(e, e->d_net_pod_key, t->pod_key, static_cast<size_t>(t->n_pods) * 4))) return rc;
  if ((rc = upload(e, e->d_net_key_flag, t->key_score_equally, static_cast<size_t>(t->n_keys)))) return rc;
  if ((rc = upload(e, e->d_net_pair_ptr, t->pair_ptr, static_cast<size_t>(t->n_keys + 1) * 4))) return rc;
  if ((rc = upload(e, e->d_net_pair_node, pairs ? static_cast<const void*>(t->pair_node) : static_cast<const void*>(&rc), pairs * 4))) return rc;
  if ((rc = upload(e, e->d_net_pair_max, pairs ? static_cast<const void*>(t->pair_max_cost) : static_cast<const void*>(&rc), pairs * 8))) return rc;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->net_pods = true;
  return SPX_OK;
}

int spx_upload_sort_keys(spx_engine* e, const spx_sort_keys_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (t->n_pods <= 0 || t->n_pods >= (int64_t{1} << 31)) return fail(e, SPX_ERR_ARG, "sort keys: n_pods must be in [1, 2^31)");
  const size_t p = static_cast<size_t>(t->n_pods);
  int rc;
  if ((rc = upload(e, e->d_sort_prio, t->priority, p * 4))) return rc;
  if ((rc = upload(e, e->d_sort_ts, t->queue_ts, p * 8))) return rc;
  if ((rc = upload(e, e->d_sort_group, t->appgroup, p * 4))) return rc;
  if ((rc = upload(e, e->d_sort_topo, t->topo_order, p * 4))) return rc;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->sort_n = t->n_pods;
  return SPX_OK;
}

int spx_sort_keys(spx_engine* e, int32_t* perm_out) {
  if (!e || !perm_out) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (e->sort_n <= 0) return fail(e, SPX_ERR_STATE, "TopologicalSort: spx_upload_sort_keys not called");
  int rc;
  if ((rc = ensure(e, e->d_sort_scratch, spx::sort_scratch_bytes(e->sort_n)))) return rc;
  if (!e->h_sort_hist) SPX_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_sort_hist), 16 * 256 * sizeof(unsigned), hipHostMallocDefault));
  spx::SortArgs a{};
  a.n = e->sort_n;
  a.priority = static_cast<const int32_t*>(e->d_sort_prio.p);
  a.queue_ts = static_cast<const int64_t*>(e->d_sort_ts.p);
  a.appgroup = static_cast<const int32_t*>(e->d_sort_group.p);
  a.topo_order = static_cast<const int32_t*>(e->d_sort_topo.p);
  SPX_HIP(e, hipEventRecord(e->ev0, e->stream));
  hipError_t st = hipSuccess;
  const int32_t* perm = spx::launch_sort_keys(a, e->d_sort_scratch.p, e->h_sort_hist, e->stream, &st);
  if (st != hipSuccess || !perm) return fail(e, SPX_ERR_HIP, std::string("spx_sort_keys: ") + hipGetErrorString(st));
  SPX_HIP(e, hipEventRecord(e->ev1, e->stream));
  e->timed = true;
  SPX_HIP(e, hipMemcpyAsync(perm_out, perm, static_cast<size_t>(e->sort_n) * 4, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_upload_quota(spx_engine* e, const spx_quota_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_pods(e, t->n_pods);
  if (rc) return rc;
  if (t->n_namespaces < 0 || !t->nom_ptr) return fail(e, SPX_ERR_ARG, "quota: namespace tables missing");
  const size_t P = static_cast<size_t>(t->n_pods), NS = static_cast<size_t>(t->n_namespaces), S = SPX_QUOTA_SLOTS;
  const size_t nn = static_cast<size_t>(t->nom_ptr[t->n_namespaces]);
  // a column may be NULL only when it has no entries (no namespaces / no nominated pods); upload() rejects the rest.  Every exit
  // after the first asynchronous copy waits for the stream: the host columns are only borrowed for the call.
  const int64_t dummy[SPX_QUOTA_SLOTS] = {0};
  auto col = [&](const void* p) { return p ? p : static_cast<const void*>(dummy); };
  struct Drain {
    spx_engine* e;
    ~Drain() { (void)hipStreamSynchronize(e->stream); }
  } drain{e};
  if ((NS > 0 && (!t->has_quota || !t->used || !t->max || !t->max_present || !t->other_nominated || !t->other_nominated_present)) ||
      (nn > 0 && (!t->nom_priority || !t->nom_pending_index || !t->nom_req || !t->nom_req_present)))
    return fail(e, SPX_ERR_ARG, "quota: NULL column in a non-empty table");
  if ((rc = upload(e, e->d_q_pod_ns, t->pod_ns, P * 4))) return rc;
  if ((rc = upload(e, e->d_q_pod_prio, t->pod_priority, P * 4))) return rc;
  if ((rc = upload(e, e->d_q_pod_req, t->pod_req, P * S * 8))) return rc;
  if ((rc = upload(e, e->d_q_pod_reqp, t->pod_req_present, P))) return rc;
  if ((rc = upload(e, e->d_q_has, col(t->has_quota), NS))) return rc;
  if ((rc = upload(e, e->d_q_used, col(t->used), NS * S * 8))) return rc;
  if (NS > 0 && !t->used_present) return fail(e, SPX_ERR_ARG, "quota: NULL column in a non-empty table");
  if ((rc = upload(e, e->d_q_usedp, col(t->used_present), NS))) return rc;
  e->q_has_min = t->min && t->min_present;
  if (e->q_has_min) {
    if ((rc = upload(e, e->d_q_min, t->min, NS * S * 8))) return rc;
    if ((rc = upload(e, e->d_q_minp, t->min_present, NS))) return rc;
  }
  if ((rc = upload(e, e->d_q_max, col(t->max), NS * S * 8))) return rc;
  if ((rc = upload(e, e->d_q_maxp, col(t->max_present), NS))) return rc;
  if ((rc = upload(e, e->d_q_other, col(t->other_nominated), NS * S * 8))) return rc;
  if ((rc = upload(e, e->d_q_otherp, col(t->other_nominated_present), NS))) return rc;
  if ((rc = upload(e, e->d_q_nom_ptr, t->nom_ptr, (NS + 1) * 4))) return rc;
  if ((rc = upload(e, e->d_q_nom_prio, col(t->nom_priority), nn * 4))) return rc;
  if ((rc = upload(e, e->d_q_nom_idx, col(t->nom_pending_index), nn * 8))) return rc;
  if ((rc = upload(e, e->d_q_nom_req, col(t->nom_req), nn * S * 8))) return rc;
  if ((rc = upload(e, e->d_q_nom_reqp, col(t->nom_req_present), nn))) return rc;
  if (!t->agg_used || !t->agg_min || !t->agg_used_present || !t->agg_min_present) return fail(e, SPX_ERR_ARG, "quota: aggregate vectors missing");
  std::memcpy(e->q_agg_used, t->agg_used, sizeof e->q_agg_used);
  std::memcpy(e->q_agg_min, t->agg_min, sizeof e->q_agg_min);
  e->q_agg_used_present = *t->agg_used_present;
  e->q_agg_min_present = *t->agg_min_present;
  e->q_n_namespaces = t->n_namespaces;
  e->q_n_nominated = nn;
  {
    int64_t agg[SPX_QUOTA_SLOTS + 1];
    std::memcpy(agg, t->agg_used, sizeof e->q_agg_used);
    agg[SPX_QUOTA_SLOTS] = *t->agg_used_present;
    if ((rc = upload(e, e->d_q_agg, agg, sizeof agg))) return rc;
    SPX_HIP(e, hipStreamSynchronize(e->stream));  // agg is a stack array
  }
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->quota = true;
  return SPX_OK;
}

int spx_fetch_prefilter(spx_engine* e, int plugin, int64_t row_begin, int64_t row_end, uint8_t* out) {
  if (!e || !out) return SPX_ERR_ARG;
  if (plugin != SPX_PLUGIN_CAPACITY || !(e->evaluated & (1u << SPX_PLUGIN_CAPACITY)))
    return fail(e, SPX_ERR_STATE, "CapacityScheduling.PreFilter has not been evaluated");
  if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
  if (int rc = rows_evaluated(e, SPX_PLUGIN_CAPACITY, row_begin, row_end)) return rc;
  SPX_HIP(e, hipSetDevice(e->device));
  SPX_HIP(e, hipMemcpy(out, static_cast<const uint8_t*>(e->d_q_status.p) + row_begin, static_cast<size_t>(row_end - row_begin),
                       hipMemcpyDeviceToHost));
  return SPX_OK;
}

int spx_upload_feasible_mask(spx_engine* e, const uint8_t* mask, int64_t n_pods, int64_t n_nodes) {
  if (!e) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  ++e->ext_gen;
  if (!mask) {  // clear
    e->ext_mask = false;
    return SPX_OK;
  }
  int rc = set_nodes(e, n_nodes);
  if (rc) return rc;
  if ((rc = set_pods(e, n_pods))) return rc;
  // stored like a Filter plugin's status table: 0 = passed, so that every consumer treats filters uniformly
  std::vector<uint8_t> st(static_cast<size_t>(n_pods) * static_cast<size_t>(e->row_stride), 1);
  for (int64_t p = 0; p < n_pods; ++p)
    for (int64_t n = 0; n < n_nodes; ++n) st[static_cast<size_t>(p * e->row_stride + n)] = mask[p * n_nodes + n] ? 0 : 1;
  if ((rc = upload(e, e->d_ext_status, st.data(), st.size()))) return rc;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->ext_mask = true;
  return SPX_OK;
}

int spx_eval(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end) {
  if (!e) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  const uint32_t known = (1u << SPX_PLUGIN_ALLOCATABLE) | (1u << SPX_PLUGIN_TLP) | (1u << SPX_PLUGIN_LVRB) | (1u << SPX_PLUGIN_NRT) |
                         (1u << SPX_PLUGIN_NETOVERHEAD) | (1u << SPX_PLUGIN_CAPACITY) | (1u << SPX_PLUGIN_LROC) | (1u << SPX_PLUGIN_PEAKS);
  if (plugin_mask == 0 || (plugin_mask & ~known)) return fail(e, SPX_ERR_ARG, "plugin mask has unsupported bits");
  const bool R = plugin_mask & (1u << SPX_PLUGIN_LROC);
  if (R && !(e->tri_nodes && e->lroc_nodes && e->lroc_pods)) return fail(e, SPX_ERR_STATE, "LowRiskOverCommitment node/pod tables not uploaded");
  const bool K = plugin_mask & (1u << SPX_PLUGIN_PEAKS);
  if (K && !(e->peaks_nodes && e->peaks_pods)) return fail(e, SPX_ERR_STATE, "Peaks node/pod tables not uploaded");
  const bool Q = plugin_mask & (1u << SPX_PLUGIN_CAPACITY);
  if (Q && !e->quota) return fail(e, SPX_ERR_STATE, "CapacityScheduling quota tables not uploaded");
  if (e->n_nodes <= 0 && plugin_mask != (1u << SPX_PLUGIN_CAPACITY)) return fail(e, SPX_ERR_STATE, "no node table uploaded");
  const bool A = plugin_mask & (1u << SPX_PLUGIN_ALLOCATABLE);
  const bool T = plugin_mask & (1u << SPX_PLUGIN_TLP);
  const bool L = plugin_mask & (1u << SPX_PLUGIN_LVRB);
  if ((T || L) && !(e->tri_nodes && e->tri_pods)) return fail(e, SPX_ERR_STATE, "trimaran node/pod tables not uploaded");
  if (e->n_pods <= 0) {
    if (T || L) return fail(e, SPX_ERR_STATE, "no pod table uploaded");
    return fail(e, SPX_ERR_STATE, "n_pods unknown: upload a pod table first");
  }
  if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
  int rc;
  const bool N = plugin_mask & (1u << SPX_PLUGIN_NRT);
  if (N && !(e->nrt_slots && e->nrt_nodes && e->nrt_pods)) return fail(e, SPX_ERR_STATE, "NRT slot/node/pod tables not uploaded");
  if (A && (rc = prepare_alloc(e))) return rc;
  const bool W = plugin_mask & (1u << SPX_PLUGIN_NETOVERHEAD);
  if (W && !(e->net_nodes && e->net_topo && e->net_pods)) return fail(e, SPX_ERR_STATE, "NetworkOverhead node/topology/pod tables not uploaded");
  // the reference accumulates a node's cost in int64 (networkoverhead.go:605-633); the sweeps add in int32 and keep the Filter
  // verdict in the sign bit, which is exact as long as (largest cost entry) x (most pairs of any workload) stays below 2^31
  if (W && e->net_max_cost * std::max<int64_t>(e->net_max_pairs, 1) >= (int64_t{1} << 31))
    return fail(e, SPX_ERR_ARG, "NetworkOverhead: accumulated cost of a node may exceed 2^31 (cost entries x dependency pairs); this build sweeps in int32");
  for (int p = 0; p < 5; ++p)
    if ((plugin_mask & (1u << p)) && (rc = ensure_score_table(e, p))) return rc;
  if (R && (rc = ensure_score_table(e, SPX_PLUGIN_LROC))) return rc;
  if (R && e->score_stride[SPX_PLUGIN_LROC] != e->row_stride)
    return fail(e, SPX_ERR_STATE, "bound score table must use the engine row stride (spx_score_table reports it)");
  if (R && (rc = ensure(e, e->d_lroc_tab, static_cast<size_t>(e->row_stride) * spx::kLrocTabCols * sizeof(double)))) return rc;
  if (K && (rc = ensure_score_table(e, SPX_PLUGIN_PEAKS))) return rc;
  if (K && e->score_stride[SPX_PLUGIN_PEAKS] != e->row_stride)
    return fail(e, SPX_ERR_STATE, "bound score table must use the engine row stride (spx_score_table reports it)");
  if (K && ((rc = ensure(e, e->d_pk_min, static_cast<size_t>(e->n_pods) * 8)) || (rc = ensure(e, e->d_pk_max, static_cast<size_t>(e->n_pods) * 8)) ||
            (rc = ensure(e, e->d_pk_rowc, static_cast<size_t>(e->n_pods) * 16)) || (rc = ensure(e, e->d_pk_tab, static_cast<size_t>(e->row_stride) * 96))))
    return rc;
  if (N && (rc = ensure_status_table(e, SPX_PLUGIN_NRT))) return rc;
  if (W && (rc = ensure_status_table(e, SPX_PLUGIN_NETOVERHEAD))) return rc;

  spx::TrimaranArgs a{};
  fill_trimaran(e, a);
  a.row_begin = row_begin;
  a.row_end = row_end;
  // all three tables share row_stride when engine-owned; bound tables must use it too
  for (int p = 0; p < 3; ++p)
    if ((plugin_mask & (1u << p)) && e->score_stride[p] != e->row_stride)
      return fail(e, SPX_ERR_STATE, "bound score table must use the engine row stride (spx_score_table reports it)");
  // Allocatable's NormalizeScore runs over each pod's feasible nodes as soon as any Filter is in play
  const bool masked = N || W || e->ext_mask;
  bool alloc_by_net = false;  // Allocatable's masked table written by the NetworkOverhead sweep (SPX_OPT_NET_ALLOC_FUSED)
  a.out_alloc = (A && !masked) ? static_cast<uint8_t*>(e->score[SPX_PLUGIN_ALLOCATABLE].p) : nullptr;
  a.out_tlp = T ? static_cast<uint8_t*>(e->score[SPX_PLUGIN_TLP].p) : nullptr;
  a.out_lvrb = L ? static_cast<uint8_t*>(e->score[SPX_PLUGIN_LVRB].p) : nullptr;
  if (L) {
    if ((rc = ensure(e, e->d_lv_exact, static_cast<size_t>(e->n_nodes) * 8 * sizeof(double)))) return rc;
    a.lv_exact = static_cast<double*>(e->d_lv_exact.p);
    if (!e->d_lv_exact.p || !e->d_lv_fast.p || !e->d_lv_amb.p) e->lv_amb_built = false;
    if ((rc = ensure(e, e->d_lv_fast, static_cast<size_t>(spx::round_up(e->row_stride, 512)) * 8 * sizeof(float)))) return rc;
    a.lv_fast = static_cast<float*>(e->d_lv_fast.p);
    if ((rc = ensure(e, e->d_lv_amb, spx::lvrb_amb_bytes()))) return rc;
    a.lv_amb = static_cast<uint32_t*>(e->d_lv_amb.p);
    a.lv_amb_built = &e->lv_amb_built;
    a.lv_amb_geom = e->lv_amb_geom;
  }
  if (T) {
    if ((rc = ensure(e, e->d_tlp_fast, static_cast<size_t>(spx::round_up(e->row_stride, 1024)) * 4 * sizeof(float)))) return rc;
    a.tlp_fast = static_cast<float*>(e->d_tlp_fast.p);
    if (!e->d_tlp_amb.p) e->tlp_amb_built = false;
    if ((rc = ensure(e, e->d_tlp_amb, static_cast<size_t>(spx::kTlpAmbSize) * 4))) return rc;
    a.tlp_amb = static_cast<uint32_t*>(e->d_tlp_amb.p);
    a.tlp_amb_size = spx::kTlpAmbSize;
    a.tlp_amb_built = &e->tlp_amb_built;
    a.tlp_amb_geom = e->tlp_amb_geom;
  }
  if (!e->hold_ev0) SPX_HIP(e, hipEventRecord(e->ev0, e->stream));
  if (Q) {
    if ((rc = ensure(e, e->d_q_status, static_cast<size_t>(e->n_pods)))) return rc;
    spx::QuotaArgs qa{};
    qa.row_begin = row_begin;
    qa.row_end = row_end;
    qa.row_ptr = e->row_indirect;
    qa.n_namespaces = e->q_n_namespaces;
    qa.pod_ns = static_cast<const int32_t*>(e->d_q_pod_ns.p);
    qa.pod_priority = static_cast<const int32_t*>(e->d_q_pod_prio.p);
    qa.pod_req = static_cast<const int64_t*>(e->d_q_pod_req.p);
    qa.pod_req_present = static_cast<const uint8_t*>(e->d_q_pod_reqp.p);
    qa.has_quota = static_cast<const uint8_t*>(e->d_q_has.p);
    qa.used = static_cast<const int64_t*>(e->d_q_used.p);
    qa.max = static_cast<const int64_t*>(e->d_q_max.p);
    qa.max_present = static_cast<const uint8_t*>(e->d_q_maxp.p);
    std::memcpy(qa.agg_used, e->q_agg_used, sizeof qa.agg_used);
    std::memcpy(qa.agg_min, e->q_agg_min, sizeof qa.agg_min);
    qa.agg_used_present = e->q_agg_used_present;
    qa.agg_min_present = e->q_agg_min_present;
    qa.agg_used_dyn = e->q_agg_dyn;
    qa.other_nominated = static_cast<const int64_t*>(e->d_q_other.p);
    qa.other_nominated_present = static_cast<const uint8_t*>(e->d_q_otherp.p);
    qa.nom_ptr = static_cast<const int32_t*>(e->d_q_nom_ptr.p);
    qa.nom_priority = static_cast<const int32_t*>(e->d_q_nom_prio.p);
    qa.nom_pending_index = static_cast<const int64_t*>(e->d_q_nom_idx.p);
    qa.nom_req = static_cast<const int64_t*>(e->d_q_nom_req.p);
    qa.nom_req_present = static_cast<const uint8_t*>(e->d_q_nom_reqp.p);
    qa.out_status = static_cast<uint8_t*>(e->d_q_status.p);
    spx::launch_quota(qa, e->stream);
    SPX_HIP(e, hipGetLastError());
  }
  if (N) {
    if (e->score_stride[SPX_PLUGIN_NRT] != e->row_stride)
      return fail(e, SPX_ERR_STATE, "bound score table must use the engine row stride (spx_score_table reports it)");
    if (e->nrt_params.strategy == SPX_NRT_LEAST_NUMA_NODES && (rc = build_ln_tab(e))) return rc;
    if (e->nrt_params.strategy == SPX_NRT_BALANCED_ALLOCATION) {
      // room for 1/32 of the cells (config #3 marks 0.7 %); what does not fit is recomputed where it is found.  Inside the
      // sequential commit loop (row_indirect: one row per launch, graph capture) the list allocated for the first pod is kept.
      const uint64_t want = std::max<uint64_t>(4096, static_cast<uint64_t>(row_end - row_begin) * static_cast<uint64_t>(e->row_stride) / 32);
      const uint32_t cap = static_cast<uint32_t>(std::min<uint64_t>(want, 1u << 28));
      if (cap > e->nrt_redo_cap) {
        if ((rc = ensure(e, e->d_nrt_redo, (2 + 2 * static_cast<size_t>(cap)) * sizeof(uint32_t)))) return rc;
        e->nrt_redo_cap = cap;
      }
    }
    if ((rc = ensure_nrt_creq(e))) return rc;
    spx::NrtArgs na{};
    fill_nrt(e, na);
    na.row_begin = row_begin;
    na.row_end = row_end;
    na.out_status = static_cast<uint8_t*>(e->status[SPX_PLUGIN_NRT].p);
    na.out_score = static_cast<uint8_t*>(e->score[SPX_PLUGIN_NRT].p);
    // one representative per pod equivalence class when the whole batch is evaluated with the float64 formulation and enough
    // rows are copies (a partial range may cut a class off from its representative)
    const bool classes = e->option[SPX_OPT_NRT_POD_CLASSES] && na.fast && !(na.opts & spx::kOptNrtGeneric) && !e->row_indirect &&
                         row_begin == 0 && row_end == e->n_pods && e->nrt_n_dups > 0 && e->nrt_n_dups * 32 >= e->n_pods &&
                         !(na.strategy == SPX_NRT_LEAST_NUMA_NODES && !na.ln_tab);
    {  // LeastAllocated's Score launch in packed float32 (only the split launch of a row range acts on it: launch_nrt_fast)
      NrtPacked pk;
      if (na.fast && !(na.opts & spx::kOptNrtGeneric) && !e->row_indirect && nrt_packed_score(e, &pk)) {
        if (pk.tab_slot >= 0) {
          const size_t bytes = (static_cast<size_t>(pk.tab_kmax) + 1) * pk.tab_words * 4;
          if (!e->d_nrt_pk_tab.p || e->d_nrt_pk_tab.bytes < bytes) e->nrt_pk_tab_built = false;
          if ((rc = ensure(e, e->d_nrt_pk_tab, bytes))) return rc;
        }
        na.pk_mode = 1, na.pk_tab_slot = pk.tab_slot;
        na.pk_tab = static_cast<uint32_t*>(e->d_nrt_pk_tab.p);
        na.pk_tab_words = pk.tab_words, na.pk_tab_kmax = pk.tab_kmax, na.pk_tab_inv_unit = pk.tab_inv_unit;
        na.pk_tab_built = &e->nrt_pk_tab_built;
      }
    }
    // the fused Filter + Score launch (kernels_nrt_fused.hip): a whole-batch LeastAllocated sweep in the packed Score's preconditions with unit
    // weights; it walks the rank stream of the class representatives or, without classes, of every row
    bool fused = e->option[SPX_OPT_NRT_FUSED] && e->option[SPX_OPT_NRT_RANK_FILTER] && na.pk_mode && !(na.opts & spx::kOptNrtSingleLaunch) &&
                 row_begin == 0 && row_end == e->n_pods;
    fused = fused && e->option[SPX_OPT_NRT_RANK_NARROW];  // (its only count layout)
    for (int i = 0; fused && i < e->nrt_n_res; ++i) fused = e->nrt_slot_weight[i] == 0 || e->nrt_slot_weight[i] == 1;
    if (classes || fused) {
      if ((rc = nrt_rank_stream(e, classes ? 1 : 2))) return rc;
    }
    const bool stream = e->nrt_rk_max_dwords && e->nrt_rk_kind == (classes ? 1 : 2) && e->option[SPX_OPT_NRT_RANK_FILTER];
    if (classes) {
      na.row_list = static_cast<const int32_t*>(e->d_nrt_uniq.p);
      na.n_list = e->nrt_n_uniq;
    }
    fused = fused && stream && e->nrt_rk_all_narrow;
    if (stream && (classes || fused)) {  // the Filter in rank space (its own launch, or inside the fused one)
      na.rk_stream = static_cast<const uint32_t*>(e->d_nrt_rk.p);
      na.rk_off = static_cast<const uint32_t*>(e->d_nrt_rk_off.p);
      na.rk_max_dwords = e->nrt_rk_max_dwords;
      na.rk_first = static_cast<const uint32_t*>(e->d_nrt_rk_first.p);
      na.rk_chunks = e->nrt_rk_chunks;
      na.rk_all_narrow = e->nrt_rk_all_narrow && e->option[SPX_OPT_NRT_FUSED];
      if (!classes) na.n_list = e->n_pods;
      if (fused) {
        if ((rc = ensure(e, e->d_nrt_fz, spx::nrt_fused_item_words(e->nrt_n_res, na.n_list) * sizeof(uint32_t)))) return rc;
        na.fz_items = static_cast<uint32_t*>(e->d_nrt_fz.p);
        const spx_engine::FzKey key{e->nrt_items_gen, classes ? 1 : 2, na.pk_tab_slot, e->d_nrt_fz.p};
        na.fz_pack = !(key == e->nrt_fz_key);
        e->nrt_fz_key = key;
      }
    }
    if (na.strategy == SPX_NRT_LEAST_NUMA_NODES && na.fast && !(na.opts & spx::kOptNrtGeneric) && !e->row_indirect) {
      // LeastNUMANodes, batch launch: per evaluated row and node scope a list of the nodes whose cell needs the complete subset
      // search (k_nrt_ln_redo) — room for 3/8 of the nodes per list by default (config #3 lists 13 % of the cells, no row more than 40 %); a
      // list that overflows sends the launch back to the complete sweep
      const int64_t rows = classes ? e->nrt_n_uniq : row_end - row_begin;
      // The lists are scratch: held to 1 GiB (config #3: 0.42 GB) by shortening them — a shorter list overflows sooner, and an
      // allocation that fails leaves the launch without lists; either way the complete sweep writes the same table, slower
      uint32_t per_row = static_cast<uint32_t>(spx::round_up(std::max<int64_t>(64, e->n_nodes * e->option[SPX_OPT_NRT_LN_LIST_PERMILLE] / 1000), 64));
      constexpr size_t kListWords = (size_t{1} << 30) / sizeof(uint32_t);
      if (rows > 0) {
        const size_t room = (kListWords - 2) / (2 * static_cast<size_t>(rows));  // 1 + per_row words per list
        if (room < 1 + static_cast<size_t>(per_row)) per_row = room > 64 ? static_cast<uint32_t>((room - 1) / 64 * 64) : 0;
      }
      const size_t words = 2 + 2 * static_cast<size_t>(rows) * (1 + static_cast<size_t>(per_row));
      if (rows > 0 && per_row >= 64) {
        std::string kept;
        {
          std::lock_guard<std::mutex> g(e->err_mu);
          kept = e->err;
        }
        if (ensure(e, e->d_nrt_redo, words * sizeof(uint32_t)) == SPX_OK &&
            ensure(e, e->d_nrt_lnrec, static_cast<size_t>(e->n_nodes) * (SPX_NRT_MAX_ZONES * (e->nrt_n_res <= 4 ? 4 : 8) * 2 + 16) * sizeof(uint32_t)) == SPX_OK) {
          na.redo_list = static_cast<uint32_t*>(e->d_nrt_redo.p);
          na.ln_rec = static_cast<uint32_t*>(e->d_nrt_lnrec.p);
          na.ln_rows = rows;
          na.ln_per_row = per_row;
        } else {
          (void)hipGetLastError();  // the failed allocation's sticky error
          std::lock_guard<std::mutex> g(e->err_mu);
          e->err = kept;
        }
      }
    }
    const bool ran_fused = spx::launch_nrt(na, e->stream);
    e->last_nrt_filter = ran_fused ? 3 : (na.rk_stream ? 2 : 1);
    if (classes)
      spx::launch_rows_expand(static_cast<const int32_t*>(e->d_nrt_dups.p), e->nrt_n_dups, na.out_status, na.out_score, e->row_stride, e->stream);
    SPX_HIP(e, hipGetLastError());
  }
  if (W) {
    if (e->score_stride[SPX_PLUGIN_NETOVERHEAD] != e->row_stride)
      return fail(e, SPX_ERR_STATE, "bound score table must use the engine row stride (spx_score_table reports it)");
    spx::NetArgs g{};
    fill_net(e, g);
    g.row_begin = row_begin;
    g.row_end = row_end;
    // upstream scores only nodes that passed every Filter plugin
    g.other_status[0] = N ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NRT].p) : nullptr;
    g.other_status[1] = e->ext_mask ? static_cast<const uint8_t*>(e->d_ext_status.p) : nullptr;
    g.out_status = static_cast<uint8_t*>(e->status[SPX_PLUGIN_NETOVERHEAD].p);
    g.out_score = static_cast<uint8_t*>(e->score[SPX_PLUGIN_NETOVERHEAD].p);
    // Allocatable's masked NormalizeScore rides on the network kernel's walks when both are evaluated over a row range (same feasible
    // set; k_alloc_masked would read the status tables again): launch_net says whether it did
    if (A && masked && !e->skip_alloc_masked && e->alloc_compact && !e->row_indirect && row_end - row_begin > 1 && e->option[SPX_OPT_NET_ALLOC_FUSED]) {
      g.alloc_rel = static_cast<const uint32_t*>(e->d_alloc_rel.p);
      g.out_alloc = static_cast<uint8_t*>(e->score[SPX_PLUGIN_ALLOCATABLE].p);
    }
    alloc_by_net = spx::launch_net(g, e->stream);
    SPX_HIP(e, hipGetLastError());
  }
  spx::launch_trimaran(a, e->stream);
  SPX_HIP(e, hipGetLastError());
  if (R) {
    spx::LrocArgs la{};
    fill_lroc(e, la);
    la.row_begin = row_begin;
    la.row_end = row_end;
    la.out_score = static_cast<uint8_t*>(e->score[SPX_PLUGIN_LROC].p);
    if (!e->lroc_tab_ready) {  // per-node riskLoad: once per (node tables, params)
      spx::launch_lroc_prepare(la, e->stream);
      SPX_HIP(e, hipGetLastError());
      e->lroc_tab_ready = true;
    }
    spx::launch_lroc(la, e->stream);
    SPX_HIP(e, hipGetLastError());
  }
  if (K) {  // after the Filter plugins: NormalizeScore runs over each pod's feasible nodes
    spx::PeaksArgs ka{};
    fill_peaks(e, ka);
    ka.row_begin = row_begin;
    ka.row_end = row_end;
    ka.other_status[0] = N ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NRT].p) : nullptr;
    ka.other_status[1] = W ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NETOVERHEAD].p) : nullptr;
    ka.other_status[2] = e->ext_mask ? static_cast<const uint8_t*>(e->d_ext_status.p) : nullptr;
    ka.out_score = static_cast<uint8_t*>(e->score[SPX_PLUGIN_PEAKS].p);
    // one row per distinct cpu request when the whole batch is swept and nothing narrows a pod's node list (NormalizeScore runs
    // over the same nodes for every pod then), provided enough rows are copies
    const bool classes = e->option[SPX_OPT_PEAKS_POD_CLASSES] && !ka.other_status[0] && !ka.other_status[1] && !ka.other_status[2] &&
                         row_begin == 0 && row_end == e->n_pods && e->pk_n_dups > 0 && e->pk_n_dups * 8 >= e->n_pods;
    if (classes) {
      ka.row_list = static_cast<const int32_t*>(e->d_pk_uniq.p);
      ka.n_list = e->pk_n_uniq;
    }
    if (e->pk_negative) ka.opts &= ~spx::kOptPeaksEstimate;  // (the float64 passes take whatever the table holds)
    if (ka.opts & spx::kOptPeaksEstimate) {  // the undecided cells' list: sized by the rows this sweep walks
      size_t seg_bytes = 0, cnt_bytes = 0;
      ka.est_pods = spx::peaks_est_plan(ka.opts, e->row_stride, classes ? ka.n_list : row_end - row_begin, &seg_bytes, &cnt_bytes);
      if ((rc = ensure(e, e->d_pk_seg, seg_bytes)) || (rc = ensure(e, e->d_pk_segn, cnt_bytes))) return rc;
      ka.seg = e->d_pk_seg.p;
      ka.seg_n = static_cast<int32_t*>(e->d_pk_segn.p);
    }
    spx::launch_peaks(ka, e->stream);
    if (classes)
      spx::launch_rows_expand(static_cast<const int32_t*>(e->d_pk_dups.p), e->pk_n_dups, ka.out_score, nullptr, e->row_stride, e->stream);
    SPX_HIP(e, hipGetLastError());
  }
  if (A && masked && !e->skip_alloc_masked && !alloc_by_net) {
    spx::ProfileArgs pa{};
    pa.n_nodes = e->n_nodes;
    pa.row_stride = e->row_stride;
    pa.row_begin = row_begin;
    pa.row_end = row_end;
    pa.row_ptr = e->row_indirect;
    pa.status[0] = N ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NRT].p) : nullptr;
    pa.status[1] = W ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NETOVERHEAD].p) : nullptr;
    pa.status[2] = e->ext_mask ? static_cast<const uint8_t*>(e->d_ext_status.p) : nullptr;
    pa.alloc_raw = static_cast<const int64_t*>(e->d_alloc_raw.p);
    pa.alloc_rel = static_cast<const uint32_t*>(e->d_alloc_rel.p);
    pa.out_alloc = static_cast<uint8_t*>(e->score[SPX_PLUGIN_ALLOCATABLE].p);
    pa.block_per_row = static_cast<int32_t>(e->option[SPX_OPT_ROW_WORKGROUP]);
    spx::launch_alloc_masked(pa, e->stream);
    SPX_HIP(e, hipGetLastError());
  }
  SPX_HIP(e, hipEventRecord(e->ev1, e->stream));
  e->timed = true;
  e->best_valid = false;
  e->evaluated |= plugin_mask;
  const bool alloc_skipped = A && masked && e->skip_alloc_masked;  // spx_decide: the table is not written — nothing to fetch
  if (alloc_skipped) {
    e->evaluated &= ~(1u << SPX_PLUGIN_ALLOCATABLE);
    e->eval_info[SPX_PLUGIN_ALLOCATABLE] = spx_engine::EvalInfo{};
  }
  if (row_end > row_begin) {
    const uint32_t filters = plugin_mask & kFilterPlugins;
    for (int p = 0; p < SPX_NUM_PLUGINS; ++p) {
      if (!((plugin_mask >> p) & 1u) || (alloc_skipped && p == SPX_PLUGIN_ALLOCATABLE)) continue;
      spx_engine::EvalInfo& i = e->eval_info[p];
      const bool same_ctx = i.filters == filters && i.ext_gen == e->ext_gen && i.end > i.begin;
      if (same_ctx && row_begin <= i.end && row_end >= i.begin) {  // overlapping or adjacent: the evaluated rows grow
        i.begin = std::min(i.begin, row_begin);
        i.end = std::max(i.end, row_end);
      } else {
        i.begin = row_begin, i.end = row_end, i.filters = filters, i.ext_gen = e->ext_gen;
      }
    }
  }
  return SPX_OK;
}

int spx_sync(spx_engine* e) {
  if (!e) return SPX_ERR_ARG;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_upload_net_commit(spx_engine* e, const spx_net_commit_soa* t) {
  if (!e || !t) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if (!e->net_pods) return fail(e, SPX_ERR_STATE, "upload the NetworkOverhead pod table first");
  if (t->n_pods != e->n_pods || !t->eff_ptr) return fail(e, SPX_ERR_ARG, "net commit table: pod count differs from the uploaded pod tables");
  const size_t P = static_cast<size_t>(t->n_pods), n = static_cast<size_t>(t->eff_ptr[P]);
  if (n && (!t->eff_key || !t->eff_max_cost)) return fail(e, SPX_ERR_ARG, "NULL column in table");
  for (size_t i = 0; i < n; ++i)
    if (t->eff_key[i] < 0 || t->eff_key[i] >= e->net_n_keys) return fail(e, SPX_ERR_ARG, "net commit table: key out of range");
  e->h_eff_ptr.assign(t->eff_ptr, t->eff_ptr + P + 1);
  e->h_eff_key.assign(t->eff_key, t->eff_key + n);
  e->h_eff_cost.assign(t->eff_max_cost, t->eff_max_cost + n);
  int rc;
  const int64_t zero = 0;
  if ((rc = upload(e, e->d_net_eff_ptr, t->eff_ptr, (P + 1) * 4))) return rc;
  if ((rc = upload(e, e->d_net_eff_key, n ? static_cast<const void*>(t->eff_key) : static_cast<const void*>(&zero), n * 4))) return rc;
  if ((rc = upload(e, e->d_net_eff_cost, n ? static_cast<const void*>(t->eff_max_cost) : static_cast<const void*>(&zero), n * 8))) return rc;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  e->net_commit = true;
  return SPX_OK;
}

namespace {

// The sequential commit of a profile with Filter plugins as one cooperative persistent launch (kernels_commit_coop.hip).  *ran stays
// false when the profile does not fit the kernel (strategy, sizes, weights, forced reference kernels, SPX_OPT_COMMIT_COOP 0): the
// caller then runs the per-pod loop.  `dyn_ptr`: the workload pair lists' starts in the layout with slack (built by the caller).
int commit_coop(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end, const std::vector<int32_t>& dyn_ptr, int32_t* node_idx,
                int64_t* weighted_score, int32_t* n_ties, int64_t* tlp_missing_out, bool* ran) {
  *ran = false;
  const bool A = plugin_mask & (1u << SPX_PLUGIN_ALLOCATABLE), T = plugin_mask & (1u << SPX_PLUGIN_TLP), Lv = plugin_mask & (1u << SPX_PLUGIN_LVRB);
  const bool N = plugin_mask & (1u << SPX_PLUGIN_NRT), W = plugin_mask & (1u << SPX_PLUGIN_NETOVERHEAD), Q = plugin_mask & (1u << SPX_PLUGIN_CAPACITY);
  if (!e->option[SPX_OPT_COMMIT_COOP] || e->option[SPX_OPT_COMMIT_FROM_MEMORY]) return SPX_OK;
  for (int p : {SPX_PLUGIN_TLP, SPX_PLUGIN_LVRB, SPX_PLUGIN_NRT, SPX_PLUGIN_NETOVERHEAD})
    if (((plugin_mask >> p) & 1u) && forced_reference(e, p)) return SPX_OK;
  const int64_t n_wg = (e->n_nodes + spx::kCoopWindow - 1) / spx::kCoopWindow;
  if (n_wg > spx::kCoopMaxWg) return SPX_OK;
  int64_t bound = 0;
  for (int k = 0; k <= SPX_PLUGIN_NETOVERHEAD; ++k)
    if ((plugin_mask >> k) & 1u) {
      if (e->plugin_weight[k] < 0 || e->plugin_weight[k] >= (int64_t{1} << 23)) return SPX_OK;
      bound += e->plugin_weight[k] * 255;
    }
  if (bound >= (int64_t{1} << 31)) return SPX_OK;
  int rc;
  if (A) {
    if ((rc = prepare_alloc(e))) return rc;
    if (!e->alloc_compact) return SPX_OK;
  }
  if (N) {
    const bool fast = e->nrt_fast_slots && e->nrt_fast_nodes && e->nrt_fast_pods;
    if (!fast || e->nrt_n_res > 4 || (e->nrt_params.strategy != SPX_NRT_LEAST_ALLOCATED && e->nrt_params.strategy != SPX_NRT_MOST_ALLOCATED)) return SPX_OK;
  }
  if (W) {
    if (e->net_n_classes <= 0 || e->net_n_classes > spx::kCoopMaxClasses || e->net_n_keys <= 0) return SPX_OK;
    for (size_t k = 0; k + 1 < dyn_ptr.size(); ++k)
      if (dyn_ptr[k + 1] - dyn_ptr[k] > spx::kCoopMaxPairs) return SPX_OK;
    for (int64_t i = row_begin; i < row_end; ++i)
      if (e->h_eff_ptr[static_cast<size_t>(i) + 1] - e->h_eff_ptr[static_cast<size_t>(i)] > spx::kCoopMaxEffects) return SPX_OK;
  }
  const size_t P = static_cast<size_t>(e->n_pods), Nn = static_cast<size_t>(e->n_nodes);
  spx::CoopArgs c{};
  c.use = plugin_mask;
  for (int k = 0; k < SPX_NUM_PLUGINS; ++k) c.w[k] = static_cast<int32_t>(e->plugin_weight[k]);
  c.n_nodes = e->n_nodes, c.n_pods = e->n_pods, c.row_stride = e->row_stride, c.row_begin = row_begin, c.row_end = row_end;
  c.n_wg = static_cast<int32_t>(n_wg);
  c.nrt_sg = e->nrt_params.strategy == SPX_NRT_MOST_ALLOCATED ? 1 : 0;
  c.alloc_rel = static_cast<const uint32_t*>(e->d_alloc_rel.p);
  fill_trimaran(e, c.t);
  if (N) fill_nrt(e, c.nrt);
  if (W) {
    fill_net(e, c.net);
    c.net.pair_ptr = static_cast<const int32_t*>(e->d_net_dyn_ptr.p);
    c.net_init_end = static_cast<const int32_t*>(e->d_net_dyn_end.p);
    c.net_init_flag = static_cast<const uint8_t*>(e->d_net_key_flag.p);
    c.net_init_node = static_cast<const int32_t*>(e->d_net_dyn_node.p);
    c.net_init_max = static_cast<const int64_t*>(e->d_net_dyn_max.p);
    c.net_cap = dyn_ptr.empty() ? 0 : dyn_ptr.back();
    c.net_n_keys = e->net_n_keys;
    c.eff_ptr = static_cast<const int32_t*>(e->d_net_eff_ptr.p);
    c.eff_key = static_cast<const int32_t*>(e->d_net_eff_key.p);
    c.eff_cost = static_cast<const int64_t*>(e->d_net_eff_cost.p);
  }
  if (Q) {
    c.q_ns = e->q_n_namespaces;
    c.q_n_nom = static_cast<int32_t>(e->q_n_nominated);
    c.q_pod_ns = static_cast<const int32_t*>(e->d_q_pod_ns.p);
    c.q_pod_prio = static_cast<const int32_t*>(e->d_q_pod_prio.p);
    c.q_pod_req = static_cast<const int64_t*>(e->d_q_pod_req.p);
    c.q_pod_reqp = static_cast<const uint8_t*>(e->d_q_pod_reqp.p);
    c.q_has = static_cast<const uint8_t*>(e->d_q_has.p);
    c.q_used = static_cast<const int64_t*>(e->d_q_used.p);
    c.q_usedp = static_cast<const uint8_t*>(e->d_q_usedp.p);
    c.q_max = static_cast<const int64_t*>(e->d_q_max.p);
    c.q_maxp = static_cast<const uint8_t*>(e->d_q_maxp.p);
    c.q_min = static_cast<const int64_t*>(e->d_q_min.p);
    c.q_minp = static_cast<const uint8_t*>(e->d_q_minp.p);
    c.q_agg = static_cast<const int64_t*>(e->d_q_agg.p);
    std::memcpy(c.q_agg_min, e->q_agg_min, sizeof c.q_agg_min);
    c.q_agg_min_present = e->q_agg_min_present;
    c.q_other = static_cast<const int64_t*>(e->d_q_other.p);
    c.q_otherp = static_cast<const uint8_t*>(e->d_q_otherp.p);
    c.q_nom_ptr = static_cast<const int32_t*>(e->d_q_nom_ptr.p);
    c.q_nom_prio = static_cast<const int32_t*>(e->d_q_nom_prio.p);
    c.q_nom_pending = static_cast<const int64_t*>(e->d_q_nom_idx.p);
    c.q_nom_req = static_cast<const int64_t*>(e->d_q_nom_req.p);
    c.q_nom_reqp = static_cast<const uint8_t*>(e->d_q_nom_reqp.p);
  }
  int lds_max = 0;
  SPX_HIP(e, hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, e->device));
  const size_t lds = spx::commit_coop_lds_bytes(c);
  if (lds + 4096 > static_cast<size_t>(lds_max)) return SPX_OK;  // (4 KB: the kernel's static LDS)
  {
    // every workgroup polls every other one's granules: all n_wg must be resident at once.  The occupancy the runtime reports for this
    // kernel at this LDS size x the CU count is the ceiling (a smaller part, a CU mask); above it the per-pod loop runs instead.
    const int resident = spx::commit_coop_max_resident(c, e->device);
    if (resident > 0 && n_wg > resident) return SPX_OK;
  }
  // ---- from here on the kernel runs
  if (Lv) {  // LVRB carries no commit state: its rows are swept once
    if ((rc = spx_eval(e, 1u << SPX_PLUGIN_LVRB, row_begin, row_end))) return rc;
    if (e->score_stride[SPX_PLUGIN_LVRB] != e->row_stride) return fail(e, SPX_ERR_STATE, "bound LVRB table must use the engine row stride");
    c.lv_table = static_cast<const uint8_t*>(e->score[SPX_PLUGIN_LVRB].p);
  }
  const size_t sync_bytes = 2 * static_cast<size_t>(spx::kCoopKinds) * spx::kCoopMaxWg * 8;
  if ((rc = ensure(e, e->d_coop_sync, sync_bytes + 64))) return rc;
  SPX_HIP(e, hipMemsetAsync(e->d_coop_sync.p, 0, sync_bytes + 64, e->stream));
  c.sync = static_cast<unsigned long long*>(e->d_coop_sync.p);
  c.err = reinterpret_cast<int32_t*>(static_cast<char*>(e->d_coop_sync.p) + sync_bytes);
  if (W) {
    const size_t cap = static_cast<size_t>(c.net_cap ? c.net_cap : 1);
    if ((rc = ensure(e, e->d_coop_node, static_cast<size_t>(n_wg) * cap * 4)) || (rc = ensure(e, e->d_coop_max, static_cast<size_t>(n_wg) * cap * 8))) return rc;
    c.net_priv_node = static_cast<int32_t*>(e->d_coop_node.p);
    c.net_priv_max = static_cast<int64_t*>(e->d_coop_max.p);
  }
  if ((rc = ensure(e, e->d_best, P * 20))) return rc;
  c.best_score = static_cast<int64_t*>(e->d_best.p);
  c.best_node = reinterpret_cast<int32_t*>(c.best_score + P);
  c.best_ties = c.best_node + P;
  c.best_feasible = c.best_ties + P;
  if (tlp_missing_out && T) {
    if ((rc = ensure(e, e->d_commit, Nn * 8))) return rc;
    c.missing_out = static_cast<int64_t*>(e->d_commit.p);
  }
  spx::launch_commit_coop(c, e->stream);
  SPX_HIP(e, hipGetLastError());
  const size_t rows = static_cast<size_t>(row_end - row_begin);
  int32_t err = 0;
  SPX_HIP(e, hipMemcpyAsync(&err, c.err, 4, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipMemcpyAsync(weighted_score, c.best_score + row_begin, rows * 8, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipMemcpyAsync(node_idx, c.best_node + row_begin, rows * 4, hipMemcpyDeviceToHost, e->stream));
  if (n_ties) SPX_HIP(e, hipMemcpyAsync(n_ties, c.best_ties + row_begin, rows * 4, hipMemcpyDeviceToHost, e->stream));
  if (tlp_missing_out && T) SPX_HIP(e, hipMemcpyAsync(tlp_missing_out, c.missing_out, Nn * 8, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  if (err != 0) {
    // a workgroup gave up waiting for another one (the device is shared with other work and not all workgroups became resident): the
    // kernel mutated nothing in the engine's tables, so the per-pod loop can still serve the call
    e->coop_gave_up += 1;
    return SPX_OK;  // *ran is false
  }
  if (tlp_missing_out && !T) std::memset(tlp_missing_out, 0, Nn * 8);
  e->best_valid = false;
  e->last_commit_path = 3;
  *ran = true;
  return SPX_OK;
}

// Sequential commit with Filter plugins in the profile: per pod one single-row evaluation of the whole plugin set on the
// CURRENT device tables, the weighted argmax, and k_commit_apply.  Everything is enqueued on the engine stream without a host
// sync; the tables the loop mutates are saved before and restored after.
int commit_with_filters(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end, int32_t* node_idx, int64_t* weighted_score,
                        int32_t* n_ties, int64_t* tlp_missing_out) {
  const bool T = plugin_mask & (1u << SPX_PLUGIN_TLP), N = plugin_mask & (1u << SPX_PLUGIN_NRT);
  const bool W = plugin_mask & (1u << SPX_PLUGIN_NETOVERHEAD), Q = plugin_mask & (1u << SPX_PLUGIN_CAPACITY);
  if ((T || (plugin_mask & (1u << SPX_PLUGIN_LVRB))) && !(e->tri_nodes && e->tri_pods)) return fail(e, SPX_ERR_STATE, "trimaran node/pod tables not uploaded");
  if (N && !(e->nrt_slots && e->nrt_nodes && e->nrt_pods)) return fail(e, SPX_ERR_STATE, "NRT slot/node/pod tables not uploaded");
  if (W && !(e->net_nodes && e->net_topo && e->net_pods && e->net_commit))
    return fail(e, SPX_ERR_STATE, "NetworkOverhead in a sequential commit needs spx_upload_net_commit (after the NetworkOverhead pod table)");
  if (Q && !(e->quota && e->q_has_min)) return fail(e, SPX_ERR_STATE, "CapacityScheduling in a sequential commit needs spx_quota_soa.min / min_present");
  if (e->ext_mask) return fail(e, SPX_ERR_STATE, "a caller feasibility mask is a frozen-snapshot input: clear it for the sequential commit");
  const size_t Nn = static_cast<size_t>(e->n_nodes), P = static_cast<size_t>(e->n_pods), R = static_cast<size_t>(e->nrt_n_res);
  const size_t NS = static_cast<size_t>(e->q_n_namespaces), S = SPX_QUOTA_SLOTS, K = static_cast<size_t>(e->net_n_keys);
  int rc;
  // ---- NetworkOverhead: pair lists with the slack the effects of this batch can fill
  std::vector<int32_t> dyn_ptr;
  if (W) {
    std::vector<int32_t> extra(K, 0);
    for (size_t i = 0; i < e->h_eff_key.size(); ++i)
      if (e->h_eff_cost[i] >= 0) ++extra[static_cast<size_t>(e->h_eff_key[i])];
    dyn_ptr.assign(K + 1, 0);
    for (size_t k = 0; k < K; ++k) dyn_ptr[k + 1] = dyn_ptr[k] + (e->h_pair_ptr[k + 1] - e->h_pair_ptr[k]) + extra[k];
    std::vector<int32_t> dyn_end(K);
    for (size_t k = 0; k < K; ++k) dyn_end[k] = dyn_ptr[k] + (e->h_pair_ptr[k + 1] - e->h_pair_ptr[k]);
    const size_t cap = static_cast<size_t>(dyn_ptr[K]);
    if (static_cast<int64_t>(e->net_max_cost) * std::max<int64_t>(1, *std::max_element(extra.begin(), extra.end()) + e->net_max_pairs) >= (int64_t{1} << 31))
      return fail(e, SPX_ERR_ARG, "NetworkOverhead: accumulated cost of a node may exceed 2^31 once the batch is bound; this build sweeps in int32");
    if ((rc = upload(e, e->d_net_dyn_ptr, dyn_ptr.data(), (K + 1) * 4))) return rc;
    if ((rc = upload(e, e->d_net_dyn_end, dyn_end.data(), K * 4))) return rc;
    if ((rc = ensure(e, e->d_net_dyn_node, cap * 4)) || (rc = ensure(e, e->d_net_dyn_max, cap * 8))) return rc;
    SPX_HIP(e, hipStreamSynchronize(e->stream));  // the vectors above are locals
    // the initial pairs into the layout with slack: one launch (round 3 issued two copies per key: 14k tiny copies for config #5's share)
    spx::launch_spread_pairs(static_cast<int32_t>(K), static_cast<const int32_t*>(e->d_net_pair_ptr.p), static_cast<const int32_t*>(e->d_net_dyn_ptr.p),
                             static_cast<const int32_t*>(e->d_net_pair_node.p), static_cast<const int64_t*>(e->d_net_pair_max.p),
                             static_cast<int32_t*>(e->d_net_dyn_node.p), static_cast<int64_t*>(e->d_net_dyn_max.p), e->stream);
    SPX_HIP(e, hipGetLastError());
  }
  // ---- the cooperative persistent kernel (kernels_commit_coop.hip) when the profile fits it: nothing is mutated in the engine's
  // tables (the state lives in the kernel's registers / LDS), so nothing is saved or restored
  {
    bool ran = false;
    if ((rc = commit_coop(e, plugin_mask, row_begin, row_end, dyn_ptr, node_idx, weighted_score, n_ties, tlp_missing_out, &ran))) return rc;
    if (ran) return SPX_OK;
  }
  e->last_commit_path = 2;
  struct LoopFlag {
    spx_engine* e;
    explicit LoopFlag(spx_engine* x) : e(x) { e->in_commit_loop = true, e->tlp_amb_built = false, e->nrt_pk_tab_built = false; }  // (k_commit_apply advances d_tlp_missing and the zone tables)
    ~LoopFlag() { e->in_commit_loop = false, e->tlp_amb_built = false, e->nrt_pk_tab_built = false; }
  } loop_flag(e);
  // ---- save what the loop mutates
  struct Saved {
    DevBuf* buf;
    size_t bytes, off;
  };
  std::vector<Saved> saved;
  size_t total = 0;
  auto keep = [&](DevBuf& b, size_t bytes) {
    if (!bytes) return;
    saved.push_back({&b, bytes, total});
    total += (bytes + 255) / 256 * 256;
  };
  if (T) keep(e->d_tlp_missing, Nn * 8);
  if (N) {
    const size_t cells = SPX_NRT_MAX_ZONES * R * Nn * 8, zn = SPX_NRT_MAX_ZONES * Nn * 8;
    keep(e->d_nrt_avail, cells), keep(e->d_nrt_fav, cells), keep(e->d_nrt_frc, cells), keep(e->d_nrt_frcv, cells), keep(e->d_nrt_fcpu, zn), keep(e->d_nrt_fbraw, zn);
  }
  if (Q) {
    keep(e->d_q_used, NS * S * 8), keep(e->d_q_usedp, NS), keep(e->d_q_agg, (S + 1) * 8), keep(e->d_q_nom_req, e->q_n_nominated * S * 8),
        keep(e->d_q_nom_reqp, e->q_n_nominated), keep(e->d_q_other, NS * S * 8), keep(e->d_q_otherp, NS);
  }
  if (W) keep(e->d_net_key_flag, K);
  if ((rc = ensure(e, e->d_commit_save, total))) return rc;
  for (const Saved& sv : saved)
    SPX_HIP(e, hipMemcpyAsync(static_cast<char*>(e->d_commit_save.p) + sv.off, sv.buf->p, sv.bytes, hipMemcpyDeviceToDevice, e->stream));
  // ---- the loop
  if ((rc = ensure(e, e->d_best, P * 20))) return rc;
  spx::CommitApplyArgs ca{};
  ca.n_nodes = e->n_nodes;
  ca.n_pods = e->n_pods;
  ca.best_node = reinterpret_cast<const int32_t*>(static_cast<const int64_t*>(e->d_best.p) + P);
  if (T) {
    ca.tlp_missing = static_cast<int64_t*>(e->d_tlp_missing.p);
    ca.tlp_pod_milli = static_cast<const int64_t*>(e->d_tlp_pod.p);
  }
  if (N) {
    ca.nrt_n_res = e->nrt_n_res;
    ca.nrt_cpu_slot = e->nrt_cpu_slot;
    ca.nrt_flags = static_cast<const uint8_t*>(e->d_nrt_flags.p);
    ca.nrt_zone_present = static_cast<const uint8_t*>(e->d_nrt_zp.p);
    ca.nrt_avail = static_cast<int64_t*>(e->d_nrt_avail.p);
    ca.f_av = static_cast<double*>(e->d_nrt_fav.p);
    ca.f_rc = static_cast<double*>(e->d_nrt_frc.p);
    ca.f_rcv = static_cast<double*>(e->d_nrt_frcv.p);
    ca.f_cpu = static_cast<double*>(e->d_nrt_fcpu.p);
    ca.f_braw = static_cast<double*>(e->d_nrt_fbraw.p);
    ca.nrt_pod_present = static_cast<const uint8_t*>(e->d_nrt_ppres.p);
    ca.nrt_pod_req = static_cast<const int64_t*>(e->d_nrt_preq.p);
  }
  if (Q) {
    ca.q_n_namespaces = e->q_n_namespaces;
    ca.q_pod_ns = static_cast<const int32_t*>(e->d_q_pod_ns.p);
    ca.q_pod_req = static_cast<const int64_t*>(e->d_q_pod_req.p);
    ca.q_pod_reqp = static_cast<const uint8_t*>(e->d_q_pod_reqp.p);
    ca.q_has = static_cast<const uint8_t*>(e->d_q_has.p);
    ca.q_used = static_cast<int64_t*>(e->d_q_used.p);
    ca.q_used_present = static_cast<uint8_t*>(e->d_q_usedp.p);
    ca.q_min = static_cast<const int64_t*>(e->d_q_min.p);
    ca.q_min_present = static_cast<const uint8_t*>(e->d_q_minp.p);
    ca.q_agg_used = static_cast<int64_t*>(e->d_q_agg.p);
    ca.q_nom_ptr = static_cast<const int32_t*>(e->d_q_nom_ptr.p);
    ca.q_nom_pending = static_cast<const int64_t*>(e->d_q_nom_idx.p);
    ca.q_nom_req = static_cast<int64_t*>(e->d_q_nom_req.p);
    ca.q_nom_reqp = static_cast<uint8_t*>(e->d_q_nom_reqp.p);
    ca.q_other = static_cast<int64_t*>(e->d_q_other.p);
    ca.q_otherp = static_cast<uint8_t*>(e->d_q_otherp.p);
    e->q_agg_dyn = static_cast<const int64_t*>(e->d_q_agg.p);
  }
  if (W) {
    ca.net_eff_ptr = static_cast<const int32_t*>(e->d_net_eff_ptr.p);
    ca.net_eff_key = static_cast<const int32_t*>(e->d_net_eff_key.p);
    ca.net_eff_cost = static_cast<const int64_t*>(e->d_net_eff_cost.p);
    ca.net_key_flag = static_cast<uint8_t*>(e->d_net_key_flag.p);
    ca.net_pair_end = static_cast<int32_t*>(e->d_net_dyn_end.p);
    ca.net_pair_node = static_cast<int32_t*>(e->d_net_dyn_node.p);
    ca.net_pair_max = static_cast<int64_t*>(e->d_net_dyn_max.p);
    e->net_dyn_active = true;
  }
  // LoadVariationRiskBalancing carries no commit state: its rows are swept once, the per-pod evaluation leaves it out
  const uint32_t lvrb_bit = 1u << SPX_PLUGIN_LVRB;
  const uint32_t step_mask = plugin_mask & ~lvrb_bit;
  rc = (plugin_mask & lvrb_bit) ? spx_eval(e, lvrb_bit, row_begin, row_end) : SPX_OK;
  auto step = [&](int64_t pod) -> int {  // one pod: sweep its row on the current tables, argmax, Reserve bookkeeping
    int r;
    bool decided = false;  // Allocatable's masked normalisation and the argmax in one kernel where that form applies
    if ((r = decide_masked(e, step_mask, plugin_mask, pod, pod + 1, &decided))) return r;
    if (!decided) {
      if ((r = spx_eval(e, step_mask, pod, pod + 1))) return r;
      if ((r = spx_eval_best(e, plugin_mask, pod, pod + 1))) return r;
    }
    ca.pod = pod;
    spx::launch_commit_apply(ca, e->stream);
    return hipGetLastError() == hipSuccess ? SPX_OK : fail(e, SPX_ERR_HIP, "k_commit_apply launch failed");
  };
  // The first pod runs as plain launches (anything still to allocate is allocated here).  The same dozen launches are then
  // captured ONCE with every sweep reading its row from a device counter that k_commit_apply advances, and the graph is replayed
  // for the remaining pods: the host enqueues one graph launch per pod instead of a dozen kernels (measured: 162 -> about 40 us
  // per pod for the full profile at 20k nodes).
  if (rc == SPX_OK) rc = step(row_begin);
  const int64_t remaining = row_end - row_begin - 1;
  if (rc == SPX_OK && remaining > 0) {
    bool replayed = false;
    if (remaining >= 4 && !e->option[SPX_OPT_COMMIT_FROM_MEMORY] && ensure(e, e->d_row_counter, 8) == SPX_OK) {
      const int64_t first = row_begin + 1;
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      if (hipMemcpyAsync(e->d_row_counter.p, &first, 8, hipMemcpyHostToDevice, e->stream) == hipSuccess && hipStreamSynchronize(e->stream) == hipSuccess &&
          hipStreamBeginCapture(e->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
        e->row_indirect = static_cast<const int64_t*>(e->d_row_counter.p);
        ca.row_counter = static_cast<int64_t*>(e->d_row_counter.p);
        const int crc = step(first);  // the row number only sizes the grids (one row); the kernels read the counter
        e->row_indirect = nullptr;
        ca.row_counter = nullptr;
        const hipError_t end = hipStreamEndCapture(e->stream, &graph);
        if (crc == SPX_OK && end == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          replayed = true;
          for (int64_t i = 0; i < remaining; ++i)
            if (hipGraphLaunch(exec, e->stream) != hipSuccess) {
              rc = fail(e, SPX_ERR_HIP, "hipGraphLaunch failed in the sequential commit loop");
              break;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
      }
    }
    if (!replayed)
      for (int64_t pod = row_begin + 1; pod < row_end && rc == SPX_OK; ++pod) rc = step(pod);
    for (int p = 0; p < SPX_NUM_PLUGINS; ++p)  // the host-side bookkeeping saw only the rows it enqueued itself
      if ((step_mask >> p) & 1u) e->eval_info[p].begin = row_begin, e->eval_info[p].end = row_end;
  }
  e->q_agg_dyn = nullptr;
  e->net_dyn_active = false;
  if (rc == SPX_OK && tlp_missing_out && T) {
    if (hipMemcpyAsync(tlp_missing_out, e->d_tlp_missing.p, Nn * 8, hipMemcpyDeviceToHost, e->stream) != hipSuccess) rc = fail(e, SPX_ERR_HIP, "copy of the missing-utilisation column failed");
  }
  // ---- restore the snapshot (also after an error: the tables must not stay half-committed)
  for (const Saved& sv : saved)
    (void)hipMemcpyAsync(sv.buf->p, static_cast<const char*>(e->d_commit_save.p) + sv.off, sv.bytes, hipMemcpyDeviceToDevice, e->stream);
  e->lroc_tab_ready = false;
  if (rc != SPX_OK) {
    (void)hipStreamSynchronize(e->stream);
    return rc;
  }
  const size_t rows = static_cast<size_t>(row_end - row_begin);
  const int64_t* ds = static_cast<const int64_t*>(e->d_best.p);
  const int32_t* dn = reinterpret_cast<const int32_t*>(ds + P);
  SPX_HIP(e, hipMemcpyAsync(weighted_score, ds + row_begin, rows * 8, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipMemcpyAsync(node_idx, dn + row_begin, rows * 4, hipMemcpyDeviceToHost, e->stream));
  if (n_ties) SPX_HIP(e, hipMemcpyAsync(n_ties, dn + P + row_begin, rows * 4, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  if (tlp_missing_out && !T) std::memset(tlp_missing_out, 0, Nn * 8);
  return SPX_OK;
}

}  // namespace

int spx_commit_sequential(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end, int32_t* node_idx,
                          int64_t* weighted_score, int32_t* n_ties, int64_t* tlp_missing_out) {
  if (!e || !node_idx || !weighted_score) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  const uint32_t allowed = (1u << SPX_PLUGIN_ALLOCATABLE) | (1u << SPX_PLUGIN_TLP) | (1u << SPX_PLUGIN_LVRB);
  const uint32_t with_filters = allowed | (1u << SPX_PLUGIN_NRT) | (1u << SPX_PLUGIN_NETOVERHEAD) | (1u << SPX_PLUGIN_CAPACITY);
  if (plugin_mask == 0 || (plugin_mask & ~with_filters))
    return fail(e, SPX_ERR_ARG, "spx_commit_sequential supports Allocatable / TargetLoadPacking / LoadVariationRiskBalancing / NodeResourceTopologyMatch / "
                                "NetworkOverhead / CapacityScheduling");
  if (plugin_mask & ~allowed) {
    if (e->n_pods <= 0 || e->n_nodes <= 0) return fail(e, SPX_ERR_STATE, "shape unknown");
    if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
    if (row_begin == row_end) return SPX_OK;
    return commit_with_filters(e, plugin_mask, row_begin, row_end, node_idx, weighted_score, n_ties, tlp_missing_out);
  }
  const bool A = plugin_mask & (1u << SPX_PLUGIN_ALLOCATABLE);
  const bool T = plugin_mask & (1u << SPX_PLUGIN_TLP);
  const bool L = plugin_mask & (1u << SPX_PLUGIN_LVRB);
  if (!(e->tri_nodes && e->tri_pods)) return fail(e, SPX_ERR_STATE, "trimaran node/pod tables not uploaded");
  if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
  int rc;
  if (A && (rc = prepare_alloc(e))) return rc;
  const size_t rows = static_cast<size_t>(row_end - row_begin), N = static_cast<size_t>(e->n_nodes);
  if (rows == 0) return SPX_OK;
  // scratch: [missing int64 N | score int64 rows | node int32 rows | ties int32 rows]
  if ((rc = ensure(e, e->d_commit, N * 8 + rows * 16))) return rc;
  spx::CommitArgs c{};
  fill_trimaran(e, c.t);
  c.t.row_begin = row_begin;
  c.t.row_end = row_end;
  if (L) {  // LVRB has no commit state: sweep its rows once (the engine's LVRB table is (re)written for this row range)
    if ((rc = spx_eval(e, 1u << SPX_PLUGIN_LVRB, row_begin, row_end))) return rc;
    c.lv_table = static_cast<const uint8_t*>(e->score[SPX_PLUGIN_LVRB].p);
    if (e->score_stride[SPX_PLUGIN_LVRB] != e->row_stride) return fail(e, SPX_ERR_STATE, "bound LVRB table must use the engine row stride");
  }
  c.use_mask = (A ? 1u : 0u) | (T ? 2u : 0u) | (L ? 4u : 0u);
  c.w_alloc = e->plugin_weight[SPX_PLUGIN_ALLOCATABLE];
  c.w_tlp = e->plugin_weight[SPX_PLUGIN_TLP];
  c.w_lvrb = e->plugin_weight[SPX_PLUGIN_LVRB];
  c.missing = static_cast<int64_t*>(e->d_commit.p);
  c.out_score = c.missing + N;
  c.out_node = reinterpret_cast<int32_t*>(c.out_score + rows);
  c.out_ties = n_ties ? c.out_node + rows : nullptr;
  SPX_HIP(e, hipMemcpyAsync(c.missing, e->d_tlp_missing.p, N * 8, hipMemcpyDeviceToDevice, e->stream));
  spx::launch_commit_trimaran(c, e->stream);
  e->last_commit_path = 1;
  SPX_HIP(e, hipGetLastError());
  SPX_HIP(e, hipMemcpyAsync(weighted_score, c.out_score, rows * 8, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipMemcpyAsync(node_idx, c.out_node, rows * 4, hipMemcpyDeviceToHost, e->stream));
  if (n_ties) SPX_HIP(e, hipMemcpyAsync(n_ties, c.out_node + rows, rows * 4, hipMemcpyDeviceToHost, e->stream));
  if (tlp_missing_out) SPX_HIP(e, hipMemcpyAsync(tlp_missing_out, c.missing, N * 8, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

// ---------------------------------------------------------------- object tables -> SoA -> device in one call
// What a cgo (or any FFI) caller wants: it holds object tables (marshalled itself, or decoded by spx_ingest_*) and should not have to
// size and own two dozen intermediate arrays per plugin.  Each function runs the host flatteners with the engine's current plugin
// parameters and uploads the result, exactly the sequence of scheduler-plugins_amd/engine.py's load_*_objects.
int spx_load_trimaran(spx_engine* e, const spx_node_objects* nodes, const spx_resource_classes* rc, const spx_pod_objects* pods, const spx_metrics_objects* metrics,
                      const spx_assigned_objects* assigned) {
  if (!e || !nodes || !pods || !metrics) return SPX_ERR_ARG;
  const size_t N = static_cast<size_t>(nodes->n_nodes), P = static_cast<size_t>(pods->n_pods), R = e->alloc_res.size();
  spx_allocatable_params ap{e->alloc_mode, static_cast<int32_t>(R), e->alloc_res.data(), e->alloc_weight.data()};
  std::vector<int64_t> alloc(R * N);
  if (spx_flatten_alloc_nodes(nodes, rc, &ap, alloc.data()) != SPX_OK) return fail(e, SPX_ERR_ARG, "spx_flatten_alloc_nodes failed");
  spx_alloc_nodes_soa an{nodes->n_nodes, static_cast<int32_t>(R), alloc.data()};
  int rc_;
  if ((rc_ = spx_upload_alloc_nodes(e, &an))) return rc_;
  std::vector<int64_t> cap(N), missing(N), acpu(N), amem(N), tpod(P), rcpu(P), rmem(P);
  std::vector<double> util(N), cavg(N), cstd(N), mavg(N), mstd(N);
  std::vector<uint8_t> valid(N), flags(N);
  if (spx_flatten_trimaran_nodes(nodes, metrics, assigned, &e->tlp, cap.data(), util.data(), missing.data(), valid.data(), acpu.data(), amem.data(), cavg.data(),
                                 cstd.data(), mavg.data(), mstd.data(), flags.data()) != SPX_OK)
    return fail(e, SPX_ERR_ARG, "spx_flatten_trimaran_nodes failed");
  spx_trimaran_nodes_soa tn{nodes->n_nodes, cap.data(), util.data(), missing.data(), valid.data(), acpu.data(), amem.data(), cavg.data(), cstd.data(), mavg.data(),
                            mstd.data(), flags.data()};
  if ((rc_ = spx_upload_trimaran_nodes(e, &tn))) return rc_;
  if (spx_flatten_trimaran_pods(pods, &e->tlp, tpod.data(), rcpu.data(), rmem.data()) != SPX_OK) return fail(e, SPX_ERR_ARG, "spx_flatten_trimaran_pods failed");
  spx_trimaran_pods_soa tp{pods->n_pods, tpod.data(), rcpu.data(), rmem.data()};
  return spx_upload_trimaran_pods(e, &tp);
}

// A new pending batch for the trimaran plugins (and Allocatable): the three pod columns are flattened by all host threads straight
// into the engine's pinned staging buffer and leave with asynchronous DMAs at link speed — through pageable memory (flatten into
// the caller's arrays, then spx_upload_trimaran_pods) the runtime copies each column a second time into its own staging first:
// 1.04 ms for 100 000 pods against the sweep's 0.42.
int spx_load_trimaran_pods(spx_engine* e, const spx_pod_objects* pods) {
  if (!e || !pods) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  int rc = set_pods(e, pods->n_pods);
  if (rc) return rc;
  const size_t p = static_cast<size_t>(pods->n_pods), col = (p * 8 + 255) & ~static_cast<size_t>(255), bytes = 3 * col;
  SPX_HIP(e, hipStreamSynchronize(e->stream));  // an earlier upload may still be reading the staging buffer
  if (e->h_stage_bytes < bytes) {
    if (e->h_stage) SPX_HIP(e, hipHostFree(e->h_stage));
    e->h_stage = nullptr, e->h_stage_bytes = 0;
    SPX_HIP(e, hipHostMalloc(&e->h_stage, bytes + 65536, hipHostMallocDefault));
    e->h_stage_bytes = bytes + 65536;
  }
  char* h = static_cast<char*>(e->h_stage);
  int64_t* tpod = reinterpret_cast<int64_t*>(h);
  int64_t* rcpu = reinterpret_cast<int64_t*>(h + col);
  int64_t* rmem = reinterpret_cast<int64_t*>(h + 2 * col);
  if (spx_flatten_trimaran_pods(pods, &e->tlp, tpod, rcpu, rmem) != SPX_OK) return fail(e, SPX_ERR_ARG, "spx_flatten_trimaran_pods failed");
  if ((rc = upload(e, e->d_tlp_pod, tpod, p * 8))) return rc;
  if ((rc = upload(e, e->d_lv_rcpu, rcpu, p * 8))) return rc;
  if ((rc = upload(e, e->d_lv_rmem, rmem, p * 8))) return rc;
  e->tri_pods = true;
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

int spx_load_nrt(spx_engine* e, const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_resource_classes* rc, const spx_pod_objects* pods,
                 const spx_nrt_params* params) {
  if (!e || !nodes || !nrt || !pods || !params) return SPX_ERR_ARG;
  using clk = std::chrono::steady_clock;
  auto since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
  for (double& x : e->load_nrt_ms) x = 0.0;
  auto t0 = clk::now();
  int32_t n_res = 0, slot_res[SPX_NRT_MAX_RES] = {0};
  uint8_t slot_flags[SPX_NRT_MAX_RES] = {0};
  int64_t slot_weight[SPX_NRT_MAX_RES] = {0};
  if (spx_flatten_nrt_slots(pods, nrt, rc, params, &n_res, slot_res, slot_flags, slot_weight) != SPX_OK) return fail(e, SPX_ERR_ARG, "spx_flatten_nrt_slots failed");
  const spx_nrt_slots slots{n_res, slot_res, slot_flags, slot_weight};
  e->load_nrt_ms[0] = since(t0);  // 0: spx_flatten_nrt_slots
  t0 = clk::now();
  int rc_;
  if ((rc_ = spx_set_nrt_params(e, params)) || (rc_ = spx_upload_nrt_slots(e, &slots))) return rc_;
  // (both halves below check the batch / node count against what the engine holds: settled here, before they run side by side)
  if ((rc_ = set_nodes(e, nodes->n_nodes)) || (rc_ = set_pods(e, pods->n_pods))) return rc_;
  e->load_nrt_ms[3] = since(t0);  // 3: params + slot table
  const size_t N = static_cast<size_t>(nodes->n_nodes), P = static_cast<size_t>(pods->n_pods), R = static_cast<size_t>(n_res > 0 ? n_res : 1), Z = SPX_NRT_MAX_ZONES,
               Cn = SPX_NRT_MAX_CTRS;
  // Round 6: the node half (flatten 1.8 ms + upload 2.7 ms at 20 000 nodes) and the pod half (0.4 + 2.7 ms at 8 192 pods) touch disjoint
  // engine state — node tables / the blob staging, pod tables / the record stream's staging — and one stream; they run on two host
  // threads (each with its own worker pool, parallel.hpp).  Stages 1 / 4 and 2 / 5 therefore overlap in time.
  int rc_pods = SPX_OK;
  std::thread pod_half([&] {
    const auto t1 = clk::now();
    std::vector<uint8_t> qos(P), nn(P), nctr(P), ckind(P * Cn), cpres(P * Cn), ppres(P);
    std::vector<int64_t> creq(P * Cn * R), preq(P * R);
    if (spx_flatten_nrt_pods(pods, rc, &slots, qos.data(), nn.data(), nctr.data(), ckind.data(), cpres.data(), creq.data(), ppres.data(), preq.data()) != SPX_OK) {
      rc_pods = fail(e, SPX_ERR_ARG, "spx_flatten_nrt_pods failed");
      return;
    }
    e->load_nrt_ms[2] = since(t1);  // 2: pod columns allocated + spx_flatten_nrt_pods
    const auto t2 = clk::now();
    const spx_nrt_pods_soa ps{pods->n_pods, n_res, qos.data(), nn.data(), nctr.data(), ckind.data(), cpres.data(), creq.data(), ppres.data(), preq.data()};
    rc_pods = spx_upload_nrt_pods(e, &ps);
    e->load_nrt_ms[5] = since(t2);  // 5: spx_upload_nrt_pods (item stream, pod classes, rank stream)
  });
  int rc_nodes = SPX_OK;
  {
    const auto t1 = clk::now();
    std::vector<uint8_t> nflags(N), nz(N), zid(N * Z), zp(N * Z), np(N);
    std::vector<int32_t> max_numa(N), zcost(N * Z * Z);
    std::vector<int64_t> zavail(N * Z * R);
    std::vector<float> minavg(N * Z);
    if (spx_flatten_nrt_nodes(nodes, nrt, &slots, nflags.data(), max_numa.data(), nz.data(), zid.data(), zp.data(), zavail.data(), zcost.data(), minavg.data(), np.data()) !=
        SPX_OK) {
      rc_nodes = fail(e, SPX_ERR_ARG, "spx_flatten_nrt_nodes failed");
    } else {
      e->load_nrt_ms[1] = since(t1);  // 1: node columns allocated + spx_flatten_nrt_nodes
      const auto t2 = clk::now();
      const spx_nrt_nodes_soa ns{nodes->n_nodes, n_res, nflags.data(), max_numa.data(), nz.data(), zid.data(), zp.data(), zavail.data(), zcost.data(), minavg.data(), np.data()};
      rc_nodes = spx_upload_nrt_nodes(e, &ns);
      e->load_nrt_ms[4] = since(t2);  // 4: spx_upload_nrt_nodes (precondition checks, window-local node order, one blob, derived columns on the device)
    }
  }
  pod_half.join();
  return rc_nodes ? rc_nodes : rc_pods;
}

// The four loaders of a full profile side by side: they fill disjoint tables of the engine (trimaran + Allocatable columns, NRT tables,
// NetworkOverhead tables, quota tables), share one stream, and each takes a worker pool of its own.  Members left NULL skip their loader.
int spx_load_profile(spx_engine* e, const spx_profile_objects* o) {
  if (!e || !o || !o->nodes || !o->pods) return SPX_ERR_ARG;
  int rc_;
  if ((rc_ = set_nodes(e, o->nodes->n_nodes)) || (rc_ = set_pods(e, o->pods->n_pods))) return rc_;
  int rcs[4] = {SPX_OK, SPX_OK, SPX_OK, SPX_OK};
  std::vector<std::thread> th;
  if (o->nrt && o->nrt_params) th.emplace_back([&] { rcs[1] = spx_load_nrt(e, o->nodes, o->nrt, o->rc, o->pods, o->nrt_params); });  // the longest first
  if (o->appgroups && o->nettopo) th.emplace_back([&] { rcs[2] = spx_load_network(e, o->nodes, o->pods, o->appgroups, o->nettopo); });
  if (o->quota) th.emplace_back([&] { rcs[3] = spx_load_quota(e, o->pods, o->rc, o->quota); });
  if (o->metrics) rcs[0] = spx_load_trimaran(e, o->nodes, o->rc, o->pods, o->metrics, o->assigned);
  for (std::thread& t : th) t.join();
  for (int r : rcs)
    if (r) return r;
  return SPX_OK;
}

int spx_last_load_nrt_ms(const spx_engine* e, double* ms6) {
  if (!e || !ms6) return SPX_ERR_ARG;
  std::memcpy(ms6, e->load_nrt_ms, sizeof e->load_nrt_ms);
  return SPX_OK;
}

int spx_load_network(spx_engine* e, const spx_node_objects* nodes, const spx_pod_objects* pods, const spx_appgroup_objects* appgroups, const spx_nettopo_objects* nettopo) {
  if (!e || !nodes || !pods || !appgroups || !nettopo) return SPX_ERR_ARG;
  const size_t P = static_cast<size_t>(pods->n_pods);
  const size_t rg = static_cast<size_t>(nettopo->n_regions), zc = static_cast<size_t>(nettopo->n_zones);
  std::vector<int32_t> rcost(rg * rg ? rg * rg : 1, -1), zcost(zc * zc ? zc * zc : 1, -1);
  if (spx_flatten_net_topo(nettopo, rcost.data(), zcost.data()) != SPX_OK) return fail(e, SPX_ERR_ARG, "spx_flatten_net_topo failed");
  int32_t n_keys = 0;
  int64_t n_pairs = 0, n_eff = 0;
  if (spx_flatten_net_keys(pods, appgroups, &n_keys, &n_pairs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != SPX_OK)
    return fail(e, SPX_ERR_ARG, "spx_flatten_net_keys failed");
  std::vector<int32_t> pod_key(P), topo(P), pair_ptr(static_cast<size_t>(n_keys) + 1), pair_node(n_pairs > 0 ? static_cast<size_t>(n_pairs) : 1);
  std::vector<uint8_t> eq(n_keys > 0 ? static_cast<size_t>(n_keys) : 1);
  std::vector<int64_t> pair_max(n_pairs > 0 ? static_cast<size_t>(n_pairs) : 1);
  if (spx_flatten_net_keys(pods, appgroups, &n_keys, &n_pairs, pod_key.data(), topo.data(), eq.data(), pair_ptr.data(), pair_node.data(), pair_max.data()) != SPX_OK)
    return fail(e, SPX_ERR_ARG, "spx_flatten_net_keys failed");
  if (spx_flatten_net_commit(pods, appgroups, &n_eff, nullptr, nullptr, nullptr) != SPX_OK) return fail(e, SPX_ERR_ARG, "spx_flatten_net_commit failed");
  std::vector<int32_t> eff_ptr(P + 1), eff_key(n_eff > 0 ? static_cast<size_t>(n_eff) : 1);
  std::vector<int64_t> eff_cost(n_eff > 0 ? static_cast<size_t>(n_eff) : 1);
  if (spx_flatten_net_commit(pods, appgroups, &n_eff, eff_ptr.data(), eff_key.data(), eff_cost.data()) != SPX_OK) return fail(e, SPX_ERR_ARG, "spx_flatten_net_commit failed");
  int rc_;
  const spx_net_nodes_soa nn{nodes->n_nodes, nodes->region, nodes->zone};
  if ((rc_ = spx_upload_net_nodes(e, &nn))) return rc_;
  const spx_net_topo_soa nt{nettopo->n_regions, nettopo->n_zones, rcost.data(), zcost.data()};
  if ((rc_ = spx_upload_net_topo(e, &nt))) return rc_;
  const spx_net_pods_soa np{pods->n_pods, n_keys, pod_key.data(), eq.data(), pair_ptr.data(), pair_node.data(), pair_max.data(), topo.data()};
  if ((rc_ = spx_upload_net_pods(e, &np))) return rc_;
  const spx_net_commit_soa nc{pods->n_pods, eff_ptr.data(), eff_key.data(), eff_cost.data()};
  return spx_upload_net_commit(e, &nc);
}

int spx_load_quota(spx_engine* e, const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* quota) {
  if (!e || !pods || !quota) return SPX_ERR_ARG;
  constexpr size_t S = SPX_QUOTA_SLOTS;
  const size_t P = static_cast<size_t>(pods->n_pods), NS = static_cast<size_t>(quota->n_namespaces), NN = quota->n_nominated > 0 ? static_cast<size_t>(quota->n_nominated) : 1;
  std::vector<int32_t> pod_ns(P), pod_prio(P), nom_ptr(NS + 1), nom_prio(NN);
  std::vector<int64_t> pod_req(P * S), agg_used(S), agg_min(S), other((NS ? NS : 1) * S), nom_pending(NN), nom_req(NN * S);
  std::vector<uint8_t> pod_reqp(P), other_p(NS ? NS : 1), nom_reqp(NN);
  uint8_t agg_used_p = 0, agg_min_p = 0;
  if (spx_flatten_quota(pods, rc, quota, pod_ns.data(), pod_prio.data(), pod_req.data(), pod_reqp.data(), agg_used.data(), &agg_used_p, agg_min.data(), &agg_min_p, other.data(),
                        other_p.data(), nom_ptr.data(), nom_prio.data(), nom_pending.data(), nom_req.data(), nom_reqp.data()) != SPX_OK)
    return fail(e, SPX_ERR_ARG, "spx_flatten_quota failed");
  spx_quota_soa q{};
  q.n_pods = pods->n_pods, q.n_namespaces = quota->n_namespaces;
  q.pod_ns = pod_ns.data(), q.pod_priority = pod_prio.data(), q.pod_req = pod_req.data(), q.pod_req_present = pod_reqp.data();
  q.has_quota = quota->has_quota, q.used = quota->used, q.used_present = quota->used_present, q.max = quota->max, q.max_present = quota->max_present;
  q.agg_used = agg_used.data(), q.agg_used_present = &agg_used_p, q.agg_min = agg_min.data(), q.agg_min_present = &agg_min_p;
  q.other_nominated = other.data(), q.other_nominated_present = other_p.data();
  q.nom_ptr = nom_ptr.data(), q.nom_priority = nom_prio.data(), q.nom_pending_index = nom_pending.data(), q.nom_req = nom_req.data(), q.nom_req_present = nom_reqp.data();
  q.min = quota->min, q.min_present = quota->min_present;
  return spx_upload_quota(e, &q);
}

int spx_nrt_filter_path(const spx_engine* e) { return e ? e->last_nrt_filter : 0; }

int spx_nrt_packed_score_slots(const spx_engine* e) {
  if (!e) return SPX_ERR_ARG;
  NrtPacked pk;
  if (!nrt_packed_score(e, &pk)) return 0;
  return static_cast<int>(0x1000000u | pk.small_slots | (static_cast<uint32_t>(pk.tab_slot + 1) << 16));
}

int spx_commit_path(const spx_engine* e) { return e ? e->last_commit_path : SPX_ERR_ARG; }

int spx_kernel_path(const spx_engine* e, int plugin) {
  if (!e) return SPX_ERR_ARG;
  if (plugin == SPX_PLUGIN_NRT)
    return (e->nrt_fast_slots && e->nrt_fast_nodes && e->nrt_fast_pods && !forced_reference(e, SPX_PLUGIN_NRT) &&
            (e->nrt_params.strategy != SPX_NRT_LEAST_NUMA_NODES || e->nrt_ln_ok)) ? 1 : 0;
  if (plugin == SPX_PLUGIN_NETOVERHEAD) return (e->net_nodes && e->net_class16 && e->net_n_classes > 0 && !forced_reference(e, SPX_PLUGIN_NETOVERHEAD)) ? 1 : 0;
  if (plugin == SPX_PLUGIN_LROC) return (lroc_exact53(e) && !e->option[SPX_OPT_LROC_FLOAT64]) ? 1 : 0;
  if (plugin == SPX_PLUGIN_TLP) return (e->tlp.target_utilization >= 1 && e->tlp.target_utilization <= 99 && !(launch_opts(e) & spx::kOptTrimaranExact)) ? 1 : 0;
  return 0;
}

int spx_last_eval_ms(spx_engine* e, float* ms) {
  if (!e || !ms) return SPX_ERR_ARG;
  if (!e->timed) return fail(e, SPX_ERR_STATE, "no spx_eval has run");
  SPX_HIP(e, hipEventSynchronize(e->ev1));
  SPX_HIP(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
  return SPX_OK;
}

namespace {
// A reader thread's own pinned staging buffer and stream (thread-local, per device): a row lands in pinned memory with an async
// copy on the reader's stream and is copied out from there — no pageable-memory path through the runtime's shared staging
// buffers, no engine stream, no engine state.  The calling thread's current device is set first (it is arbitrary on a reader
// thread; with spx_multi the engines live on different devices).
struct ReaderSlot {
  int device = -1;
  hipStream_t stream = nullptr;
  void* pinned = nullptr;
  size_t bytes = 0;
  ~ReaderSlot() {
    if (device < 0) return;
    if (hipSetDevice(device) != hipSuccess) return;
    if (pinned) (void)hipHostFree(pinned);
    if (stream) (void)hipStreamDestroy(stream);
  }
};
thread_local ReaderSlot tl_reader[8];  // by device id modulo 8

int reader_copy(spx_engine* e, void* out, const void* src, size_t bytes) {
  SPX_HIP(e, hipSetDevice(e->device));
  ReaderSlot& r = tl_reader[static_cast<unsigned>(e->device) & 7u];
  if (r.device != e->device) {
    if (r.device >= 0) {  // slot taken by another device id (more than 8 devices): the plain copy
      SPX_HIP(e, hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
      return SPX_OK;
    }
    SPX_HIP(e, hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    r.device = e->device;
  }
  if (r.bytes < bytes) {
    if (r.pinned) SPX_HIP(e, hipHostFree(r.pinned));
    r.pinned = nullptr, r.bytes = 0;
    const size_t want = (bytes + 65535) & ~static_cast<size_t>(65535);
    SPX_HIP(e, hipHostMalloc(&r.pinned, want, hipHostMallocDefault));
    r.bytes = want;
  }
  SPX_HIP(e, hipMemcpyAsync(r.pinned, src, bytes, hipMemcpyDeviceToHost, r.stream));
  SPX_HIP(e, hipStreamSynchronize(r.stream));
  std::memcpy(out, r.pinned, bytes);
  return SPX_OK;
}
}  // namespace

int spx_fetch_scores(spx_engine* e, int plugin, int64_t pod_row, uint8_t* out) {
  if (!e || !out) return SPX_ERR_ARG;
  if (plugin < 0 || plugin >= SPX_NUM_PLUGINS || !(e->evaluated & (1u << plugin)))
    return fail(e, SPX_ERR_STATE, "plugin has not been evaluated");
  if (pod_row < 0 || pod_row >= e->n_pods) return fail(e, SPX_ERR_ARG, "pod_row out of range");
  if (int rc = rows_evaluated(e, plugin, pod_row, pod_row + 1)) return rc;
  // one row, D2H, from any number of reader threads after spx_sync() (no engine state is touched: reader_copy)
  const uint8_t* src = static_cast<const uint8_t*>(e->score[plugin].p) + pod_row * e->score_stride[plugin];
  return reader_copy(e, out, src, static_cast<size_t>(e->n_nodes));
}

int spx_fetch_status(spx_engine* e, int plugin, int64_t pod_row, uint8_t* out) {
  if (!e || !out) return SPX_ERR_ARG;
  if (plugin < 0 || plugin >= SPX_NUM_PLUGINS || !e->status[plugin].p || !(e->evaluated & (1u << plugin)))
    return fail(e, SPX_ERR_STATE, "plugin has no evaluated Filter table");
  if (pod_row < 0 || pod_row >= e->n_pods) return fail(e, SPX_ERR_ARG, "pod_row out of range");
  if (int rc = rows_evaluated(e, plugin, pod_row, pod_row + 1)) return rc;
  const uint8_t* src = static_cast<const uint8_t*>(e->status[plugin].p) + pod_row * e->row_stride;
  return reader_copy(e, out, src, static_cast<size_t>(e->n_nodes));
}

int spx_fetch_raw(spx_engine* e, int plugin, int which, int64_t pod_row, int64_t* out) {
  if (!e || !out) return SPX_ERR_ARG;
  // a raw row is computed on demand (a single-row launch on the engine stream into one scratch row): concurrent readers are
  // serialised here — correct from any thread, but not a fan-out path; the uint8 tables are
  std::lock_guard<std::mutex> raw_guard(e->raw_mu);
  SPX_HIP(e, hipSetDevice(e->device));
  if (e->n_nodes <= 0) return fail(e, SPX_ERR_STATE, "no node table uploaded");
  int rc;
  const size_t bytes = static_cast<size_t>(e->n_nodes) * sizeof(int64_t);
  if (plugin == SPX_PLUGIN_ALLOCATABLE) {
    if ((rc = prepare_alloc(e))) return rc;
    SPX_HIP(e, hipMemcpyAsync(out, e->d_alloc_raw.p, bytes, hipMemcpyDeviceToHost, e->stream));
    SPX_HIP(e, hipStreamSynchronize(e->stream));
    return SPX_OK;
  }
  if (plugin == SPX_PLUGIN_NRT) {  // TopologyMatch has no NormalizeScore (score.go:104-106)
    if (!(e->nrt_slots && e->nrt_nodes && e->nrt_pods)) return fail(e, SPX_ERR_STATE, "NRT slot/node/pod tables not uploaded");
    if (pod_row < 0 || pod_row >= e->n_pods) return fail(e, SPX_ERR_ARG, "pod_row out of range");
    if ((rc = ensure(e, e->d_raw_row, bytes))) return rc;
    if (e->nrt_params.strategy == SPX_NRT_LEAST_NUMA_NODES && (rc = build_ln_tab(e))) return rc;
    if ((rc = ensure_nrt_creq(e))) return rc;
    spx::NrtArgs na{};
    fill_nrt(e, na);
    na.row_begin = pod_row;
    na.row_end = pod_row + 1;
    na.out_raw = static_cast<int64_t*>(e->d_raw_row.p);
    spx::launch_nrt(na, e->stream);
    SPX_HIP(e, hipGetLastError());
    SPX_HIP(e, hipMemcpyAsync(out, e->d_raw_row.p, bytes, hipMemcpyDeviceToHost, e->stream));
    SPX_HIP(e, hipStreamSynchronize(e->stream));
    return SPX_OK;
  }
  if (plugin == SPX_PLUGIN_NETOVERHEAD) {  // raw accumulated cost / satisfied / violated (PreFilterState maps)
    if (!(e->net_nodes && e->net_topo && e->net_pods)) return fail(e, SPX_ERR_STATE, "NetworkOverhead tables not uploaded");
    if (pod_row < 0 || pod_row >= e->n_pods) return fail(e, SPX_ERR_ARG, "pod_row out of range");
    if (which < SPX_NET_RAW_COST || which > SPX_NET_RAW_VIOLATED) return fail(e, SPX_ERR_ARG, "which: 0 cost, 1 satisfied, 2 violated");
    if ((rc = ensure(e, e->d_raw_row, bytes))) return rc;
    spx::NetArgs g{};
    fill_net(e, g);
    g.row_begin = pod_row;
    g.row_end = pod_row + 1;
    g.out_raw = static_cast<int64_t*>(e->d_raw_row.p);
    g.raw_which = which;
    spx::launch_net(g, e->stream);
    SPX_HIP(e, hipGetLastError());
    SPX_HIP(e, hipMemcpyAsync(out, e->d_raw_row.p, bytes, hipMemcpyDeviceToHost, e->stream));
    SPX_HIP(e, hipStreamSynchronize(e->stream));
    return SPX_OK;
  }
  if (plugin == SPX_PLUGIN_PEAKS) {  // Peaks.Score before NormalizeScore: the power jump x 1e15
    if (!(e->peaks_nodes && e->peaks_pods)) return fail(e, SPX_ERR_STATE, "Peaks node/pod tables not uploaded");
    if (pod_row < 0 || pod_row >= e->n_pods) return fail(e, SPX_ERR_ARG, "pod_row out of range");
    if ((rc = ensure(e, e->d_raw_row, bytes))) return rc;
    spx::PeaksArgs ka{};
    fill_peaks(e, ka);
    ka.row_begin = pod_row;
    ka.row_end = pod_row + 1;
    ka.out_raw = static_cast<int64_t*>(e->d_raw_row.p);
    spx::launch_peaks(ka, e->stream);
    SPX_HIP(e, hipGetLastError());
    SPX_HIP(e, hipMemcpyAsync(out, e->d_raw_row.p, bytes, hipMemcpyDeviceToHost, e->stream));
    SPX_HIP(e, hipStreamSynchronize(e->stream));
    return SPX_OK;
  }
  if (plugin != SPX_PLUGIN_TLP && plugin != SPX_PLUGIN_LVRB) return fail(e, SPX_ERR_ARG, "raw rows: unsupported plugin");
  if (!(e->tri_nodes && e->tri_pods)) return fail(e, SPX_ERR_STATE, "trimaran node/pod tables not uploaded");
  if (pod_row < 0 || pod_row >= e->n_pods) return fail(e, SPX_ERR_ARG, "pod_row out of range");
  if ((rc = ensure(e, e->d_raw_row, bytes))) return rc;
  spx::TrimaranArgs a{};
  fill_trimaran(e, a);
  spx::launch_trimaran_raw(a, plugin, pod_row, static_cast<int64_t*>(e->d_raw_row.p), e->stream);
  SPX_HIP(e, hipGetLastError());
  SPX_HIP(e, hipMemcpyAsync(out, e->d_raw_row.p, bytes, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  return SPX_OK;
}

namespace {
// rows [row_begin,row_end) of a uint8 table into a caller buffer with its own row stride: one strided D2H
int fetch_rows(spx_engine* e, const uint8_t* table, int64_t stride, int64_t row_begin, int64_t row_end, uint8_t* out, int64_t out_stride) {
  if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
  if (out_stride < e->n_nodes) return fail(e, SPX_ERR_ARG, "out_stride is smaller than n_nodes");
  if (row_begin == row_end) return SPX_OK;
  SPX_HIP(e, hipSetDevice(e->device));  // reader threads: the current device is per thread
  SPX_HIP(e, hipMemcpy2D(out, static_cast<size_t>(out_stride), table + row_begin * stride, static_cast<size_t>(stride),
                         static_cast<size_t>(e->n_nodes), static_cast<size_t>(row_end - row_begin), hipMemcpyDeviceToHost));
  return SPX_OK;
}
}  // namespace

int spx_fetch_score_rows(spx_engine* e, int plugin, int64_t row_begin, int64_t row_end, uint8_t* out, int64_t out_stride) {
  if (!e || !out) return SPX_ERR_ARG;
  if (plugin < 0 || plugin >= SPX_NUM_PLUGINS || !e->score[plugin].p || !(e->evaluated & (1u << plugin)))
    return fail(e, SPX_ERR_STATE, "plugin has not been evaluated");
  if (row_end > row_begin)
    if (int rc = rows_evaluated(e, plugin, row_begin, row_end)) return rc;
  return fetch_rows(e, static_cast<const uint8_t*>(e->score[plugin].p), e->score_stride[plugin], row_begin, row_end, out, out_stride);
}

int spx_fetch_status_rows(spx_engine* e, int plugin, int64_t row_begin, int64_t row_end, uint8_t* out, int64_t out_stride) {
  if (!e || !out) return SPX_ERR_ARG;
  if (plugin < 0 || plugin >= SPX_NUM_PLUGINS || !e->status[plugin].p || !(e->evaluated & (1u << plugin)))
    return fail(e, SPX_ERR_STATE, "plugin has no evaluated Filter table");
  if (row_end > row_begin)
    if (int rc = rows_evaluated(e, plugin, row_begin, row_end)) return rc;
  return fetch_rows(e, static_cast<const uint8_t*>(e->status[plugin].p), e->row_stride, row_begin, row_end, out, out_stride);
}

int spx_fetch_stats(spx_engine* e, int64_t* reevaluated_cells, int reset) {
  if (!e || !reevaluated_cells) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  std::vector<unsigned long long> h(spx::kStatBytes / sizeof(unsigned long long));
  SPX_HIP(e, hipMemcpyAsync(h.data(), e->d_stats.p, spx::kStatBytes, hipMemcpyDeviceToHost, e->stream));
  if (reset) SPX_HIP(e, hipMemsetAsync(e->d_stats.p, 0, spx::kStatBytes, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  for (int p = 0; p < SPX_NUM_PLUGINS; ++p) {  // the kernels spread their counts over kStatSlots lines per plugin
    unsigned long long sum = 0;
    for (int k = 0; k < spx::kStatSlots; ++k) sum += h[static_cast<size_t>(p * spx::kStatSlots + k) * spx::kStatStride];
    reevaluated_cells[p] = static_cast<int64_t>(sum);
  }
  return SPX_OK;
}

int spx_score_table(spx_engine* e, int plugin, void** dptr, int64_t* row_stride, int64_t* n_rows) {
  if (!e || plugin < 0 || plugin >= SPX_NUM_PLUGINS) return SPX_ERR_ARG;
  if (e->n_nodes <= 0 || e->n_pods <= 0) return fail(e, SPX_ERR_STATE, "shape unknown: upload node and pod tables first");
  int rc = ensure_score_table(e, plugin);
  if (rc) return rc;
  if (dptr) *dptr = e->score[plugin].p;
  if (row_stride) *row_stride = e->score_stride[plugin];
  if (n_rows) *n_rows = e->score_rows[plugin];
  return SPX_OK;
}

int spx_bind_score_table(spx_engine* e, int plugin, void* dptr, int64_t row_stride, int64_t n_rows) {
  if (!e || plugin < 0 || plugin >= SPX_NUM_PLUGINS) return SPX_ERR_ARG;
  DevBuf& b = e->score[plugin];
  if (!dptr) {  // unbind
    if (b.external) b = DevBuf{};
    e->score_rows[plugin] = e->score_stride[plugin] = 0;
    return SPX_OK;
  }
  if (row_stride % spx::kRowAlign != 0 || (reinterpret_cast<uintptr_t>(dptr) % spx::kRowAlign) != 0)
    return fail(e, SPX_ERR_ARG, "bound table must be 16-byte aligned with a 16-byte multiple row stride");
  if (b.p && !b.external) SPX_HIP(e, hipFree(b.p));
  b.p = dptr;
  b.bytes = static_cast<size_t>(row_stride) * static_cast<size_t>(n_rows);
  b.external = true;
  e->score_rows[plugin] = n_rows;
  e->score_stride[plugin] = row_stride;
  return SPX_OK;
}

int spx_bind_status_table(spx_engine* e, int plugin, void* dptr, int64_t row_stride, int64_t n_rows) {
  if (!e || (plugin != SPX_PLUGIN_NRT && plugin != SPX_PLUGIN_NETOVERHEAD)) return SPX_ERR_ARG;
  DevBuf& b = e->status[plugin];
  if (!dptr) {  // unbind
    if (b.external) b = DevBuf{};
    return SPX_OK;
  }
  if (e->row_stride <= 0) return fail(e, SPX_ERR_STATE, "shape unknown: upload the node table first");
  if (row_stride != e->row_stride || (reinterpret_cast<uintptr_t>(dptr) % spx::kRowAlign) != 0)
    return fail(e, SPX_ERR_ARG, "bound status table must be 16-byte aligned and use the engine row stride (spx_score_table reports it)");
  if (b.p && !b.external) SPX_HIP(e, hipFree(b.p));
  b.p = dptr;
  b.bytes = static_cast<size_t>(row_stride) * static_cast<size_t>(n_rows);
  b.external = true;
  return SPX_OK;
}

}  // extern "C"

namespace spx {
EngineView engine_view(spx_engine* e) {
  EngineView v{};
  v.device = e->device;
  v.stream = e->stream;
  v.n_nodes = e->n_nodes;
  v.n_pods = e->n_pods;
  v.row_stride = e->row_stride;
  v.best = e->d_best.p;
  v.best_valid = e->best_valid;
  v.evaluated = e->evaluated;
  v.ev0 = e->ev0;
  v.ev1 = e->ev1;
  v.timed = e->timed;
  return v;
}
}  // namespace spx

namespace {
// spx_decide for a profile with Filter plugins (NRT / NetworkOverhead / a caller mask): the sweeps of `eval_mask` write their
// status and score tables as in spx_eval; Allocatable's feasibility-aware normalisation is folded into the argmax kernel
// (k_decide_masked) over the scoring plugins of `score_mask` — its table is not written, and ALLOCATABLE is left "not evaluated"
// for the fetch functions.  eval_mask differs from score_mask in the sequential commit loop (LVRB's rows are swept once, up
// front).  *done = false: the form does not apply (no Filter in play, wide Allocatable range, weights) — nothing was launched.
int decide_masked(spx_engine* e, uint32_t eval_mask, uint32_t score_mask, int64_t row_begin, int64_t row_end, bool* done) {
  *done = false;
  const uint32_t A = 1u << SPX_PLUGIN_ALLOCATABLE;
  const bool masked = (score_mask & ((1u << SPX_PLUGIN_NRT) | (1u << SPX_PLUGIN_NETOVERHEAD))) || e->ext_mask;
  if (!(score_mask & A) || !(eval_mask & A) || !masked || e->option[SPX_OPT_DECIDE_UNFUSED] || e->n_nodes <= 0 || e->n_pods <= 0 || row_begin < 0 ||
      row_end > e->n_pods || row_begin >= row_end)
    return SPX_OK;
  int rc;
  if ((rc = prepare_alloc(e))) return rc;
  const size_t P = static_cast<size_t>(e->n_pods);
  spx::ProfileArgs pa{};
  pa.n_nodes = e->n_nodes;
  pa.row_stride = e->row_stride;
  pa.row_begin = row_begin;
  pa.row_end = row_end;
  pa.row_ptr = e->row_indirect;
  pa.alloc_rel = static_cast<const uint32_t*>(e->d_alloc_rel.p);
  for (int k = 0; k < SPX_NUM_PLUGINS; ++k) {
    const bool has_score = k <= SPX_PLUGIN_NETOVERHEAD || k == SPX_PLUGIN_LROC || k == SPX_PLUGIN_PEAKS;
    // the tables the sweep below will have written by the time the kernel runs (engine-owned or bound: same row stride)
    if ((score_mask & (1u << k)) && has_score && k != SPX_PLUGIN_ALLOCATABLE) pa.score[k] = reinterpret_cast<const uint8_t*>(uintptr_t{1});
    pa.weight[k] = e->plugin_weight[k];
  }
  if (!e->alloc_compact || !spx::decide_masked_ok(pa)) return SPX_OK;
  if ((rc = ensure(e, e->d_best, P * 20))) return rc;
  SPX_HIP(e, hipEventRecord(e->ev0, e->stream));
  e->hold_ev0 = e->skip_alloc_masked = true;
  rc = spx_eval(e, eval_mask, row_begin, row_end);
  e->hold_ev0 = e->skip_alloc_masked = false;
  if (rc) return rc;
  for (int k = 0; k < SPX_NUM_PLUGINS; ++k)
    if (pa.score[k]) {
      if (!(e->evaluated & (1u << k)) || e->score_stride[k] != e->row_stride)
        return fail(e, SPX_ERR_STATE, "spx_decide: a scoring plugin of the mask has no evaluated table with the engine row stride");
      pa.score[k] = static_cast<const uint8_t*>(e->score[k].p);
    }
  pa.status[0] = (score_mask & (1u << SPX_PLUGIN_NRT)) ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NRT].p) : nullptr;
  pa.status[1] = (score_mask & (1u << SPX_PLUGIN_NETOVERHEAD)) ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NETOVERHEAD].p) : nullptr;
  pa.status[2] = e->ext_mask ? static_cast<const uint8_t*>(e->d_ext_status.p) : nullptr;
  pa.prefilter = (score_mask & (1u << SPX_PLUGIN_CAPACITY)) ? static_cast<const uint8_t*>(e->d_q_status.p) : nullptr;
  pa.best_score = static_cast<int64_t*>(e->d_best.p);
  pa.best_node = reinterpret_cast<int32_t*>(pa.best_score + P);
  pa.best_ties = pa.best_node + P;
  pa.best_feasible = pa.best_ties + P;
  pa.block_per_row = static_cast<int32_t>(e->option[SPX_OPT_ROW_WORKGROUP]);
  spx::launch_decide_masked(pa, e->stream);
  SPX_HIP(e, hipGetLastError());
  SPX_HIP(e, hipEventRecord(e->ev1, e->stream));
  e->timed = true;
  e->best_valid = true;
  *done = true;
  return SPX_OK;
}
}  // namespace

extern "C" {

int spx_eval_best(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end) {
  if (!e) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  if ((plugin_mask & ~e->evaluated) != 0) return fail(e, SPX_ERR_STATE, "spx_eval_best: plugin in the mask has not been evaluated");
  if (e->n_nodes <= 0 || e->n_pods <= 0) return fail(e, SPX_ERR_STATE, "shape unknown");
  if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
  int rc;
  // every table in the sum must cover the rows, and the normalising plugins must have been evaluated under exactly the Filter
  // set this argmax uses (their NormalizeScore ran over the nodes that passed those Filters, as upstream's RunScorePlugins does)
  for (int p = 0; p < SPX_NUM_PLUGINS && row_end > row_begin; ++p) {
    if (!((plugin_mask >> p) & 1u)) continue;
    if ((rc = rows_evaluated(e, p, row_begin, row_end))) return rc;
    const spx_engine::EvalInfo& i = e->eval_info[p];
    const bool ctx_matters = ((kNormalizingPlugins >> p) & 1u) != 0;
    uint32_t want = plugin_mask & kFilterPlugins;
    if (p == SPX_PLUGIN_NETOVERHEAD) want &= ~(1u << SPX_PLUGIN_NETOVERHEAD), want |= i.filters & (1u << SPX_PLUGIN_NETOVERHEAD);  // its own Filter is implied
    if (ctx_matters && (i.filters != want || i.ext_gen != e->ext_gen))
      return fail(e, SPX_ERR_STATE, "spx_eval_best: a normalising plugin (Allocatable / NetworkOverhead / Peaks) was evaluated under a different Filter set "
                                    "or feasibility mask than this argmax uses; evaluate the whole profile in one spx_eval");
  }
  const size_t P = static_cast<size_t>(e->n_pods);
  if ((rc = ensure(e, e->d_best, P * 20))) return rc;
  spx::ProfileArgs pa{};
  pa.n_nodes = e->n_nodes;
  pa.row_stride = e->row_stride;
  pa.row_begin = row_begin;
  pa.row_end = row_end;
  pa.row_ptr = e->row_indirect;
  pa.status[0] = (plugin_mask & (1u << SPX_PLUGIN_NRT)) ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NRT].p) : nullptr;
  pa.status[1] = (plugin_mask & (1u << SPX_PLUGIN_NETOVERHEAD)) ? static_cast<const uint8_t*>(e->status[SPX_PLUGIN_NETOVERHEAD].p) : nullptr;
  pa.status[2] = e->ext_mask ? static_cast<const uint8_t*>(e->d_ext_status.p) : nullptr;
  pa.prefilter = (plugin_mask & (1u << SPX_PLUGIN_CAPACITY)) ? static_cast<const uint8_t*>(e->d_q_status.p) : nullptr;
  for (int k = 0; k < SPX_NUM_PLUGINS; ++k) {
    const bool has_score = k <= SPX_PLUGIN_NETOVERHEAD || k == SPX_PLUGIN_LROC || k == SPX_PLUGIN_PEAKS;
    if ((plugin_mask & (1u << k)) && has_score) {
      if (e->score_stride[k] != e->row_stride) return fail(e, SPX_ERR_STATE, "score table stride differs from the engine row stride");
      pa.score[k] = static_cast<const uint8_t*>(e->score[k].p);
    }
    pa.weight[k] = e->plugin_weight[k];
  }
  pa.best_score = static_cast<int64_t*>(e->d_best.p);
  pa.best_node = reinterpret_cast<int32_t*>(pa.best_score + P);
  pa.best_ties = pa.best_node + P;
  pa.best_feasible = pa.best_ties + P;
  spx::launch_best(pa, e->stream);
  SPX_HIP(e, hipGetLastError());
  e->best_valid = true;
  return SPX_OK;
}

int spx_decide(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end) {
  if (!e) return SPX_ERR_ARG;
  SPX_HIP(e, hipSetDevice(e->device));
  const uint32_t A = 1u << SPX_PLUGIN_ALLOCATABLE, T = 1u << SPX_PLUGIN_TLP;
  // Score-only plugins whose tables the fused sweep can fold in: no Filter, and their bytes are final once evaluated
  // (Peaks normalises inside its own sweep)
  const int kExtra[3] = {SPX_PLUGIN_LVRB, SPX_PLUGIN_LROC, SPX_PLUGIN_PEAKS};
  uint32_t extra_mask = 0;
  int64_t w_sum = 0;
  bool w_ok = true;
  for (int p = 0; p < SPX_NUM_PLUGINS; ++p)
    if (plugin_mask & (1u << p)) w_sum += e->plugin_weight[p], w_ok &= e->plugin_weight[p] >= 0;
  for (int x = 0; x < 3; ++x) extra_mask |= plugin_mask & (1u << kExtra[x]);
  const int64_t wa = e->plugin_weight[SPX_PLUGIN_ALLOCATABLE], wt = e->plugin_weight[SPX_PLUGIN_TLP];
  const bool fusable = (plugin_mask & T) && !(plugin_mask & ~(A | T | extra_mask)) && !e->ext_mask && e->tri_nodes && e->tri_pods &&
                       e->tlp.target_utilization >= 1 && e->tlp.target_utilization <= 99 && !(launch_opts(e) & spx::kOptTrimaranExact) &&
                       !e->option[SPX_OPT_DECIDE_UNFUSED] && w_ok && w_sum <= 8000;  // (every weighted total in 21 bits: the sweep's 32-bit key)
  int rc;
  if (!fusable) {  // a profile with Filter plugins: see decide_masked
    bool done = false;
    if ((rc = decide_masked(e, plugin_mask, plugin_mask, row_begin, row_end, &done)) || done) return rc;
  }
  if (!fusable) {
    if ((rc = spx_eval(e, plugin_mask, row_begin, row_end))) return rc;
    return spx_eval_best(e, plugin_mask, row_begin, row_end);
  }
  if (e->n_nodes <= 0 || e->n_pods <= 0) return fail(e, SPX_ERR_STATE, "shape unknown");
  if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
  const bool use_alloc = plugin_mask & A;
  if (use_alloc && (rc = prepare_alloc(e))) return rc;
  const size_t P = static_cast<size_t>(e->n_pods);
  if ((rc = ensure(e, e->d_best, P * 20))) return rc;
  if ((rc = ensure(e, e->d_tlp_fast, static_cast<size_t>(spx::round_up(e->row_stride, 1024)) * 4 * sizeof(float)))) return rc;
  if ((rc = ensure(e, e->d_decide, spx::decide_scratch_bytes(e->row_stride, row_end - row_begin)))) return rc;
  spx::DecideLaunch d{};
  fill_trimaran(e, d.t);
  d.t.row_begin = row_begin;
  d.t.row_end = row_end;
  d.t.tlp_fast = static_cast<float*>(e->d_tlp_fast.p);
  if (!e->d_tlp_amb.p) e->tlp_amb_built = false;
  if ((rc = ensure(e, e->d_tlp_amb, static_cast<size_t>(spx::kTlpAmbSize) * 4))) return rc;
  d.t.tlp_amb = static_cast<uint32_t*>(e->d_tlp_amb.p);
  d.t.tlp_amb_size = spx::kTlpAmbSize;
  d.t.tlp_amb_built = &e->tlp_amb_built;
  d.t.tlp_amb_geom = e->tlp_amb_geom;
  d.use_alloc = use_alloc;
  d.w_alloc = static_cast<int32_t>(use_alloc ? wa : 0);
  d.w_tlp = static_cast<int32_t>(wt);
  d.scratch = e->d_decide.p;
  d.best_score = static_cast<int64_t*>(e->d_best.p);
  d.best_node = reinterpret_cast<int32_t*>(d.best_score + P);
  d.best_ties = d.best_node + P;
  d.best_feasible = d.best_ties + P;
  SPX_HIP(e, hipEventRecord(e->ev0, e->stream));
  if (extra_mask) {
    e->hold_ev0 = true;
    rc = spx_eval(e, extra_mask, row_begin, row_end);
    e->hold_ev0 = false;
    if (rc) return rc;
    for (int x = 0; x < 3; ++x)
      if (extra_mask & (1u << kExtra[x])) {
        d.w_extra[d.n_extra] = static_cast<int32_t>(e->plugin_weight[kExtra[x]]);
        d.extra[d.n_extra++] = static_cast<const uint8_t*>(e->score[kExtra[x]].p);
      }
  }
  spx::launch_decide_trimaran(d, e->stream);
  SPX_HIP(e, hipGetLastError());
  SPX_HIP(e, hipEventRecord(e->ev1, e->stream));
  e->timed = true;
  e->best_valid = true;
  return SPX_OK;
}

int spx_fetch_best(spx_engine* e, int64_t row_begin, int64_t row_end, int32_t* node_idx, int64_t* weighted_score, int32_t* n_ties,
                   int32_t* n_feasible) {
  if (!e || !node_idx || !weighted_score) return SPX_ERR_ARG;
  if (!e->best_valid) return fail(e, SPX_ERR_STATE, "spx_eval_best has not run since the last spx_eval");
  if (row_begin < 0 || row_end > e->n_pods || row_begin > row_end) return fail(e, SPX_ERR_ARG, "row range out of bounds");
  const size_t n = static_cast<size_t>(row_end - row_begin);
  const size_t P = static_cast<size_t>(e->n_pods);
  // one D2H of the whole decision block into pinned memory, then scatter into the caller's arrays
  if (e->h_best_bytes < P * 20) {
    if (e->h_best) SPX_HIP(e, hipHostFree(e->h_best));
    e->h_best = nullptr;
    e->h_best_bytes = 0;
    SPX_HIP(e, hipHostMalloc(&e->h_best, P * 20, hipHostMallocDefault));
    e->h_best_bytes = P * 20;
  }
  SPX_HIP(e, hipMemcpyAsync(e->h_best, e->d_best.p, P * 20, hipMemcpyDeviceToHost, e->stream));
  SPX_HIP(e, hipStreamSynchronize(e->stream));
  const int64_t* hs = static_cast<const int64_t*>(e->h_best);
  const int32_t* hn = reinterpret_cast<const int32_t*>(hs + P);
  std::memcpy(weighted_score, hs + row_begin, n * 8);
  std::memcpy(node_idx, hn + row_begin, n * 4);
  if (n_ties) std::memcpy(n_ties, hn + P + row_begin, n * 4);
  if (n_feasible) std::memcpy(n_feasible, hn + 2 * P + row_begin, n * 4);
  return SPX_OK;
}

}  // extern "C"
