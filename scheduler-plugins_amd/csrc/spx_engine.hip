// spx_engine.hip — the C-ABI engine of libspx.so: lifecycle, options and parameters, the evaluation (spx_eval, spx_decide, spx_eval_best)
// and the fetch calls.  Tables and deltas into HBM: spx_uploads.hip; the one-pod-at-a-time loops: spx_commit.hip; shared state and
// helpers: spx_engine.h.  See include/spx.h for the contract.
//
// There is deliberately no CPU fallback: spx_create() fails with SPX_ERR_NOGPU when no HIP device
// is usable, and every compute entry point needs an engine.
#include "spx_engine.h"

thread_local std::string g_create_error;
thread_local std::string tl_err;
thread_local const spx_engine* tl_err_engine = nullptr;

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
                    &e->d_nrt_uniq, &e->d_nrt_dups, &e->d_pk_uniq, &e->d_pk_dups, &e->d_delta, &e->d_nrt_lnrec, &e->d_net_pair_node2, &e->d_net_pair_max2, &e->d_nrt_rk, &e->d_nrt_rk_off, &e->d_nrt_rk_first, &e->d_nrt_fz, &e->d_nrt_wsort, &e->d_nrt_wrank};
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
    // (BalancedAllocation, round 6b: the walk carries its float32 Score too — no packed-Score preconditions, its own list of cells for float64)
    const bool fz_balanced = na.strategy == SPX_NRT_BALANCED_ALLOCATION && na.fast && !(na.opts & spx::kOptNrtGeneric) && !e->row_indirect;
    bool fused = e->option[SPX_OPT_NRT_FUSED] && e->option[SPX_OPT_NRT_RANK_FILTER] && (na.pk_mode || fz_balanced) && !(na.opts & spx::kOptNrtSingleLaunch) &&
                 row_begin == 0 && row_end == e->n_pods;
    fused = fused && e->option[SPX_OPT_NRT_RANK_NARROW];  // (its only count layout)
    for (int i = 0; fused && !fz_balanced && i < e->nrt_n_res; ++i) fused = e->nrt_slot_weight[i] == 0 || e->nrt_slot_weight[i] == 1;
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
    if (na.rk_stream && na.rk_all_narrow && e->nrt_n_res <= 4) {  // the walk's block start reads the windows' sorted quantities: (re)built when the zone tables changed
      size_t rank_bytes = 0;
      const size_t sort_bytes = spx::nrt_window_sort_bytes(e->n_nodes, &rank_bytes);
      if (!e->d_nrt_wsort.p || e->d_nrt_wsort.bytes < sort_bytes || !e->d_nrt_wrank.p || e->d_nrt_wrank.bytes < rank_bytes) e->nrt_wsort_built = false;
      if ((rc = ensure(e, e->d_nrt_wsort, sort_bytes)) || (rc = ensure(e, e->d_nrt_wrank, rank_bytes))) return rc;
      if (!e->nrt_wsort_built) {
        spx::launch_nrt_window_sort(na, static_cast<double*>(e->d_nrt_wsort.p), static_cast<uint16_t*>(e->d_nrt_wrank.p), e->stream);
        e->nrt_wsort_built = true;
      }
      na.wsort = static_cast<const double*>(e->d_nrt_wsort.p);
      na.wrank = static_cast<const uint16_t*>(e->d_nrt_wrank.p);
    }
    const bool ran_fused = spx::launch_nrt(na, e->stream);
    e->last_nrt_filter = ran_fused ? 3 : (na.rk_stream ? 2 : 1);
    if (classes)
      spx::launch_rows_expand(static_cast<const int32_t*>(e->d_nrt_dups.p), static_cast<const int32_t*>(e->d_nrt_dups.p) + 2 * e->nrt_n_dups, e->nrt_n_tasks, na.out_status, na.out_score,
                              e->row_stride, e->stream);
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
      spx::launch_rows_expand(static_cast<const int32_t*>(e->d_pk_dups.p), static_cast<const int32_t*>(e->d_pk_dups.p) + 2 * e->pk_n_dups, e->pk_n_tasks, ka.out_score, nullptr,
                              e->row_stride, e->stream);
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
  if (plugin == SPX_PLUGIN_LROC) return lroc_f32_ok(e) ? 1 : 0;
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

extern "C" {
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
}  // extern "C" (decide_masked: hidden, shared with spx_commit.hip)

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

