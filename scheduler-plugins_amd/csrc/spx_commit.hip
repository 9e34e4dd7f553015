// spx_commit.hip — the one-pod-at-a-time loops: spx_upload_net_commit and spx_commit_sequential with its three forms
// (one-workgroup chain, per-pod launches replayed from a graph, cooperative persistent kernel).  Engine state and shared helpers: spx_engine.h.
#include "spx_engine.h"

extern "C" {

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
    explicit LoopFlag(spx_engine* x) : e(x) { e->in_commit_loop = true, e->tlp_amb_built = false, e->nrt_pk_tab_built = e->nrt_wsort_built = false; }  // (k_commit_apply advances d_tlp_missing and the zone tables)
    ~LoopFlag() { e->in_commit_loop = false, e->tlp_amb_built = false, e->nrt_pk_tab_built = e->nrt_wsort_built = false; }
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

}  // extern "C"
