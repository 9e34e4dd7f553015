// spx_uploads.hip — tables into HBM: spx_upload_* (SoA columns), spx_update_* (snapshot deltas), the derived host-built streams of the
// NRT sweeps (host/nrt_streams.cc), and spx_load_* / spx_load_profile (object tables -> SoA -> device inside the library).
// Engine state and shared helpers: spx_engine.h.
#include "spx_engine.h"

extern "C" {

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
  e->lv_alloc_f32 = all_below_2p47(t->lv_alloc_cpu_milli, n) && all_below_2p47(t->lv_alloc_mem, n);
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
  e->lv_alloc_f32 = e->lv_alloc_f32 && all_below_2p47(t->lv_alloc_cpu_milli, m) && all_below_2p47(t->lv_alloc_mem, m);
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
  e->nrt_pk_tab_built = e->nrt_wsort_built = false;  // zone capacities changed
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
  e->lroc_nodes_f32 = all_below_2p47(t->req_cpu_milli, n) && all_below_2p47(t->req_mem, n) && all_below_2p47(t->lim_cpu_milli, n) && all_below_2p47(t->lim_mem, n) &&
                      none_below(t->lim_cpu_milli, t->req_cpu_milli, n) && none_below(t->lim_mem, t->req_mem, n);
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
  e->lroc_pods_f32 = e->lroc_pods_exact && all_below_2p47(t->req_cpu_milli, p) && all_below_2p47(t->req_mem, p) && all_below_2p47(t->lim_cpu_milli, p) &&
                     all_below_2p47(t->lim_mem, p) && none_below(t->lim_cpu_milli, t->req_cpu_milli, p) && none_below(t->lim_mem, t->req_mem, p);
  if (e->lroc_pods_f32) {
    // the float32 sweep's pod records, 32 bytes each (one scalar load): the limits' high float32 parts (cpu, memory), their low parts (exact below
    // 2^47), limit - request as the float32 it is used as, and a marker for the pod without requests and limits
    std::vector<float> f(8 * p);
    for (size_t i = 0; i < p; ++i) {
      float* r = &f[8 * i];
      const bool none = t->req_cpu_milli[i] == 0 && t->req_mem[i] == 0 && t->lim_cpu_milli[i] == 0 && t->lim_mem[i] == 0;
      const double lc = static_cast<double>(t->lim_cpu_milli[i]), lm = static_cast<double>(t->lim_mem[i]);
      r[0] = static_cast<float>(lc), r[2] = static_cast<float>(lc - static_cast<double>(r[0]));
      r[1] = static_cast<float>(lm), r[3] = static_cast<float>(lm - static_cast<double>(r[1]));
      r[4] = static_cast<float>(static_cast<double>(t->lim_cpu_milli[i] - t->req_cpu_milli[i]));
      r[5] = static_cast<float>(static_cast<double>(t->lim_mem[i] - t->req_mem[i]));
      r[6] = none ? 1.0f : 0.0f, r[7] = 0.0f;
    }
    if ((rc = upload(e, e->d_lroc_podf, f.data(), f.size() * sizeof(float)))) return rc;
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
      const int64_t n_dups = static_cast<int64_t>(dups.size() / 2), n_tasks = expand_tasks(dups, static_cast<int64_t>(p));
      if ((rc = upload(e, e->d_pk_uniq, uniq.data(), uniq.size() * sizeof(int32_t)))) return rc;
      if ((rc = upload(e, e->d_pk_dups, dups.data(), dups.size() * sizeof(int32_t)))) return rc;
      SPX_HIP(e, hipStreamSynchronize(e->stream));  // the vectors go out of scope
      e->pk_n_uniq = static_cast<int64_t>(uniq.size());
      e->pk_n_dups = n_dups;
      e->pk_n_tasks = n_tasks;
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
    e->nrt_pk_tab_built = e->nrt_wsort_built = false;
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
    e->nrt_pk_tab_built = e->nrt_wsort_built = false;
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

int build_ln_tab(spx_engine* e) {
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
        const int64_t n_dups = static_cast<int64_t>(dups.size() / 2), n_tasks = expand_tasks(dups, static_cast<int64_t>(p));
        if ((rc = upload(e, e->d_nrt_uniq, uniq.data(), uniq.size() * sizeof(int32_t)))) return rc;
        if ((rc = upload(e, e->d_nrt_dups, dups.data(), dups.size() * sizeof(int32_t)))) return rc;
        SPX_HIP(e, hipStreamSynchronize(e->stream));
        e->nrt_n_uniq = static_cast<int64_t>(uniq.size());
        e->nrt_n_dups = n_dups;
        e->nrt_n_tasks = n_tasks;
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

}  // extern "C"
