// flatten_trimaran.cc — object tables -> SoA columns for Allocatable / TargetLoadPacking /
// LoadVariationRiskBalancing.  Host-side product code (runs once per snapshot, O(N + P)), the
// C++ twin of what the Go shim does before spx_upload_*.
//
// What is hoisted out of the per-(pod,node) path, and where the reference does it per call:
//   Allocatable : NodeInfo.Allocatable lookup per weighted resource   resource_allocation.go:79-100
//   TLP node    : metric selection (last CPU AVG/Latest wins)          targetloadpacking.go:131-145
//                 Capacity cpu millis                                  targetloadpacking.go:146
//                 missing utilisation from ScheduledPodsCache          targetloadpacking.go:151-168
//   TLP pod     : Σ PredictUtilisation(containers) + overhead          targetloadpacking.go:122-129,198-205
//   LVRB node   : GetResourceData for cpu / memory                     resourcestats.go:89-107
//   LVRB pod    : GetResourceRequested                                 resourcestats.go:110-146
#include <cmath>
#include <cstdint>

#include "../../include/spx.h"
#include "parallel.hpp"

namespace {

inline bool find_qty(const int32_t* res, const int64_t* qty, int32_t lo, int32_t hi, int32_t want, int64_t* out) {
  for (int32_t i = lo; i < hi; ++i) {
    if (res[i] == want) {
      *out = qty[i];
      return true;
    }
  }
  return false;
}

// Collector.GetNodeMetrics: false when the reference would see `metrics == nil`.
inline bool node_metrics(const spx_metrics_objects* m, int64_t n, int32_t* lo, int32_t* hi) {
  if (m == nullptr || m->map_is_nil) return false;
  if (!m->node_present[n]) return false;
  if (m->node_metrics_nil != nullptr && m->node_metrics_nil[n]) return false;
  *lo = m->m_ptr[n];
  *hi = m->m_ptr[n + 1];
  return true;
}

int64_t predict_utilisation(const spx_pod_objects* pods, int32_t c, const spx_tlp_params* p) {
  int64_t q;
  if (find_qty(pods->lim_res, pods->lim_qty, pods->lim_ptr[c], pods->lim_ptr[c + 1], SPX_RES_CPU, &q)) return q;
  if (find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q))
    return static_cast<int64_t>(std::round(static_cast<double>(q) * p->requests_multiplier));
  return p->default_requests_milli;
}

int64_t tlp_pod_milli(const spx_pod_objects* pods, int64_t i, const spx_tlp_params* p) {
  int64_t cur = 0;
  for (int32_t c = pods->ctr_ptr[i]; c < pods->ctr_ptr[i + 1]; ++c)
    if (pods->ctr_kind[c] == SPX_CTR_APP) cur += predict_utilisation(pods, c, p);
  int64_t ovh;
  if (pods->ovh_ptr != nullptr &&
      find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[i], pods->ovh_ptr[i + 1], SPX_RES_CPU, &ovh))
    cur += ovh;
  return cur;
}

}  // namespace

extern "C" int spx_flatten_alloc_nodes(const spx_node_objects* nodes, const spx_resource_classes* rc,
                                       const spx_allocatable_params* p, int64_t* alloc_out) {
  if (!nodes || !p || !alloc_out) return SPX_ERR_ARG;
  const int64_t n = nodes->n_nodes;
  for (int32_t r = 0; r < p->n_res; ++r) {
    int64_t* col = alloc_out + static_cast<int64_t>(r) * n;
    const int32_t res = p->res[r];
    for (int64_t i = 0; i < n; ++i) {
      int64_t v = 0;
      if (res == SPX_RES_CPU) {
        v = nodes->alloc_cpu_milli[i];
      } else if (res == SPX_RES_MEMORY) {
        v = nodes->alloc_mem[i];
      } else if (res == SPX_RES_EPHEMERAL) {
        v = nodes->alloc_eph[i];
      } else if (rc != nullptr && res >= 0 && res < rc->n_res && (rc->flags[res] & SPX_RC_SCALAR)) {
        find_qty(nodes->scalar_res, nodes->scalar_qty, nodes->scalar_ptr[i], nodes->scalar_ptr[i + 1], res, &v);
      }  // any other name: "not considered for node score calculation" -> 0
      col[i] = v;
    }
  }
  return SPX_OK;
}

namespace {
struct TrimaranNodeCols {
  int64_t* cap_cpu_milli;
  double* tlp_cpu_util;
  int64_t* tlp_missing_milli;
  uint8_t* tlp_valid;
  int64_t* lv_alloc_cpu_milli;
  int64_t* lv_alloc_mem;
  double* lv_cpu_avg;
  double* lv_cpu_std;
  double* lv_mem_avg;
  double* lv_mem_std;
  uint8_t* lv_flags;
};

// node i's trimaran columns into row j of the output (j == i for the whole table, the position in the index list for a delta)
inline void trimaran_node_row(const spx_node_objects* nodes, const spx_metrics_objects* metrics, const spx_assigned_objects* assigned,
                              const spx_tlp_params* tlp, int64_t i, int64_t j, const TrimaranNodeCols& c) {
    int32_t lo = 0, hi = 0;
    const bool have = node_metrics(metrics, i, &lo, &hi);
    // ---- TLP columns
    double util = 0;
    bool cpu_found = false;
    if (have) {
      for (int32_t k = lo; k < hi; ++k) {  // no break: the last CPU AVG/Latest metric wins
        if (metrics->m_type[k] == SPX_MT_CPU && (metrics->m_op[k] == SPX_MO_AVG || metrics->m_op[k] == SPX_MO_LATEST)) {
          util = metrics->m_value[k];
          cpu_found = true;
        }
      }
    }
    int64_t missing = 0;
    if (have && cpu_found && assigned != nullptr && assigned->e_ptr != nullptr) {
      const int64_t end = metrics->window_end;
      for (int32_t e = assigned->e_ptr[i]; e < assigned->e_ptr[i + 1]; ++e) {
        const int64_t ts = assigned->e_ts_unix[e];
        if (ts > end || (ts <= end && (end - ts) < 60)) missing += tlp_pod_milli(assigned->pods, assigned->e_pod[e], tlp);
      }
    }
    if (c.cap_cpu_milli) c.cap_cpu_milli[j] = nodes->cap_cpu_milli[i];
    if (c.tlp_cpu_util) c.tlp_cpu_util[j] = util;
    if (c.tlp_missing_milli) c.tlp_missing_milli[j] = missing;
    if (c.tlp_valid) c.tlp_valid[j] = (have && cpu_found) ? 1 : 0;
    // ---- LVRB columns (GetResourceData: AVG wins over Latest/"" regardless of order)
    double avg[2] = {0, 0}, sd[2] = {0, 0};
    bool valid[2] = {false, false};
    if (have) {
      for (int t = 0; t < 2; ++t) {
        bool avg_found = false;
        for (int32_t k = lo; k < hi; ++k) {
          if (metrics->m_type[k] != t) continue;
          const uint8_t op = metrics->m_op[k];
          if (op == SPX_MO_AVG) {
            avg[t] = metrics->m_value[k];
            avg_found = true;
          } else if (op == SPX_MO_STD) {
            sd[t] = metrics->m_value[k];
          } else if ((op == SPX_MO_EMPTY || op == SPX_MO_LATEST) && !avg_found) {
            avg[t] = metrics->m_value[k];
          }
          valid[t] = true;
        }
      }
    }
    if (c.lv_alloc_cpu_milli) c.lv_alloc_cpu_milli[j] = nodes->alloc_cpu_milli[i];
    if (c.lv_alloc_mem) c.lv_alloc_mem[j] = nodes->alloc_mem[i];
    if (c.lv_cpu_avg) c.lv_cpu_avg[j] = avg[0];
    if (c.lv_cpu_std) c.lv_cpu_std[j] = sd[0];
    if (c.lv_mem_avg) c.lv_mem_avg[j] = avg[1];
    if (c.lv_mem_std) c.lv_mem_std[j] = sd[1];
    if (c.lv_flags)
      c.lv_flags[j] = static_cast<uint8_t>((have ? SPX_LV_HAS_METRICS : 0) | (valid[0] ? SPX_LV_CPU_VALID : 0) |
                                         (valid[1] ? SPX_LV_MEM_VALID : 0));
}
}  // namespace

extern "C" int spx_flatten_trimaran_nodes(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                                          const spx_assigned_objects* assigned, const spx_tlp_params* tlp,
                                          int64_t* cap_cpu_milli, double* tlp_cpu_util, int64_t* tlp_missing_milli,
                                          uint8_t* tlp_valid, int64_t* lv_alloc_cpu_milli, int64_t* lv_alloc_mem,
                                          double* lv_cpu_avg, double* lv_cpu_std, double* lv_mem_avg,
                                          double* lv_mem_std, uint8_t* lv_flags) {
  if (!nodes || !tlp) return SPX_ERR_ARG;
  const TrimaranNodeCols c{cap_cpu_milli, tlp_cpu_util, tlp_missing_milli, tlp_valid, lv_alloc_cpu_milli, lv_alloc_mem,
                           lv_cpu_avg, lv_cpu_std, lv_mem_avg, lv_mem_std, lv_flags};
  for (int64_t i = 0; i < nodes->n_nodes; ++i) trimaran_node_row(nodes, metrics, assigned, tlp, i, i, c);
  return SPX_OK;
}

// the same columns for the listed nodes only (row j = node idx[j]): the input of spx_update_trimaran_nodes.  A cycle's delta is
// a few hundred nodes (collector.go:139-150 refreshes metrics per node, handler.go:131-139 adds an assigned pod to one node):
// walking the whole node list for them cost 0.46 ms at 10k nodes, the listed rows cost microseconds.
extern "C" int spx_flatten_trimaran_node_rows(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                                              const spx_assigned_objects* assigned, const spx_tlp_params* tlp, const int64_t* idx,
                                              int64_t n_rows, int64_t* cap_cpu_milli, double* tlp_cpu_util,
                                              int64_t* tlp_missing_milli, uint8_t* tlp_valid, int64_t* lv_alloc_cpu_milli,
                                              int64_t* lv_alloc_mem, double* lv_cpu_avg, double* lv_cpu_std, double* lv_mem_avg,
                                              double* lv_mem_std, uint8_t* lv_flags) {
  if (!nodes || !tlp || (n_rows > 0 && !idx) || n_rows < 0) return SPX_ERR_ARG;
  const TrimaranNodeCols c{cap_cpu_milli, tlp_cpu_util, tlp_missing_milli, tlp_valid, lv_alloc_cpu_milli, lv_alloc_mem,
                           lv_cpu_avg, lv_cpu_std, lv_mem_avg, lv_mem_std, lv_flags};
  for (int64_t j = 0; j < n_rows; ++j) {
    if (idx[j] < 0 || idx[j] >= nodes->n_nodes) return SPX_ERR_ARG;
    trimaran_node_row(nodes, metrics, assigned, tlp, idx[j], j, c);
  }
  return SPX_OK;
}

extern "C" int spx_flatten_trimaran_pods(const spx_pod_objects* pods, const spx_tlp_params* tlp,
                                         int64_t* tlp_pod_milli_out, int64_t* lv_req_cpu_milli, int64_t* lv_req_mem) {
  if (!pods || !tlp) return SPX_ERR_ARG;
  spx_host::parallel_rows(pods->n_pods, [&](int64_t row0, int64_t row1) {
  for (int64_t i = row0; i < row1; ++i) {
    if (tlp_pod_milli_out) tlp_pod_milli_out[i] = tlp_pod_milli(pods, i, tlp);
    int64_t cpu = 0, mem = 0, q;
    for (int32_t c = pods->ctr_ptr[i]; c < pods->ctr_ptr[i + 1]; ++c) {
      if (pods->ctr_kind[c] != SPX_CTR_APP) continue;
      if (find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q)) cpu += q;
      if (find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_MEMORY, &q)) mem += q;
    }
    for (int32_t c = pods->ctr_ptr[i]; c < pods->ctr_ptr[i + 1]; ++c) {
      if (pods->ctr_kind[c] == SPX_CTR_APP) continue;
      if (find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q) && q > cpu) cpu = q;
      if (find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_MEMORY, &q) && q > mem) mem = q;
    }
    if (pods->ovh_ptr != nullptr) {
      if (find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[i], pods->ovh_ptr[i + 1], SPX_RES_CPU, &q)) cpu += q;
      if (find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[i], pods->ovh_ptr[i + 1], SPX_RES_MEMORY, &q)) mem += q;
    }
    if (lv_req_cpu_milli) lv_req_cpu_milli[i] = cpu;
    if (lv_req_mem) lv_req_mem[i] = mem;
  }
  }, 4096);
  return SPX_OK;
}
