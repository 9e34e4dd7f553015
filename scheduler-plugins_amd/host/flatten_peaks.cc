// flatten_peaks.cc — object tables -> SoA columns for trimaran Peaks.  Host-side product code (once per snapshot).
//
// What is hoisted out of the per-(pod,node) path, and where the reference does it per call:
//   node : GetNodeMetrics + the scan for the first CPU metric with operator AVG or Latest     peaks.go:108-131
//          node.Status.Capacity cpu in millicores                                             peaks.go:132
//          getPowerModel(nodeName, NodePowerModel)                                            peaks.go:190-196
//   pod  : resource.GetResourceRequestQuantity(pod, cpu).MilliValue()                         peaks.go:114-115
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

}  // namespace

extern "C" int spx_flatten_peaks_nodes(const spx_node_objects* nodes, const spx_metrics_objects* metrics, const spx_power_model_objects* models,
                                       int64_t* cap_cpu_milli, double* cpu_util, uint8_t* valid, double* k1, double* k2) {
  if (!nodes || !cap_cpu_milli || !cpu_util || !valid || !k1 || !k2) return SPX_ERR_ARG;
  if (models && (!models->k1 || !models->k2)) return SPX_ERR_ARG;
  const bool have_map = metrics != nullptr && !metrics->map_is_nil;
  spx_host::parallel_rows(nodes->n_nodes, [&](int64_t row0, int64_t row1) {
    for (int64_t n = row0; n < row1; ++n) {
      cap_cpu_milli[n] = nodes->cap_cpu_milli[n];
      k1[n] = models ? models->k1[n] : 0.0;
      k2[n] = models ? models->k2[n] : 0.0;
      cpu_util[n] = 0.0;
      valid[n] = 0;
      // Collector.GetNodeMetrics returns nil metrics for a nil map, an absent node or a nil slice (collector.go:110-123)
      if (!have_map || !metrics->node_present[n] || (metrics->node_metrics_nil && metrics->node_metrics_nil[n])) continue;
      for (int32_t i = metrics->m_ptr[n]; i < metrics->m_ptr[n + 1]; ++i) {
        if (metrics->m_type[i] == SPX_MT_CPU && (metrics->m_op[i] == SPX_MO_AVG || metrics->m_op[i] == SPX_MO_LATEST)) {
          cpu_util[n] = metrics->m_value[i];
          valid[n] = 1;
          break;  // the first one wins here (TLP keeps the last, targetloadpacking.go:133-140)
        }
      }
    }
  });
  return SPX_OK;
}

extern "C" int spx_flatten_peaks_pods(const spx_pod_objects* pods, int64_t* cpu_milli) {
  if (!pods || !cpu_milli) return SPX_ERR_ARG;
  spx_host::parallel_rows(pods->n_pods, [&](int64_t row0, int64_t row1) {
    for (int64_t i = row0; i < row1; ++i) {
      int64_t total = 0, q;
      for (int32_t c = pods->ctr_ptr[i]; c < pods->ctr_ptr[i + 1]; ++c)  // app containers add up
        if (pods->ctr_kind[c] == SPX_CTR_APP && find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q)) total += q;
      for (int32_t c = pods->ctr_ptr[i]; c < pods->ctr_ptr[i + 1]; ++c)  // any init container can raise the total
        if (pods->ctr_kind[c] != SPX_CTR_APP && find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q) && total < q) total = q;
      // the overhead counts only on top of a non-zero total
      if (pods->ovh_ptr != nullptr && total != 0 && find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[i], pods->ovh_ptr[i + 1], SPX_RES_CPU, &q)) total += q;
      cpu_milli[i] = total;
    }
  });
  return SPX_OK;
}
