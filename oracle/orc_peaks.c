/*
 * orc_peaks.c — restatement of trimaran Peaks (TEST INFRASTRUCTURE, see spx_oracle.h).
 *
 * Follows pkg/trimaran/peaks/peaks.go:103-144 (Score), :150-166 (NormalizeScore), :168-184 (getMinMaxScores),
 * :186-188 (getPowerJumpForUtilisation), :190-196 (getPowerModel).
 *
 * Third-party piece: peaks.go:114 calls resource.GetResourceRequestQuantity of k8s.io/kubernetes v1.35.7
 * (pkg/api/v1/resource/helpers.go, go.mod:28), not vendored under /root/reference.  Its published behaviour: the sum of
 * the app containers' requests of the resource, raised to the largest single init container request, plus the pod
 * overhead of that resource only when the total so far is non-zero.  peaks_test.go pins two corners of it: a pod whose
 * only cpu figure is an overhead scores as requesting nothing (:300-326, "Pod with Overhead"), and limits are not
 * requests (:327-353).
 *
 * math.Exp: Go's amd64 assembly vs libm differ in the last digit at most; the raw score is that value times 1e15, so
 * raw scores agree to ~1e-15 relative and normalized scores to +-1 (the parity tolerance).
 */
#include <math.h>
#include <stdint.h>

#include "spx_oracle.h"

/* resource.GetResourceRequestQuantity(pod, v1.ResourceCPU).MilliValue() */
int64_t orc_get_resource_request_quantity_cpu_milli(const spx_pod_objects* pods, int64_t pod) {
  int64_t total = 0, q;
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c)
    if (pods->ctr_kind[c] == SPX_CTR_APP &&
        orc_find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q))
      total += q;
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c)
    if (pods->ctr_kind[c] != SPX_CTR_APP &&
        orc_find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q) && total < q)
      total = q;
  if (pods->ovh_ptr && total != 0 &&
      orc_find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[pod], pods->ovh_ptr[pod + 1], SPX_RES_CPU, &q))
    total += q;
  return total;
}

/* Peaks.Score peaks.go:103-144 */
int64_t orc_peaks_score(const spx_node_objects* nodes, const spx_metrics_objects* metrics, const spx_power_model_objects* models,
                        const spx_pod_objects* pods, int64_t pod, int64_t node) {
  int32_t lo, hi;
  if (!orc_node_metrics(metrics, node, &lo, &hi)) return 0; /* :108-112 */
  int64_t cur_pod_cpu = orc_get_resource_request_quantity_cpu_milli(pods, pod); /* :114-115 */
  double util_percent = 0;
  int found = 0;
  for (int32_t i = lo; i < hi; ++i) { /* :119-127: the first CPU metric that is AVG or Latest */
    if (metrics->m_type[i] == SPX_MT_CPU && (metrics->m_op[i] == SPX_MO_AVG || metrics->m_op[i] == SPX_MO_LATEST)) {
      util_percent = metrics->m_value[i];
      found = 1;
      break;
    }
  }
  if (!found) return 0; /* :128-131 */
  double cap_millis = (double)nodes->cap_cpu_milli[node]; /* :132 node.Status.Capacity */
  double util_millis = (util_percent / 100) * cap_millis;
  double predicted = 0;
  if (cap_millis != 0) predicted = 100 * (util_millis + (double)cur_pod_cpu) / cap_millis; /* :135-138 */
  if (predicted > 100) return 0;                                                            /* :139-140 */
  double k1 = models ? models->k1[node] : 0, k2 = models ? models->k2[node] : 0;            /* :190-196 */
  double jump = k1 * (exp(k2 * predicted) - exp(k2 * util_percent));                        /* :186-188 */
  return (int64_t)(jump * 1e15);                                                             /* :143, math.Pow(10, 15) */
}

/* Peaks.NormalizeScore peaks.go:150-166 */
void orc_peaks_normalize(int64_t* scores, int64_t n) {
  int64_t max = INT64_MIN, min = INT64_MAX;
  for (int64_t i = 0; i < n; ++i) {
    if (scores[i] > max) max = scores[i];
    if (scores[i] < min) min = scores[i];
  }
  if (min == 0 && max == 0) return;
  for (int64_t i = 0; i < n; ++i) {
    double norm;
    if (max != min)
      norm = 100.0 * (double)(scores[i] - min) / (double)(max - min);
    else
      norm = (double)(scores[i] - min);
    scores[i] = 100 - (int64_t)norm;
  }
}
