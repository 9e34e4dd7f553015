/*
 * orc_trimaran.c — restatement of trimaran TargetLoadPacking and LoadVariationRiskBalancing
 * (TEST INFRASTRUCTURE, see spx_oracle.h).
 *
 * Follows pkg/trimaran/targetloadpacking/targetloadpacking.go:107-205,
 * pkg/trimaran/loadvariationriskbalancing/loadvariationriskbalancing.go:84-122,
 * pkg/trimaran/loadvariationriskbalancing/analysis.go:34-60, pkg/trimaran/resourcestats.go:45-146,
 * pkg/trimaran/collector.go:110-123 and the cache read at targetloadpacking.go:151-168.
 */
#include <math.h>

#include "spx_oracle.h"

/* ---------------------------------------------------------------- shared helpers */

/* Collector.GetNodeMetrics (collector.go:110-123): returns 0 when `metrics == nil` for the node
 * (nil map, node absent, or a nil Metrics slice), 1 otherwise with [*lo,*hi) the metric range. */
int orc_node_metrics(const spx_metrics_objects* m, int64_t node, int32_t* lo, int32_t* hi) {
  if (m == 0 || m->map_is_nil) return 0;  /* :113-116 -> (nil, nil) */
  if (!m->node_present[node]) return 0;   /* :118-121 -> (nil, allMetrics) */
  if (m->node_metrics_nil && m->node_metrics_nil[node]) return 0;
  *lo = m->m_ptr[node];
  *hi = m->m_ptr[node + 1];
  return 1;
}

int orc_find_qty(const int32_t* res, const int64_t* qty, int32_t lo, int32_t hi, int32_t want, int64_t* out) {
  for (int32_t i = lo; i < hi; ++i)
    if (res[i] == want) {
      *out = qty[i];
      return 1;
    }
  return 0;
}

/* ---------------------------------------------------------------- TargetLoadPacking */

/* PredictUtilisation (targetloadpacking.go:198-205) */
int64_t orc_tlp_predict_utilisation(const spx_pod_objects* pods, int32_t c, const spx_tlp_params* p) {
  int64_t q;
  if (orc_find_qty(pods->lim_res, pods->lim_qty, pods->lim_ptr[c], pods->lim_ptr[c + 1], SPX_RES_CPU, &q))
    return q; /* Limits.Cpu().MilliValue() */
  if (orc_find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q))
    return (int64_t)round((double)q * p->requests_multiplier);
  return p->default_requests_milli;
}

/* Σ PredictUtilisation over pod.Spec.Containers (+ overhead cpu) — :122-129 and :160-163.
 * Only app containers: the reference ranges pod.Spec.Containers, not InitContainers. */
static int64_t tlp_pod_cpu(const spx_pod_objects* pods, int64_t pod, const spx_tlp_params* p) {
  int64_t cur = 0;
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c)
    if (pods->ctr_kind[c] == SPX_CTR_APP) cur += orc_tlp_predict_utilisation(pods, c, p);
  int64_t ovh;
  if (pods->ovh_ptr && orc_find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[pod], pods->ovh_ptr[pod + 1], SPX_RES_CPU, &ovh))
    cur += ovh; /* pod.Spec.Overhead.Cpu().MilliValue(); nil map / missing key yield 0 */
  return cur;
}

/* the walk of ScheduledPodsCache[nodeName] in TargetLoadPacking.Score (targetloadpacking.go:151-168) over one node's entries
 * given as a slice: (bind time, pod index into `entry_pods`) pairs in cache order.  orc_tlp_score runs it on the snapshot's
 * image of the cache; orc_commit.c runs it once more on the entries the one-pod-at-a-time cycle appended since
 * (handler.go:131-139 appends to the same per-node slice, and the sum does not depend on the order). */
int64_t orc_tlp_missing_entries(const int64_t* e_ts_unix, const int32_t* e_pod, int32_t n_entries, const spx_pod_objects* entry_pods,
                                int64_t window_end, const spx_tlp_params* p) {
  int64_t missing = 0;
  for (int32_t e = 0; e < n_entries; ++e) {
    int64_t ts = e_ts_unix[e];
    int64_t end = window_end;
    /* Go precedence: a || (b && c) */
    if (ts > end || (ts <= end && (end - ts) < 60 /* metricsAgentReportingIntervalSeconds */)) {
      missing += tlp_pod_cpu(entry_pods, e_pod[e], p);
    }
  }
  return missing;
}

/* TargetLoadPacking.Score (targetloadpacking.go:107-187) */
int64_t orc_tlp_score(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                      const spx_assigned_objects* assigned, const spx_pod_objects* pods,
                      const spx_tlp_params* p, int64_t pod, int64_t node) {
  return orc_tlp_score_appended(nodes, metrics, assigned, pods, p, pod, node, 0, 0, 0, 0);
}

/* the same with `n_more` further cache entries of this node (appended after the snapshot's) */
int64_t orc_tlp_score_appended(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                               const spx_assigned_objects* assigned, const spx_pod_objects* pods,
                               const spx_tlp_params* p, int64_t pod, int64_t node,
                               const int64_t* more_ts_unix, const int32_t* more_pod, int32_t n_more, const spx_pod_objects* more_pods) {
  const int64_t min_node_score = 0;
  int32_t lo, hi;
  if (!orc_node_metrics(metrics, node, &lo, &hi)) return min_node_score; /* :114-120 */

  int64_t cur_pod_cpu = tlp_pod_cpu(pods, pod, p); /* :122-129 */

  double node_cpu_util_percent = 0;
  int cpu_metric_found = 0;
  for (int32_t i = lo; i < hi; ++i) { /* :133-140 — no break: the LAST matching metric wins */
    if (metrics->m_type[i] == SPX_MT_CPU) {
      if (metrics->m_op[i] == SPX_MO_AVG || metrics->m_op[i] == SPX_MO_LATEST) {
        node_cpu_util_percent = metrics->m_value[i];
        cpu_metric_found = 1;
      }
    }
  }
  if (!cpu_metric_found) return min_node_score; /* :142-145 */

  double node_cpu_cap_millis = (double)nodes->cap_cpu_milli[node];                 /* Capacity, :146 */
  double node_cpu_util_millis = (node_cpu_util_percent / 100) * node_cpu_cap_millis; /* :147 */

  int64_t missing = 0; /* :151-168 */
  if (assigned && assigned->e_ptr) {
    const int32_t e0 = assigned->e_ptr[node];
    missing += orc_tlp_missing_entries(assigned->e_ts_unix + e0, assigned->e_pod + e0, assigned->e_ptr[node + 1] - e0, assigned->pods,
                                       metrics->window_end, p);
  }
  if (n_more > 0) missing += orc_tlp_missing_entries(more_ts_unix, more_pod, n_more, more_pods, metrics->window_end, p);

  double predicted = 0; /* :169-173 */
  if (node_cpu_cap_millis != 0)
    predicted = 100 * (node_cpu_util_millis + (double)cur_pod_cpu + (double)missing) / node_cpu_cap_millis;

  double t = (double)p->target_utilization;
  if (predicted > t) { /* :174-181 */
    if (predicted > 100) return min_node_score;
    return (int64_t)round(t * (100 - predicted) / (100 - t));
  }
  return (int64_t)round((100 - t) * predicted / t + t); /* :183-186 */
}

/* ---------------------------------------------------------------- LoadVariationRiskBalancing */

/* math.Pow of Go 1.25 (src/math/pow.go), restated for the cases the plugin reaches.
 * Special cases follow Go's order.  The general fractional-exponent branch of Go is
 * Exp(yf*Log(x)) * x**yi with an amd64 assembly Exp that cannot be reproduced bit-for-bit
 * here; it is delegated to libm pow (a last-ulp difference can move a score by at most 1,
 * inside the ±1 tolerance north_star grants).  Integer exponents and ±0.5 are exact. */
double orc_go_pow(double x, double y) {
  if (y == 0 || x == 1) return 1;
  if (y == 1) return x;
  if (isnan(x) || isnan(y)) return NAN;
  if (x == 0) {
    if (y < 0) return INFINITY; /* sign handling for odd integers is irrelevant for x = +0 */
    return 0;
  }
  if (isinf(y)) {
    if (x == -1) return 1;
    if ((fabs(x) < 1) == (y > 0)) return 0;
    return INFINITY;
  }
  if (isinf(x)) {
    if (x < 0) return pow(x, y);
    return y < 0 ? 0 : INFINITY;
  }
  if (y == 0.5) return sqrt(x);
  if (y == -0.5) return 1 / sqrt(x);
  double yi, yf = modf(fabs(y), &yi);
  if (yf == 0 && yi < 1024) {
    /* ans *= x**yi via frexp square-and-multiply (pow.go:107-133) */
    double a1 = 1.0;
    int ae = 0;
    int xe;
    double x1 = frexp(x, &xe);
    for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
      if (i & 1) {
        a1 *= x1;
        ae += xe;
      }
      x1 *= x1;
      xe <<= 1;
      if (x1 < .5) {
        x1 += x1;
        xe--;
      }
    }
    if (y < 0) {
      a1 = 1 / a1;
      ae = -ae;
    }
    return ldexp(a1, ae);
  }
  return pow(x, y);
}

static double fmin_go(double a, double b) { return a < b ? a : b; } /* builtin min, finite inputs */
static double fmax_go(double a, double b) { return a > b ? a : b; }

/* GetMuSigma resourcestats.go:77-86 */
void orc_get_mu_sigma(const orc_resource_stats* rs, double* mu, double* sigma) {
  if (rs->capacity <= 0) {
    *mu = 0;
    *sigma = 0;
    return;
  }
  double m = (rs->used_avg + rs->req) / rs->capacity;
  m = fmax_go(fmin_go(m, 1), 0);
  double s = rs->used_stdev / rs->capacity;
  s = fmax_go(fmin_go(s, 1), 0);
  *mu = m;
  *sigma = s;
}

/* computeScore analysis.go:34-60 */
double orc_lvrb_compute_score(orc_resource_stats* rs, double margin, double sensitivity) {
  if (rs->capacity <= 0) return 0;
  rs->req = fmax_go(rs->req, 0);
  rs->used_avg = fmax_go(fmin_go(rs->used_avg, rs->capacity), 0);
  rs->used_stdev = fmax_go(fmin_go(rs->used_stdev, rs->capacity), 0);
  double mu, sigma;
  orc_get_mu_sigma(rs, &mu, &sigma);
  if (sensitivity >= 0) sigma = orc_go_pow(sigma, 1 / sensitivity);
  sigma *= margin;
  sigma = fmax_go(fmin_go(sigma, 1), 0);
  double risk = (mu + sigma) / 2;
  return (1. - risk) * 100.0; /* float64(fwk.MaxNodeScore) */
}

/* GetResourceData resourcestats.go:89-107 over the node's metric list */
int orc_get_resource_data(const spx_metrics_objects* metrics, int64_t node, int type, double* avg, double* stdev) {
  int32_t lo, hi;
  *avg = 0;
  *stdev = 0;
  if (!orc_node_metrics(metrics, node, &lo, &hi)) return 0;
  int avg_found = 0, is_valid = 0;
  for (int32_t i = lo; i < hi; ++i) {
    if (metrics->m_type[i] == type) {
      uint8_t op = metrics->m_op[i];
      if (op == SPX_MO_AVG) {
        *avg = metrics->m_value[i];
        avg_found = 1;
      } else if (op == SPX_MO_STD) {
        *stdev = metrics->m_value[i];
      } else if ((op == SPX_MO_EMPTY || op == SPX_MO_LATEST) && !avg_found) {
        *avg = metrics->m_value[i];
      }
      is_valid = 1;
    }
  }
  return is_valid;
}

/* GetResourceRequested -> GetEffectiveResource resourcestats.go:110-146 (cpu and memory only;
 * framework.Resource.Add: MilliCPU += MilliValue(), Memory += Value()) */
void orc_get_resource_requested(const spx_pod_objects* pods, int64_t pod, int64_t* milli_cpu, int64_t* memory) {
  int64_t cpu = 0, mem = 0, q;
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) {
    if (pods->ctr_kind[c] != SPX_CTR_APP) continue;
    if (orc_find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q)) cpu += q;
    if (orc_find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_MEMORY, &q)) mem += q;
  }
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) { /* :129-139 setMax per init container */
    if (pods->ctr_kind[c] == SPX_CTR_APP) continue;
    if (orc_find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_CPU, &q) && q > cpu) cpu = q;
    if (orc_find_qty(pods->req_res, pods->req_qty, pods->req_ptr[c], pods->req_ptr[c + 1], SPX_RES_MEMORY, &q) && q > mem) mem = q;
  }
  if (pods->ovh_ptr) { /* :141-143 */
    if (orc_find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[pod], pods->ovh_ptr[pod + 1], SPX_RES_CPU, &q)) cpu += q;
    if (orc_find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[pod], pods->ovh_ptr[pod + 1], SPX_RES_MEMORY, &q)) mem += q;
  }
  *milli_cpu = cpu;
  *memory = mem;
}

/* CreateResourceStats resourcestats.go:45-74 */
int orc_create_resource_stats(const spx_node_objects* nodes, const spx_metrics_objects* metrics, int64_t node,
                                 int64_t req_cpu, int64_t req_mem, int type, orc_resource_stats* rs) {
  const double mega_factor = 1. / 1024. / 1024.; /* resourcestats.go:29 */
  double node_util, node_std;
  if (!orc_get_resource_data(metrics, node, type, &node_util, &node_std)) return 0;
  if (type == SPX_MT_CPU) {
    rs->capacity = (double)nodes->alloc_cpu_milli[node]; /* node.Status.Allocatable cpu MilliValue */
    rs->req = (double)req_cpu;
  } else {
    rs->capacity = (double)nodes->alloc_mem[node];
    rs->capacity *= mega_factor;
    rs->req = (double)req_mem * mega_factor;
  }
  rs->used_avg = node_util * rs->capacity / 100;
  rs->used_stdev = node_std * rs->capacity / 100;
  return 1;
}

/* LoadVariationRiskBalancing.Score loadvariationriskbalancing.go:84-122 */
int64_t orc_lvrb_score(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                       const spx_pod_objects* pods, const spx_lvrb_params* p, int64_t pod, int64_t node) {
  int32_t lo, hi;
  if (!orc_node_metrics(metrics, node, &lo, &hi)) return 0; /* :90-94 */
  int64_t req_cpu, req_mem;
  orc_get_resource_requested(pods, pod, &req_cpu, &req_mem);
  double cpu_score = 0, memory_score = 0;
  orc_resource_stats cpu_stats, mem_stats;
  int cpu_ok = orc_create_resource_stats(nodes, metrics, node, req_cpu, req_mem, SPX_MT_CPU, &cpu_stats);
  if (cpu_ok) cpu_score = orc_lvrb_compute_score(&cpu_stats, p->safe_variance_margin, p->safe_variance_sensitivity);
  int mem_ok = orc_create_resource_stats(nodes, metrics, node, req_cpu, req_mem, SPX_MT_MEMORY, &mem_stats);
  if (mem_ok) memory_score = orc_lvrb_compute_score(&mem_stats, p->safe_variance_margin, p->safe_variance_sensitivity);
  double total;
  if (mem_ok && cpu_ok)
    total = fmin_go(memory_score, cpu_score);
  else
    total = fmax_go(memory_score, cpu_score);
  return (int64_t)round(total);
}
