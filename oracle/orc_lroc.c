/*
 * orc_lroc.c — restatement of trimaran LowRiskOverCommitment (TEST INFRASTRUCTURE, see spx_oracle.h).
 *
 * Follows pkg/trimaran/lowriskovercommitment/lowriskovercommitment.go:96-141 (PreScore, Score), :158-255
 * (computeRank, computeRisk), :259-275 (CreatePodResourcesStateData); .../beta.go:41-69, :85-126, :152-191;
 * pkg/trimaran/resourcestats.go:45-86 (CreateResourceStats, GetMuSigma), :109-146 (GetResourceRequested / Limits),
 * :163-225 (GetNodeRequestsAndLimits), :229-232 (SetMaxLimits).
 *
 * Third-party piece: beta.go:158-171 calls gonum.org/v1/gonum v0.12.0 mathext.RegIncBeta (go.mod:18), which is not
 * vendored under /root/reference.  gonum documents it as the regularized incomplete beta function I_x(a,b), a port of
 * the Cephes `incbet` routine (S. Moshier): power series when b*x <= 1 and x <= 0.95, otherwise one of two continued
 * fractions chosen by the sign of x*(a+b-2)-(a-1) after the usual x <-> 1-x reflection, with the prefactor
 * x^a (1-x)^b / (a B(a,b)) taken directly or through logarithms.  orc_reg_inc_beta restates that published algorithm;
 * it is pinned (tests/test_oracle_golden_lroc.py) by the reference's own known answers (beta_test.go:236-327,
 * lowriskovercommitment_test.go:341-395) and, because those are few, cross-checked against an independent
 * implementation of the same function (scipy.special.betainc; fixture + generating script under tests/golden/).
 * A last-digit difference from gonum's floating-point evaluation can move round(rank*100) by at most 1 at an exact
 * rounding boundary, inside the +-1 tolerance of the parity bar.
 */
#include <float.h>
#include <math.h>

#include "spx_oracle.h"

/* Go builtin min/max on float64 (Go 1.21+): NaN if any argument is NaN */
static double go_min(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a < b ? a : b); }
static double go_max(double a, double b) { return (isnan(a) || isnan(b)) ? NAN : (a > b ? a : b); }

/* ---------------------------------------------------------------- regularized incomplete beta (Cephes incbet) */

static const double kMachEp = 1.11022302462515654042e-16; /* 2^-53 */
static const double kMaxLog = 7.09782712893383996843e2;
static const double kMinLog = -7.451332191019412076235e2;
static const double kMaxGam = 171.624376956302725;
static const double kBig = 4.503599627370496e15;
static const double kBigInv = 2.22044604925031308085e-16;

/* ln B(a,b) and 1/B(a,b) for positive arguments.  Cephes writes 1/B as gamma(a+b)/(gamma(a)*gamma(b)), whose
 * denominator overflows for a+b just below MAXGAM with one tiny argument (e.g. a = 0.003, b = 171.5) and then yields 0;
 * whether gonum v0.12.0 keeps that form cannot be checked here (dependency not vendored), so this restatement takes
 * the overflow-safe route for that corner and the synthetic generators stay away from it. */
static double ln_beta(double a, double b) { return lgamma(a) + lgamma(b) - lgamma(a + b); }
static double inv_beta(double a, double b) {
  double den = tgamma(a) * tgamma(b);
  return isinf(den) ? exp(-ln_beta(a, b)) : tgamma(a + b) / den;
}

/* x^a (1-x)^b / (a B(a,b)) * w, directly when nothing overflows, through logarithms otherwise */
static double incbet_prefactor(double a, double b, double x, double xc, double w) {
  double y = a * log(x);
  double t = b * log(xc);
  if (a + b < kMaxGam && fabs(y) < kMaxLog && fabs(t) < kMaxLog) {
    t = pow(xc, b);
    t *= pow(x, a);
    t /= a;
    t *= w;
    t *= inv_beta(a, b);
    return t;
  }
  y += t - ln_beta(a, b);
  y += log(w / a);
  return y < kMinLog ? 0.0 : exp(y);
}

/* power series  x^a/(a B(a,b)) * (1 + a * sum_n ((1-b)(2-b)...(n-b) x^n) / (n! (a+n))), for b*x <= 1 */
static double incbet_pseries(double a, double b, double x) {
  double ai = 1.0 / a;
  double u = (1.0 - b) * x;
  double v = u / (a + 1.0);
  double t1 = v;
  double t = u;
  double n = 2.0;
  double s = 0.0;
  double z = kMachEp * ai;
  while (fabs(v) > z) {
    u = (n - b) * x / n;
    t *= u;
    v = t / (a + n);
    s += v;
    n += 1.0;
  }
  s += t1;
  s += ai;
  u = a * log(x);
  if (a + b < kMaxGam && fabs(u) < kMaxLog) {
    t = inv_beta(a, b);
    s = s * t * pow(x, a);
  } else {
    t = -ln_beta(a, b) + u + log(s);
    s = t < kMinLog ? 0.0 : exp(t);
  }
  return s;
}

/* the two continued fractions share the evaluation loop; they differ in the partial numerators' coefficients:
 * k[] start values, their increments, and the variable (x or x/(1-x)) */
static double incbet_cf(const double k0[8], const double dk[8], double z) {
  double k[8];
  for (int i = 0; i < 8; ++i) k[i] = k0[i];
  double pkm2 = 0.0, qkm2 = 1.0, pkm1 = 1.0, qkm1 = 1.0;
  double ans = 1.0, r = 1.0;
  const double thresh = 3.0 * kMachEp;
  for (int n = 0; n < 300; ++n) {
    double xk = -(z * k[0] * k[1]) / (k[2] * k[3]);
    double pk = pkm1 + pkm2 * xk;
    double qk = qkm1 + qkm2 * xk;
    pkm2 = pkm1, pkm1 = pk, qkm2 = qkm1, qkm1 = qk;
    xk = (z * k[4] * k[5]) / (k[6] * k[7]);
    pk = pkm1 + pkm2 * xk;
    qk = qkm1 + qkm2 * xk;
    pkm2 = pkm1, pkm1 = pk, qkm2 = qkm1, qkm1 = qk;
    if (qk != 0) r = pk / qk;
    double t;
    if (r != 0) {
      t = fabs((ans - r) / r);
      ans = r;
    } else {
      t = 1.0;
    }
    if (t < thresh) break;
    for (int i = 0; i < 8; ++i) k[i] += dk[i];
    if (fabs(qk) + fabs(pk) > kBig) {
      pkm2 *= kBigInv, pkm1 *= kBigInv, qkm2 *= kBigInv, qkm1 *= kBigInv;
    }
    if (fabs(qk) < kBigInv || fabs(pk) < kBigInv) {
      pkm2 *= kBig, pkm1 *= kBig, qkm2 *= kBig, qkm1 *= kBig;
    }
  }
  return ans;
}

/* mathext.RegIncBeta(a, b, x) for a, b > 0 and 0 <= x <= 1 */
double orc_reg_inc_beta(double a, double b, double x) {
  if (a <= 0 || b <= 0) return NAN;
  if (x <= 0) return x == 0 ? 0.0 : NAN;
  if (x >= 1) return x == 1 ? 1.0 : NAN;
  if (b * x <= 1.0 && x <= 0.95) return incbet_pseries(a, b, x);
  double w = 1.0 - x;
  int flipped = 0;
  double xc = w;
  if (x > a / (a + b)) { /* past the mean: evaluate the complement */
    flipped = 1;
    double tmp = a;
    a = b, b = tmp;
    xc = x;
    x = w;
  }
  double t;
  if (flipped && b * x <= 1.0 && x <= 0.95) {
    t = incbet_pseries(a, b, x);
  } else {
    double y = x * (a + b - 2.0) - (a - 1.0);
    if (y < 0.0) {
      const double k0[8] = {a, a + b, a, a + 1.0, 1.0, b - 1.0, a + 1.0, a + 2.0};
      const double dk[8] = {1.0, 1.0, 2.0, 2.0, 1.0, -1.0, 2.0, 2.0};
      w = incbet_cf(k0, dk, x);
    } else {
      const double k0[8] = {a, b - 1.0, a, a + 1.0, 1.0, a + b, a + 1.0, a + 2.0};
      const double dk[8] = {1.0, -1.0, 2.0, 2.0, 1.0, 1.0, 2.0, 2.0};
      w = incbet_cf(k0, dk, x / (1.0 - x)) / xc;
    }
    t = incbet_prefactor(a, b, x, xc, w);
  }
  if (flipped) t = t <= kMachEp ? 1.0 - kMachEp : 1.0 - t;
  return t;
}

/* ---------------------------------------------------------------- beta.go */

/* RegularizedIncomplete beta.go:158-171 */
static double regularized_incomplete(double x, double a, double b) {
  if (a <= 0 || b <= 0 || x < 0 || x > 1) return NAN;
  if (x == 0) return 0;
  if (x == 1) return 1;
  return orc_reg_inc_beta(a, b, x);
}

/* BetaDistribution.DistributionFunction beta.go:85-91 */
double orc_beta_distribution_function(double alpha, double beta, double x) {
  double p = regularized_incomplete(x, alpha, beta);
  if (isnan(p) || p < 0 || p > 1) p = 0;
  return p;
}

/* GetMaxVariance beta.go:120-125 */
double orc_beta_max_variance(double m1) { return (m1 > 0 && m1 < 1) ? m1 * (1 - m1) : 0; }

/* BetaDistribution.MatchMoments beta.go:107-117 on a distribution built by NewBetaDistribution(1, 1) */
int orc_beta_match_moments(double m1, double m2, double* alpha, double* beta) {
  double variance = m2 - m1 * m1;
  if (m1 < 0 || m1 > 1 || variance < 0 || variance >= m1 * (1 - m1)) return 0;
  double temp = (m1 * (1 - m1) / variance) - 1;
  temp = go_max(temp, 4.9406564584124654e-324); /* math.SmallestNonzeroFloat64 */
  *alpha = m1 * temp;
  *beta = (1 - m1) * temp;
  return 1; /* computeMoments: isValid is true since NewBetaDistribution(1,1) */
}

/* ComputeProbability beta.go:173-191; *has_dist = 1 when a fitted distribution is returned */
double orc_lroc_compute_probability(double mu, double sigma, double threshold, int* has_dist, double* alpha, double* beta) {
  *has_dist = 0;
  if (mu == 0 || (sigma == 0 && mu <= threshold)) return 1;
  if (sigma == 0 && mu > threshold) return 0;
  double m1 = mu;
  double m2 = (sigma * sigma) + (mu * mu);
  if (!orc_beta_match_moments(m1, m2, alpha, beta)) return 0;
  double below = orc_beta_distribution_function(*alpha, *beta, threshold);
  *has_dist = 1;
  if (isnan(below)) return 1;
  return below;
}

/* ---------------------------------------------------------------- resourcestats.go */

/* GetResourceLimits -> GetEffectiveResource resourcestats.go:116-146 over the containers' Limits */
void orc_get_resource_limits(const spx_pod_objects* pods, int64_t pod, int64_t* milli_cpu, int64_t* memory) {
  int64_t cpu = 0, mem = 0, q;
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) {
    if (pods->ctr_kind[c] != SPX_CTR_APP) continue;
    if (orc_find_qty(pods->lim_res, pods->lim_qty, pods->lim_ptr[c], pods->lim_ptr[c + 1], SPX_RES_CPU, &q)) cpu += q;
    if (orc_find_qty(pods->lim_res, pods->lim_qty, pods->lim_ptr[c], pods->lim_ptr[c + 1], SPX_RES_MEMORY, &q)) mem += q;
  }
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) {
    if (pods->ctr_kind[c] == SPX_CTR_APP) continue;
    if (orc_find_qty(pods->lim_res, pods->lim_qty, pods->lim_ptr[c], pods->lim_ptr[c + 1], SPX_RES_CPU, &q) && q > cpu) cpu = q;
    if (orc_find_qty(pods->lim_res, pods->lim_qty, pods->lim_ptr[c], pods->lim_ptr[c + 1], SPX_RES_MEMORY, &q) && q > mem) mem = q;
  }
  if (pods->ovh_ptr) {
    if (orc_find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[pod], pods->ovh_ptr[pod + 1], SPX_RES_CPU, &q)) cpu += q;
    if (orc_find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[pod], pods->ovh_ptr[pod + 1], SPX_RES_MEMORY, &q)) mem += q;
  }
  *milli_cpu = cpu;
  *memory = mem;
}

/* requests and limits of one pod with SetMaxLimits applied (resourcestats.go:229-232 for cpu and memory) */
static void pod_requests_limits(const spx_pod_objects* pods, int64_t pod, int64_t out[4]) {
  orc_get_resource_requested(pods, pod, &out[0], &out[1]);
  orc_get_resource_limits(pods, pod, &out[2], &out[3]);
  if (out[2] < out[0]) out[2] = out[0];
  if (out[3] < out[1]) out[3] = out[1];
}

/* GetNodeRequestsAndLimits resourcestats.go:163-225.  pod_rl = {reqCPU, reqMem, limCPU, limMem} of the pending pod as
 * handed in by the caller (podRequests / podLimits). */
void orc_node_requests_and_limits(const spx_node_objects* nodes, const spx_node_pods_objects* node_pods, int64_t node,
                                  const int64_t* pod_rl, orc_node_requests_limits* out) {
  int64_t req_cpu = 0, req_mem = 0, lim_cpu = 0, lim_mem = 0;
  int64_t cap_cpu = nodes->alloc_cpu_milli[node]; /* node.Status.Allocatable: MilliValue / Value */
  int64_t cap_mem = nodes->alloc_mem[node];
  if (node_pods && node_pods->p_ptr) {
    for (int32_t i = node_pods->p_ptr[node]; i < node_pods->p_ptr[node + 1]; ++i) { /* :184-206, pods on the node */
      int64_t rl[4];
      pod_requests_limits(node_pods->pods, node_pods->p_pod[i], rl);
      req_cpu += rl[0], req_mem += rl[1], lim_cpu += rl[2], lim_mem += rl[3];
    }
  }
  out->req_minus_pod_cpu = req_cpu, out->req_minus_pod_mem = req_mem; /* :188-190, the pending pod comes last */
  out->lim_minus_pod_cpu = lim_cpu, out->lim_minus_pod_mem = lim_mem;
  req_cpu += pod_rl[0], req_mem += pod_rl[1], lim_cpu += pod_rl[2], lim_mem += pod_rl[3];
  if (req_cpu > cap_cpu) req_cpu = cap_cpu; /* :208-211 setMin */
  if (req_mem > cap_mem) req_mem = cap_mem;
  if (out->req_minus_pod_cpu > cap_cpu) out->req_minus_pod_cpu = cap_cpu;
  if (out->req_minus_pod_mem > cap_mem) out->req_minus_pod_mem = cap_mem;
  out->req_cpu = req_cpu, out->req_mem = req_mem, out->lim_cpu = lim_cpu, out->lim_mem = lim_mem;
  out->cap_cpu = cap_cpu, out->cap_mem = cap_mem;
}

/* ---------------------------------------------------------------- lowriskovercommitment.go */

/* computeRisk lowriskovercommitment.go:173-255 */
double orc_lroc_compute_risk(const spx_node_objects* nodes, const spx_metrics_objects* metrics, int64_t node, int type,
                             const orc_node_requests_limits* nrl, const spx_lroc_params* p) {
  double risk_limit = 0, risk_load = 0;
  int64_t request, limit, capacity, request_minus_pod, limit_minus_pod;
  if (type == SPX_MT_CPU) {
    request = nrl->req_cpu, limit = nrl->lim_cpu, capacity = nrl->cap_cpu;
    request_minus_pod = nrl->req_minus_pod_cpu, limit_minus_pod = nrl->lim_minus_pod_cpu;
  } else {
    request = nrl->req_mem, limit = nrl->lim_mem, capacity = nrl->cap_mem;
    request_minus_pod = nrl->req_minus_pod_mem, limit_minus_pod = nrl->lim_minus_pod_mem;
  }
  if (limit > capacity) risk_limit = (double)(limit - capacity) / (double)(limit - request); /* :206-208 */

  orc_resource_stats rs;
  if (orc_create_resource_stats(nodes, metrics, node, 0, 0, type, &rs)) { /* :212-214, zeroRequest */
    double mu, sigma;
    orc_get_mu_sigma(&rs, &mu, &sigma);
    sigma *= orc_go_pow((double)p->smoothing_window_size, 0.5);                 /* :218 */
    sigma = go_min(sigma, sqrt(orc_beta_max_variance(mu) * 0.99));              /* :220, MaxVarianceAllowance */
    double alloc_threshold = (double)request_minus_pod / (double)capacity;     /* :223 */
    alloc_threshold = go_min(go_max(alloc_threshold, 0), 1);
    int has_dist;
    double alpha = 0, beta = 0;
    double alloc_prob = orc_lroc_compute_probability(mu, sigma, alloc_threshold, &has_dist, &alpha, &beta);
    if (limit_minus_pod < capacity && request_minus_pod <= limit_minus_pod) {  /* :230-241 */
      double limit_threshold = (double)limit_minus_pod / (double)capacity;
      if (limit_threshold == 0) {
        alloc_prob = 1;
      } else if (has_dist) {
        double limit_prob = orc_beta_distribution_function(alpha, beta, limit_threshold);
        if (limit_prob > 0) {
          alloc_prob /= limit_prob;
          alloc_prob = go_min(go_max(alloc_prob, 0), 1);
        }
      }
    }
    risk_load = 1 - alloc_prob; /* :244 */
  }
  double w = type == SPX_MT_CPU ? p->risk_limit_weight_cpu : p->risk_limit_weight_mem; /* :250 */
  double total = w * risk_limit + (1 - w) * risk_load;
  return go_min(go_max(total, 0), 1);
}

/* LowRiskOverCommitment.Score lowriskovercommitment.go:105-141 (PreScore state = CreatePodResourcesStateData) */
int64_t orc_lroc_score(const spx_node_objects* nodes, const spx_node_pods_objects* node_pods, const spx_metrics_objects* metrics,
                       const spx_pod_objects* pods, const spx_lroc_params* p, int64_t pod, int64_t node) {
  int64_t rl[4];
  pod_requests_limits(pods, pod, rl);                               /* :259-268 */
  if (rl[0] == 0 && rl[1] == 0 && rl[2] == 0 && rl[3] == 0) return 0; /* :124-128 best-effort pods */
  int32_t lo, hi;
  if (!orc_node_metrics(metrics, node, &lo, &hi)) return 0;         /* :130-134 */
  orc_node_requests_limits nrl;
  orc_node_requests_and_limits(nodes, node_pods, node, rl, &nrl);   /* :162 */
  double risk_cpu = orc_lroc_compute_risk(nodes, metrics, node, SPX_MT_CPU, &nrl, p);
  double risk_mem = orc_lroc_compute_risk(nodes, metrics, node, SPX_MT_MEMORY, &nrl, p);
  double rank = 1 - go_max(risk_cpu, risk_mem);                     /* :165 */
  return (int64_t)round(rank * 100.0);                              /* :136-137, fwk.MaxNodeScore */
}
