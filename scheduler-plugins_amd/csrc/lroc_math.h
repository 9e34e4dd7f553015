// lroc_math.h — the pod-independent half of trimaran LowRiskOverCommitment, once per node: the "risk of measured
// overcommitment" (riskLoad, lowriskovercommitment.go:210-246), which is a Beta-distribution tail fitted to the node's
// load statistics and evaluated at thresholds that only involve the node's own sums (the *MinusPod fields of
// resourcestats.go:188-190).  Hoisting it turns the P x N sweep into a handful of float64 operations per cell and
// leaves the special-function work (regularized incomplete beta, log-gamma) at N evaluations per snapshot.
//
// Compiles for the device (k_lroc_prepare) and for the host: tests/cpp/lroc_math_check.cc runs the same source on the
// CPU against the oracle, so the logic is checked without a GPU; only libm (host) vs OCML (device) last-digit behaviour
// differs between the two builds.
//
// The incomplete beta function follows the published Cephes scheme that gonum's mathext.RegIncBeta (the function
// beta.go:158-171 calls) ports: power series for b*x <= 1, else Lentz-free forward evaluation of one of two continued
// fractions after reflecting to the side of the mean that converges, 300 terms at most — so the values, including the
// slow-convergence behaviour for very peaked distributions, track the reference's.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SPX_HD __host__ __device__ __forceinline__
#else
#define SPX_HD inline
#endif

namespace spx {
namespace lroc {

constexpr double kEps = 1.11022302462515654042e-16;   // 2^-53
constexpr double kLogMax = 7.09782712893383996843e2;
constexpr double kLogMin = -7.451332191019412076235e2;
constexpr double kGammaMax = 171.624376956302725;
constexpr double kHuge = 4.503599627370496e15;         // 2^52
constexpr double kHugeInv = 2.22044604925031308085e-16;
constexpr double kMega = 1.0 / 1024.0 / 1024.0;        // resourcestats.go:29
constexpr double kMaxVarianceAllowance = 0.99;         // lowriskovercommitment.go:47

// Go's builtin min/max on float64 return NaN when either side is NaN
SPX_HD double gmin(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
SPX_HD double gmax(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }

SPX_HD double log_beta(double a, double b) { return lgamma(a) + lgamma(b) - lgamma(a + b); }
// 1/B(a,b); the gamma product overflows for a+b < kGammaMax with one tiny argument — go through logarithms there
SPX_HD double recip_beta(double a, double b) {
  const double den = tgamma(a) * tgamma(b);
  return isinf(den) ? exp(-log_beta(a, b)) : tgamma(a + b) / den;
}

// I_x(a,b) by its power series in x (b*x <= 1)
SPX_HD double series(double a, double b, double x) {
  const double inv_a = 1.0 / a;
  double coeff = (1.0 - b) * x;   // running product (1-b)(2-b)...(n-b) x^n / n!
  double term = coeff / (a + 1.0);
  const double first = term;
  double tail = 0.0;
  const double stop = kEps * inv_a;
  for (double n = 2.0; fabs(term) > stop; n += 1.0) {
    coeff *= (n - b) * x / n;
    term = coeff / (a + n);
    tail += term;
  }
  double sum = tail + first;
  sum += inv_a;
  const double lx = a * log(x);
  if (a + b < kGammaMax && fabs(lx) < kLogMax) return sum * recip_beta(a, b) * pow(x, a);
  const double l = -log_beta(a, b) + lx + log(sum);
  return l < kLogMin ? 0.0 : exp(l);
}

// forward three-term recurrence of a continued fraction whose two interleaved partial numerators are
//   -(z * c0 * c1) / (c2 * c3)   and   (z * c4 * c5) / (c6 * c7),
// every c advancing by its own step per round; at most 300 rounds, rescaled to stay inside the exponent range
struct CfCoeff {
  double c[8];
  double step[8];
};
SPX_HD double continued_fraction(CfCoeff k, double z) {
  double p_prev2 = 0.0, q_prev2 = 1.0, p_prev = 1.0, q_prev = 1.0;
  double value = 1.0, ratio = 1.0;
  for (int round = 0; round < 300; ++round) {
    double num = -(z * k.c[0] * k.c[1]) / (k.c[2] * k.c[3]);
    double p = p_prev + p_prev2 * num, q = q_prev + q_prev2 * num;
    p_prev2 = p_prev, p_prev = p, q_prev2 = q_prev, q_prev = q;
    num = (z * k.c[4] * k.c[5]) / (k.c[6] * k.c[7]);
    p = p_prev + p_prev2 * num, q = q_prev + q_prev2 * num;
    p_prev2 = p_prev, p_prev = p, q_prev2 = q_prev, q_prev = q;
    if (q != 0.0) ratio = p / q;
    double change = 1.0;
    if (ratio != 0.0) {
      change = fabs((value - ratio) / ratio);
      value = ratio;
    }
    if (change < 3.0 * kEps) break;
    for (int i = 0; i < 8; ++i) k.c[i] += k.step[i];
    if (fabs(q) + fabs(p) > kHuge) p_prev2 *= kHugeInv, p_prev *= kHugeInv, q_prev2 *= kHugeInv, q_prev *= kHugeInv;
    if (fabs(q) < kHugeInv || fabs(p) < kHugeInv) p_prev2 *= kHuge, p_prev *= kHuge, q_prev2 *= kHuge, q_prev *= kHuge;
  }
  return value;
}

// regularized incomplete beta I_x(a,b), a,b > 0, 0 < x < 1
SPX_HD double reg_inc_beta(double a, double b, double x) {
  if (b * x <= 1.0 && x <= 0.95) return series(a, b, x);
  double comp = 1.0 - x;
  const bool reflect = x > a / (a + b);
  if (reflect) {  // I_x(a,b) = 1 - I_{1-x}(b,a)
    const double t = a;
    a = b, b = t;
    const double u = x;
    x = comp, comp = u;
  }
  double r;
  if (reflect && b * x <= 1.0 && x <= 0.95) {
    r = series(a, b, x);
  } else {
    double cf;
    if (x * (a + b - 2.0) - (a - 1.0) < 0.0) {
      cf = continued_fraction(CfCoeff{{a, a + b, a, a + 1.0, 1.0, b - 1.0, a + 1.0, a + 2.0}, {1, 1, 2, 2, 1, -1, 2, 2}}, x);
    } else {
      cf = continued_fraction(CfCoeff{{a, b - 1.0, a, a + 1.0, 1.0, a + b, a + 1.0, a + 2.0}, {1, -1, 2, 2, 1, 1, 2, 2}}, x / (1.0 - x)) / comp;
    }
    // x^a (1-x)^b / (a B(a,b)) * cf
    double lx = a * log(x);
    const double lc = b * log(comp);
    if (a + b < kGammaMax && fabs(lx) < kLogMax && fabs(lc) < kLogMax) {
      r = pow(comp, b);
      r *= pow(x, a);
      r /= a;
      r *= cf;
      r *= recip_beta(a, b);
    } else {
      lx += lc - log_beta(a, b);
      lx += log(cf / a);
      r = lx < kLogMin ? 0.0 : exp(lx);
    }
  }
  if (reflect) r = r <= kEps ? 1.0 - kEps : 1.0 - r;
  return r;
}

// BetaDistribution.DistributionFunction (beta.go:85-91) over RegularizedIncomplete (:158-171)
SPX_HD double beta_cdf(double alpha, double beta, double x) {
  double p;
  if (alpha <= 0 || beta <= 0 || x < 0 || x > 1 || x != x) p = NAN;
  else if (x == 0) p = 0;
  else if (x == 1) p = 1;
  else p = reg_inc_beta(alpha, beta, x);
  return (p != p || p < 0 || p > 1) ? 0.0 : p;
}

struct NodeResource {   // one resource of one node, everything pod-independent
  bool metric_valid;    // GetResourceData found a metric of the type (and the node has metrics at all)
  double capacity_stat; // CreateResourceStats capacity: allocatable millicores, or allocatable bytes * kMega
  double avg, stdev;    // the metric values (percent)
  int64_t capacity;     // allocatable, canonical integer units
  int64_t requested;    // sum of the requests of the pods on the node (uncapped)
  int64_t limits;       // sum of their limits
};

// riskLoad of computeRisk (lowriskovercommitment.go:210-246) for a node that has metrics
SPX_HD double risk_load(const NodeResource& r, double sqrt_window) {
  if (!r.metric_valid) return 0.0;  // CreateResourceStats !ok (:213)
  // GetMuSigma (resourcestats.go:77-86) on CreateResourceStats (:45-74) with a zero pod request
  double mu = 0.0, sigma = 0.0;
  if (r.capacity_stat > 0) {
    const double used_avg = r.avg * r.capacity_stat / 100;
    const double used_std = r.stdev * r.capacity_stat / 100;
    mu = (used_avg + 0.0) / r.capacity_stat;
    mu = gmax(gmin(mu, 1.0), 0.0);
    sigma = used_std / r.capacity_stat;
    sigma = gmax(gmin(sigma, 1.0), 0.0);
  }
  sigma *= sqrt_window;  // math.Pow(window, 0.5) is math.Sqrt (:218)
  const double max_var = (mu > 0 && mu < 1) ? mu * (1 - mu) : 0.0;  // GetMaxVariance beta.go:120-125
  sigma = gmin(sigma, sqrt(max_var * kMaxVarianceAllowance));       // :220
  const int64_t req_minus_pod = r.requested < r.capacity ? r.requested : r.capacity;  // resourcestats.go:210-211
  const int64_t lim_minus_pod = r.limits;
  double alloc_threshold = static_cast<double>(req_minus_pod) / static_cast<double>(r.capacity);  // :223
  alloc_threshold = gmin(gmax(alloc_threshold, 0.0), 1.0);
  // ComputeProbability beta.go:173-191
  bool fitted = false;
  double alpha = 0.0, beta = 0.0, alloc_prob;
  if (mu == 0 || (sigma == 0 && mu <= alloc_threshold)) {
    alloc_prob = 1.0;
  } else if (sigma == 0 && mu > alloc_threshold) {
    alloc_prob = 0.0;
  } else {
    const double m2 = (sigma * sigma) + (mu * mu);
    const double variance = m2 - mu * mu;  // MatchMoments :107-117
    if (mu < 0 || mu > 1 || variance < 0 || variance >= mu * (1 - mu)) {
      alloc_prob = 0.0;
    } else {
      double t = (mu * (1 - mu) / variance) - 1;
      t = gmax(t, 4.9406564584124654e-324);
      alpha = mu * t;
      beta = (1 - mu) * t;
      fitted = true;
      alloc_prob = beta_cdf(alpha, beta, alloc_threshold);
    }
  }
  if (lim_minus_pod < r.capacity && req_minus_pod <= lim_minus_pod) {  // :230-241
    const double limit_threshold = static_cast<double>(lim_minus_pod) / static_cast<double>(r.capacity);
    if (limit_threshold == 0) {
      alloc_prob = 1.0;
    } else if (fitted) {
      const double limit_prob = beta_cdf(alpha, beta, limit_threshold);
      if (limit_prob > 0) {
        alloc_prob /= limit_prob;
        alloc_prob = gmin(gmax(alloc_prob, 0.0), 1.0);
      }
    }
  }
  return 1 - alloc_prob;  // :244
}

}  // namespace lroc
}  // namespace spx
