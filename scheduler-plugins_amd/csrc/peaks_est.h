// peaks_est.h — the float32 interval of trimaran Peaks' raw score (SPX_OPT_PEAKS_ESTIMATE): per-node constants and the per-cell interval.
// Compiles for the device (kernels_peaks.hip: k_peaks_nodetab, k_peaks_minmax_est, k_peaks_write_est) and for the host:
// tests/cpp/peaks_est_check.cc runs the same source on the CPU, where tests/test_exactness_arguments.py holds it bit for bit against the
// numpy replay whose intervals are checked to contain the float64 sequence's scores — only the exponential differs between the two
// builds (v_exp_f32 on the device, a correctly rounded 2^y on the host; the replay perturbs it by +-3 ulp).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SPX_PK_HD __host__ __device__ __forceinline__
#else
#define SPX_PK_HD inline
#endif

namespace spx {
namespace peaks_est {

// Interval estimates (round 5, SPX_OPT_PEAKS_ESTIMATE, default on).
//
// Both passes above spend ~45 float64 instructions per cell on raw_score (a division, OCML's exp).  Neither needs the value of most
// cells: the min/max pass needs the row's two extremes, the write pass needs floor(100 * (raw - min) / span), which a value known to
// a few parts in 10^5 decides for all but the cells next to a step.  So both passes first compute, per cell, a float32 INTERVAL
// [lo, hi] that is guaranteed to contain the float64 raw score the reference sequence yields — 13 full-rate float32 instructions and
// one v_exp_f32 — and evaluate raw_score itself (the code above, node constants re-read from the tables) only where the interval
// cannot decide:
//   min/max pass   the cells with hi >= (largest lo of the wave's nodes) or lo <= (smallest hi): the wave's extremes are among them;
//   write pass     the cells whose interval of 100 * (raw - min) / span straddles an integer.
// Everything that reaches a table or the row statistic is therefore either the float64 sequence's own value or a value the interval
// proves equal to it: the tables are byte-identical to k_peaks' (tests/test_gpu_peaks.py::test_estimate_*: every cell, both ways).
//
// The interval.  e^(K2 predicted) - e^(K2 util) = e_now (e^(K2 (predicted - util)) - 1) and predicted - util = 100 pod / cap, so with
// C0 = 100 util_m / cap, C1 = 100 / cap, QL = K2 C1 log2(e), KE = 1e15 K1 e_now (float64, rounded once to float32) and u = 2^-24:
//   y = ql * pod                          relative error 3u (ql, pod, the product)
//   e = v_exp_f32(y)                      |e - 2^Y| <= e (2.2u |y| + 4.1u)                       (<= 3 ulp hardware exp2)
//   est = ke * (e - 1),  B = |ke| (e + 1) (5u |y| + 12u) + sigma            (needed: 2.2u |y| + 8.2u)
// — the difference is formed AFTER the common factor is taken out, so the interval is a few 10^-6 of the score itself even where the
// jump is a thousandth of the two exponentials.  The float64 sequence's own roundings (of predicted, of the two exponentials, of the
// products: below |KE| (e + 1) (1 + |K2| dmax) 2^-47) disappear in what the constants have to spare; sigma = 2 covers the truncation to an integer (0 for a node without a power model: it scores exactly 0).
// A TAME node: cap > 0, every constant finite, |util| <= 400, |K2| log2(e) dmax <= 40 with dmax = 101 + |util|, K1 = 0 or
// 1e7 <= |KE| <= 1e24 — conditions on the node alone: a cell with 100 pod / cap > dmax has predicted > 101, i.e. is beyond the
// band below and scores 0 whatever its interval says (y is clamped at 41 so that nothing overflows there; |p - predicted| grows with the
// request, but so does predicted - 100).  For the other cells |C1| pod + |C0| <= 1024 and |y| <= 40, which is what the bounds above use.
// Replayed on the CPU against the float64 sequence with the exponential perturbed by +-3 ulp: the largest |raw - est| / B over 10^7
// cells stays below one half (tests/test_exactness_arguments.py).
// `predicted > 100` scores 0 (peaks.go:139-140).  p = fma(c1, pod, c0) is within 4u * 1024 of predicted; g = clamp(102400.5 - 1024 p, 0, 1)
// is 1 below 100 - 2^-11, 0 above 100 + 2^-11 (the band is twice that bound), and in between a = g - g^2 > 0 blows the interval up to +-1e37:
//   lo, hi = est g -+ (B g + a 1e38).
// A node that is not tame gets the constants of a cell inside the band (c0 = 100, the rest 0): always "undecided", always evaluated
// by raw_score.  A node without metrics (valid = 0) or with K1 = 0 has lo = hi = 0, which is its exact score; so has every cell
// with g = 0.  Such cells take part in the row statistic as the value 0 and are never evaluated.  Columns past the table carry NaN
// constants: v_max_f32 / v_min_f32 and every comparison ignore them.
constexpr float kEstBeta = 5.0f * 0x1p-24f, kEstAlpha = 12.0f * 0x1p-24f;
constexpr float kEstHuge = 1e38f;
constexpr float kEstBpInv = 1024.0f;
constexpr float kEstGc = 102400.5f;  // 0.5 + 100 * kEstBpInv, exact in float32
constexpr double kEstUtilMax = 400.0, kEstYMax = 40.0, kEstMagMin = 1e7, kEstMagMax = 1e24;
constexpr float kEstYClamp = 41.0f;
constexpr float kEstYFloor = -200.0f;

struct NodeE {
  float c0, c1, ql, ke, sigma;
};

SPX_PK_HD float est_exp2(float y) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(y);
#else
  return static_cast<float>(exp2(static_cast<double>(y)));
#endif
}

// ND: load_node's view of a node (cap, util_m, e_now, k1, k2, valid); util = the node's cpu utilisation in percent
template <class ND>
SPX_PK_HD NodeE est_node_compute(const ND& nd, double util) {
  NodeE ne;
  double c0 = 0.0, c1 = 0.0;
  if (nd.cap != 0) {
    c0 = 100 * nd.util_m / nd.cap;
    c1 = 100 / nd.cap;
  }
  const double ql = nd.k2 * c1 * 1.4426950408889634, ke = nd.k1 * 1e15 * nd.e_now;
  // a cell that is not beyond the band has 100 pod / cap <= 101 + |util|: the preconditions need no word about the pods
  const double dmax = 101.0 + fabs(util), ymax = fabs(nd.k2) * 1.4426950408889634 * dmax;
  const double sigma = nd.k1 == 0 ? 0.0 : 2.0;  // the truncation to an integer (a node without a power model scores exactly 0)
  const bool fin = __builtin_isfinite(c0) && __builtin_isfinite(c1) && __builtin_isfinite(ql) && __builtin_isfinite(ke) && __builtin_isfinite(sigma);
  const bool mag = nd.k1 == 0 || (fabs(ke) >= kEstMagMin && fabs(ke) <= kEstMagMax);
  const bool tame = nd.cap > 0 && fin && fabs(util) <= kEstUtilMax && ymax <= kEstYMax && mag;
  if (!nd.valid) {  // raw_score is 0 whatever the rest says
    ne.c0 = ne.c1 = ne.ql = ne.ke = ne.sigma = 0.0f;
  } else if (!tame) {  // a cell inside the band: g = 1/2
    ne.c0 = 100.0f;
    ne.c1 = ne.ql = ne.ke = ne.sigma = 0.0f;
  } else {
    ne.c0 = static_cast<float>(c0), ne.c1 = static_cast<float>(c1), ne.ql = static_cast<float>(ql);
    ne.ke = nd.k1 == 0 ? 0.0f : static_cast<float>(ke);
    ne.sigma = static_cast<float>(sigma);
  }
  return ne;
}

// the interval given p ~ predicted, the (clamped) exponent y and e ~ 2^y: what follows the exponential (the host check supplies its own e)
SPX_PK_HD void est_interval_from(const NodeE& ne, float p, float y, float e, float& lo, float& hi) {
  const float est = ne.ke * (e - 1.0f);
  const float w = __builtin_fmaf(__builtin_fabsf(ne.ke), e, __builtin_fabsf(ne.ke));
  const float b = __builtin_fmaf(w, __builtin_fmaf(__builtin_fabsf(y), kEstBeta, kEstAlpha), ne.sigma);
  const float g = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(p, -kEstBpInv, kEstGc), 0.0f), 1.0f);
  const float am = __builtin_fmaf(-g, g, g);
  const float bg = __builtin_fmaf(b, g, am * kEstHuge);
  const float eg = est * g;
  lo = eg - bg;
  hi = eg + bg;
}

// Clamped from above (beyond the clamp the cell is beyond the band: g = 0, and nothing overflows) and, round 6, from below: 2^y is 0 in
// float32 from y = -150 on whatever the request, so est and the true score agree to |KE| 2^-200 there, while an unclamped |y| — a huge
// request against a negative K2 — entered B = |KE| (e + 1) (5u |y| + 12u) and could carry it past FLT_MAX, where lo / hi turn NaN and a
// NaN interval reads as "outside the table" instead of "undecided" (advisor, round 5).  |y| <= 200 keeps B below 2e20.  One v_med3_f32.
SPX_PK_HD float est_exponent(const NodeE& ne, float pod32) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(ne.ql * pod32, kEstYFloor, kEstYClamp);
#else
  return fmaxf(fminf(ne.ql * pod32, kEstYClamp), kEstYFloor);
#endif
}

SPX_PK_HD float est_predicted(const NodeE& ne, float pod32) { return __builtin_fmaf(ne.c1, pod32, ne.c0); }

SPX_PK_HD void est_interval(const NodeE& ne, float pod32, float& lo, float& hi) {
  const float p = est_predicted(ne, pod32);
  const float y = est_exponent(ne, pod32);
  const float e = est_exp2(y);
  est_interval_from(ne, p, y, e, lo, hi);
}

}  // namespace peaks_est
}  // namespace spx
