// Host build of scheduler-plugins_amd/csrc/peaks_est.h for tests/test_exactness_arguments.py: the per-node constants and the per-cell
// interval of the Peaks estimate exactly as the kernels compute them, minus the exponential (supplied by the caller: the device has
// v_exp_f32, the replay a correctly rounded 2^y perturbed by a few ulp).  A test vehicle only: nothing in the product links it.
#include "../../scheduler-plugins_amd/csrc/peaks_est.h"

namespace {
struct Node {  // load_node's view (kernels_peaks.hip: NodeP)
  double cap, util_m, k1, k2, e_now;
  bool valid;
};
}  // namespace

extern "C" {

// n nodes: cap, util (percent), e_now = exp(k2 * util) as the caller computed it, k1, k2, valid -> out[n][5] = c0, c1, ql, ke, sigma
void peaks_est_check_nodes(int64_t n, const double* cap, const double* util, const double* e_now, const double* k1, const double* k2, const uint8_t* valid,
                           float* out) {
  for (int64_t i = 0; i < n; ++i) {
    Node nd;
    nd.cap = cap[i];
    nd.util_m = (util[i] / 100) * nd.cap;  // load_node, peaks.go:133
    nd.k1 = k1[i], nd.k2 = k2[i], nd.e_now = e_now[i];
    nd.valid = valid[i] != 0;
    const spx::peaks_est::NodeE ne = spx::peaks_est::est_node_compute(nd, util[i]);
    out[5 * i + 0] = ne.c0, out[5 * i + 1] = ne.c1, out[5 * i + 2] = ne.ql, out[5 * i + 3] = ne.ke, out[5 * i + 4] = ne.sigma;
  }
}

// the exponents y[p][n] the kernels hand to v_exp_f32
void peaks_est_check_exponents(int64_t n_pods, int64_t n_nodes, const float* consts, const float* pod32, float* y) {
  for (int64_t p = 0; p < n_pods; ++p)
    for (int64_t i = 0; i < n_nodes; ++i) {
      const float* c = consts + 5 * i;
      const spx::peaks_est::NodeE ne{c[0], c[1], c[2], c[3], c[4]};
      y[p * n_nodes + i] = spx::peaks_est::est_exponent(ne, pod32[p]);
    }
}

// the intervals given e[p][n] ~ 2^y
void peaks_est_check_intervals(int64_t n_pods, int64_t n_nodes, const float* consts, const float* pod32, const float* e, float* lo, float* hi) {
  for (int64_t p = 0; p < n_pods; ++p)
    for (int64_t i = 0; i < n_nodes; ++i) {
      const float* c = consts + 5 * i;
      const spx::peaks_est::NodeE ne{c[0], c[1], c[2], c[3], c[4]};
      const int64_t k = p * n_nodes + i;
      spx::peaks_est::est_interval_from(ne, spx::peaks_est::est_predicted(ne, pod32[p]), spx::peaks_est::est_exponent(ne, pod32[p]), e[k], lo[k], hi[k]);
    }
}

// est_interval itself (host exponential): for the smoke comparison with the replay's unperturbed intervals
void peaks_est_check_intervals_host_exp(int64_t n_pods, int64_t n_nodes, const float* consts, const float* pod32, float* lo, float* hi) {
  for (int64_t p = 0; p < n_pods; ++p)
    for (int64_t i = 0; i < n_nodes; ++i) {
      const float* c = consts + 5 * i;
      const spx::peaks_est::NodeE ne{c[0], c[1], c[2], c[3], c[4]};
      spx::peaks_est::est_interval(ne, pod32[p], lo[p * n_nodes + i], hi[p * n_nodes + i]);
    }
}

}  // extern "C"
