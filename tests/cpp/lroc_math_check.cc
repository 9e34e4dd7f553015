// lroc_math_check.cc — TEST-ONLY host build of scheduler-plugins_amd/csrc/lroc_math.h (the source k_lroc_prepare
// compiles for the device), exported with C linkage so tests/test_lroc_math_host.py can compare it with the oracle on
// the CPU.  Not part of libspx.so: the product evaluates this header on the GPU only.
#include "../../scheduler-plugins_amd/csrc/lroc_math.h"

extern "C" double lroc_check_reg_inc_beta(double a, double b, double x) { return spx::lroc::reg_inc_beta(a, b, x); }
extern "C" double lroc_check_beta_cdf(double a, double b, double x) { return spx::lroc::beta_cdf(a, b, x); }
extern "C" double lroc_check_risk_load(int valid, double capacity_stat, double avg, double stdev, int64_t capacity, int64_t requested,
                                       int64_t limits, double sqrt_window) {
  spx::lroc::NodeResource r;
  r.metric_valid = valid != 0;
  r.capacity_stat = capacity_stat;
  r.avg = avg;
  r.stdev = stdev;
  r.capacity = capacity;
  r.requested = requested;
  r.limits = limits;
  return spx::lroc::risk_load(r, sqrt_window);
}
