// trimaran_math.h — TargetLoadPacking arithmetic shared by the sweep kernels (kernels_trimaran.hip) and the sequential
// commit kernels (kernels_commit_trimaran.hip): the reference's float64 sequence and the float32 formulation's per-node
// constants.  Everything here is inlined device code; the translation units are compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spx {
namespace trimath {

// ------------------------------------------------------------------------------------------------
// per-node state held in registers

struct TlpNode {
  double util_millis;  // (util% / 100) * cap   targetloadpacking.go:147
  double missing;      // float64(missingCPUUtilMillis)
  double cap;          // float64(Capacity.Cpu().MilliValue())
  bool valid;          // metrics != nil && cpuMetricFound
};

__device__ __forceinline__ double tlp_predicted(const TlpNode& n, double pod_milli) {
  double predicted = 0.0;
  if (n.cap != 0.0) predicted = 100.0 * ((n.util_millis + pod_milli) + n.missing) / n.cap;  // :170-173
  return predicted;
}

// float64 value the reference rounds; *zero is set when the reference returns MinNodeScore outright
__device__ __forceinline__ double tlp_unrounded(const TlpNode& n, double pod_milli, double t, bool* zero) {
  *zero = false;
  if (!n.valid) {
    *zero = true;
    return 0.0;
  }
  const double predicted = tlp_predicted(n, pod_milli);
  if (predicted > t) {  // :174-181
    if (predicted > 100.0) {
      *zero = true;
      return 0.0;
    }
    return t * (100.0 - predicted) / (100.0 - t);
  }
  return (100.0 - t) * predicted / t + t;  // :183-184
}

__device__ __forceinline__ uint32_t to_u8(double unrounded) {
  // int64(math.Round(x)) then saturate into the uint8 table cell
  int v = static_cast<int>(round(unrounded));
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  return static_cast<uint32_t>(v);
}

// float32 formulation (derivation: kernels_trimaran.hip, k_tlp_fast2): a cell is provably the reference's result when its
// value is at least kTol32 away from a rounding tie and |u| at least kTolU away from the branch point
constexpr float kTol32 = 4e-5f;
constexpr float kTolU = 1e-6f;

// float32 constants of one node for the TLP fast formula (same derivation as k_tlp_prepare_fast / k_tlp_fast2):
// (b2h, b2l, coefficient for u > 0, coefficient for u <= 0); NaN b2h = always the exact path
__device__ __forceinline__ float4 tlp_fast_consts(double cap, double util_pct, double missing, bool valid, double t, double c1, double c2) {
  double b = 1e30;
  float f1 = -1.0f, f2 = 0.0f;
  bool split = false;
  if (valid) {
    const double um = (util_pct / 100.0) * cap;
    if (cap == 0.0) {
      b = 1.0;
      f1 = 0.0f;
    } else if (!(um >= 0.0) || !(missing >= 0.0) || !(cap > 0.0) || !(um < 1e15) || !(missing < 1e15)) {
      b = __builtin_nan("");
    } else {
      const double k = 100.0 / cap;
      b = (um + missing) - t * cap / 100.0;
      f1 = static_cast<float>(-c1 * k);
      f2 = static_cast<float>(c2 * k);
      split = __builtin_fabs(b) < 8388607.0;
      if (!split) b = __builtin_nan("");
    }
  }
  const double bh = split ? __builtin_rint(b) : b;
  return float4{static_cast<float>(bh), split ? static_cast<float>(b - bh) : 0.0f, f1, f2};
}
}  // namespace trimath
}  // namespace spx
