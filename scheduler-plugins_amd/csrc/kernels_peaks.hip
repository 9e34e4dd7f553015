// kernels_peaks.hip — trimaran Peaks on gfx950.
//
// Reference: Peaks.Score pkg/trimaran/peaks/peaks.go:103-144 once per (pod, node) — the jump in node power
// K1 * (e^(K2 * predicted) - e^(K2 * current)) scaled by 1e15 and truncated to int64 — then Peaks.NormalizeScore
// (:150-166) once per pod over its node list: min-max rescale to 0..100, inverted (the smallest jump scores 100).
//
// The normalisation needs each pod row's min and max before any byte of the row can be written, while the sweeps of this
// engine keep node state in registers and walk pods (a wave owns 256 nodes x 64 pods).  So the row statistic is gathered
// across tiles: three launches on the engine stream,
//   k_peaks_init     row_min / row_max <- +inf / -inf
//   k_peaks<false>   raw scores of the wave's nodes for each of its pods, reduced over the wave (min, max of the
//                    feasible nodes), one 64-bit atomic min and max per (tile, pod)
//   k_peaks<true>    the same raw scores again (recomputing beats storing 8 B per cell), normalised against the row's
//                    min/max, one byte per cell written.
// Per cell and pass: one float64 division, one exp, one trunc (the int64 conversion is never materialised, see raw_score);
// the write pass adds the normalising division.  VALU-bound, like LowRiskOverCommitment.
// The arithmetic is the reference's, operation for operation; exp is OCML's, so raw scores can differ from a Go
// evaluation in the last digits (relative ~1e-16) and normalised scores by 1 at an exact truncation boundary.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>

#include "spx_internal.h"

namespace spx {
namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kPodsPerChunk = 64;
// nodes per lane (template parameter of k_peaks): 4 = one dword of scores and of each status table per pod row, 8 = two
constexpr double kInf = __builtin_huge_val();

template <typename T>
__device__ __forceinline__ T uload(const T* p) {  // wave-uniform read of immutable input -> scalar load
  typedef const T __attribute__((address_space(4))) CT;
  return *reinterpret_cast<CT*>(reinterpret_cast<uintptr_t>(p));
}

// a / b correctly rounded from y = RN(1/b): q = RN(a*y) is within an ulp of the quotient, the residual a - b*q is exact in
// one fma, and RN(q + r*y) is RN(a/b) (Markstein's theorem; the exception, a divisor whose significand is all ones, cannot
// occur for the integer-valued divisors used here).  Three full-rate instructions instead of the ~14 of the IEEE division
// sequence; tests/test_exactness_arguments.py replays it against true division in exact rational arithmetic.
__device__ __forceinline__ double div_rn(double a, double b, double y) {
  const double q = a * y;
  const double r = fma(-b, q, a);
  return fma(r, y, q);
}

struct NodeP {
  double cap;       // float64(node.Status.Capacity.Cpu().MilliValue())   peaks.go:132
  double rcap;      // RN(1 / cap), for div_rn
  double util_m;    // (util / 100) * cap                                  :133
  double k1, k2;    // power model                                          :190-196
  double e_now;     // exp(K2 * util)                                       :187
  bool valid;       // metrics present and a CPU AVG/Latest metric found   :108-131
};

__device__ __forceinline__ NodeP load_node(const PeaksArgs& a, int64_t n) {
  NodeP nd;
  const bool in = n < a.n_nodes;
  nd.valid = in && a.valid[n] != 0;
  nd.cap = in ? static_cast<double>(a.cap_cpu_milli[n]) : 0.0;
  const double util = in ? a.cpu_util[n] : 0.0;
  nd.rcap = 1.0 / nd.cap;
  nd.util_m = (util / 100) * nd.cap;
  nd.k1 = in ? a.k1[n] : 0.0;
  nd.k2 = in ? a.k2[n] : 0.0;
  nd.e_now = exp(nd.k2 * util);
  return nd;
}

// Peaks.Score for one node given float64(curPodCPUUsage), as an integer-valued float64: int64(x) truncates toward zero and
// |x| < 2^63 here, so trunc(x) is that int64 exactly (a float64 at or above 2^53 is an integer already).  Staying in float64
// saves the two multi-instruction conversions per cell; differences of two such values taken in float64 are the correctly
// rounded exact difference, i.e. the very float64(score - minCost) the reference forms (peaks.go:158).
__device__ __forceinline__ double raw_score(const NodeP& nd, double pod_cpu) {
  double predicted = 0.0;
  if (nd.cap != 0) predicted = div_rn(100 * (nd.util_m + pod_cpu), nd.cap, nd.rcap);  // :135-138
  const double jump = nd.k1 * (exp(nd.k2 * predicted) - nd.e_now);     // :186-188
  const double v = trunc(jump * 1e15);                                 // :143
  return (nd.valid && !(predicted > 100)) ? v : 0.0;                   // :108-112, :128-131, :139-140
}

__global__ void k_peaks_init(int64_t* row_min, int64_t* row_max, int64_t row_begin, int64_t row_end) {
  const int64_t i = row_begin + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < row_end) {
    row_min[i] = INT64_MAX;
    row_max[i] = INT64_MIN;
  }
}

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  const int lo = __shfl_xor(__double2loint(v), m);
  const int hi = __shfl_xor(__double2hiint(v), m);
  return __hiloint2double(hi, lo);
}

// feasibility of the lane's NPL nodes for `pod` (bit j set = node j does not count): every Filter status table must say 0;
// columns past n_nodes never count
template <int NPL>
__device__ __forceinline__ uint32_t infeasible_mask(const PeaksArgs& a, int64_t pod, int64_t node0, bool active) {
  uint32_t m = 0;
#pragma unroll
  for (int w = 0; w < NPL / 4; ++w) {
    uint32_t bad = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (a.other_status[t] != nullptr && active)
        bad |= *reinterpret_cast<const uint32_t*>(a.other_status[t] + pod * a.row_stride + node0 + 4 * w);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 4 * w + q;
      const bool out = !active || node0 + j >= a.n_nodes || ((bad >> (8 * q)) & 0xffu) != 0;
      m |= (out ? 1u : 0u) << j;
    }
  }
  return m;
}

template <bool kWrite, int kNpl>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_peaks(PeaksArgs a, int n_tiles) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  // rows of this chunk: positions [i0, i1) of the row list when there is one, else of [row_begin, row_end)
  const int64_t n_rows = a.row_list ? a.n_list : a.row_end - a.row_begin;
  const int64_t i0 = chunk * kPodsPerChunk;
  if (i0 >= n_rows) return;  // wave-uniform
  const int64_t i1 = (i0 + kPodsPerChunk < n_rows) ? i0 + kPodsPerChunk : n_rows;
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * kNpl;
  const bool active = node0 < a.row_stride;  // all 64 lanes stay in the loop: the reduction below shuffles across them

  NodeP nd[kNpl];
#pragma unroll
  for (int j = 0; j < kNpl; ++j) nd[j] = load_node(a, active ? node0 + j : a.n_nodes);

  for (int64_t i = i0; i < i1; ++i) {
    const int64_t pod = a.row_list ? static_cast<int64_t>(uload(a.row_list + i)) : a.row_begin + i;
    const double pod_cpu = static_cast<double>(uload(a.pod_cpu_milli + pod));
    const uint32_t bad = infeasible_mask<kNpl>(a, pod, node0, active);
    double raw[kNpl];
#pragma unroll
    for (int j = 0; j < kNpl; ++j) raw[j] = raw_score(nd[j], pod_cpu);
    if constexpr (!kWrite) {
      double mn = kInf, mx = -kInf;
#pragma unroll
      for (int j = 0; j < kNpl; ++j) {
        if (!((bad >> j) & 1u)) {
          mn = fmin(mn, raw[j]);
          mx = fmax(mx, raw[j]);
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        mn = fmin(mn, shfl_xor_f64(mn, m));
        mx = fmax(mx, shfl_xor_f64(mx, m));
      }
      if (lane == 0 && mn <= mx) {  // at least one feasible node in this tile; the values are integers below 2^63
        atomicMin(reinterpret_cast<long long*>(a.row_min + pod), static_cast<long long>(mn));
        atomicMax(reinterpret_cast<long long*>(a.row_max + pod), static_cast<long long>(mx));
      }
    } else {
      const int64_t mni = uload(a.row_min + pod), mxi = uload(a.row_max + pod);
      uint32_t word[kNpl / 4] = {};
      if (!(mni == 0 && mxi == 0)) {  // :152-154: all raw scores are 0 and stay 0
        const double mn = static_cast<double>(mni);          // exact: it was stored from an integer-valued float64
        const double span = static_cast<double>(mxi - mni);  // float64(maxCost - minCost)
        const bool flat = mxi == mni;
        const double rspan = 1.0 / span;                      // once per pod and wave
#pragma unroll
        for (int j = 0; j < kNpl; ++j) {
          const double diff = raw[j] - mn;                   // float64(score - minCost), see raw_score
          const double norm = flat ? diff : div_rn(100.0 * diff, span, rspan);  // :158, :161
          const int sc = 100 - static_cast<int>(norm);       // :159, :162 (|norm| <= 100 for a feasible node)
          const uint32_t b = ((bad >> j) & 1u) ? 0u : static_cast<uint32_t>(sc < 0 ? 0 : (sc > 100 ? 100 : sc));
          word[j >> 2] |= b << (8 * (j & 3));
        }
      }
      if (active) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(a.out_score + pod * a.row_stride + node0);
        if constexpr (kNpl == 8) {
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<u32x2*>(dst) = u32x2{word[0], word[1]};
        } else {
          dst[0] = word[0];
        }
      }
    }
  }
}

__global__ void k_peaks_raw(PeaksArgs a) {  // Score() of one pod row as int64 (spx_fetch_raw)
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.n_nodes) return;
  a.out_raw[n] = static_cast<int64_t>(raw_score(load_node(a, n), static_cast<double>(a.pod_cpu_milli[a.row_begin])));
}

}  // namespace

void launch_peaks(const PeaksArgs& a, hipStream_t s) {
  if (a.out_raw != nullptr) {
    hipLaunchKernelGGL(k_peaks_raw, dim3(static_cast<unsigned>((a.n_nodes + 255) / 256)), dim3(256), 0, s, a);
    return;
  }
  const int64_t rows = a.row_end - a.row_begin;
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_peaks_init, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0, s, a.row_min, a.row_max, a.row_begin, a.row_end);
  const int64_t swept = a.row_list ? a.n_list : rows;
  const int64_t chunks = (swept + kPodsPerChunk - 1) / kPodsPerChunk;
  auto grid = [&](int npl, int* n_tiles) {
    const int tile_nodes = kWave * npl;
    *n_tiles = static_cast<int>((a.row_stride + tile_nodes - 1) / tile_nodes);
    return dim3(static_cast<unsigned>((chunks * *n_tiles + kWavesPerBlock - 1) / kWavesPerBlock));
  };
  // experiment knob (SPX_OPT_PEAKS_TILE): 44, 84, 48, 88 = nodes per lane of (min/max pass, write pass).  Measured on 10k x 100k: 3.49 /
  // 3.50 / 3.55 / 3.57 ms — the wider tiles halve the cross-lane reductions per cell but cost occupancy (145 VGPRs): no gain
  const int npl_a = (a.opts & kOptPeaksWideA) ? 8 : 4, npl_b = (a.opts & kOptPeaksWideB) ? 8 : 4;
  int nt;
  if (npl_a == 8) {
    const dim3 g = grid(8, &nt);
    hipLaunchKernelGGL((k_peaks<false, 8>), g, dim3(kWave * kWavesPerBlock), 0, s, a, nt);
  } else {
    const dim3 g = grid(4, &nt);
    hipLaunchKernelGGL((k_peaks<false, 4>), g, dim3(kWave * kWavesPerBlock), 0, s, a, nt);
  }
  if (npl_b == 8) {
    const dim3 g = grid(8, &nt);
    hipLaunchKernelGGL((k_peaks<true, 8>), g, dim3(kWave * kWavesPerBlock), 0, s, a, nt);
  } else {
    const dim3 g = grid(4, &nt);
    hipLaunchKernelGGL((k_peaks<true, 4>), g, dim3(kWave * kWavesPerBlock), 0, s, a, nt);
  }
}

}  // namespace spx
