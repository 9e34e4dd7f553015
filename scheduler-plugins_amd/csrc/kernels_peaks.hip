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
//
// Round 5 (default, SPX_OPT_PEAKS_ESTIMATE): the same two passes with a float32 interval per cell in front of that arithmetic —
// k_peaks_nodetab, k_peaks_minmax_est + k_peaks_fix_minmax, k_peaks_rowconst, k_peaks_write_est + k_peaks_fix_write further down;
// the float64 sequence above (raw_score) then runs only for the cells an interval cannot decide.  Same tables, byte for byte.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>

#include "spx_internal.h"
#include "peaks_est.h"

namespace spx {
namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kPodsPerChunk = 64;
// nodes per lane (template parameter of k_peaks): 4 = one dword of scores and of each status table per pod row, 8 = two
constexpr double kInf = __builtin_huge_val();
using peaks_est::NodeE;
using peaks_est::est_interval;
using peaks_est::est_node_compute;

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

// ------------------------------------------------------------------------------------------------------------------------------
// Interval estimates (round 5, SPX_OPT_PEAKS_ESTIMATE, default on): the arithmetic and its error budget are in peaks_est.h.
// pod rows per chunk (PeaksArgs::est_pods, a power of two in [16, 128] chosen per launch by peaks_est_plan so that the sweep has some 8 waves per
// SIMD to offer: one row per distinct request of config #2 is 12 120 rows x 10 tiles — at 128 rows per chunk that is 950 waves for 1024 SIMDs)
constexpr int kEstPodsMax = 128, kEstPodsMin = 16;
constexpr int64_t kEstWavesWanted = 8192;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Per-node tables of the launches below (k_peaks_nodetab, once per launch_peaks; S = row_stride):
//   doubles [8][S]   cap, RN(1 / cap), (util / 100) * cap, exp(K2 * util) — load_node's own expressions — K1, K2, valid, and raw_score for a
//                    pod that requests no cpu (every such row is this row); column-major: the tile launches read 64 consecutive nodes per load
//   floats  [S][8]   the interval's constants (est_node), one 32-byte row per node: the sweeps read 16 consecutive nodes per lane
constexpr int kTabCols = 8;
__device__ __forceinline__ const float* est_tab(const PeaksArgs& a) { return reinterpret_cast<const float*>(a.node_tab + kTabCols * a.row_stride); }
__device__ __forceinline__ NodeP load_node_tab(const PeaksArgs& a, int64_t n) {  // n < n_nodes
  NodeP nd;
  const double* t = a.node_tab + n;
  const int64_t S = a.row_stride;
  nd.cap = t[0], nd.rcap = t[S], nd.util_m = t[2 * S], nd.e_now = t[3 * S], nd.k1 = t[4 * S], nd.k2 = t[5 * S];
  nd.valid = t[6 * S] != 0.0;
  return nd;
}

// wave-wide max / min of a float32 (NaN lanes are ignored: v_max_f32 / v_min_f32 return the other operand); DPP steps inside the
// 16-lane rows as wave_umax (kernels_trimaran.hip), the four rows through v_readlane.  Every lane must be live.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <bool kMax>
__device__ __forceinline__ float wave_fminmax(float v) {
  auto op = [](float x, float y) { return kMax ? __builtin_fmaxf(x, y) : __builtin_fminf(x, y); };
  v = op(v, dpp_f32<0xB1>(v));
  v = op(v, dpp_f32<0x4E>(v));
  v = op(v, dpp_f32<0x141>(v));
  v = op(v, dpp_f32<0x140>(v));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return op(op(r0, r1), op(r2, r3));
}

__global__ void k_peaks_nodetab(PeaksArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.n_nodes) return;
  const NodeP nd = load_node(a, n);
  double* t = a.node_tab + n;
  const int64_t S = a.row_stride;
  t[0] = nd.cap, t[S] = nd.rcap, t[2 * S] = nd.util_m, t[3 * S] = nd.e_now, t[4 * S] = nd.k1, t[5 * S] = nd.k2, t[6 * S] = nd.valid ? 1.0 : 0.0;
  t[7 * S] = raw_score(nd, 0.0);
  const NodeE ne = est_node_compute(nd, a.cpu_util[n]);
  f32x4* et = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.node_tab + kTabCols * S) + n * 8);
  et[0] = f32x4{ne.c0, ne.c1, ne.ql, ne.ke};
  et[1] = f32x4{ne.sigma, 0.0f, 0.0f, 0.0f};
}

__device__ __forceinline__ NodeE est_node(const PeaksArgs& a, int64_t n, bool counted) {
  NodeE ne;
  const float q = __builtin_nanf("");
  ne.c0 = ne.c1 = ne.ql = ne.ke = ne.sigma = q;
  if (!counted) return ne;
  const f32x4* et = reinterpret_cast<const f32x4*>(est_tab(a) + n * 8);
  const f32x4 u = et[0];
  ne.c0 = u.x, ne.c1 = u.y, ne.ql = u.z, ne.ke = u.w;
  ne.sigma = et[1].x;
  return ne;
}

// The cells an estimate pass cannot decide leave it in one of two ways (per wave and pod row, wave-uniform):
//   sparse   a handful of lanes hold one or two such cells (the usual case: the tile's largest score, the cells next to a step of the
//            normalised score): they are APPENDED to the wave's private segment of a list — (row, node) pairs, no counter shared
//            between waves — and a second launch (k_peaks_fix_*) runs raw_score on the listed cells with every lane busy.  Evaluating
//            them where they are found costs a dependent table read and ~100 float64 instructions with one lane of 64 active per cell:
//            measured, 1.7 + 1.2 ms on top of the 0.8 ms the estimates take for 10 000 x 100 000;
//   dense    many (a pod that requests no cpu: every jump is rounding noise and the interval decides nothing; a tile of nodes outside
//            the preconditions; a full segment): ONE entry (row, -1) names the whole tile, and the second launch runs every cell of it
//            through raw_score, a wave per tile as k_peaks does.  The sweeps themselves hold no float64 code: 2 -> 4-5 waves per SIMD.
constexpr int kEstSegPerPod = 3;                       // segment entries per pod row of a chunk (a segment: 3 * est_pods entries)
constexpr int kEstSparseLanes = 12;                    // more lanes than this with undecided cells: dense

struct SegEntry {
  int32_t row, node;
};

__device__ __forceinline__ int lanes_below(unsigned long long mask) {  // set bits of mask below this lane
  return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u)));
}

template <int kNpl, bool kMask>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_peaks_minmax_est(PeaksArgs a, int n_tiles) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t n_rows = a.row_list ? a.n_list : a.row_end - a.row_begin;
  const int est_pods = a.est_pods, seg_cap = kEstSegPerPod * est_pods;  // uniform
  const int64_t i0 = chunk * est_pods;
  if (i0 >= n_rows) return;  // wave-uniform
  const int64_t i1 = (i0 + est_pods < n_rows) ? i0 + est_pods : n_rows;
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * kNpl;
  const bool active = node0 < a.row_stride;  // all 64 lanes stay in the loop: the reductions below run across them
  SegEntry* const seg = reinterpret_cast<SegEntry*>(a.seg) + unit * seg_cap;
  int seg_n = 0;  // wave-uniform

  NodeE ne[kNpl];
#pragma unroll
  for (int j = 0; j < kNpl; ++j) ne[j] = est_node(a, node0 + j, active && node0 + j < a.n_nodes);

  for (int64_t i = i0; i < i1; ++i) {
    const int64_t pod = a.row_list ? static_cast<int64_t>(uload(a.row_list + i)) : a.row_begin + i;
    const int64_t pod_i = uload(a.pod_cpu_milli + pod);
    const float pod32 = static_cast<float>(pod_i);
    uint32_t bad = 0;
    if constexpr (kMask) bad = infeasible_mask<kNpl>(a, pod, node0, active);
    float lo[kNpl], hi[kNpl];
    float maxlo = -__builtin_inff(), minhi = __builtin_inff();
#pragma unroll
    for (int j = 0; j < kNpl; ++j) {
      est_interval(ne[j], pod32, lo[j], hi[j]);
      if constexpr (kMask) {
        if ((bad >> j) & 1u) lo[j] = hi[j] = __builtin_nanf("");
      }
      maxlo = __builtin_fmaxf(maxlo, lo[j]);
      minhi = __builtin_fminf(minhi, hi[j]);
    }
    maxlo = wave_fminmax<true>(maxlo);
    minhi = wave_fminmax<false>(minhi);
    // undecided cells: the row's extremes over this tile are among them and the cells known to score exactly 0
    uint32_t cand = 0;
    bool zero = false;
#pragma unroll
    for (int j = 0; j < kNpl; ++j) {
      const bool c = (hi[j] >= maxlo) || (lo[j] <= minhi);
      cand |= (c && hi[j] > lo[j]) ? (1u << j) : 0u;
      zero = zero || (lo[j] == hi[j]);
    }
#if defined(SPX_PEAKS_DIAG) && (SPX_PEAKS_DIAG & 1)
    cand = 0;  // (timing experiment: no float64 evaluations for the min/max pass — wrong tables)
#endif
    const unsigned long long any = __ballot(cand != 0u);
    const int n_any = __builtin_popcountll(any);
    const bool multi = __ballot((cand & (cand - 1u)) != 0u) != 0ull;
    const bool zero_any = __ballot(zero) != 0ull;
    const int left = static_cast<int>(i1 - i - 1);  // every later row of the chunk keeps room for one entry
    if (!multi && n_any <= kEstSparseLanes && seg_n + n_any + left <= seg_cap) {  // uniform
      if (cand != 0u) seg[seg_n + lanes_below(any)] = SegEntry{static_cast<int32_t>(pod), static_cast<int32_t>(node0 + __builtin_ctz(cand))};
      seg_n += n_any;
      if (lane == 0 && zero_any) {
        atomicMin(reinterpret_cast<long long*>(a.row_min + pod), 0ll);
        atomicMax(reinterpret_cast<long long*>(a.row_max + pod), 0ll);
      }
    } else {  // the whole tile, by k_peaks_fix_minmax
      if (lane == 0) seg[seg_n] = SegEntry{static_cast<int32_t>(pod), -1};
      seg_n += 1;
    }
  }
  if (lane == 0) a.seg_n[unit] = seg_n;
}

// whether node `node` counts for row `pod`: inside the table and passed by every Filter status table in play
template <bool kMask>
__device__ __forceinline__ bool node_counts(const PeaksArgs& a, int64_t pod, int64_t node) {
  bool c = node < a.n_nodes;
  if constexpr (kMask) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (a.other_status[t] != nullptr && c) c = a.other_status[t][pod * a.row_stride + node] == 0;
  }
  return c;
}

// The second launch of the min/max pass, a wave per segment: the listed cells through raw_score, one lane per cell, into the row
// statistic; a listed TILE by the whole wave — lane l takes the nodes base + 64 j + l (consecutive lanes, consecutive nodes: every table
// column is read in whole lines), and the rows of pods that request no cpu read their raw scores from the table's last column.
template <int kNpl, bool kMask>
__global__ __launch_bounds__(kWave) void k_peaks_fix_minmax(PeaksArgs a, int n_tiles) {
  const int lane = threadIdx.x;
  const int n = uload(a.seg_n + blockIdx.x);
  const SegEntry* seg = reinterpret_cast<const SegEntry*>(a.seg) + static_cast<int64_t>(blockIdx.x) * (kEstSegPerPod * a.est_pods);
  const int tile = static_cast<int>(blockIdx.x % static_cast<unsigned>(n_tiles));
  const int64_t base = static_cast<int64_t>(tile) * kWave * kNpl + lane;
  const double* raw0 = a.node_tab + 7 * a.row_stride;
  for (int k0 = 0; k0 < n; k0 += kWave) {  // uniform
    const bool have = k0 + lane < n;
    SegEntry e = SegEntry{0, 0};
    if (have) e = seg[k0 + lane];
    const bool is_tile = have && e.node < 0;
    if (have && !is_tile) {
      const double v = raw_score(load_node_tab(a, e.node), static_cast<double>(a.pod_cpu_milli[e.row]));
      atomicMin(reinterpret_cast<long long*>(a.row_min + e.row), static_cast<long long>(v));
      atomicMax(reinterpret_cast<long long*>(a.row_max + e.row), static_cast<long long>(v));
    }
    unsigned long long tm = __ballot(is_tile);
    while (tm != 0ull) {  // uniform
      const int src = __builtin_ctzll(tm);
      tm &= tm - 1ull;
      const int64_t pod = __builtin_amdgcn_readlane(e.row, src);
      const int64_t pod_i = uload(a.pod_cpu_milli + pod);
      double mn = kInf, mx = -kInf;
      if (pod_i == 0) {  // uniform: all the loads first
        double r[kNpl];
        bool c[kNpl];
#pragma unroll
        for (int j = 0; j < kNpl; ++j) {
          c[j] = node_counts<kMask>(a, pod, base + j * kWave);
          r[j] = c[j] ? raw0[base + j * kWave] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kNpl; ++j) {
          mn = c[j] ? fmin(mn, r[j]) : mn;
          mx = c[j] ? fmax(mx, r[j]) : mx;
        }
      } else {
        const double pod_cpu = static_cast<double>(pod_i);
#pragma unroll 2
        for (int j = 0; j < kNpl; ++j) {
          const int64_t node = base + j * kWave;
          if (node_counts<kMask>(a, pod, node)) {
            const double v = raw_score(load_node_tab(a, node), pod_cpu);
            mn = fmin(mn, v);
            mx = fmax(mx, v);
          }
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        mn = fmin(mn, shfl_xor_f64(mn, m));
        mx = fmax(mx, shfl_xor_f64(mx, m));
      }
      if (lane == 0 && mn <= mx) {
        atomicMin(reinterpret_cast<long long*>(a.row_min + pod), static_cast<long long>(mn));
        atomicMax(reinterpret_cast<long long*>(a.row_max + pod), static_cast<long long>(mx));
      }
    }
  }
}

// per swept row, once the row statistic is complete: what the write pass needs of it in float32 — r = 100 / span, and the two
// constants of n(x) = fma(x, r, c): c_lo / c_hi = -min * r -+ en, en covering the roundings of r, of min * r and of the fma for results in
// [-1, 101] (outside that range both ends of a cell's interval land on the same side of every integer that matters).
// kind: 0 = every raw score of the row is 0 (peaks.go:152-154), 1 = min == max, 2 = the general case
__global__ void k_peaks_rowconst(PeaksArgs a) {
  const int64_t n_rows = a.row_list ? a.n_list : a.row_end - a.row_begin;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const int64_t pod = a.row_list ? static_cast<int64_t>(a.row_list[i]) : a.row_begin + i;
  const int64_t mni = a.row_min[pod], mxi = a.row_max[pod];
  f32x4 c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (mni == 0 && mxi == 0) {
  } else if (mni == mxi) {
    c.w = 1.0f;
  } else {
    const double mn = static_cast<double>(mni);
    const double span = static_cast<double>(static_cast<int64_t>(static_cast<uint64_t>(mxi) - static_cast<uint64_t>(mni)));
    const double r = 100.0 / span, mr = mn * r;
    const float en = static_cast<float>(6.0 * 0x1p-24 * (fabs(mr) + 101.0));
    c.x = static_cast<float>(r);
    c.y = static_cast<float>(-mr) - en;
    c.z = static_cast<float>(-mr) + en;
    c.w = 2.0f;
  }
  reinterpret_cast<f32x4*>(a.row_c)[pod] = c;
}

template <int kNpl, bool kMask>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_peaks_write_est(PeaksArgs a, int n_tiles) {
  constexpr int kWords = kNpl / 4;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t n_rows = a.row_list ? a.n_list : a.row_end - a.row_begin;
  const int est_pods = a.est_pods, seg_cap = kEstSegPerPod * est_pods;  // uniform
  const int64_t i0 = chunk * est_pods;
  if (i0 >= n_rows) return;  // wave-uniform
  const int64_t i1 = (i0 + est_pods < n_rows) ? i0 + est_pods : n_rows;
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * kNpl;
  const bool active = node0 < a.row_stride;
  SegEntry* const seg = reinterpret_cast<SegEntry*>(a.seg) + unit * seg_cap;
  int seg_n = 0;  // wave-uniform

  NodeE ne[kNpl];
  uint32_t keep0[kWords];  // 0xff per column inside the table
#pragma unroll
  for (int w = 0; w < kWords; ++w) keep0[w] = 0u;
#pragma unroll
  for (int j = 0; j < kNpl; ++j) {
    const bool counted = active && node0 + j < a.n_nodes;
    ne[j] = est_node(a, node0 + j, counted);
    keep0[j >> 2] |= counted ? (0xffu << (8 * (j & 3))) : 0u;
  }

  for (int64_t i = i0; i < i1; ++i) {
    const int64_t pod = a.row_list ? static_cast<int64_t>(uload(a.row_list + i)) : a.row_begin + i;
    const f32x4 rc = uload(reinterpret_cast<const f32x4*>(a.row_c) + pod);
    uint32_t word[kWords];
#pragma unroll
    for (int w = 0; w < kWords; ++w) word[w] = 0u;
    if (rc.w != 0.0f) {  // uniform
      uint32_t keep[kWords];
#pragma unroll
      for (int w = 0; w < kWords; ++w) {
        keep[w] = keep0[w];
        if constexpr (kMask) {
          uint32_t st = 0;
#pragma unroll
          for (int t = 0; t < 3; ++t)
            if (a.other_status[t] != nullptr && active) st |= *reinterpret_cast<const uint32_t*>(a.other_status[t] + pod * a.row_stride + node0 + 4 * w);
          const uint32_t nz = (((st & 0x7f7f7f7fu) + 0x7f7f7f7fu) | st) & 0x80808080u;  // bit 7 of every non-zero byte
          keep[w] &= ~((nz >> 7) * 0xffu);
        }
      }
      if (rc.w == 1.0f) {  // min == max: every feasible node's raw score is the minimum, norm = 0
#pragma unroll
        for (int w = 0; w < kWords; ++w) word[w] = 0x64646464u;
      } else {
        const int64_t pod_i = uload(a.pod_cpu_milli + pod);
        const float pod32 = static_cast<float>(pod_i);
        uint32_t amb = 0;
#pragma unroll
        for (int j = 0; j < kNpl; ++j) {
          float lo, hi;
          est_interval(ne[j], pod32, lo, hi);
          const float nlo = __builtin_fmaf(lo, rc.x, rc.y), nhi = __builtin_fmaf(hi, rc.x, rc.z);
          const uint32_t klo = static_cast<uint32_t>(__builtin_fminf(__builtin_fmaxf(nlo, 0.0f), 1.0e6f));
          const uint32_t khi = static_cast<uint32_t>(__builtin_fminf(__builtin_fmaxf(nhi, 0.0f), 1.0e6f));
          amb |= (klo != khi) ? (1u << j) : 0u;
          const uint32_t sc = 100u - (khi < 100u ? khi : 100u);
          word[j >> 2] |= sc << (8 * (j & 3));
        }
        // the undecided cells of columns that count
        uint32_t feas = 0;
#pragma unroll
        for (int w = 0; w < kWords; ++w) feas |= ((((keep[w] & 0x01010101u) * 0x00204081u) >> 21) & 0xfu) << (4 * w);
        amb &= feas;
#if defined(SPX_PEAKS_DIAG) && (SPX_PEAKS_DIAG & 2)
        amb = 0;  // (timing experiment: no float64 evaluations for the write pass — wrong tables)
#endif
        unsigned long long any = __ballot(amb != 0u);
        if (any != 0ull) {  // uniform
          const int left = static_cast<int>(i1 - i - 1);  // every later row of the chunk keeps room for one entry
          const int mark = seg_n;
          bool tile_entry = __builtin_popcountll(any) > kEstSparseLanes;
          while (!tile_entry && any != 0ull) {  // listed: the first undecided cell of every lane that has one, until none is left
            const int n_any = __builtin_popcountll(any);
            if (seg_n + n_any + left > seg_cap) {
              tile_entry = true;
              break;
            }
            if (amb != 0u) {
              seg[seg_n + lanes_below(any)] = SegEntry{static_cast<int32_t>(pod), static_cast<int32_t>(node0 + __builtin_ctz(amb))};
              amb &= amb - 1u;
            }
            seg_n += n_any;
            any = __ballot(amb != 0u);
          }
          if (tile_entry) {  // the whole tile, by k_peaks_fix_write (cells listed so far for this row are dropped again)
            seg_n = mark;
            if (lane == 0) seg[seg_n] = SegEntry{static_cast<int32_t>(pod), -1};
            seg_n += 1;
          }
        }
      }
#pragma unroll
      for (int w = 0; w < kWords; ++w) word[w] &= keep[w];
    }
    if (active) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(a.out_score + pod * a.row_stride + node0);
      if constexpr (kWords == 4) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(dst) = u32x4{word[0], word[1], word[2], word[3]};
      } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(dst) = u32x2{word[0], word[1]};
      }
    }
  }
  if (lane == 0) a.seg_n[unit] = seg_n;
}

// NormalizeScore of a raw score (k_peaks<true>'s expressions; span != 0 on this path)
__device__ __forceinline__ uint32_t norm_byte(double raw, double mn, double span, double rspan) {
  const double norm = div_rn(100.0 * (raw - mn), span, rspan);  // :161
  const int sc = 100 - static_cast<int>(norm);
  return static_cast<uint32_t>(sc < 0 ? 0 : (sc > 100 ? 100 : sc));
}

// The second launch of the write pass, a wave per segment: the listed cells through the float64 NormalizeScore, one lane per cell, one
// byte each into the table; a listed tile by the whole wave as in k_peaks_fix_minmax, every column of the tile written (0 where the
// node does not count).
template <int kNpl, bool kMask>
__global__ __launch_bounds__(kWave) void k_peaks_fix_write(PeaksArgs a, int n_tiles) {
  const int lane = threadIdx.x;
  const int n = uload(a.seg_n + blockIdx.x);
  const SegEntry* seg = reinterpret_cast<const SegEntry*>(a.seg) + static_cast<int64_t>(blockIdx.x) * (kEstSegPerPod * a.est_pods);
  const int tile = static_cast<int>(blockIdx.x % static_cast<unsigned>(n_tiles));
  const int64_t base = static_cast<int64_t>(tile) * kWave * kNpl + lane;
  const double* raw0 = a.node_tab + 7 * a.row_stride;
  for (int k0 = 0; k0 < n; k0 += kWave) {  // uniform
    const bool have = k0 + lane < n;
    SegEntry e = SegEntry{0, 0};
    if (have) e = seg[k0 + lane];
    const bool is_tile = have && e.node < 0;
    if (have && !is_tile) {
      const int64_t mni = a.row_min[e.row], mxi = a.row_max[e.row];
      const double mn = static_cast<double>(mni);
      const double span = static_cast<double>(mxi - mni);
      const double raw = raw_score(load_node_tab(a, e.node), static_cast<double>(a.pod_cpu_milli[e.row]));
      a.out_score[static_cast<int64_t>(e.row) * a.row_stride + e.node] = static_cast<uint8_t>(norm_byte(raw, mn, span, 1.0 / span));
    }
    unsigned long long tm = __ballot(is_tile);
    while (tm != 0ull) {  // uniform
      const int src = __builtin_ctzll(tm);
      tm &= tm - 1ull;
      const int64_t pod = __builtin_amdgcn_readlane(e.row, src);
      const int64_t pod_i = uload(a.pod_cpu_milli + pod);
      const int64_t mni = uload(a.row_min + pod), mxi = uload(a.row_max + pod);
      const double mn = static_cast<double>(mni);
      const double span = static_cast<double>(mxi - mni);
      const double rspan = 1.0 / span;
      uint8_t* out = a.out_score + pod * a.row_stride;
      if (pod_i == 0) {  // uniform: all the loads first
        double r[kNpl];
        bool c[kNpl];
#pragma unroll
        for (int j = 0; j < kNpl; ++j) {
          c[j] = node_counts<kMask>(a, pod, base + j * kWave);
          r[j] = c[j] ? raw0[base + j * kWave] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kNpl; ++j) {
          const int64_t node = base + j * kWave;
          if (node < a.row_stride) out[node] = static_cast<uint8_t>(c[j] ? norm_byte(r[j], mn, span, rspan) : 0u);
        }
      } else {
        const double pod_cpu = static_cast<double>(pod_i);
#pragma unroll 2
        for (int j = 0; j < kNpl; ++j) {
          const int64_t node = base + j * kWave;
          if (node >= a.row_stride) continue;
          uint32_t b = 0;
          if (node_counts<kMask>(a, pod, node)) b = norm_byte(raw_score(load_node_tab(a, node), pod_cpu), mn, span, rspan);
          out[node] = static_cast<uint8_t>(b);
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

// the interval-estimate passes for a sweep of `swept` rows: pod rows per chunk, and the scratch — one list segment and one counter per wave
int peaks_est_plan(uint32_t opts, int64_t row_stride, int64_t swept, size_t* seg_bytes, size_t* cnt_bytes) {
  const int npl = (opts & kOptPeaksEst8) ? 8 : 16;
  const int64_t nt = (row_stride + kWave * npl - 1) / (kWave * npl);
  int est_pods = kEstPodsMax;
  while (est_pods > kEstPodsMin && ((swept + est_pods - 1) / est_pods) * nt < kEstWavesWanted) est_pods >>= 1;
  const int64_t waves = ((swept + est_pods - 1) / est_pods) * nt;
  *seg_bytes = static_cast<size_t>(waves) * kEstSegPerPod * est_pods * sizeof(SegEntry);
  *cnt_bytes = static_cast<size_t>(waves) * sizeof(int32_t);
  return est_pods;
}

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
  // interval estimates, raw_score where they cannot decide — when the caller planned them (peaks_est_plan) and brought the scratch
  if ((a.opts & kOptPeaksEstimate) && a.est_pods > 0 && a.seg && a.seg_n && a.node_tab && a.row_c) {
    const bool mask = a.other_status[0] || a.other_status[1] || a.other_status[2];
    const int npl = (a.opts & kOptPeaksEst8) ? 8 : 16;
    const int nt = static_cast<int>((a.row_stride + kWave * npl - 1) / (kWave * npl));
    const int64_t waves = ((swept + a.est_pods - 1) / a.est_pods) * nt;  // (est_pods: peaks_est_plan) one list segment each
    const dim3 g(static_cast<unsigned>((waves + kWavesPerBlock - 1) / kWavesPerBlock)), blk(kWave * kWavesPerBlock);
    hipLaunchKernelGGL(k_peaks_nodetab, dim3(static_cast<unsigned>((a.n_nodes + 255) / 256)), dim3(256), 0, s, a);
    if (npl == 8) {
      if (mask) hipLaunchKernelGGL((k_peaks_minmax_est<8, true>), g, blk, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_minmax_est<8, false>), g, blk, 0, s, a, nt);
    } else {
      if (mask) hipLaunchKernelGGL((k_peaks_minmax_est<16, true>), g, blk, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_minmax_est<16, false>), g, blk, 0, s, a, nt);
    }
    const dim3 gf(static_cast<unsigned>(waves)), bf(kWave);
    if (npl == 8) {
      if (mask) hipLaunchKernelGGL((k_peaks_fix_minmax<8, true>), gf, bf, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_fix_minmax<8, false>), gf, bf, 0, s, a, nt);
    } else {
      if (mask) hipLaunchKernelGGL((k_peaks_fix_minmax<16, true>), gf, bf, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_fix_minmax<16, false>), gf, bf, 0, s, a, nt);
    }
    hipLaunchKernelGGL(k_peaks_rowconst, dim3(static_cast<unsigned>((swept + 255) / 256)), dim3(256), 0, s, a);
    if (npl == 8) {
      if (mask) hipLaunchKernelGGL((k_peaks_write_est<8, true>), g, blk, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_write_est<8, false>), g, blk, 0, s, a, nt);
    } else {
      if (mask) hipLaunchKernelGGL((k_peaks_write_est<16, true>), g, blk, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_write_est<16, false>), g, blk, 0, s, a, nt);
    }
    if (npl == 8) {
      if (mask) hipLaunchKernelGGL((k_peaks_fix_write<8, true>), gf, bf, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_fix_write<8, false>), gf, bf, 0, s, a, nt);
    } else {
      if (mask) hipLaunchKernelGGL((k_peaks_fix_write<16, true>), gf, bf, 0, s, a, nt);
      else hipLaunchKernelGGL((k_peaks_fix_write<16, false>), gf, bf, 0, s, a, nt);
    }
    return;
  }
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
