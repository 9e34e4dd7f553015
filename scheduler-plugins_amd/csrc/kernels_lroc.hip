// kernels_lroc.hip — trimaran LowRiskOverCommitment on gfx950.
//
// Reference: LowRiskOverCommitment.Score pkg/trimaran/lowriskovercommitment/lowriskovercommitment.go:105-141 calling
// computeRank / computeRisk (:158-255) once per (pod, node), each call walking the node's pod list
// (GetNodeRequestsAndLimits, resourcestats.go:163-225) and fitting a Beta distribution to the node's load statistics
// (beta.go:173-191 over gonum's RegIncBeta).
//
// Split used here: of the two risk components only riskLimit (:205-208) involves the pending pod; riskLoad (:210-246)
// sees the node through its metrics and through the sums *without* the pod.  So
//   k_lroc_prepare  one thread per node: both resources' riskLoad (lroc_math.h; incomplete beta, log-gamma),
//                   stored pre-multiplied by (1 - weight), plus float64 images of the node's integer columns;
//   k_lroc          the P x N sweep: per cell two divisions and a dozen float64 operations, one byte written.
// The sweep keeps the reference's operations and their order (w*riskLimit + (1-w)*riskLoad, clamp, 1 - max, *100,
// round), so given the same riskLoad the scores are bit-identical; riskLoad itself can differ from a Go evaluation in
// the last digits of the special functions, which moves a score only across an exact rounding boundary (parity +-1).
//
// Integer arithmetic: limit - capacity and limit - request are int64 in the reference, then converted.  When every
// column is in [0, 2^52) (checked by the engine at upload, LrocArgs::exact53) the sums and differences are exact in
// float64 and the kernel never leaves the float64 pipe; otherwise the int64 form runs, operation for operation.
//
// k_lroc_fast (the default when exact53 holds) is the cheap formulation of the same sweep, in the manner of the TLP / LVRB
// kernels: numerator and denominator of riskLimit are still formed exactly in float64 — with the per-node differences
//   A = nodeLimit - capacity,  D = nodeLimit - nodeRequest   and per pod   d = podLimit - podRequest
// they are  over = A + podLimit  and  den = limit - min(request, cap) = max(D + d, over)  — but the quotient, the weighted
// sum and the final 100*(1 - max) run in float32 (v_rcp_f32, packed fma).  The float32 value s differs from the
// reference's float64 value by less than 6e-5 (error budget in DESIGN.md 3.8; tests/test_exactness_arguments.py), so
// whenever s is farther than kBand from a rounding boundary k + 0.5 the rounded score is provably the reference's; the
// remaining cells (~3e-4 of those with a non-trivial score) are recomputed with the float64 form from the node table.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lroc_math.h"
#include "spx_internal.h"

namespace spx {
namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kPodsPerChunk = 64;
constexpr int kNpl = 4;  // nodes per lane: one dword of scores per pod row
constexpr int kNplFast = 8;           // k_lroc_fast: two dwords per lane and row
constexpr int kTabCols = kLrocTabCols;
constexpr double kNoOver = -1e30;     // "limit - capacity" of a node that must not contribute a riskLimit
constexpr float kBand = 1.5e-4f;      // ambiguity band around k + 0.5 (float32 error of s < 6e-5)
typedef float F32x2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ T uload(const T* p) {  // wave-uniform read of immutable input -> scalar load
  typedef const T __attribute__((address_space(4))) CT;
  return *reinterpret_cast<CT*>(reinterpret_cast<uintptr_t>(p));
}

__global__ __launch_bounds__(256) void k_lroc_prepare(LrocArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.row_stride) return;
  double* tab = a.node_tab + n;
  const int64_t s = a.row_stride;
  if (n >= a.n_nodes) {  // padding columns score 0
    tab[0] = __builtin_nan("");
    for (int k = 1; k < kTabCols; ++k) tab[k * s] = 0.0;
    tab[8 * s] = kNoOver, tab[10 * s] = kNoOver, tab[12 * s] = 1.0;  // fast form: riskLimit 0, total risk 1 -> score 0
    return;
  }
  const uint8_t f = a.flags[n];
  const bool has = (f & SPX_LV_HAS_METRICS) != 0;
  lroc::NodeResource c, m;
  c.metric_valid = has && (f & SPX_LV_CPU_VALID) != 0;
  c.capacity = a.alloc_cpu_milli[n];
  c.capacity_stat = static_cast<double>(c.capacity);  // resourcestats.go:60-61
  c.avg = a.cpu_avg[n];
  c.stdev = a.cpu_std[n];
  c.requested = a.node_req_cpu[n];
  c.limits = a.node_lim_cpu[n];
  m.metric_valid = has && (f & SPX_LV_MEM_VALID) != 0;
  m.capacity = a.alloc_mem[n];
  m.capacity_stat = static_cast<double>(m.capacity);  // :63-65
  m.capacity_stat *= lroc::kMega;
  m.avg = a.mem_avg[n];
  m.stdev = a.mem_std[n];
  m.requested = a.node_req_mem[n];
  m.limits = a.node_lim_mem[n];
  // a node without metrics scores MinNodeScore (lowriskovercommitment.go:130-134): flagged by NaN in slot 0
  tab[0 * s] = has ? (1 - a.w_cpu) * lroc::risk_load(c, a.sqrt_window) : __builtin_nan("");
  tab[1 * s] = has ? (1 - a.w_mem) * lroc::risk_load(m, a.sqrt_window) : 0.0;
  tab[2 * s] = static_cast<double>(c.requested);
  tab[3 * s] = static_cast<double>(c.limits);
  tab[4 * s] = static_cast<double>(c.capacity);
  tab[5 * s] = static_cast<double>(m.requested);
  tab[6 * s] = static_cast<double>(m.limits);
  tab[7 * s] = static_cast<double>(m.capacity);
  // fast form (meaningful when exact53): A and D per resource, (1-w)*riskLoad as the float32 it is used as
  tab[8 * s] = has ? static_cast<double>(c.limits - c.capacity) : kNoOver;
  tab[9 * s] = has ? static_cast<double>(c.limits - c.requested) : 0.0;
  tab[10 * s] = has ? static_cast<double>(m.limits - m.capacity) : kNoOver;
  tab[11 * s] = has ? static_cast<double>(m.limits - m.requested) : 0.0;
  tab[12 * s] = has ? static_cast<double>(static_cast<float>(tab[0])) : 1.0;  // no metrics: total risk 1 -> score 0
  tab[13 * s] = has ? static_cast<double>(static_cast<float>(tab[s])) : 0.0;
}

struct NodeF {  // float64 form of one node
  double kl_c, kl_m;                 // (1 - w) * riskLoad
  double req_c, lim_c, cap_c;
  double req_m, lim_m, cap_m;
};
struct NodeI {  // int64 form
  double kl_c, kl_m;
  int64_t req_c, lim_c, cap_c;
  int64_t req_m, lim_m, cap_m;
};

// totalRisk of one resource (lowriskovercommitment.go:205-208, :250-253) given (1-w)*riskLoad
__device__ __forceinline__ double total_risk(double w, double kl, double node_req, double node_lim, double cap, double pod_req, double pod_lim) {
  const double limit = node_lim + pod_lim;                        // resourcestats.go:204-205
  const double request = fmin(node_req + pod_req, cap);           // :202-203, :208-209
  const double over = limit - cap;
  const double risk_limit = over > 0.0 ? over / (limit - request) : 0.0;
  const double total = w * risk_limit + kl;
  return fmax(fmin(total, 1.0), 0.0);
}
__device__ __forceinline__ double total_risk(double w, double kl, int64_t node_req, int64_t node_lim, int64_t cap, int64_t pod_req, int64_t pod_lim) {
  const int64_t limit = node_lim + pod_lim;
  int64_t request = node_req + pod_req;
  if (request > cap) request = cap;
  const double risk_limit = limit > cap ? static_cast<double>(limit - cap) / static_cast<double>(limit - request) : 0.0;
  const double total = w * risk_limit + kl;
  return fmax(fmin(total, 1.0), 0.0);
}

__device__ __forceinline__ uint32_t score_byte(bool has, double risk_c, double risk_m) {
  const double rank = 1 - fmax(risk_c, risk_m);                   // :165
  const int v = static_cast<int>(round(rank * 100.0));            // :136-137
  return has ? static_cast<uint32_t>(v < 0 ? 0 : (v > 100 ? 100 : v)) : 0u;
}

template <bool F64>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_lroc(LrocArgs a, int n_tiles) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t pod0 = a.row_begin + chunk * kPodsPerChunk;
  if (pod0 >= a.row_end) return;
  const int64_t pod1 = (pod0 + kPodsPerChunk < a.row_end) ? pod0 + kPodsPerChunk : a.row_end;
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * kNpl;
  if (node0 >= a.row_stride) return;  // row_stride is a multiple of 16

  using Node = typename std::conditional<F64, NodeF, NodeI>::type;
  Node nd[kNpl];
  bool has[kNpl];
  const int64_t s = a.row_stride;
#pragma unroll
  for (int j = 0; j < kNpl; ++j) {
    const int64_t n = node0 + j;  // < row_stride: the table is padded
    const double* tab = a.node_tab + n;
    const double k0 = tab[0];
    has[j] = k0 == k0;
    nd[j].kl_c = has[j] ? k0 : 0.0;
    nd[j].kl_m = tab[s];
    if constexpr (F64) {
      nd[j].req_c = tab[2 * s], nd[j].lim_c = tab[3 * s], nd[j].cap_c = tab[4 * s];
      nd[j].req_m = tab[5 * s], nd[j].lim_m = tab[6 * s], nd[j].cap_m = tab[7 * s];
    } else {
      const bool in = n < a.n_nodes;
      nd[j].req_c = in ? a.node_req_cpu[n] : 0, nd[j].lim_c = in ? a.node_lim_cpu[n] : 0, nd[j].cap_c = in ? a.alloc_cpu_milli[n] : 0;
      nd[j].req_m = in ? a.node_req_mem[n] : 0, nd[j].lim_m = in ? a.node_lim_mem[n] : 0, nd[j].cap_m = in ? a.alloc_mem[n] : 0;
    }
  }

  for (int64_t pod = pod0; pod < pod1; ++pod) {
    const int64_t prc = uload(a.pod_req_cpu + pod), prm = uload(a.pod_req_mem + pod);
    const int64_t plc = uload(a.pod_lim_cpu + pod), plm = uload(a.pod_lim_mem + pod);
    uint32_t word = 0;
    if (!(prc == 0 && prm == 0 && plc == 0 && plm == 0)) {  // best-effort pods score MinNodeScore (:124-128); wave-uniform
#pragma unroll
      for (int j = 0; j < kNpl; ++j) {
        double rc, rm;
        if constexpr (F64) {
          rc = total_risk(a.w_cpu, nd[j].kl_c, nd[j].req_c, nd[j].lim_c, nd[j].cap_c, static_cast<double>(prc), static_cast<double>(plc));
          rm = total_risk(a.w_mem, nd[j].kl_m, nd[j].req_m, nd[j].lim_m, nd[j].cap_m, static_cast<double>(prm), static_cast<double>(plm));
        } else {
          rc = total_risk(a.w_cpu, nd[j].kl_c, nd[j].req_c, nd[j].lim_c, nd[j].cap_c, prc, plc);
          rm = total_risk(a.w_mem, nd[j].kl_m, nd[j].req_m, nd[j].lim_m, nd[j].cap_m, prm, plm);
        }
        word |= score_byte(has[j], rc, rm) << (8 * j);
      }
    }
    *reinterpret_cast<uint32_t*>(a.out_score + pod * a.row_stride + node0) = word;
  }
}


// exact float64 score of one cell from the node table (fallback of k_lroc_fast)
__device__ __forceinline__ uint32_t exact_cell(const LrocArgs& a, int64_t n, int64_t pod) {
  const double* tab = a.node_tab + n;
  const int64_t s = a.row_stride;
  const double k0 = tab[0];
  const bool has = k0 == k0;
  const double prc = static_cast<double>(a.pod_req_cpu[pod]), prm = static_cast<double>(a.pod_req_mem[pod]);
  const double plc = static_cast<double>(a.pod_lim_cpu[pod]), plm = static_cast<double>(a.pod_lim_mem[pod]);
  const double rc = total_risk(a.w_cpu, has ? k0 : 0.0, tab[2 * s], tab[3 * s], tab[4 * s], prc, plc);
  const double rm = total_risk(a.w_mem, tab[s], tab[5 * s], tab[6 * s], tab[7 * s], prm, plm);
  return score_byte(has, rc, rm);
}

__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_lroc_fast(LrocArgs a, int n_tiles) {
  constexpr int NPL = kNplFast;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t pod0 = a.row_begin + chunk * kPodsPerChunk;
  if (pod0 >= a.row_end) return;
  const int64_t pod1 = (pod0 + kPodsPerChunk < a.row_end) ? pod0 + kPodsPerChunk : a.row_end;
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * NPL;
  if (node0 >= a.row_stride) return;  // row_stride is a multiple of 16; no cross-lane operation below

  double A_c[NPL], D_c[NPL], A_m[NPL], D_m[NPL];
  F32x2 kl[NPL];
  float klmax[NPL];
  const int64_t s = a.row_stride;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const double* tab = a.node_tab + node0 + j;
    A_c[j] = tab[8 * s], D_c[j] = tab[9 * s], A_m[j] = tab[10 * s], D_m[j] = tab[11 * s];
    kl[j] = F32x2{static_cast<float>(tab[12 * s]), static_cast<float>(tab[13 * s])};
    klmax[j] = __builtin_fmaxf(kl[j].x, kl[j].y);
  }
  const F32x2 w2{static_cast<float>(a.w_cpu), static_cast<float>(a.w_mem)};
  const int64_t np = a.n_pods_total;
  constexpr float kHalf = 0.5f - kBand;

  unsigned redone = 0;
  for (int64_t pod = pod0; pod < pod1; ++pod) {
    // per pod (host-prepared float64, scalar loads): podLimit and podLimit - podRequest for cpu, memory; NaN marks a
    // pod without requests or limits (MinNodeScore, lowriskovercommitment.go:124-128)
    const double plc = uload(a.pod_f64 + pod), dc = uload(a.pod_f64 + np + pod);
    const double plm = uload(a.pod_f64 + 2 * np + pod), dm = uload(a.pod_f64 + 3 * np + pod);
    uint32_t w[NPL / 4] = {};
    if (plc == plc) {  // wave-uniform
      float worst = 0.0f;
#pragma unroll
      for (int j = 0; j < NPL / 4; ++j) {
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = j * 4 + q;
          const double over_c = A_c[i] + plc, over_m = A_m[i] + plm;        // limit - capacity, exact
          const double den_c = fmax(D_c[i] + dc, over_c), den_m = fmax(D_m[i] + dm, over_m);  // limit - request, exact
          const F32x2 ov{static_cast<float>(over_c), static_cast<float>(over_m)};
          const F32x2 rc{__builtin_amdgcn_rcpf(static_cast<float>(den_c)), __builtin_amdgcn_rcpf(static_cast<float>(den_m))};
          // w * max(over/den, 0) + kl  ==  max(fma(w, over/den, kl), kl): a negative, infinite or NaN quotient (over <= 0,
          // possibly with den == 0) is absorbed by the max, which returns its non-NaN operand
          const F32x2 t2 = __builtin_elementwise_fma(w2, ov * rc, kl[i]);
          const float m = __builtin_fmaxf(__builtin_fmaxf(t2.x, t2.y), klmax[i]);
          const float sc = __builtin_fmaf(-100.0f, m, 100.0f);            // 100 * (1 - max risk)
          const float rr = __builtin_rintf(sc);
          worst = __builtin_fmaxf(worst, __builtin_fabsf(sc - rr));
          acc = __builtin_amdgcn_cvt_pk_u8_f32(rr, q, acc);
        }
        w[j] = acc;
      }
      if (__builtin_expect(!(worst < kHalf), 0)) {  // rare: some cell of this lane is within the band of a rounding boundary
#pragma unroll
        for (int j = 0; j < NPL / 4; ++j) {
#pragma unroll 1
          for (int q = 0; q < 4; ++q) {  // not unrolled: the slow path must not cost the sweep its registers
            const uint32_t b = exact_cell(a, node0 + j * 4 + q, pod);
            w[j] = (w[j] & ~(0xffu << (8 * q))) | (b << (8 * q));
          }
        }
        ++redone;  // spx_fetch_stats: the whole lane (NPL cells) is redone, not only the cells inside the band
      }
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(a.out_score + pod * a.row_stride + node0);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(dst) = u32x2{w[0], w[1]};
  }
  // one atomic per lane that met a band, each lane on its own counter line (lanes past the row have left: no wave reduction)
  if (redone && a.stats) {
    const int64_t cells = min<int64_t>(NPL, max<int64_t>(a.n_nodes - node0, 0));
    atomicAdd(a.stats + (SPX_PLUGIN_LROC * kStatSlots + lane) * kStatStride, static_cast<unsigned long long>(redone) * static_cast<unsigned long long>(cells));
  }
}

}  // namespace

void launch_lroc_prepare(const LrocArgs& a, hipStream_t s) {
  const unsigned blocks = static_cast<unsigned>((a.row_stride + 255) / 256);
  hipLaunchKernelGGL(k_lroc_prepare, dim3(blocks), dim3(256), 0, s, a);
}

void launch_lroc(const LrocArgs& a, hipStream_t s) {
  const int tile_nodes = kWave * kNpl;
  const int n_tiles = static_cast<int>((a.row_stride + tile_nodes - 1) / tile_nodes);
  const int64_t rows = a.row_end - a.row_begin;
  if (rows <= 0) return;
  const int64_t chunks = (rows + kPodsPerChunk - 1) / kPodsPerChunk;
  const int64_t units = chunks * n_tiles;
  const unsigned blocks = static_cast<unsigned>((units + kWavesPerBlock - 1) / kWavesPerBlock);
  if (a.exact53 && a.pod_f64 != nullptr) {
    const int tn = kWave * kNplFast;
    const int nt = static_cast<int>((a.row_stride + tn - 1) / tn);
    const unsigned nb = static_cast<unsigned>((chunks * nt + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(k_lroc_fast, dim3(nb), dim3(kWave * kWavesPerBlock), 0, s, a, nt);
  } else if (a.exact53)
    hipLaunchKernelGGL((k_lroc<true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles);
  else
    hipLaunchKernelGGL((k_lroc<false>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles);
}

}  // namespace spx
