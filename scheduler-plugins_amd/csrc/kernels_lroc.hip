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
// k_lroc_fast (round 6; the default when its preconditions hold: every column in [0, 2^47), no limit below its request) is the
// float32 formulation of the same sweep.  With the per-node differences
//   A = nodeLimit - capacity,  D = nodeLimit - nodeRequest   and per pod   d = podLimit - podRequest   (D, d >= 0)
// the reference's  over / (limit - min(request, cap))  for over = A + podLimit > 0, else 0, is  clamp(over / (D + d), 0, 1):
//   * over: A and podLimit each as the sum of two float32 (exact below 2^47); high parts, low parts, then both — three additions,
//     within 3 ulp of the exact sum even when it cancels (the high sum is exact whenever it cancels, the low sum always);
//   * both resources' quotients from ONE v_rcp_f32 of the product of the two denominators (D is held at >= 2^-30: a zero sum of
//     whole numbers then yields quotient 1 or 0 through the clamp, never NaN);
//   * the two resources ride the two halves of packed float32 operations (v_pk_add / v_pk_mul ... clamp / v_pk_fma);
//   * 100 * (1 - max risk) is formed in units of 2^-16 by one fma whose sum with 2^23 rounds it to a whole number: the mantissa
//     reads score << 16 | fraction, and "fraction within 8 units of 1/2" is one AND and one comparison whose lane mask is the
//     answer (no per-lane flag).
// The float32 value differs from the reference's float64 value by less than 8.7e-5 (error budget in DESIGN.md 3.8;
// tests/test_exactness_arguments.py replays it with the reciprocal pushed one ulp either way), so a cell farther than 8 units
// (1.22e-4) from a rounding boundary k + 0.5 has provably the reference's score; the remaining CELLS (~2.4e-4 of those with a
// non-trivial score) — not their whole lanes: a lane's eight cells cost eight dependent trips to the node table, which was a
// third of the round-5 kernel's time — are recomputed with the float64 form from the node table.
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
constexpr float kBand = 1.5e-4f;      // upper limit of the ambiguity band around k + 0.5 (float32 error of the score < 8.7e-5)
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

__global__ __launch_bounds__(kWave* kWavesPerBlock, 4) void k_lroc_fast(LrocArgs a, int n_tiles) {
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

  // per node: A as the sum of two float32 (exact: |A| < 2^47; its sum with the pod's limit cancels), D as the float32 it is used as —
  // never below 2^-30, so that the reciprocal of D + d is finite: a sum of whole numbers that is 0 stands for "any excess is the whole
  // denominator" (quotient 1)
  // (cpu in .x, memory in .y: the two resources' chains run as packed float32 operations)
  F32x2 Ah[NPL], Al[NPL], D[NPL], kl[NPL];
  const int64_t s = a.row_stride;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const double* tab = a.node_tab + node0 + j;
    const double ac = tab[8 * s], am = tab[10 * s];
    Ah[j] = F32x2{static_cast<float>(ac), static_cast<float>(am)};
    Al[j] = F32x2{static_cast<float>(ac - static_cast<double>(Ah[j].x)), static_cast<float>(am - static_cast<double>(Ah[j].y))};
    D[j] = F32x2{__builtin_fmaxf(static_cast<float>(tab[9 * s]), 0x1p-30f), __builtin_fmaxf(static_cast<float>(tab[11 * s]), 0x1p-30f)};
    kl[j] = F32x2{static_cast<float>(tab[12 * s]), static_cast<float>(tab[13 * s])};
  }
  const F32x2 w2{static_cast<float>(a.w_cpu), static_cast<float>(a.w_mem)};
  const int64_t np = a.n_pods_total;
  constexpr float kMagic = 8388608.0f;   // 2^23: a sum in [2^23, 2^24) is a whole number, and the mantissa bits are that number - 2^23
  constexpr float kScale = 6553600.0f;   // 100 * 2^16
  constexpr float kBandUnits = 8.0f;     // the band around k + 1/2 in units of 2^-16: 1.22e-4 (float32 error of the score < 8.7e-5)
  static_assert(kBandUnits * 0x1p-16f < kBand && kBandUnits == 8.0f, "the mask below clears log2(2 * kBandUnits) bits");

  unsigned redone = 0;
  // per pod (host-prepared, one 32-byte scalar load): podLimit as two float32 and podLimit - podRequest for cpu, memory; the seventh
  // word marks a pod without requests or limits (MinNodeScore, lowriskovercommitment.go:124-128).  The next pod's is asked for
  // before this one's cells.
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x8* recs = reinterpret_cast<const f32x8*>(a.pod_f32);
  f32x8 next = uload(recs + pod0);
  for (int64_t pod = pod0; pod < pod1; ++pod) {
    const f32x8 rec = next;
    next = uload(recs + (pod + 1 < pod1 ? pod + 1 : pod));
    const F32x2 plh{rec[0], rec[1]}, pll{rec[2], rec[3]}, df{rec[4], rec[5]};
    uint32_t w[NPL / 4] = {};
    if (rec[6] == 0.0f) {  // wave-uniform
      unsigned long long near[NPL];  // per cell: the lanes whose value is within the band of a rounding boundary
#pragma unroll
      for (int j = 0; j < NPL / 4; ++j) {
        uint32_t kb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = j * 4 + q;
          // riskLimit = over / max(D + d, over) for over > 0, else 0  ==  clamp(over / (D + d), 0, 1): over exact, then float32
          // (high parts, low parts, then both: the high sum is exact whenever it cancels, the low sum always — within 3 ulp of A + limit)
          const F32x2 ov = (Ah[i] + plh) + (Al[i] + pll);
          const F32x2 dd = D[i] + df;
          const float rr = __builtin_amdgcn_rcpf(dd.x * dd.y);  // one reciprocal for both quotients
          const F32x2 x = ov * __builtin_shufflevector(dd, dd, 1, 0);
          F32x2 qq, r2;
          r2.x = rr;  // (.y is not read: op_sel_hi takes the low half for both products)
          // (inline: the compiler has no packed clamp pattern; s_nop: the wait state a transcendental's consumer needs, which the
          // hazard pass cannot add inside an asm statement)
          asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] clamp" : "=v"(qq) : "v"(x), "v"(r2));  // both quotients, clamped to [0, 1] (NaN cannot occur: rr and x are finite or x is an infinity)
          const F32x2 t2 = __builtin_elementwise_fma(w2, qq, kl[i]);
          const float t_c = t2.x, t_m = t2.y;
          const float m = __builtin_amdgcn_fmed3f(__builtin_fmaxf(t_c, t_m), 0.0f, 1.0f);  // totalRisk's clamp (:252), after the max
          // 100 * (1 - max risk) in units of 2^-16, + 1/2 + the band, rounded to a whole number by the sum with 2^23 (the fma rounds once):
          // the mantissa then reads  score << 16 | fraction,  and a fraction below 2 * kBandUnits means "within the band of k + 1/2"
          const float k = __builtin_fmaf(m, -kScale, kMagic + kScale + 32768.0f + kBandUnits);
          kb[q] = __float_as_uint(k);
          near[i] = __ballot((kb[q] & (0xffffu & ~(2u * static_cast<uint32_t>(kBandUnits) - 1u))) == 0u);  // (the comparison's lane mask: no vector work)
        }
        const uint32_t lo = __builtin_amdgcn_perm(kb[1], kb[0], 0x0c0c0602u), hi = __builtin_amdgcn_perm(kb[3], kb[2], 0x0c0c0602u);  // the scores: byte 2
        w[j] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      unsigned long long any = 0;
#pragma unroll
      for (int i = 0; i < NPL; ++i) any |= near[i];
      if (__builtin_expect(any != 0ull, 0)) {  // rare (uniform): some cell of some lane is within the band — those cells in float64
        uint32_t amb = 0;  // this lane's cells: cell i -> bit NPL - 1 - i
#pragma unroll
        for (int i = 0; i < NPL; ++i) amb |= ((near[i] >> lane) & 1ull) ? 1u << (NPL - 1 - i) : 0u;
        while (amb != 0u) {
          const int i = NPL - 1 - (31 - __builtin_clz(amb));  // (per lane)
          amb &= ~(1u << (NPL - 1 - i));
          const uint32_t b = exact_cell(a, node0 + i, pod);
          const uint32_t keep = ~(0xffu << (8 * (i & 3))), put = b << (8 * (i & 3));
#pragma unroll
          for (int j = 0; j < NPL / 4; ++j) w[j] = (i >> 2) == j ? (w[j] & keep) | put : w[j];
          ++redone;  // spx_fetch_stats: cells re-evaluated
        }
      }
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(a.out_score + pod * a.row_stride + node0);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(dst) = u32x2{w[0], w[1]};
  }
  // one atomic per lane that met a band, each lane on its own counter line (lanes past the row have left: no wave reduction)
  if (redone && a.stats) {
    atomicAdd(a.stats + (SPX_PLUGIN_LROC * kStatSlots + lane) * kStatStride, static_cast<unsigned long long>(redone));
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
  if (a.exact53 && a.pod_f32 != nullptr) {
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
