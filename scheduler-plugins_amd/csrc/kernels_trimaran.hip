// kernels_trimaran.hip — gfx950 kernels for NodeResourcesAllocatable + trimaran TargetLoadPacking +
// LoadVariationRiskBalancing, evaluated as one dense pods x nodes sweep.
//
// Work decomposition (one wavefront = one unit of work):
//   unit = (node tile of 64*NPL nodes) x (chunk of kPodsPerChunk pod rows)
//   lane l owns NPL consecutive nodes; their per-node parameters are derived ONCE in the wave
//   prologue and live in VGPRs for the whole chunk, so the steady state reads nothing from memory
//   except one scalar (wave-uniform) pod record per row and writes NPL result bytes per lane per
//   plugin as one coalesced vector store (64*NPL contiguous bytes per wave per row).
//   HBM traffic is therefore the compulsory P*N bytes per plugin of output; node tables (O(N)) are
//   re-read per unit from L2.  No LDS is needed: there is no cross-lane reuse — every lane's node
//   state is private — and the only reduction (Allocatable's min/max) is pod-independent.
//
// Arithmetic is the reference's, operation for operation, in IEEE double with contraction off
// (the TU is compiled with -ffp-contract=off): Go on amd64 never fuses a*b+c.
//   TLP  : pkg/trimaran/targetloadpacking/targetloadpacking.go:146-186
//   LVRB : pkg/trimaran/loadvariationriskbalancing/analysis.go:34-60, pkg/trimaran/resourcestats.go:45-86
//   Alloc: pkg/noderesources/allocatable.go:117-168
#include <cstdlib>

#include "spx_internal.h"
#include "trimaran_math.h"

namespace spx {

namespace {

using namespace trimath;

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kPodsPerChunk = 64;

__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int m) {
  int lo = __shfl_xor(static_cast<int>(v & 0xffffffffLL), m, kWave);
  int hi = __shfl_xor(static_cast<int>(v >> 32), m, kWave);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

// ------------------------------------------------------------------------------------------------
// Allocatable: raw score per node + NormalizeScore over the whole node list (pod independent when
// no per-row feasibility mask is installed: allocatable.go:118-126 never reads the pod).
// One block; N is O(10^4).
__global__ __launch_bounds__(1024) void k_alloc_prepare(AllocPrepArgs a) {
  __shared__ int64_t s_lo[16];
  __shared__ int64_t s_hi[16];
  const int tid = threadIdx.x;
  uint64_t wsum = 0;
  for (int r = 0; r < a.n_res; ++r) wsum += static_cast<uint64_t>(a.weight[r]);
  int64_t lo = INT64_MAX;
  int64_t hi = -INT64_MAX;
  for (int64_t i = tid; i < a.n_nodes; i += 1024) {
    uint64_t acc = 0;  // Go int64 wraps; do the sums in uint64
    for (int r = 0; r < a.n_res; ++r) {
      const int64_t v = a.alloc[static_cast<int64_t>(r) * a.n_nodes + i];
      int64_t sc = 0;
      if (a.mode == SPX_MODE_LEAST) sc = static_cast<int64_t>(0 - static_cast<uint64_t>(v));
      else if (a.mode == SPX_MODE_MOST) sc = v;
      acc += static_cast<uint64_t>(sc) * static_cast<uint64_t>(a.weight[r]);
    }
    const int64_t raw = static_cast<int64_t>(acc) / static_cast<int64_t>(wsum);  // truncates toward zero
    a.raw[i] = raw;
    lo = raw < lo ? raw : lo;
    hi = raw > hi ? raw : hi;
  }
  for (int m = 32; m >= 1; m >>= 1) {
    const int64_t olo = shfl_xor_i64(lo, m);
    const int64_t ohi = shfl_xor_i64(hi, m);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  if ((tid & 63) == 0) {
    s_lo[tid >> 6] = lo;
    s_hi[tid >> 6] = hi;
  }
  __syncthreads();
  lo = s_lo[0];
  hi = s_hi[0];
  for (int w = 1; w < 16; ++w) {
    lo = s_lo[w] < lo ? s_lo[w] : lo;
    hi = s_hi[w] > hi ? s_hi[w] : hi;
  }
  const int64_t range = static_cast<int64_t>(static_cast<uint64_t>(hi) - static_cast<uint64_t>(lo));
  for (int64_t i = tid; i < a.row_stride; i += 1024) {
    int64_t v = 0;
    if (i < a.n_nodes && range != 0) {
      const uint64_t d = static_cast<uint64_t>(a.raw[i]) - static_cast<uint64_t>(lo);
      v = static_cast<int64_t>(d * 100ull) / range;
    }
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    a.norm[i] = static_cast<uint8_t>(v);
    // compact form for the per-row masked normalisation (k_alloc_masked): offsets from the global minimum as
    // uint32 (valid when the global range fits; the flag sits behind the last padded entry)
    if (a.rel) a.rel[i] = i < a.n_nodes ? static_cast<uint32_t>(static_cast<uint64_t>(a.raw[i]) - static_cast<uint64_t>(lo)) : 0u;
  }
  if (a.rel && tid == 0) a.rel[a.row_stride] = static_cast<uint64_t>(range) < (1ull << 32) ? 1u : 0u;
}

// LVRB per resource: state 0 = metric type absent (CreateResourceStats !ok), 1 = capacity <= 0
// (computeScore returns 0), 2 = regular
struct LvRes {
  double cap;
  double used_avg;  // clamped to [0, cap]   analysis.go:42
  double sigma;     // final sigma (pod independent) analysis.go:43-54
  int state;
};

// math.Pow as the plugin reaches it (x in [0,1]); exact for y in {0,1,2,0.5,+Inf}; see oracle note.
__device__ double go_pow01(double x, double y) {
  if (y == 0.0 || x == 1.0) return 1.0;
  if (y == 1.0) return x;
  if (x != x || y != y) return x + y;
  if (x == 0.0) return y < 0.0 ? __builtin_inf() : 0.0;
  if (__builtin_isinf(y)) return ((fabs(x) < 1.0) == (y > 0.0)) ? 0.0 : __builtin_inf();
  if (y == 0.5) return sqrt(x);
  if (y == -0.5) return 1.0 / sqrt(x);
  if (y == 2.0) return x * x;
  return pow(x, y);
}

__device__ __forceinline__ LvRes lv_make(bool valid, double cap, double util, double sd, double margin, double sens) {
  LvRes r;
  r.cap = cap;
  r.used_avg = 0.0;
  r.sigma = 0.0;
  if (!valid) {
    r.state = 0;
    return r;
  }
  if (cap <= 0.0) {
    r.state = 1;
    return r;
  }
  r.state = 2;
  double used_avg = util * cap / 100.0;  // resourcestats.go:68
  double used_sd = sd * cap / 100.0;     // :69
  used_avg = fmax(fmin(used_avg, cap), 0.0);
  used_sd = fmax(fmin(used_sd, cap), 0.0);
  double sigma = used_sd / cap;
  sigma = fmax(fmin(sigma, 1.0), 0.0);
  if (sens >= 0.0) sigma = go_pow01(sigma, 1.0 / sens);
  sigma *= margin;
  sigma = fmax(fmin(sigma, 1.0), 0.0);
  r.used_avg = used_avg;
  r.sigma = sigma;
  return r;
}

__device__ __forceinline__ double lv_res_score(const LvRes& r, double req) {
  if (r.state != 2) return 0.0;
  double mu = (r.used_avg + req) / r.cap;
  mu = fmax(fmin(mu, 1.0), 0.0);
  const double risk = (mu + r.sigma) / 2.0;
  return (1.0 - risk) * 100.0;
}

__device__ __forceinline__ double lv_total(bool has_metrics, const LvRes& c, const LvRes& m, double req_cpu, double req_mem) {
  if (!has_metrics) return 0.0;
  const double cs = lv_res_score(c, req_cpu);
  const double ms = lv_res_score(m, req_mem);
  return (c.state != 0 && m.state != 0) ? fmin(ms, cs) : fmax(ms, cs);
}

constexpr double kMega = 1.0 / 1024.0 / 1024.0;  // resourcestats.go:29

typedef float F32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NPL>
struct Vec;
template <>
struct Vec<4> { using T = uint32_t; };
template <>
struct Vec<8> { using T = u32x2; };
template <>
struct Vec<16> { using T = u32x4; };

template <int NPL>
__device__ __forceinline__ void store_bytes(uint8_t* dst, const uint32_t (&w)[NPL / 4]) {
  using V = typename Vec<NPL>::T;
  V v;
  __builtin_memcpy(&v, w, sizeof(V));
  *reinterpret_cast<V*>(dst) = v;  // non-temporal stores measured: no gain (0.457 vs 0.464 ms on config #2)
}

template <int NPL, bool A, bool T, bool L>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_trimaran(TrimaranArgs a, int n_tiles) {
  SPX_RESOLVE_ROWS(a);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t pod0 = a.row_begin + chunk * kPodsPerChunk;
  if (pod0 >= a.row_end) return;
  const int64_t pod1 = (pod0 + kPodsPerChunk < a.row_end) ? pod0 + kPodsPerChunk : a.row_end;
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * NPL;
  const bool active = node0 < a.row_stride;  // row_stride is a multiple of 16 >= n_nodes

  // ---- prologue: per-node state into registers
  uint32_t alloc_w[NPL / 4];
  TlpNode tn[NPL];
  LvRes lc[NPL], lm[NPL];
  bool lhas[NPL];
  if constexpr (A) {
#pragma unroll
    for (int j = 0; j < NPL / 4; ++j) alloc_w[j] = active ? reinterpret_cast<const uint32_t*>(a.alloc_norm + node0)[j] : 0u;
  }
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int64_t n = node0 + j;
    const bool in = n < a.n_nodes;
    if constexpr (T) {
      const double cap = in ? static_cast<double>(a.cap_cpu_milli[n]) : 0.0;
      const double util = in ? a.tlp_cpu_util[n] : 0.0;
      tn[j].cap = cap;
      tn[j].util_millis = (util / 100.0) * cap;
      tn[j].missing = in ? static_cast<double>(a.tlp_missing_milli[n]) : 0.0;
      tn[j].valid = in && a.tlp_valid[n] != 0;
    }
    if constexpr (L) {
      const uint8_t f = in ? a.lv_flags[n] : 0;
      lhas[j] = (f & SPX_LV_HAS_METRICS) != 0;
      const double ccap = in ? static_cast<double>(a.lv_alloc_cpu_milli[n]) : 0.0;
      double mcap = in ? static_cast<double>(a.lv_alloc_mem[n]) : 0.0;
      mcap *= kMega;
      lc[j] = lv_make((f & SPX_LV_CPU_VALID) != 0, ccap, in ? a.lv_cpu_avg[n] : 0.0, in ? a.lv_cpu_std[n] : 0.0,
                      a.lv_margin, a.lv_sensitivity);
      lm[j] = lv_make((f & SPX_LV_MEM_VALID) != 0, mcap, in ? a.lv_mem_avg[n] : 0.0, in ? a.lv_mem_std[n] : 0.0,
                      a.lv_margin, a.lv_sensitivity);
    }
  }
  if (!active) return;

  // ---- steady state: one pod row per iteration
  for (int64_t pod = pod0; pod < pod1; ++pod) {
    const int64_t row = pod * a.row_stride + node0;
    if constexpr (A) store_bytes<NPL>(a.out_alloc + row, alloc_w);
    if constexpr (T) {
      const double pod_milli = static_cast<double>(a.tlp_pod_milli[pod]);
      uint32_t w[NPL / 4];
#pragma unroll
      for (int j = 0; j < NPL / 4; ++j) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bool zero;
          const double x = tlp_unrounded(tn[j * 4 + k], pod_milli, a.tlp_target, &zero);
          acc |= (zero ? 0u : to_u8(x)) << (8 * k);
        }
        w[j] = acc;
      }
      store_bytes<NPL>(a.out_tlp + row, w);
    }
    if constexpr (L) {
      const double req_cpu = fmax(static_cast<double>(a.lv_req_cpu_milli[pod]), 0.0);           // analysis.go:41
      const double req_mem = fmax(static_cast<double>(a.lv_req_mem[pod]) * kMega, 0.0);          // resourcestats.go:64
      uint32_t w[NPL / 4];
#pragma unroll
      for (int j = 0; j < NPL / 4; ++j) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = j * 4 + k;
          acc |= to_u8(lv_total(lhas[q], lc[q], lm[q], req_cpu, req_mem)) << (8 * k);
        }
        w[j] = acc;
      }
      store_bytes<NPL>(a.out_lvrb + row, w);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TargetLoadPacking fast path (bit-exact by construction): float32 per cell, float64 only where it is needed.
//
// The exact expression costs two float64 divisions per (pod,node) and makes the sweep VALU-bound at ~2 ms for 10^9
// evals.  The result however is a small integer, round(g(pred)), with g piecewise linear — so a cheaper evaluation x' of
// the same real number decides the SAME integer whenever x' is not within its error bound of a rounding tie or of the
// pred == T branch point; cells that are ("ambiguous") re-read the node's original columns and recompute with the
// reference's exact operation sequence (tlp_unrounded above).  Only speed depends on how many cells are ambiguous.
//
// With u = (util_millis + missing - T*cap/100) + pod   [millicores above the target line]
//      pred - T = k*u,  k = 100/cap, and the reference's two branches become
//      u > 0 :  x = T   - (c1*k)*u          u <= 0 :  x = 100 + (c2*k)*u
// u is ~1e5 millicores with a fractional part, but pod is an integer and b2 is a per-node constant: with
// b2 = b2h + b2l (b2h integer-valued, |b2l| <= 0.5) the sum u = (pod + b2h) + b2l costs two float32 adds, the first
// exact, the second rounding once — no float64 op is left in the per-cell path.  After it:
// select(coef, off), fma, rndne, tie test, v_cvt_pk_u8_f32 (convert+clamp+pack in one op).
// |x' - x| <= |coef*u| * 1.2e-7 + ulp32(100)/2 <= 1.7e-5 for x >= -1 (for x < -1 only the sign
// matters and it cannot flip), so lanes with |frac(x') - .5| >= kTol32 = 4e-5 and |u| >= 1e-6 are
// provably the exact result; the others ("ambiguous", ~8e-5 of cells on continuous inputs) are
// recomputed per cell with the reference's exact float64 sequence from the node's original columns.
// The 64 pod records of a chunk are fetched with one coalesced load and broadcast with v_readlane.

// node slot n of a sweep kernel with NPL nodes per lane -> index of its record in the tile-transposed constant tables
// ([tile][j][lane]: for a fixed j the 64 lanes of a wavefront read consecutive records)
template <int NPL>
__device__ __forceinline__ int64_t tile_slot(int64_t n) {
  const int64_t tile = n / (kWave * NPL);
  const int lane = static_cast<int>((n / NPL) % kWave), j = static_cast<int>(n % NPL);
  return (tile * NPL + j) * kWave + lane;
}

constexpr int kLvNpl = 8, kTlpNpl = 16;  // nodes per lane of k_lvrb_fast / k_tlp_fast2
constexpr int kLvAmbCpu = 1 << 16, kLvAmbMem = 1 << 17;  // k_lvrb_amb_build's table: cpu millicores / memory MiB it covers
constexpr double kAmbMargin = 1.25;          // k_tlp_amb_build lists a cell when it is within 1.25 x the tolerance of a rounding tie
constexpr double kAmbMinSlope = 2.5 * 4e-5;  // score units per millicore below which a node is always exact (see k_tlp_amb_build)

// (see k_lvrb_prepare_fast) TargetLoadPacking's per-node float32 constants: b2 = b2h + b2l and the two branch coefficients
__global__ void k_tlp_prepare_fast(TrimaranArgs a, int64_t n_slots, double c1, double c2) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= n_slots) return;
  const double t = a.tlp_target;
  const bool in = n < a.n_nodes;
  double b = 1e30;       // invalid / padding: u huge -> x' = T - u -> cvt_pk_u8 gives 0, never ambiguous
  float f1 = -1.0f, f2 = 0.0f;
  bool split = false;
  if (in && a.tlp_valid[n] != 0) {
    const double cap = static_cast<double>(a.cap_cpu_milli[n]);
    const double um = (a.tlp_cpu_util[n] / 100.0) * cap;
    const double miss = static_cast<double>(a.tlp_missing_milli[n]);
    if (cap == 0.0) {
      b = 1.0;  // predicted stays 0 (targetloadpacking.go:171): x = T for every pod
      f1 = 0.0f;
    } else if (!(um >= 0.0) || !(miss >= 0.0) || !(cap > 0.0) || !(um < 1e15) || !(miss < 1e15)) {
      b = __builtin_nan("");
    } else {
      const double k = 100.0 / cap;
      b = (um + miss) - t * cap / 100.0;
      f1 = static_cast<float>(-c1 * k);
      f2 = static_cast<float>(c2 * k);
      split = __builtin_fabs(b) < 8388607.0 && c1 * k >= kAmbMinSlope && c2 * k >= kAmbMinSlope;
      if (!split) b = __builtin_nan("");  // beyond the exact float32 integer range (or a slope so flat that runs of pod values are ambiguous): always the exact path
    }
  }
  // u = (pod + b2h) + b2l: the first add is exact (two integers below 2^23), the second rounds once — the same
  // single float32 rounding a float64 add followed by a conversion would make, without the two float64-rate ops
  const double bh = split ? __builtin_rint(b) : b;
  reinterpret_cast<float4*>(a.tlp_fast)[tile_slot<kTlpNpl>(n)] =
      float4{static_cast<float>(bh), split ? static_cast<float>(b - bh) : 0.0f, f1, f2};
}

// Which (pod value, node tile) pairs hold a cell the float32 sweep cannot prove (k_tlp_fast2<..., AMB>).  Per node the unrounded
// score is x(p) = T - c1*k*u (u > 0), 100 + c2*k*u (u <= 0), u = p + b, p the pod's integer millicores: x comes within the
// tolerance tau of a rounding tie h = 0.5, 1.5, ... 99.5 only for the integers next to the real solution p* of x(p*) = h, and only
// when |p* - rint(p*)| * slope < tau; the branch point is ambiguous for the one integer within kTolU of -b.  One thread per (node,
// h, branch); a hit sets bit (tile & 31) of amb[p].  Conservative by the factor kAmbMargin (a superset costs speed, not results):
// an unflagged cell has |frac(x) - 0.5| >= 1.25 * 4e-5 > the float32 formula's error bound 1.7e-5 and |u| >= 2e-6.  A node whose
// slope is below 2.5 * tau (more than ~10^6 millicores of capacity) could be ambiguous for runs of integers: k_tlp_prepare_fast
// sends such nodes to the always-exact path instead.
__global__ __launch_bounds__(256) void k_tlp_amb_build(TrimaranArgs a, double c1, double c2, int tile_nodes) {
  const int64_t n = blockIdx.x;
  const int j = threadIdx.x;
  if (n >= a.n_nodes || a.tlp_valid[n] == 0 || j > 200) return;
  const double t = a.tlp_target;
  const double cap = static_cast<double>(a.cap_cpu_milli[n]);
  const double um = (a.tlp_cpu_util[n] / 100.0) * cap;
  const double miss = static_cast<double>(a.tlp_missing_milli[n]);
  if (!(cap > 0.0) || !(um >= 0.0) || !(miss >= 0.0) || !(um < 1e15) || !(miss < 1e15)) return;  // cap == 0: x = T for every pod; the others: always exact
  const double k = 100.0 / cap;
  const double b = (um + miss) - t * cap / 100.0;
  double p_star, slope;
  if (j < 100) {  // u > 0: T - c1*k*u = h
    slope = c1 * k;
    p_star = (t - (j + 0.5)) / slope - b;
  } else if (j < 200) {  // u <= 0: 100 + c2*k*u = h
    slope = c2 * k;
    p_star = ((j - 100 + 0.5) - 100.0) / slope - b;
  } else {  // the branch point u = 0
    slope = 2e-6 / (4e-5 * kAmbMargin);  // |p + b| < 2e-6 (kTolU is 1e-6)
    p_star = -b;
  }
  const double pn = __builtin_rint(p_star);
  if (!(__builtin_fabs(p_star - pn) * slope < 4e-5 * kAmbMargin)) return;
  if (!(pn >= 0.0) || !(pn < static_cast<double>(a.tlp_amb_size))) return;
  const uint32_t bit = 1u << (static_cast<uint32_t>(n / tile_nodes) & 31u);
  atomicOr(a.tlp_amb + static_cast<int64_t>(pn), bit);
}

// Decisions-only mode (template flag D): nothing is written to the score tables; each wave folds the weighted sum
// w_alloc * alloc + w_tlp * tlp (+ the bytes of further Score-only tables) of its 64 x NPL nodes into ONE 32-bit key per pod,
//     key = total << 11 | (1024 - node's index in the tile),        0 = no node of the tile is a node
// so that "highest total, lowest node" is a plain unsigned maximum: per cell one v_mad_u32_u24 on the byte the table mode would
// have stored (the cell code up to that byte is shared, so the decisions equal spx_eval + spx_eval_best by construction), the
// lane's maximum by v_max3, the wave's by four DPP row steps and a scalar combine, and the tie count as scalar population counts
// of "key > best total << 11" — no cross-lane traffic.  dec.key / dec.ties [tile][row]; k_decide_reduce merges the tiles.
// Round 3's form (per-cell int accumulators, 64-bit keys, three 6-step ds_bpermute butterflies per row; 175 VGPRs, 2 waves per
// SIMD) took 0.84 ms for 10k x 100k against the table mode's 0.41.  The launcher takes this form when every weighted total fits
// 21 bits (sum of the weights <= 8000), otherwise the tables are written and k_best reads them.
constexpr int kDecideExtras = 3;
constexpr int kKeyShift = 11;
struct DecideArgs {
  int32_t w_alloc, w_tlp;
  // score tables of further Score-only plugins of the profile, already evaluated for these rows (LVRB, LowRiskOverCommitment,
  // Peaks): their bytes are read once and folded into the total, so neither this sweep's tables nor a second pass over
  // all tables (k_best) is needed
  int32_t n_extra;
  int32_t w_extra[kDecideExtras];
  const uint8_t* extra[kDecideExtras];  // [pods][row_stride]
  uint32_t* key;   // [n_tiles][rows]
  int32_t* ties;   // [n_tiles][rows]: cells of the tile reaching the tile's best total
  int64_t rows;
};

// every lane of a 16-lane row ends with the row's maximum (DPP quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror); the four
// rows are combined on the scalar unit.  Every lane must be live.
template <int CTRL>
__device__ __forceinline__ uint32_t dmax_dpp(uint32_t v) {
  const uint32_t o = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xf, 0xf, true));
  return o > v ? o : v;
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
  v = dmax_dpp<0xB1>(v);
  v = dmax_dpp<0x4E>(v);
  v = dmax_dpp<0x141>(v);
  v = dmax_dpp<0x140>(v);
  const uint32_t r0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 0));
  const uint32_t r1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 16));
  const uint32_t r2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 32));
  const uint32_t r3 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
  const uint32_t a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}

// Exactness bookkeeping (spx_fetch_stats): each lane counts the cells it re-evaluated; the wave adds them up once, at its
// end, into one of kStatSlots counters per plugin.  One atomic per (rare) cell on a single address cost the config #2 sweep
// 0.6 ms — the L2 serialises same-address atomics — hence per wave and spread over slots.  Every lane must be live here.
__device__ __forceinline__ void flush_stats(unsigned long long* stats, int plugin, unsigned count, int64_t unit) {
  if (!stats || __ballot(count != 0) == 0) return;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) count += static_cast<unsigned>(__shfl_xor(static_cast<int>(count), m));
  if ((threadIdx.x & (kWave - 1)) == 0)
    atomicAdd(stats + (plugin * kStatSlots + static_cast<int>(unit & (kStatSlots - 1))) * kStatStride, static_cast<unsigned long long>(count));
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  const uint32_t lo = static_cast<uint32_t>(__shfl_xor(static_cast<int>(static_cast<uint32_t>(v)), m));
  const uint32_t hi = static_cast<uint32_t>(__shfl_xor(static_cast<int>(static_cast<uint32_t>(v >> 32)), m));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// one cell through the reference's float64 sequence, from the original node columns (the fast sweep's rare path)
__device__ __forceinline__ uint32_t tlp_cell_exact(const TrimaranArgs& a, int64_t n, double pod_milli) {
  TlpNode tn;
  tn.cap = static_cast<double>(a.cap_cpu_milli[n]);
  tn.util_millis = (a.tlp_cpu_util[n] / 100.0) * tn.cap;
  tn.missing = static_cast<double>(a.tlp_missing_milli[n]);
  tn.valid = a.tlp_valid[n] != 0;
  bool zero;
  const double x = tlp_unrounded(tn, pod_milli, a.tlp_target, &zero);
  return zero ? 0u : to_u8(x);
}

// AMB (round 5): the ambiguity bookkeeping leaves the cell.  Which (node, pod value) pairs can be ambiguous is a property of the
// node alone — the score is piecewise linear in the pod's integer millicores, so it comes within the tolerance of a rounding tie
// only at isolated integers (k_tlp_amb_build lists them, per launch, as a bit per (pod value, node tile)).  A row whose pod value
// has no such node in this wave's tile runs the streamlined cell: add, add, compare, fma, select, v_cvt_pk_u8_f32 (which rounds to
// nearest even and clamps by itself: tools/micro/cvt_pk_u8.hip) — 5 instructions instead of ~9.5; the other rows (~8 % of the
// (row, tile) pairs on continuous inputs), rows of pods outside the table and waves holding an always-exact node take the
// checked cell exactly as before.
template <int NPL, bool A, bool D = false, bool AMB = false>
__global__ __launch_bounds__(kWave* kWavesPerBlock, D ? 3 : 1) void k_tlp_fast2(TrimaranArgs a, int n_tiles, double c1, double c2, DecideArgs dec) {
  SPX_RESOLVE_ROWS(a);
  static_assert(kPodsPerChunk == kWave, "one pod record per lane");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * static_cast<int>(blockDim.x >> 6) + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t pod0 = a.row_begin + chunk * kPodsPerChunk;
  if (pod0 >= a.row_end) return;
  const int n_rows = static_cast<int>((pod0 + kPodsPerChunk < a.row_end) ? kPodsPerChunk : a.row_end - pod0);
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * NPL;
  const bool active = node0 < a.row_stride;

  // one pod record per lane: predicted millicores as float32 (exact below 2^23; larger or negative values send
  // the whole row to the exact path)
  const int64_t my_pod_i = (lane < n_rows) ? a.tlp_pod_milli[pod0 + lane] : 0;
  const int pod_bits = __float_as_int(static_cast<float>(my_pod_i));
  const int pod_bad = (my_pod_i < 0 || my_pod_i >= (1 << 23)) ? 1 : 0;
  int pod_slow = 1;  // this row takes the checked cell in this tile
  if constexpr (AMB) {
    if (lane < n_rows && !pod_bad && my_pod_i < a.tlp_amb_size) pod_slow = static_cast<int>((a.tlp_amb[my_pod_i] >> (tile & 31)) & 1u);
  }

  uint32_t alloc_w[NPL / 4];
  // b2 = b2h + b2l, b2h integer-valued with |b2h| < 2^23, |b2l| <= 0.5 — kept as pairs of nodes so that the two adds
  // of u run as v_pk_add_f32; kc = (coefficient of the u > 0 branch, coefficient of the u <= 0 branch) per node so that
  // both branches come out of one v_pk_fma_f32
  F32x2 b2h[NPL / 2], b2l[NPL / 2];
  F32x2 kc[NPL];
  bool lane_nan = false;  // one of this lane's nodes always takes the exact path
  const double t = a.tlp_target;
  if constexpr (A) {
#pragma unroll
    for (int j = 0; j < NPL / 4; ++j) alloc_w[j] = active ? reinterpret_cast<const uint32_t*>(a.alloc_norm + node0)[j] : 0u;
  }
  // decisions-only mode: per node (w_alloc * Allocatable's byte) << 11 | 1024 - (index in the tile), 0 for a slot past the node
  // list (its TLP constants score 0, so its key stays 0); which bytes of a row's 16 are nodes (masks the other tables' padding)
  uint32_t cbase[D ? NPL : 1], inmask[D ? NPL / 4 : 1];
  if constexpr (D) {
    static_assert(kWave * NPL <= 1024, "the in-tile index takes the key's low 11 bits");
#pragma unroll
    for (int j = 0; j < NPL / 4; ++j) inmask[j] = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const bool in = active && node0 + i < a.n_nodes;
      uint32_t ab = 0;
      if constexpr (A) ab = (alloc_w[i >> 2] >> (8 * (i & 3))) & 0xffu;
      cbase[i] = in ? ((static_cast<uint32_t>(dec.w_alloc) * ab) << kKeyShift) | (1024u - static_cast<uint32_t>(lane * NPL + i)) : 0u;
      inmask[i >> 2] |= in ? 0xffu << (8 * (i & 3)) : 0u;
    }
  }
  const uint32_t wt_sh = D ? static_cast<uint32_t>(dec.w_tlp) << kKeyShift : 0u;
  static_assert(NPL == kTlpNpl, "k_tlp_prepare_fast lays the constants out for this NPL");
  {
    const float4* tab = reinterpret_cast<const float4*>(a.tlp_fast) + static_cast<int64_t>(tile) * NPL * kWave + lane;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float4 v = tab[static_cast<int64_t>(j) * kWave];  // (b2h, b2l, coefficient for u > 0, coefficient for u <= 0)
      lane_nan |= v.x != v.x;
      b2h[j >> 1][j & 1] = v.x;
      b2l[j >> 1][j & 1] = v.y;
      kc[j] = F32x2{v.z, v.w};
    }
  }
  // no early exit for lanes past the row: every lane stays live so that the v_readlane broadcasts below always read
  // registers that were written under a full exec mask (stores are guarded by `active` instead)
  const float tf = static_cast<float>(t);
  constexpr float kHalf = 0.5f - kTol32;
  unsigned reevaluated = 0;  // cells this lane sent through the exact path (spx_fetch_stats), flushed once per wave
  const bool wave_nan = AMB ? __ballot(lane_nan) != 0 : true;  // a node that always takes the exact path: every row of the wave is checked

  for (int r = 0; r < n_rows; ++r) {
    const int64_t row = (pod0 + r) * a.row_stride + node0;
    if constexpr (A && !D) {
      if (active) store_bytes<NPL>(a.out_alloc + row, alloc_w);
    }
    const float pod_f = __int_as_float(__builtin_amdgcn_readlane(pod_bits, r));
    const bool row_bad = __builtin_amdgcn_readlane(pod_bad, r) != 0;
    bool any = row_bad;
    uint32_t w[NPL / 4];
    uint32_t tb[D ? NPL : 1];  // decisions-only mode: the byte of each cell on its own (the table mode packs four per dword)
    const F32x2 pod2{pod_f, pod_f};
    const F32x2 off2{tf, 100.0f};
    const bool row_slow = !AMB || wave_nan || __builtin_amdgcn_readlane(pod_slow, r) != 0;  // wave-uniform
    if (!row_slow) {
      // streamlined: no cell of this row in this tile can be ambiguous (k_tlp_amb_build), so the float32 value rounds to the
      // reference's integer — nothing to track
#pragma unroll
      for (int j = 0; j < NPL / 4; ++j) {
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = j * 4 + q;
          const F32x2 u2 = (pod2 + b2h[i >> 1]) + b2l[i >> 1];
          const float u = u2[i & 1];
          const bool gt = __float_as_int(u) > 0;
          const F32x2 x12 = __builtin_elementwise_fma(kc[i], F32x2{u, u}, off2);
          const float x = gt ? x12.x : x12.y;
          if constexpr (D) tb[i] = __builtin_amdgcn_cvt_pk_u8_f32(x, 0, 0u);
          else acc = __builtin_amdgcn_cvt_pk_u8_f32(x, q, acc);
        }
        w[j] = acc;
      }
    } else {
    // one cell: rounded float32 score and whether it is provably the reference's result
    auto cell = [&](int i, const F32x2& pod2, float* rr) -> bool {
      const F32x2 u2 = (pod2 + b2h[i >> 1]) + b2l[i >> 1];  // shared by the two cells of the pair
      const float u = u2[i & 1];
      const bool gt = __float_as_int(u) > 0;  // u > 0 on the float's bit pattern (NaN is caught by the tie test)
      const F32x2 x12 = __builtin_elementwise_fma(kc[i], F32x2{u, u}, off2);
      const float x = gt ? x12.x : x12.y;
      *rr = __builtin_rintf(x);
      return !(__builtin_fabsf(x - *rr) < kHalf) || !(__builtin_fabsf(u) > kTolU);
    };
    // the row's worst rounding margin and smallest |u| as running float max/min (v_max3/v_min3: half an instruction per
    // cell) instead of 32 compares and a chain of lane-mask ORs; NaN cells (nodes outside the float32 range) are
    // invisible to max/min and are flagged per lane by lane_nan
    float worst = 0.0f, minu = 1e30f;
#pragma unroll
    for (int j = 0; j < NPL / 4; ++j) {
      uint32_t acc = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = j * 4 + q;
        const F32x2 u2 = (pod2 + b2h[i >> 1]) + b2l[i >> 1];
        const float u = u2[i & 1];
        const bool gt = __float_as_int(u) > 0;
        const F32x2 x12 = __builtin_elementwise_fma(kc[i], F32x2{u, u}, off2);
        const float x = gt ? x12.x : x12.y;
        const float rr = __builtin_rintf(x);
        worst = __builtin_fmaxf(worst, __builtin_fabsf(x - rr));
        minu = __builtin_fminf(minu, __builtin_fabsf(u));
        if constexpr (D) tb[i] = __builtin_amdgcn_cvt_pk_u8_f32(rr, 0, 0u);
        else acc = __builtin_amdgcn_cvt_pk_u8_f32(rr, q, acc);
      }
      w[j] = acc;
    }
    any |= lane_nan || !(worst < kHalf) || !(minu > kTolU);
    if (__builtin_expect(any, 0)) {
      // rare (~8e-5 of cells on continuous inputs): find the ambiguous cells again and re-evaluate them exactly from
      // the original node columns.  Recomputing the flags here is cheaper than carrying 16 of them across the branch.
      const double pod_milli = static_cast<double>(a.tlp_pod_milli[pod0 + r]);
      float pf = pod_f;
      asm volatile("" : "+v"(pf));  // opaque copy: keeps the compiler from carrying the 16 flags across the branch instead
      const F32x2 pod2s{pf, pf};
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        float rr;
        if (cell(i, pod2s, &rr) || row_bad) {
          const int64_t n = node0 + i;
          uint32_t b = 0;
          if (n < a.n_nodes) {
            ++reevaluated;
            b = tlp_cell_exact(a, n, pod_milli);
          }
          if constexpr (D) {
            tb[i] = b;
          } else {
            const int sh = (i & 3) * 8;
            w[i >> 2] = (w[i >> 2] & ~(0xffu << sh)) | (b << sh);
          }
        }
      }
    }
    }  // checked row
    if constexpr (!D) {
      if (active) store_bytes<NPL>(a.out_tlp + row, w);
    } else {
      // one key per cell: total << 11 | 1024 - index in the tile (the cells' registers are reused)
#pragma unroll
      for (int i = 0; i < NPL; ++i) tb[i] = __umul24(tb[i], wt_sh) + cbase[i];
      for (int x = 0; x < dec.n_extra; ++x) {
        const uint32_t wx = static_cast<uint32_t>(dec.w_extra[x]) << kKeyShift;
        uint32_t xw[NPL / 4];
#pragma unroll
        for (int j = 0; j < NPL / 4; ++j) xw[j] = active ? reinterpret_cast<const uint32_t*>(dec.extra[x] + row)[j] & inmask[j] : 0u;
#pragma unroll
        for (int i = 0; i < NPL; ++i) tb[i] += __umul24((xw[i >> 2] >> (8 * (i & 3))) & 0xffu, wx);
      }
      uint32_t kmax = 0;
#pragma unroll
      for (int i = 0; i < NPL; ++i) kmax = tb[i] > kmax ? tb[i] : kmax;
      // wave: every lane is live here (see the note above the loop)
      const uint32_t wkey = wave_umax(kmax);
      // cells of the tile reaching its best total: scalar population counts of the per-cell compare masks
      int t = 0;
      if (wkey != 0u) {
        const uint32_t thr = wkey & ~((1u << kKeyShift) - 1u);
#pragma unroll
        for (int i = 0; i < NPL; ++i) t += __builtin_popcountll(__ballot(tb[i] > thr));  // (a node's low bits are >= 1: a slot past the list, key 0, never counts)
      }
      if (lane == 0) {
        const int64_t slot = static_cast<int64_t>(tile) * dec.rows + (pod0 + r - a.row_begin);
        dec.key[slot] = wkey;
        dec.ties[slot] = t;
      }
    }
  }
  flush_stats(a.stats, SPX_PLUGIN_TLP, reevaluated, unit);
}

// merges the per-tile (key, tie count) pairs of the decisions-only sweep into the layout spx_fetch_best reads: the best total, the
// lowest tile holding it (= the lowest node), the tie counts of every tile holding it
__global__ void k_decide_reduce(DecideArgs dec, int n_tiles, int64_t row_begin, int64_t n_nodes, int64_t* best_score, int32_t* best_node,
                                int32_t* best_ties, int32_t* best_feasible) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= dec.rows) return;
  uint32_t best = 0;
  int best_tile = -1;
  for (int t = 0; t < n_tiles; ++t) {
    const uint32_t k = dec.key[static_cast<int64_t>(t) * dec.rows + r];
    if (k != 0u && (best_tile < 0 || (k >> kKeyShift) > (best >> kKeyShift))) best = k, best_tile = t;
  }
  int32_t ties = 0;
  for (int t = 0; t < n_tiles; ++t) {
    const int64_t slot = static_cast<int64_t>(t) * dec.rows + r;
    const uint32_t k = dec.key[slot];
    if (best_tile >= 0 && k != 0u && (k >> kKeyShift) == (best >> kKeyShift)) ties += dec.ties[slot];
  }
  const int64_t pod = row_begin + r;
  const bool any = best_tile >= 0;
  best_score[pod] = any ? static_cast<int64_t>(best >> kKeyShift) : 0;
  best_node[pod] = any ? best_tile * (kWave * kTlpNpl) + static_cast<int32_t>(1024u - (best & ((1u << kKeyShift) - 1u))) : -1;
  best_ties[pod] = any ? ties : 0;
  best_feasible[pod] = static_cast<int32_t>(n_nodes);
}

// ------------------------------------------------------------------------------------------------
// LoadVariationRiskBalancing fast path (bit-exact by construction, same scheme as k_tlp_fast2).
//
// For a resource in its regular state the reference's score is affine in the pod's request between two clamps:
//     score_r = (1 - (mu + sigma)/2) * 100 = A - 50 * clamp01(B*(usedAvg + req)),   A = 100 - 50*sigma, B = 1/cap  (round 6: the clamp is the fma's)
// A, B and C = B*usedAvg are per-node constants (sigma includes math.Pow / margin, evaluated once per node in
// float64); per cell the kernel evaluates two float32 fma + v_med3, min/max, rndne and a tie test.  Non-regular
// states (metric absent, capacity <= 0) are encoded as A = B = C = 0 (score_r = 0), and "both resources valid"
// (min instead of max, loadvariationriskbalancing.go:112-116) is one bit per node.
// |x' - x| <= 50*2.4e-7 + ulp32(100) <= 2e-5; cells within kTolLv of a rounding tie are recomputed exactly
// (lv_total) from the node's original columns.
constexpr float kTolLv = 6e-5f;

// per-node exact LVRB state {cap, usedAvg, sigma, state} x {cpu, memory} + has_metrics, computed once per launch so
// that the per-cell exact fallback is two loads and one division per resource instead of the full lv_make
__global__ void k_lvrb_prepare(TrimaranArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.n_nodes) return;
  const uint8_t f = a.lv_flags[n];
  double mcap = static_cast<double>(a.lv_alloc_mem[n]);
  mcap *= kMega;
  const LvRes c = lv_make((f & SPX_LV_CPU_VALID) != 0, static_cast<double>(a.lv_alloc_cpu_milli[n]), a.lv_cpu_avg[n], a.lv_cpu_std[n],
                          a.lv_margin, a.lv_sensitivity);
  const LvRes m = lv_make((f & SPX_LV_MEM_VALID) != 0, mcap, a.lv_mem_avg[n], a.lv_mem_std[n], a.lv_margin, a.lv_sensitivity);
  double* o = a.lv_exact + n * 8;
  o[0] = c.cap; o[1] = c.used_avg; o[2] = c.sigma; o[3] = static_cast<double>(c.state);
  o[4] = m.cap; o[5] = m.used_avg; o[6] = m.sigma; o[7] = static_cast<double>(m.state + ((f & SPX_LV_HAS_METRICS) ? 8 : 0));
}

// The float32 per-node constants of the two fast sweeps, once per launch instead of once per (node tile, pod chunk) unit
// — for LVRB that setup (lv_make: divisions, math.Pow) cost as much as the 64 rows it served.  One thread per padded
// node slot; slots past n_nodes get constants that score 0 and never look ambiguous.
__global__ void k_lvrb_prepare_fast(TrimaranArgs a, int64_t n_slots) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= n_slots) return;
  float4 v0{0.0f, 0.0f, 0.0f, 0.0f}, v1{0.0f, 0.0f, __builtin_inff(), 0.0f};
  if (n < a.n_nodes) {
    const uint8_t f = a.lv_flags[n];
    const bool has = (f & SPX_LV_HAS_METRICS) != 0;
    double mcap = static_cast<double>(a.lv_alloc_mem[n]);
    mcap *= kMega;
    const LvRes c = lv_make(has && (f & SPX_LV_CPU_VALID), static_cast<double>(a.lv_alloc_cpu_milli[n]), a.lv_cpu_avg[n], a.lv_cpu_std[n],
                            a.lv_margin, a.lv_sensitivity);
    const LvRes m = lv_make(has && (f & SPX_LV_MEM_VALID), mcap, a.lv_mem_avg[n], a.lv_mem_std[n], a.lv_margin, a.lv_sensitivity);
    auto consts = [](const LvRes& r, float* fa, float* fb, float* fc) {
      if (r.state != 2) {
        *fa = *fb = *fc = 0.0f;
        return;
      }
      // round 6: slope and offset of t / 50 = (used + request) / cap, so that the sweep clamps with the fma's output modifier (0..1)
      // instead of a v_med3_f32 (0..50) per resource; A - 50 * clamp01(t / 50) is the same real number
      const double b = 1.0 / r.cap;
      *fa = static_cast<float>(100.0 - 50.0 * r.sigma);
      *fb = static_cast<float>(b);
      *fc = static_cast<float>(b * r.used_avg);
    };
    float ca, cb, cc, ma, mb, mc;
    consts(c, &ca, &cb, &cc);
    consts(m, &ma, &mb, &mc);
    // both resources valid: the reference takes the min of the two scores, else the max (loadvariationriskbalancing.go:112-116) — the
    // sweep's v_med3_f32(x_cpu, x_mem, L) with L = -inf / +inf (round 6b; before: a per-node sign on A and on the result)
    const float pick = (has && c.state != 0 && m.state != 0) ? -__builtin_inff() : __builtin_inff();
    v0 = float4{cb, mb, cc, mc};
    v1 = float4{ca, ma, pick, 0.0f};
  }
  float4* out = reinterpret_cast<float4*>(a.lv_fast) + tile_slot<kLvNpl>(n) * 2;
  out[0] = v0;
  out[1] = v1;
}

// Which (request value, node tile) pairs hold a cell k_lvrb_fast cannot prove, as k_tlp_amb_build does it for TargetLoadPacking.  Per node
// and resource the float32 value is y_r(q) = s * (A_r - clamp(B_r * q + C_r, 0, 50)) in the pod's INTEGER request q (cpu: millicores; memory:
// MiB — rows whose memory request is not a whole number of MiB keep the checked cell), and the cell is x = s * max(y_cpu, y_mem): it can
// only be near a rounding tie when the maximum's argument is, so a cell neither of whose y_r is within the tolerance of k + 0.5 is provable.
// On the linear piece y_r meets k + 0.5 at one real q*, and only the integer next to it can be within tau / B_r of it; on the clamped
// pieces y_r is constant — a constant that is itself near a tie (integer-valued metrics make that common) makes every request beyond the
// clamp ambiguous, and sends the node's whole tile to the checked cell (bit tile & 31 of the table's last word), as does a slope so
// flat that more than 16 integers either side of a tie would have to be listed.  Layout: [cpu values 0 .. kLvAmbCpu) | [memory MiB 0 .. kLvAmbMem) | one word of always-checked tiles.
// One block per node: thread j < 64 -> cpu, tie k = j; 64 <= j < 128 -> memory.
__global__ __launch_bounds__(128) void k_lvrb_amb_build(TrimaranArgs a, int tile_nodes) {
  const int64_t n = blockIdx.x;
  if (n >= a.n_nodes) return;
  const int j = threadIdx.x & 63;
  const bool is_mem = threadIdx.x >= 64;
  const double* o = a.lv_exact + n * 8;  // k_lvrb_prepare's exact state: {cap, usedAvg, sigma, state} x {cpu, memory}; o[7] also carries has_metrics
  const int ms = static_cast<int>(o[7]);
  if (!(ms & 8)) return;  // no metrics: the score is 0 for every pod
  const double cap = is_mem ? o[4] : o[0], used = is_mem ? o[5] : o[1], sigma = is_mem ? o[6] : o[2];
  const int state = is_mem ? (ms & 7) : static_cast<int>(o[3]);
  if (state != 2) return;  // this resource scores 0: an integer, never near a tie
  const uint32_t bit = 1u << (static_cast<uint32_t>(n / tile_nodes) & 31u);
  uint32_t* const tab = a.lv_amb + (is_mem ? kLvAmbCpu : 0);
  const int size = is_mem ? kLvAmbMem : kLvAmbCpu;
  uint32_t* const always = a.lv_amb + kLvAmbCpu + kLvAmbMem;
  constexpr double tau = 6e-5 * kAmbMargin;  // kTolLv with the margin
  const double big_a = 100.0 - 50.0 * sigma, b = 50.0 / cap;
  const double radius = tau / b;  // integers within this distance of a tie's real solution are listed (a 1 TiB node: B = 4.8e-5 per MiB, radius 1.6)
  if (!(radius <= 16.0)) {    // flatter than that (more than ~10 TiB / 10^7 millicores, or a NaN capacity): the whole tile keeps the checked cell
    if (j == 0) atomicOr(always, bit);
    return;
  }
  if (j == 0) {  // the clamped pieces: y = A (t = 0; only q with B q + C <= 0, i.e. the linear piece's own end) and y = A - 50 (every large q)
    const double lo = big_a - 50.0;
    if (__builtin_fabs(lo - __builtin_floor(lo) - 0.5) < tau) atomicOr(always, bit);
  }
  if (j > 51) return;
  // ties of |y| = A - t at k + 0.5 for t in [0, 50]: t = A - (k + 0.5) with k = floor(A - 0.5) - j  (the sign s only mirrors y)
  const double k_top = __builtin_floor(big_a - 0.5);
  const double t = big_a - (k_top - j + 0.5);
  if (!(t >= -1e-3) || !(t <= 50.001)) return;
  const double q_star = (t - b * used) / b;
  const double q_lo = __builtin_ceil(q_star - radius), q_hi = __builtin_floor(q_star + radius);
  for (double q = q_lo; q <= q_hi; q += 1.0)
    if (q >= 0.0 && q < static_cast<double>(size)) atomicOr(tab + static_cast<int64_t>(q), bit);
}

template <int NPL, bool A, bool AMB = false>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_lvrb_fast(TrimaranArgs a, int n_tiles) {
  SPX_RESOLVE_ROWS(a);
  static_assert(kPodsPerChunk == kWave, "one pod record per lane");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t pod0 = a.row_begin + chunk * kPodsPerChunk;
  if (pod0 >= a.row_end) return;
  const int n_rows = static_cast<int>((pod0 + kPodsPerChunk < a.row_end) ? kPodsPerChunk : a.row_end - pod0);
  const int64_t node0 = (static_cast<int64_t>(tile) * kWave + lane) * NPL;
  const bool active = node0 < a.row_stride;

  // one pod record per lane: requests as float32 (cpu millicores; memory in MiB like the reference), and whether
  // the row must take the exact path (negative or huge requests)
  float my_cpu = 0.0f, my_mem = 0.0f;
  int my_bad = 0;
  if (lane < n_rows) {
    const double rc = static_cast<double>(a.lv_req_cpu_milli[pod0 + lane]);
    const double rm = static_cast<double>(a.lv_req_mem[pod0 + lane]) * kMega;
    my_cpu = static_cast<float>(rc);
    my_mem = static_cast<float>(rm);
    my_bad = (!(rc >= 0.0) || !(rm >= 0.0) || !(rc < 1e15) || !(rm < 1e15)) ? 1 : 0;
  }
  int my_slow = 1;  // this row takes the checked cell in this tile (k_lvrb_amb_build)
  if constexpr (AMB) {
    if (lane < n_rows && !my_bad) {
      const int64_t qc = a.lv_req_cpu_milli[pod0 + lane], qm_bytes = a.lv_req_mem[pod0 + lane];
      if (qc < kLvAmbCpu && (qm_bytes & 0xfffff) == 0 && (qm_bytes >> 20) < kLvAmbMem)
        my_slow = static_cast<int>(((a.lv_amb[qc] | a.lv_amb[kLvAmbCpu + (qm_bytes >> 20)] | a.lv_amb[kLvAmbCpu + kLvAmbMem]) >> (tile & 31)) & 1u);
    }
  }
  const int cpu_bits = __float_as_int(my_cpu), mem_bits = __float_as_int(my_mem);

  uint32_t alloc_w[NPL / 4];
  // per node, (cpu, memory) pairs for v_pk_fma_f32: slope B, offset C = B*usedAvg (of t / 50: the clamp is the fma's 0..1), A; and L = -inf
  // when both resources are valid (the reference takes the min of the two scores then), +inf otherwise (max):
  //   x_r = A_r - 50 clamp01(B_r q + C_r),  x = v_med3_f32(x_cpu, x_mem, L)   — one instruction and one register per node fewer than the
  //   per-node sign of rounds 2-6a (y_r = s x_r, x = s max(y_cpu, y_mem)); the same float32 values: an fma is odd in its sign
  F32x2 kb[NPL], kc[NPL], ka[NPL];
  float kl[NPL];
  if constexpr (A) {
#pragma unroll
    for (int j = 0; j < NPL / 4; ++j) alloc_w[j] = active ? reinterpret_cast<const uint32_t*>(a.alloc_norm + node0)[j] : 0u;
  }
  static_assert(NPL == kLvNpl, "k_lvrb_prepare_fast lays the constants out for this NPL");
  {
    const float4* tab = reinterpret_cast<const float4*>(a.lv_fast) + (static_cast<int64_t>(tile) * NPL * kWave + lane) * 2;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const float4 v0 = tab[static_cast<int64_t>(j) * kWave * 2], v1 = tab[static_cast<int64_t>(j) * kWave * 2 + 1];
      kb[j] = F32x2{v0.x, v0.y};
      kc[j] = F32x2{v0.z, v0.w};
      ka[j] = F32x2{v1.x, v1.y};
      kl[j] = v1.z;
    }
  }
  constexpr float kHalf = 0.5f - kTolLv;  // (no early exit: see k_tlp_fast2)
  unsigned reevaluated = 0;  // as in k_tlp_fast2

  for (int r = 0; r < n_rows; ++r) {
    const int64_t row = (pod0 + r) * a.row_stride + node0;
    if constexpr (A) {
      if (active) store_bytes<NPL>(a.out_alloc + row, alloc_w);
    }
    const float req_cpu = __int_as_float(__builtin_amdgcn_readlane(cpu_bits, r));
    const float req_mem = __int_as_float(__builtin_amdgcn_readlane(mem_bits, r));
    const bool row_bad = __builtin_amdgcn_readlane(my_bad, r) != 0;
    bool any = row_bad;
    uint32_t w[NPL / 4];
    // one cell: x, its rounding, and the distance to the rounding tie
    const F32x2 m50{-50.0f, -50.0f};
    auto cell = [&](int i, const F32x2& req, float* rx) -> float {
      const F32x2 cl{__builtin_amdgcn_fmed3f(__builtin_fmaf(kb[i].x, req.x, kc[i].x), 0.0f, 1.0f),   // v_fma_f32 .. clamp, full rate
                     __builtin_amdgcn_fmed3f(__builtin_fmaf(kb[i].y, req.y, kc[i].y), 0.0f, 1.0f)};
      const F32x2 x2 = __builtin_elementwise_fma(m50, cl, ka[i]);
      const float x = __builtin_amdgcn_fmed3f(x2.x, x2.y, kl[i]);
      *rx = __builtin_rintf(x);
      return __builtin_fabsf(x - *rx);
    };
    const F32x2 req2{req_cpu, req_mem};
    const bool row_slow = !AMB || __builtin_amdgcn_readlane(my_slow, r) != 0;  // wave-uniform
    if (!row_slow) {
      // streamlined: no cell of this row in this tile can be near a tie (k_lvrb_amb_build) — v_cvt_pk_u8_f32 rounds to nearest even itself
#pragma unroll
      for (int j = 0; j < NPL / 4; ++j) {
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = j * 4 + q;
          const F32x2 cl{__builtin_amdgcn_fmed3f(__builtin_fmaf(kb[i].x, req2.x, kc[i].x), 0.0f, 1.0f),
                         __builtin_amdgcn_fmed3f(__builtin_fmaf(kb[i].y, req2.y, kc[i].y), 0.0f, 1.0f)};
          const F32x2 x2 = __builtin_elementwise_fma(m50, cl, ka[i]);
          acc = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(x2.x, x2.y, kl[i]), q, acc);
        }
        w[j] = acc;
      }
      if (active) store_bytes<NPL>(a.out_lvrb + row, w);
      continue;
    }
    float worst = 0.0f;  // running max of the rounding margins (v_max3_f32), as in k_tlp_fast2
#pragma unroll
    for (int j = 0; j < NPL / 4; ++j) {
      uint32_t acc = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = j * 4 + q;
        float rx;
        worst = __builtin_fmaxf(worst, cell(i, req2, &rx));
        acc = __builtin_amdgcn_cvt_pk_u8_f32(rx, q, acc);
      }
      w[j] = acc;
    }
    any |= !(worst < kHalf);
    if (__builtin_expect(any, 0)) {
      const double req_cpu_d = fmax(static_cast<double>(a.lv_req_cpu_milli[pod0 + r]), 0.0);
      const double req_mem_d = fmax(static_cast<double>(a.lv_req_mem[pod0 + r]) * kMega, 0.0);
      float rc = req_cpu, rm = req_mem;
      asm volatile("" : "+v"(rc), "+v"(rm));  // opaque copies: recompute the flags here instead of carrying them across the branch
      const F32x2 req2s{rc, rm};
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        float ry;
        if (!(cell(i, req2s, &ry) < kHalf) || row_bad) {  // exact re-evaluation of this cell from the original node columns
          const int64_t n = node0 + i;
          uint32_t b = 0;
          if (n < a.n_nodes) {
            ++reevaluated;
            const double* o = a.lv_exact + n * 8;
            const int ms = static_cast<int>(o[7]);
            const LvRes c{o[0], o[1], o[2], static_cast<int>(o[3])};
            const LvRes m{o[4], o[5], o[6], ms & 7};
            b = to_u8(lv_total((ms & 8) != 0, c, m, req_cpu_d, req_mem_d));
          }
          const int sh = (i & 3) * 8;
          w[i >> 2] = (w[i >> 2] & ~(0xffu << sh)) | (b << sh);
        }
      }
    }
    if (active) store_bytes<NPL>(a.out_lvrb + row, w);
  }
  flush_stats(a.stats, SPX_PLUGIN_LVRB, reevaluated, unit);
}

// raw int64 Score() of one row (parity harness / direct-call tests); one thread per node
__global__ void k_trimaran_raw(TrimaranArgs a, int plugin, int64_t pod, int64_t* out) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.n_nodes) return;
  double x = 0.0;
  bool zero = false;
  if (plugin == SPX_PLUGIN_TLP) {
    TlpNode t;
    t.cap = static_cast<double>(a.cap_cpu_milli[n]);
    t.util_millis = (a.tlp_cpu_util[n] / 100.0) * t.cap;
    t.missing = static_cast<double>(a.tlp_missing_milli[n]);
    t.valid = a.tlp_valid[n] != 0;
    x = tlp_unrounded(t, static_cast<double>(a.tlp_pod_milli[pod]), a.tlp_target, &zero);
  } else {
    const uint8_t f = a.lv_flags[n];
    double mcap = static_cast<double>(a.lv_alloc_mem[n]);
    mcap *= kMega;
    const LvRes c = lv_make((f & SPX_LV_CPU_VALID) != 0, static_cast<double>(a.lv_alloc_cpu_milli[n]), a.lv_cpu_avg[n],
                            a.lv_cpu_std[n], a.lv_margin, a.lv_sensitivity);
    const LvRes m = lv_make((f & SPX_LV_MEM_VALID) != 0, mcap, a.lv_mem_avg[n], a.lv_mem_std[n], a.lv_margin,
                            a.lv_sensitivity);
    const double req_cpu = fmax(static_cast<double>(a.lv_req_cpu_milli[pod]), 0.0);
    const double req_mem = fmax(static_cast<double>(a.lv_req_mem[pod]) * kMega, 0.0);
    x = lv_total((f & SPX_LV_HAS_METRICS) != 0, c, m, req_cpu, req_mem);
  }
  out[n] = zero ? 0 : static_cast<int64_t>(round(x));
}

template <int NPL>
void launch_npl(const TrimaranArgs& a, hipStream_t s) {
  const int tile_nodes = kWave * NPL;
  const int n_tiles = static_cast<int>((a.row_stride + tile_nodes - 1) / tile_nodes);
  const int64_t rows = a.row_end - a.row_begin;
  const int64_t chunks = (rows + kPodsPerChunk - 1) / kPodsPerChunk;
  const int64_t units = chunks * n_tiles;
  const unsigned blocks = static_cast<unsigned>((units + kWavesPerBlock - 1) / kWavesPerBlock);
  const dim3 block(kWave * kWavesPerBlock);
  const bool A = a.out_alloc != nullptr, T = a.out_tlp != nullptr, L = a.out_lvrb != nullptr;
#define SPX_CASE(AA, TT, LL)                                                                      \
  if (A == AA && T == TT && L == LL) {                                                            \
    hipLaunchKernelGGL((k_trimaran<NPL, AA, TT, LL>), dim3(blocks), block, 0, s, a, n_tiles);    \
    return;                                                                                       \
  }
  SPX_CASE(true, false, false)
  SPX_CASE(false, true, false)
  SPX_CASE(false, false, true)
  SPX_CASE(true, true, false)
  SPX_CASE(true, false, true)
  SPX_CASE(false, true, true)
  SPX_CASE(true, true, true)
#undef SPX_CASE
}

}  // namespace

void launch_alloc_prepare(const AllocPrepArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_alloc_prepare, dim3(1), dim3(1024), 0, s, a);
}

// the table of k_tlp_fast2<..., AMB>: cleared and rebuilt when the owner says the node columns or the target changed since it was
// built (uploads, deltas, commits, spx_set_tlp_params: TrimaranArgs::tlp_amb_built).  Worth it for multi-row launches only; a
// single-row launch (the commit loop's: row_ptr) and launches of fewer than 256 rows keep the checked cell.
static bool tlp_amb_eligible(const TrimaranArgs& a, int64_t rows) {
  return a.tlp_amb && a.tlp_amb_size > 0 && !a.row_ptr && rows >= 256 && !(a.opts & kOptTlpNoAmbTable);
}
// k_tlp_prepare_fast's constants (always) and, for an eligible launch, k_tlp_amb_build's table — unless the owner's flag says both
// still describe the node columns in place (they are built together and invalidated together); true = launch the AMB variant
static bool tlp_prepare(const TrimaranArgs& a, int64_t rows, int64_t n_slots, double c1, double c2, int tile_nodes, hipStream_t s) {
  const bool amb = tlp_amb_eligible(a, rows);
  int64_t tbits;
  static_assert(sizeof tbits == sizeof a.tlp_target, "the target's bits");
  __builtin_memcpy(&tbits, &a.tlp_target, sizeof tbits);
  const int64_t geom[3] = {tile_nodes, a.row_stride, tbits ^ (static_cast<int64_t>(a.tlp_amb_size) << 1)};
  const bool same_geom = !a.tlp_amb_geom || (a.tlp_amb_geom[0] == geom[0] && a.tlp_amb_geom[1] == geom[1] && a.tlp_amb_geom[2] == geom[2]);
  if (amb && a.tlp_amb_built && *a.tlp_amb_built && same_geom) return true;  // 21 us per sweep otherwise
  if (a.tlp_amb_geom) a.tlp_amb_geom[0] = geom[0], a.tlp_amb_geom[1] = geom[1], a.tlp_amb_geom[2] = geom[2];
  hipLaunchKernelGGL(k_tlp_prepare_fast, dim3(static_cast<unsigned>((n_slots + 255) / 256)), dim3(256), 0, s, a, n_slots, c1, c2);
  if (!amb) {
    if (a.tlp_amb_built) *a.tlp_amb_built = false;  // (a single-row launch of the commit loop writes constants of its own state)
    return false;
  }
  (void)hipMemsetAsync(a.tlp_amb, 0, static_cast<size_t>(a.tlp_amb_size) * 4, s);
  hipLaunchKernelGGL(k_tlp_amb_build, dim3(static_cast<unsigned>(a.n_nodes)), dim3(256), 0, s, a, c1, c2, tile_nodes);
  if (a.tlp_amb_built) *a.tlp_amb_built = true;
  return true;
}

template <int NPL>
void launch_tlp_fast(const TrimaranArgs& a, hipStream_t s) {
  const int tile_nodes = kWave * NPL;
  const int n_tiles = static_cast<int>((a.row_stride + tile_nodes - 1) / tile_nodes);
  const int64_t rows = a.row_end - a.row_begin;
  const int64_t chunks = (rows + kPodsPerChunk - 1) / kPodsPerChunk;
  const unsigned blocks = static_cast<unsigned>((chunks * n_tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  const double t = a.tlp_target;
  const double c1 = t / (100.0 - t), c2 = (100.0 - t) / t;
  const int64_t n_slots = static_cast<int64_t>(n_tiles) * tile_nodes;
  if (tlp_prepare(a, rows, n_slots, c1, c2, tile_nodes, s)) {
    if (a.out_alloc)
      hipLaunchKernelGGL((k_tlp_fast2<NPL, true, false, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, DecideArgs{});
    else
      hipLaunchKernelGGL((k_tlp_fast2<NPL, false, false, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, DecideArgs{});
    return;
  }
  if (a.out_alloc)
    hipLaunchKernelGGL((k_tlp_fast2<NPL, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, DecideArgs{});
  else
    hipLaunchKernelGGL((k_tlp_fast2<NPL, false>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, DecideArgs{});
}

template <int NPL>
void launch_lvrb_fast(const TrimaranArgs& a, hipStream_t s) {
  const int tile_nodes = kWave * NPL;
  const int n_tiles = static_cast<int>((a.row_stride + tile_nodes - 1) / tile_nodes);
  const int64_t chunks = (a.row_end - a.row_begin + kPodsPerChunk - 1) / kPodsPerChunk;
  const unsigned blocks = static_cast<unsigned>((chunks * n_tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  const int64_t n_slots = static_cast<int64_t>(n_tiles) * tile_nodes;
  // the per-node constants and, for a multi-row launch, the ambiguity table: built together, kept while the owner's flag says the
  // columns they were built from (and the margin / sensitivity) are the ones in place — as launch_tlp_fast does
  const bool amb = a.lv_amb && !a.row_ptr && a.row_end - a.row_begin >= 256 && !(a.opts & kOptTlpNoAmbTable);
  const bool lv_same_geom = !a.lv_amb_geom || (a.lv_amb_geom[0] == tile_nodes && a.lv_amb_geom[1] == a.row_stride);
  if (a.lv_amb_geom) a.lv_amb_geom[0] = tile_nodes, a.lv_amb_geom[1] = a.row_stride;
  if (!(amb && a.lv_amb_built && *a.lv_amb_built && lv_same_geom)) {
    hipLaunchKernelGGL(k_lvrb_prepare, dim3(static_cast<unsigned>((a.n_nodes + 255) / 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_lvrb_prepare_fast, dim3(static_cast<unsigned>((n_slots + 255) / 256)), dim3(256), 0, s, a, n_slots);
    if (amb) {
      (void)hipMemsetAsync(a.lv_amb, 0, static_cast<size_t>(kLvAmbCpu + kLvAmbMem + 1) * 4, s);
      hipLaunchKernelGGL(k_lvrb_amb_build, dim3(static_cast<unsigned>(a.n_nodes)), dim3(128), 0, s, a, tile_nodes);
    }
    if (a.lv_amb_built) *a.lv_amb_built = amb;
  }
  if (amb) {
    if (a.out_alloc)
      hipLaunchKernelGGL((k_lvrb_fast<NPL, true, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles);
    else
      hipLaunchKernelGGL((k_lvrb_fast<NPL, false, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles);
  } else if (a.out_alloc)
    hipLaunchKernelGGL((k_lvrb_fast<NPL, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles);
  else
    hipLaunchKernelGGL((k_lvrb_fast<NPL, false>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles);
}

size_t lvrb_amb_bytes() { return static_cast<size_t>(kLvAmbCpu + kLvAmbMem + 1) * 4; }

void launch_trimaran(const TrimaranArgs& a, hipStream_t s) {
  if (a.row_end <= a.row_begin) return;
  if (!a.out_alloc && !a.out_tlp && !a.out_lvrb) return;
  const bool exact_only = (a.opts & kOptTrimaranExact) != 0;  // SPX_OPT_REFERENCE_KERNELS
  const bool tlp_fast_ok = a.tlp_target >= 1.0 && a.tlp_target <= 99.0;
  if (!exact_only && (a.out_tlp || a.out_lvrb) && (!a.out_tlp || (tlp_fast_ok && a.tlp_fast)) && (!a.out_lvrb || (a.lv_exact && a.lv_fast))) {
    // one bit-exact fast kernel per plugin; Allocatable's broadcast row rides with the first of them
    TrimaranArgs t = a;
    if (a.out_tlp) {
      t.out_lvrb = nullptr;
      launch_tlp_fast<16>(t, s);
    }
    if (a.out_lvrb) {
      t = a;
      t.out_tlp = nullptr;
      if (a.out_tlp) t.out_alloc = nullptr;
      launch_lvrb_fast<8>(t, s);
    }
    return;
  }
  // nodes per lane: wide stores when only TLP state (3 doubles/node) must stay resident,
  // narrower when LVRB adds 6 more doubles per node
  const bool L = a.out_lvrb != nullptr;
  const bool T = a.out_tlp != nullptr;
  if (L && T) launch_npl<4>(a, s);
  else if (L) launch_npl<8>(a, s);
  else launch_npl<16>(a, s);
}

size_t decide_scratch_bytes(int64_t row_stride, int64_t rows) {
  const int n_tiles = static_cast<int>((row_stride + kWave * kTlpNpl - 1) / (kWave * kTlpNpl));
  return static_cast<size_t>(n_tiles) * static_cast<size_t>(rows) * (sizeof(uint32_t) + sizeof(int32_t));
}

void launch_decide_trimaran(const DecideLaunch& d, hipStream_t s) {
  const TrimaranArgs& a = d.t;
  constexpr int NPL = kTlpNpl;
  const int tile_nodes = kWave * NPL;
  const int n_tiles = static_cast<int>((a.row_stride + tile_nodes - 1) / tile_nodes);
  const int64_t rows = a.row_end - a.row_begin;
  if (rows <= 0) return;
  const int64_t chunks = (rows + kPodsPerChunk - 1) / kPodsPerChunk;
  const unsigned blocks = static_cast<unsigned>((chunks * n_tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  const double t = a.tlp_target;
  const double c1 = t / (100.0 - t), c2 = (100.0 - t) / t;
  const int64_t n_slots = static_cast<int64_t>(n_tiles) * tile_nodes;
  DecideArgs dec;
  dec.w_alloc = d.w_alloc;
  dec.w_tlp = d.w_tlp;
  dec.n_extra = d.n_extra;
  for (int x = 0; x < kDecideExtras; ++x) dec.w_extra[x] = d.w_extra[x], dec.extra[x] = d.extra[x];
  dec.key = static_cast<uint32_t*>(d.scratch);
  dec.ties = reinterpret_cast<int32_t*>(dec.key + static_cast<int64_t>(n_tiles) * rows);
  dec.rows = rows;
  if (tlp_prepare(a, rows, n_slots, c1, c2, tile_nodes, s)) {
    if (d.use_alloc)
      hipLaunchKernelGGL((k_tlp_fast2<NPL, true, true, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, dec);
    else
      hipLaunchKernelGGL((k_tlp_fast2<NPL, false, true, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, dec);
  } else if (d.use_alloc)
    hipLaunchKernelGGL((k_tlp_fast2<NPL, true, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, dec);
  else
    hipLaunchKernelGGL((k_tlp_fast2<NPL, false, true>), dim3(blocks), dim3(kWave * kWavesPerBlock), 0, s, a, n_tiles, c1, c2, dec);
  hipLaunchKernelGGL(k_decide_reduce, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0, s, dec, n_tiles, a.row_begin, a.n_nodes,
                     d.best_score, d.best_node, d.best_ties, d.best_feasible);
}

void launch_trimaran_raw(const TrimaranArgs& a, int plugin, int64_t pod_row, int64_t* out, hipStream_t s) {
  const unsigned blocks = static_cast<unsigned>((a.n_nodes + 255) / 256);
  hipLaunchKernelGGL(k_trimaran_raw, dim3(blocks), dim3(256), 0, s, a, plugin, pod_row, out);
}

}  // namespace spx
