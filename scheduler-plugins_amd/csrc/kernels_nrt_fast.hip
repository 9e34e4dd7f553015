// kernels_nrt_fast.hip — float64 formulation of the NodeResourceTopologyMatch sweep (Filter + Score for the
// LeastAllocated / MostAllocated / BalancedAllocation strategies).
//
// Same decomposition as kernels_nrt.hip (lane = node, pod record wave-uniform) and the same results, bit for
// bit; what changes is the arithmetic.  The generic kernel works on int64 quantities exactly like the
// reference (two VALU instructions per add/compare, ~25 per truncating division, 64-bit id bitmasks).  When
// the engine has verified at upload time that
//   * every NUMA zone's id equals its list position (createNUMANodeList, pluginhelpers.go:105-131, with the
//     usual node-0..node-(Z-1) zones), so "lowest NUMA id" == "lowest list position",
//   * every zone quantity and every request lies in [0, 2^42), and 100 * sum(weights) < 2^42,
// all of those integers are exact in float64 and:
//   * compare / subtract / add-back are single v_cmp_f64 / v_fma_f64 instructions;
//   * the truncating divisions by a capacity c < 2^42 (quotient x <= 100, numerator an integer) become one
//     multiplication by b = RN(100 / c), precomputed per (node, zone, resource), biased so that the float64
//     value t lands in [x, x + 2^-42): since frac(x) <= 1 - 1/c < 1 - 2^-42 for a non-integer x, floor(t) ==
//     floor(x) with no fix-up.  LeastAllocated: (c - v) * 100 / c = 100 - v * (100 / c), t = fma(-v, b, 100 + 2^-43),
//     total rounding error < 2^-45.6; MostAllocated: t = (v * (1 + 2^-49)) * b, relative error within
//     2^-49 +- 3 * 2^-53; the final acc / sum(weights) uses the same biased reciprocal;
//   * Quantity.Value() of a cpu capacity (ceil(milli / 1000)) is folded into b.
// Snapshots that fail the check run the generic kernel.  LeastNUMANodes shares the Filter and replaces the per-lane
// subset enumeration by one wave-uniform search (numa_required_fast).
//
// Reference: pkg/noderesourcetopology/filter.go:42-245, score.go:62-191, least_allocated.go:25-55,
// most_allocated.go:25-54, balanced_allocation.go:27-54, numaresources.go:105-182.
#include <cstdlib>

#include "spx_internal.h"

namespace spx {

namespace {

constexpr int kZ = SPX_NRT_MAX_ZONES;
constexpr int kC = SPX_NRT_MAX_CTRS;
constexpr int kPodsPerUnit = 32;
constexpr int kWindow = 256;  // nodes per block (4 wavefronts)
constexpr int kXcdMapWindows = 32;  // from this many node windows on (8k nodes), blocks are mapped XCD-aware (see k_nrt_fast)
constexpr int kSgLeast = 0;
constexpr int kSgMost = 1;
constexpr int kSgBalanced = 2;
constexpr int kSgLeastNuma = 3;
constexpr double kNoCap = kNrtNoCap;  // b[][] of a cell whose capacity is not positive

template <int RM>
struct FastNode {
  double av[kZ][RM];     // zone reports the resource ? available : -1
  double b[kZ][RM];      // RN(100 / Value(capacity)); kNoCap when the capacity is not positive
  uint32_t rep[RM / 4];  // per resource: 8-bit mask of the zones that report it
  uint32_t fill[RM / 4]; // per resource: 0xff when no zone reports a host-level resource (the check is skipped), else 0
  uint32_t node_present;
  int nz;
  __device__ __forceinline__ uint32_t repmask(int r) const { return (rep[r >> 2] >> (8 * (r & 3))) & 0xffu; }
  __device__ __forceinline__ uint32_t fillmask(int r) const { return (fill[r >> 2] >> (8 * (r & 3))) & 0xffu; }
};

// Wave-uniform read of immutable input through the constant address space: the backend may then use scalar
// loads (s_load_dwordxN into SGPRs).  Through a plain global pointer it cannot — the kernel's own table stores
// might alias — and every pod-record access becomes a vector load with a uniform address (measured: 54 VMEM
// reads per wave per pod, 56 % of wave cycles waiting).
template <typename T>
__device__ __forceinline__ T uload(const T* p) {
  typedef const T __attribute__((address_space(4))) CT;
  return *reinterpret_cast<CT*>(reinterpret_cast<uintptr_t>(p));
}

// The pod record stream (built by the engine at upload, spx_engine.hip: nrt_pod_items): per pod 10 items of IW dwords
// (16 for <= 4 resource slots, else 32) — item 0 the header (2 dwords used), item 1 the pod-level request, items 2..9 the
// containers in order (init containers first).  A request item: doubles raw[RM] (dwords 0..2RM-1), the slot-set dword
// (2RM), a pad, then the three doubles only the Score reads: Value() of the cpu request, the sum of the weights of the
// requested slots and its biased reciprocal.  Items are fetched with scalar loads, the next one before the current one is
// processed so that the SMEM latency overlaps with the VALU work.  How much of an item a kernel keeps in scalar registers
// matters: the Filter needs 2RM + 1 dwords of it, and with three items in flight (pod level, current and next container)
// loading all 16 costs scalar registers and SMEM bandwidth for nothing (measured: config #3 3.2 -> 3.0 ms).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
constexpr int kItemsPerPod = 2 + kC;
template <int RM>
constexpr int item_words() { return RM == 4 ? 16 : 32; }

template <int RM, bool FULL>
struct ItemRegs;
template <>
struct ItemRegs<4, false> {  // the request quantities and the slot sets
  u32x8 raw;
  u32x2 tail;
};
template <>
struct ItemRegs<4, true> {
  u32x16 w;
};
template <bool FULL>
struct ItemRegs<8, FULL> {
  u32x16 lo, hi;
};

template <int RM, bool FULL>
__device__ __forceinline__ ItemRegs<RM, FULL> load_item(const uint32_t* items, int64_t index) {
  const uint32_t* p = items + index * item_words<RM>();
  ItemRegs<RM, FULL> r;
  if constexpr (RM == 4 && !FULL) {
    r.raw = uload(reinterpret_cast<const u32x8*>(p));
    r.tail = uload(reinterpret_cast<const u32x2*>(p + 8));
  } else if constexpr (RM == 4) {
    r.w = uload(reinterpret_cast<const u32x16*>(p));
  } else {
    r.lo = uload(reinterpret_cast<const u32x16*>(p));
    r.hi = uload(reinterpret_cast<const u32x16*>(p + 16));
  }
  return r;
}

template <int RM>
struct Item {
  double raw[RM];  // requests as written (cpu in millicores); 0 for absent slots
  double cpu_v;    // Quantity.Value() of the cpu request (whole cores, rounded up)
  double wsum;     // sum of the weights of the requested slots
  double wrc;      // its biased reciprocal
  uint32_t used;   // requested slots (Score iterates these)
  uint32_t fit;    // non-zero requests compared per zone: available >= quantity
  uint32_t always; // non-zero requests of a non-Guaranteed pod for a NUMA-affine resource: any reporting zone suits
  uint32_t kind;   // SPX_CTR_*
};

template <int RM, bool FULL>
__device__ __forceinline__ Item<RM> decode_item(const ItemRegs<RM, FULL>& g) {
  Item<RM> it;
  auto word = [&](int i) -> uint32_t {
    if constexpr (RM == 4 && !FULL) return i < 8 ? g.raw[i] : g.tail[i - 8];
    else if constexpr (RM == 4) return g.w[i];
    else return i < 16 ? g.lo[i] : g.hi[i - 16];
  };
  auto f64 = [&](int i) { return __hiloint2double(static_cast<int>(word(i + 1)), static_cast<int>(word(i))); };
#pragma unroll
  for (int r = 0; r < RM; ++r) it.raw[r] = f64(2 * r);
  const uint32_t s = word(2 * RM);
  it.used = s & 0xffu;
  it.fit = (s >> 8) & 0xffu;
  it.always = (s >> 16) & 0xffu;
  it.kind = s >> 24;
  if constexpr (FULL) {
    it.cpu_v = f64(2 * RM + 2);
    it.wsum = f64(2 * RM + 4);
    it.wrc = f64(2 * RM + 6);
  } else {
    it.cpu_v = it.wsum = it.wrc = 0.0;  // Score-only fields
  }
  return it;
}

// resourcesAvailableInAnyNUMANodes filter.go:93-163 with ids == positions
template <int RM>
__device__ __forceinline__ bool fits_fast(const FastNode<RM>& ns, const Item<RM>& it, uint32_t* pos) {
  const uint32_t need = it.fit | it.always;
  const bool ok = (need & ~ns.node_present) == 0;  // requested but not reported at node level -> cannot meet request
  uint32_t mask = 0xffu;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((need >> r) & 1u)) continue;  // uniform
    uint32_t rb;
    if ((it.always >> r) & 1u) {
      rb = ns.repmask(r);
    } else {
      rb = 0;
#pragma unroll
      for (int z = 0; z < kZ; ++z) rb |= ns.av[z][r] >= it.raw[r] ? (1u << z) : 0u;
    }
    mask &= rb | ns.fillmask(r);
  }
  *pos = mask ? static_cast<uint32_t>(__builtin_ctz(mask)) : 0u;
  return ok && mask != 0;
}

// subtractResourcesFromNUMANodeList numaresources.go:145-182 (sign -1) / its inverse (+1).  Unreported cells
// hold a negative value and stay negative, which is all any reader tests.
template <int RM>
__device__ __forceinline__ void adjust_fast(FastNode<RM>& ns, const Item<RM>& it, uint32_t pos, bool apply, double sign) {
  if (it.fit == 0) return;  // uniform
  double sel[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) sel[z] = (apply && pos == static_cast<uint32_t>(z)) ? sign : 0.0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((it.fit >> r) & 1u)) continue;
#pragma unroll
    for (int z = 0; z < kZ; ++z) ns.av[z][r] = __builtin_fma(sel[z], it.raw[r], ns.av[z][r]);
  }
}

// scoreForEachNUMANode score.go:110-124: the minimum of the non-zero zone scores, 0 when there is none (the
// reference's running rule `min == 0 || (s != 0 && s < min)` is order-independent).  Zones past the node's
// count hold no capacity and score 0 under Least/MostAllocated, so they drop out by themselves.
template <int RM, int SG>
__device__ __forceinline__ int score_each_fast(const FastNode<RM>& ns, const NrtArgs& a, const Item<RM>& it,
                                               const double* __restrict__ cpu_v, const double* __restrict__ braw) {
  const uint32_t used = it.used;
  uint32_t m = 0xffffffffu;  // min over zones of (score - 1) as unsigned: a zero score wraps to the maximum
  double value[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) value[r] = r == a.cpu_slot ? it.cpu_v : it.raw[r];
  if constexpr (SG == kSgBalanced) {
    // The reference's float64 divisions, correctly rounded, without the hardware's ~10-instruction division sequence: with
    // y = RN(1 / b) (per zone and resource from the engine's table — Balanced scores on the pristine zone table, so the
    // divisors are node constants; per container for the two uniform divisors), q0 = RN(a * y), r = a - b * q0 (exact in one
    // fma) and RN(q0 + r * y) is RN(a / b) (Markstein; the only exception, a divisor whose 53-bit significand is all ones,
    // cannot occur for integers below 2^42).  Replayed against exact rationals in tests/test_exactness_arguments.py.
    // A single requested resource (n == 1) divides by n - 1 == 0: that row keeps the real divisions and their NaN.
    const int n_used = __builtin_popcount(used);
    const double n = static_cast<double>(n_used);
    const bool multi = n_used >= 2;  // uniform
    const double yn = 1.0 / n, ym = 1.0 / (n - 1.0);
    auto div_rn = [](double x, double b, double y) {
      const double q0 = x * y;
      return __builtin_fma(__builtin_fma(-b, q0, x), y, q0);
    };
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      double fr[RM];
      bool over = false;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        fr[r] = 0.0;
        if (!((used >> r) & 1u)) continue;
        const double cap = ns.av[z][r];
        const double cap_v = r == a.cpu_slot ? cpu_v[z] : cap;
        const double f = cap > 0.0 ? div_rn(value[r], cap_v, ns.b[z][r]) : 1.0;  // fractionOfCapacity balanced_allocation.go:49-54
        over |= f > 1.0;
        fr[r] = f;
      }
      // gonum stat.Variance (corrected two-pass, unbiased), fractions in ascending resource id
      double sum = 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) sum += fr[r];
      const double mean = multi ? div_rn(sum, n, yn) : sum / n;
      double ss = 0.0, comp = 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const double d = ((used >> r) & 1u) ? fr[r] - mean : 0.0;
        ss += d * d;
        comp += d;
      }
      const double variance = multi ? div_rn(ss - div_rn(comp * comp, n, yn), n - 1.0, ym) : (ss - comp * comp / n) / (n - 1.0);
      const int s = (over || z >= ns.nz) ? 0 : static_cast<int>((1.0 - variance) * 100.0);
      const uint32_t s1 = static_cast<uint32_t>(s) - 1u;
      m = s1 < m ? s1 : m;
    }
  } else {
    if (__double_as_longlong(it.wsum) == 0) return 0;
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        if (!((used >> r) & 1u)) continue;
        double rs;
        if constexpr (SG == kSgLeast) {
          // (cap_v - req_v) * 100 / cap_v == 100 - req_v * (100 / cap_v); see the header for why the floor is exact.
          // Cells without capacity hold b = +inf: -v * inf is -inf (v > 0) or NaN (explicit zero request), and
          // max(., 0) turns both into the reference's 0
          rs = __builtin_fmax(__builtin_floor(__builtin_fma(-value[r], ns.b[z][r], 100.0 + 0x1p-43)), 0.0);
        } else {
          // req_v * 100 / cap_v, zero when the request exceeds the capacity.  "request <= capacity" is read off the same
          // kind of product instead of the mutable table (so MostAllocated, like LeastAllocated, scores from b alone):
          // t = (q * (1 + 2^-49)) * RN(100 / c) <= 100 * (1 + 2^-48)  <=>  q <= c   for integers q, c < 2^42
          // (q <= c gives t <= 100 * (1 + 1.2 * 2^-49); q >= c + 1 gives t >= 100 * (1 + 2^-42)).  The cpu slot compares the
          // raw millicore quantities (braw), the score uses whole cores (b).
          const double tp = (value[r] * (1.0 + 0x1p-49)) * ns.b[z][r];
          const double chk = r == a.cpu_slot ? (it.raw[r] * (1.0 + 0x1p-49)) * braw[z] : tp;
          rs = chk <= 100.0 * (1.0 + 0x1p-48) ? __builtin_floor(tp) : 0.0;
        }
        acc = __builtin_fma(rs, a.slot_weight_f[r], acc);
      }
      const uint32_t s1 = static_cast<uint32_t>(static_cast<int>(acc * it.wrc)) - 1u;  // floor(acc / wsum) in 0..100
      m = s1 < m ? s1 : m;
    }
  }
  return static_cast<int>(m + 1u);
}

// ---------------------------------------------------------------- LeastNUMANodes (least_numa.go:35-233)
__constant__ Combo8 kCombo8 = make_combo8();

// numaNodesRequired + findSuitableCombination: the smallest subset size for which some subset of zones holds the
// request, and among the fitting subsets of that size the one the reference returns: the first (lexicographic) whose
// average distance equals the node's minimum for the size (is_min), else the first with the smallest distance.
// The search runs as ONE wave-uniform loop over the (size, lexicographic) table of 8-position subsets — the generic
// kernel's per-lane loops made the wave execute the union of all lanes' iterations, each of them divergent; here a
// subset costs every lane the same few VALU operations and the loop ends as soon as every lane has its answer.
// Subsets with positions past a node's zone count fail the "every member reports every requested resource" test by
// themselves.
template <int RM>
__device__ __forceinline__ uint32_t numa_required_fast(const FastNode<RM>& ns, const NrtArgs& a, const Item<RM>& it, int64_t n, bool active,
                                                       bool* is_min) {
  uint32_t result = 0;
  bool done = !active;
  bool hit_min = false;
  const uint32_t used = it.used, need = it.fit | it.always;
  // Bounds that let the wave skip whole subset sizes.  Every valid subset lies inside V = the zones reporting all
  // requested resources, and sums grow with the subset: if even V cannot hold the request nothing can (answer 0 at
  // once); at least ceil(request / largest zone) zones are needed per resource; at most |V| can be used.
  uint32_t v_all = 0xffu;
#pragma unroll
  for (int r = 0; r < RM; ++r)
    if ((used >> r) & 1u) v_all &= ns.repmask(r);
  int k_lo = 1;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((need >> r) & 1u)) continue;
    double total = 0.0, largest = 0.0;
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      const double v = ((v_all >> z) & 1u) ? ns.av[z][r] : 0.0;
      total += v;
      largest = __builtin_fmax(largest, v);
    }
    if (total < it.raw[r]) done = true;  // no subset fits
    // zones needed if all were as large as the largest one (a float estimate rounded down is still a lower bound)
    const float need_f = static_cast<float>(it.raw[r]) / static_cast<float>(largest > 0.0 ? largest : 1.0);
    const int need_k = static_cast<int>(need_f * 0.999f);
    k_lo = need_k + 1 > k_lo ? (need_k + 1 > kZ ? kZ : need_k + 1) : k_lo;
  }
  const int k_hi = __builtin_popcount(v_all);
  const uint32_t allrep = v_all;  // zones reporting every requested resource
  for (int k = 1; k <= kZ; ++k) {
    // skip sizes no unfinished lane can use (ballots, not shuffles: part of the wave may be masked off here, and a
    // butterfly reduction through inactive lanes loses values)
    while (k <= kZ && __ballot(!done && k_lo <= k && k <= k_hi) == 0) ++k;
    if (k > kZ) break;
    const float min_avg = a.min_avg[static_cast<int64_t>(k - 1) * a.n_nodes + n];
    uint32_t best = 0;
    float min_distance = 256.0f;
    // Subsets of size k in lexicographic order = for every (k-1)-prefix in lexicographic order, every last element
    // above the prefix's highest one, ascending.  The prefix's sums cost R*8 multiply-adds (membership as uniform 0/1
    // weights) once; each extension by zone j is then one add and one compare per resource with j a compile-time index.
    int ci = kCombo8.start[k - 1];                                // table index of the next subset (for the distance)
    const int p0 = k == 1 ? -1 : kCombo8.start[k - 2], p1 = k == 1 ? 0 : kCombo8.start[k - 1];
    for (int pi = p0; pi < p1; ++pi) {
      const uint32_t pm = pi < 0 ? 0u : kCombo8.mask[pi];         // wave-uniform prefix (empty for k == 1)
      const int last = pm ? 31 - __builtin_clz(pm) : -1;
      if (last >= kZ - 1) continue;                                // nothing above its highest element
      bool pvalid = !done && (allrep & pm) == pm;                  // isValidCombineResources for the prefix
      double psum[RM];
      double w[kZ];
#pragma unroll
      for (int z = 0; z < kZ; ++z) w[z] = ((pm >> z) & 1u) ? 1.0 : 0.0;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        psum[r] = 0.0;
        if (!((need >> r) & 1u)) continue;
#pragma unroll
        for (int z = 0; z < kZ; ++z) psum[r] = __builtin_fma(w[z], ns.av[z][r], psum[r]);
      }
#pragma unroll
      for (int j = 0; j < kZ; ++j) {
        if (j <= last) continue;                                   // uniform
        bool ok = pvalid && ((allrep >> j) & 1u);
#pragma unroll
        for (int r = 0; r < RM; ++r)
          if ((need >> r) & 1u) ok &= psum[r] + ns.av[j][r] >= it.raw[r];  // combineResources + checkResourcesFit
        if (__ballot(ok) != 0) {
          const uint32_t m = pm | (1u << j);
          const float d = a.dist[static_cast<int64_t>(ci) * a.n_nodes + n];
          if (ok && d == min_avg) {
            result = m;
            hit_min = true;
            done = true;
            pvalid = false;
          } else if (ok && d < min_distance) {
            min_distance = d;
            best = m;
          }
        }
        ++ci;
      }
    }
    if (!done && best != 0) {
      result = best;
      done = true;
    }
    if (__ballot(!done) == 0) break;
  }
  *is_min = hit_min;
  return result;
}

// subtractFromNUMAs numaresources.go:184-215 with ids == positions: walk the chosen zones in order, taking from each
// what it has until the request is covered
template <int RM>
__device__ __forceinline__ void subtract_from_numas_fast(FastNode<RM>& ns, const Item<RM>& it, uint32_t m) {
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((it.used >> r) & 1u)) continue;
    double quantity = it.raw[r];
    const uint32_t members = m & ns.repmask(r);
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      const bool member = ((members >> z) & 1u) != 0;
      const double available = ns.av[z][r];
      const double take = member ? __builtin_fmin(quantity, available) : 0.0;
      ns.av[z][r] = available - take;
      quantity -= take;
    }
  }
}

// PH: 0 = Filter and Score in one launch; 1 = Filter only, 2 = Score only (LeastAllocated: its Score reads only b, the
// Filter only the mutable table, so each half keeps 64 instead of 128 state registers and runs at higher occupancy)
constexpr int kPhBoth = 0, kPhFilter = 1, kPhScore = 2;

template <int RM, int SG, int PH>
__global__ __launch_bounds__(256, RM == 4 ? (PH == kPhScore ? (SG == kSgBalanced ? 2 : (SG == kSgMost ? 4 : 5)) : (PH == kPhFilter ? 5 : (SG == kSgLeast ? 3 : 2))) : 1) void k_nrt_fast(NrtArgs a, int n_tiles) {
  SPX_RESOLVE_ROWS(a);
  constexpr bool FULL = PH != kPhFilter;  // only the Score reads the second half of a request item
  typedef ItemRegs<RM, FULL> Regs;
  // A block owns a window of 256 consecutive nodes and a chunk of pod rows.  Inside the window the engine has
  // ordered the nodes by (aligned, scope) — perm[] — so that a wavefront's 64 nodes mostly share one code path
  // (measured before: 49 % of the VALU lanes active, pod-scope and container-scope nodes being interleaved).
  // Results are staged in LDS at the nodes' original positions and leave as whole 256-byte row segments.
  __shared__ __align__(16) uint8_t stage[2][kPodsPerUnit][kWindow];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Blocks are dealt to the 8 XCDs round-robin (block b runs on XCD b % 8) and every XCD has its own L2.  Each XCD
  // therefore gets its own set of node windows — {x, x + 8, ...} — and walks the pod chunks over them: its share of the
  // node tables (about 130 KB per window) stays in its L2 instead of all windows cycling through all eight (at 20k nodes
  // the Filter launch used to fetch 10.7 GB to write 1.25 GB).  Measured: 20k nodes 17.9 -> 17.6 ms for the full profile; at 5k
  // nodes (20 windows, 4 of 24 slots idle) the same map costs 17 %, hence the threshold.
  const int n_windows = n_tiles;  // (the launch passes the window count)
  int window;
  int64_t chunk;
  if (n_windows >= kXcdMapWindows) {
    const int wpx = (n_windows + 7) >> 3;
    const int64_t seq = blockIdx.x >> 3;
    window = static_cast<int>(blockIdx.x & 7u) + 8 * static_cast<int>(seq % wpx);
    chunk = seq / wpx;
    if (window >= n_windows) return;  // block-uniform: the XCDs' window sets differ by at most one
  } else {  // the tables fit every XCD's L2 anyway (5k nodes: 2.6 MB); the plain order keeps all slots busy
    window = static_cast<int>(blockIdx.x % n_windows);
    chunk = blockIdx.x / n_windows;
  }
  const int64_t pod0 = a.row_begin + chunk * kPodsPerUnit;
  if (pod0 >= a.row_end) return;  // block-uniform
  const int64_t pod1 = pod0 + kPodsPerUnit < a.row_end ? pod0 + kPodsPerUnit : a.row_end;
  const int64_t base = static_cast<int64_t>(window) * kWindow;
  const int32_t pn = a.perm[base + threadIdx.x];
  const bool in = pn >= 0;
  const int64_t n = in ? pn : 0;
  const int pos = in ? static_cast<int>(n - base) : 0;
  const int R = a.n_res;
  if (a.out_raw == nullptr) {
    uint4* z = reinterpret_cast<uint4*>(&stage[0][0][0]) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < static_cast<int>(sizeof(stage) / 16 / 256); ++i) z[i * 256] = uint4{0, 0, 0, 0};
    __syncthreads();
  }

  FastNode<RM> ns;
  double cpu_v[kZ], braw[kZ];
  const uint32_t flags = in ? a.flags[n] : 0u;
  ns.nz = in ? a.n_zones[n] : 0;
  ns.node_present = in ? a.node_present[n] : 0u;
#pragma unroll
  for (int i = 0; i < RM / 4; ++i) ns.rep[i] = ns.fill[i] = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= R) continue;
    const uint32_t rep = in ? a.f_rep[static_cast<int64_t>(r) * a.n_nodes + n] : 0u;
    ns.rep[r >> 2] |= rep << (8 * (r & 3));
    if ((a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL) && rep == 0) ns.fill[r >> 2] |= 0xffu << (8 * (r & 3));
  }
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    cpu_v[z] = (SG == kSgBalanced && in && a.cpu_slot >= 0) ? a.f_cpu[static_cast<int64_t>(z) * a.n_nodes + n] : 0.0;
    braw[z] = (SG == kSgMost && in && a.cpu_slot >= 0) ? a.f_braw[static_cast<int64_t>(z) * a.n_nodes + n] : kNoCap;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int64_t i = (static_cast<int64_t>(z) * R + r) * a.n_nodes + n;
      ns.av[z][r] = (in && r < R) ? a.f_av[i] : -1.0;
      const double b = (SG != kSgBalanced && SG != kSgLeastNuma && in && r < R) ? a.f_rc[i] : kNoCap;
      ns.b[z][r] = (SG == kSgLeast && b == kNoCap) ? __builtin_inf() : b;
      if constexpr (SG == kSgBalanced) ns.b[z][r] = (in && r < R) ? a.f_rcv[i] : 1.0;  // RN(1 / Value(capacity)) for div_rn
    }
  }
  const int nns = 100 / (in ? a.max_numa[n] : 8);  // normalizeScore's per-zone step, least_numa.go:90-100
  const bool fresh = flags & SPX_NRT_F_FRESH;
  const bool has_nrt = flags & SPX_NRT_F_HAS_NRT;
  const bool single = flags & SPX_NRT_F_SINGLE_NUMA;
  const bool pod_scope = flags & SPX_NRT_F_POD_SCOPE;
  const bool aligned = fresh && has_nrt && single;  // the node's NUMA table decides Filter and Score

  const uint32_t* items = a.pod_items;
  auto header = [&](int64_t pod) { return uload(reinterpret_cast<const u32x2*>(items + pod * kItemsPerPod * item_words<RM>())); };
  u32x2 hw = header(pod0);
  for (int64_t pod = pod0; pod < pod1; ++pod) {
    // ---- wave-uniform pod record: this pod's request items now, the next pod's header for the next iteration
    const int64_t pi = pod * kItemsPerPod;  // index of the pod's first item
    const Regs pw = load_item<RM, FULL>(items, pi + 1);
    Regs cw = load_item<RM, FULL>(items, pi + 2);
    const u32x2 hnext = header(pod + 1 < pod1 ? pod + 1 : pod);
    const int qos = hw[0] & 0xffu;
    const bool non_native = ((hw[0] >> 8) & 0xffu) != 0;
    const int n_ctr = (hw[0] >> 16) & 0xffu;
    const int last_app = static_cast<int>(hw[0] >> 24) == 0xff ? -1 : static_cast<int>(hw[0] >> 24);
    const uint32_t inv_n = hw[1];  // ceil(2^16 / n_ctr)
    const bool non_g = qos != SPX_QOS_GUARANTEED;
    const bool filtered = !(qos == SPX_QOS_BESTEFFORT && !non_native);  // filter.go:186-190

    uint32_t status = (filtered && !fresh) ? SPX_NRT_ST_INVALID_TOPOLOGY : 0u;
    int score = non_g ? 100 : 0;
    const bool want_filter = PH != kPhScore && filtered && aligned;
    const bool want_score = SG != kSgLeastNuma && PH != kPhFilter && !non_g && aligned;

    if ((want_filter || want_score) && pod_scope) {  // singleNUMAPodLevelHandler / podScopeScore
      const Item<RM> it = decode_item<RM, FULL>(pw);
      if constexpr (PH != kPhScore) {
        if (want_filter) {
          uint32_t pos;
          if (!fits_fast(ns, it, &pos)) status = SPX_NRT_ST_POD;
        }
      }
      if constexpr (PH != kPhFilter) {
        if (want_score) score = score_each_fast<RM, SG>(ns, a, it, cpu_v, braw);
      }
    }
    if ((want_filter || want_score) && !pod_scope) {  // singleNUMAContainerLevelHandler / containerScopeScore
      // One pass in container order (init containers come first — checked at upload): an init container must fit
      // and is never subtracted; an app container is placed on the lowest fitting zone and subtracted from it.
      // Least/MostAllocated's zone scores read only b (never the mutable table), so they score in the same pass;
      // BalancedAllocation scores after the undo.
      uint32_t chosen = 0;  // per app container: the zone it was subtracted from + 1 (0 = not placed), 4 bits each, for the undo
      int sum = 0;
      for (int c = 0; c < n_ctr; ++c) {
        const Regs nw = load_item<RM, FULL>(items, pi + 2 + (c + 1 < kC ? c + 1 : c));  // prefetch the next container's item
        const Item<RM> it = decode_item<RM, FULL>(cw);
        if constexpr (PH != kPhScore) if (want_filter) {
          uint32_t pos;
          const bool ok = fits_fast(ns, it, &pos);
          const bool live = status == 0;
          if (it.kind != SPX_CTR_APP) {
            if (live && !ok) status = it.kind == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER;
          } else {
            if (live && !ok) status = SPX_NRT_ST_CONTAINER;
            if (c != last_app) {  // nothing reads the table after the last app container
              const bool apply = live && ok;
              adjust_fast(ns, it, pos, apply, -1.0);
              chosen |= (apply ? pos + 1u : 0u) << (4 * c);
            }
          }
        }
        if constexpr ((SG == kSgLeast || SG == kSgMost) && PH != kPhFilter) {
          if (want_score) sum += score_each_fast<RM, SG>(ns, a, it, cpu_v, braw);
        }
        cw = nw;
      }
      if constexpr (PH != kPhScore) if (want_filter && last_app > 0) {  // undo: Filter works on a private copy in the reference
        cw = load_item<RM, FULL>(items, pi + 2);
        for (int c = 0; c < last_app; ++c) {
          const Regs nw = load_item<RM, FULL>(items, pi + 2 + c + 1);
          const Item<RM> it = decode_item<RM, FULL>(cw);
          if (it.kind == SPX_CTR_APP) adjust_fast(ns, it, ((chosen >> (4 * c)) & 0xfu) - 1u, ((chosen >> (4 * c)) & 0xfu) != 0, 1.0);
          cw = nw;
        }
      }
      if constexpr (SG != kSgLeast && SG != kSgMost && PH != kPhFilter) {
        if (want_score) {
          cw = load_item<RM, FULL>(items, pi + 2);
          for (int c = 0; c < n_ctr; ++c) {
            const Regs nw = load_item<RM, FULL>(items, pi + 2 + (c + 1 < kC ? c + 1 : c));
            sum += score_each_fast<RM, SG>(ns, a, decode_item<RM, FULL>(cw), cpu_v, braw);
            cw = nw;
          }
        }
      }
      if (want_score) score = static_cast<int>((static_cast<uint32_t>(sum) * inv_n) >> 16);  // int64(mean): sum / n_ctr, sum <= 800
    }

    if constexpr (SG == kSgLeastNuma) {
      // LeastNUMANodes scores every node that has a fresh NRT, whatever its topology-manager policy (score.go:167-191)
      const bool want_ln = !non_g && fresh && has_nrt;
      if (want_ln && pod_scope) {  // leastNUMAPodScopeScore
        const Item<RM> it = decode_item<RM, FULL>(pw);
        uint32_t any_rep = 0;
#pragma unroll
        for (int r = 0; r < RM; ++r)
          if ((it.used >> r) & 1u) any_rep |= ns.repmask(r);
        const bool non_numa = any_rep == 0;  // onlyNonNUMAResources
        bool is_min;
        const uint32_t m = numa_required_fast(ns, a, it, n, !non_numa, &is_min);
        const int cnt = __builtin_popcount(m);
        score = non_numa ? 100 : (m ? 100 - cnt * nns + (is_min ? nns / 2 : 0) : 0);
      }
      if (want_ln && !pod_scope) {  // leastNUMAContainerScopeScore
        int max_count = 0;
        bool all_min = true, failed = false, dirty = false;
        cw = load_item<RM, FULL>(items, pi + 2);
        for (int c = 0; c < n_ctr; ++c) {
          const Regs nw = load_item<RM, FULL>(items, pi + 2 + (c + 1 < kC ? c + 1 : c));
          const Item<RM> it = decode_item<RM, FULL>(cw);
          uint32_t any_rep = 0;
#pragma unroll
          for (int r = 0; r < RM; ++r)
            if ((it.used >> r) & 1u) any_rep |= ns.repmask(r);
          const bool go = !failed && any_rep != 0;
          bool is_min;
          const uint32_t m = numa_required_fast(ns, a, it, n, go, &is_min);
          if (go) {
            if (m == 0) {
              failed = true;
            } else {
              all_min &= is_min;
              const int cnt = __builtin_popcount(m);
              max_count = cnt > max_count ? cnt : max_count;
              subtract_from_numas_fast(ns, it, m);
              dirty = true;
            }
          }
          cw = nw;
        }
        score = failed ? 0 : (max_count == 0 ? 100 : 100 - max_count * nns + (all_min ? nns / 2 : 0));
        if (__ballot(dirty) != 0) {  // the reference scored on a private NUMANodeList: restore this lane's table
#pragma unroll
          for (int z = 0; z < kZ; ++z)
#pragma unroll
            for (int r = 0; r < RM; ++r)
              ns.av[z][r] = (in && r < R) ? a.f_av[(static_cast<int64_t>(z) * R + r) * a.n_nodes + n] : -1.0;
        }
      }
    }

    if (in && a.out_raw != nullptr) {
      a.out_raw[n] = score;
    } else if (in) {
      if constexpr (SG == kSgLeastNuma) score = score < 0 ? 0 : score;  // 100 - count*(100/maxNUMA) can go negative; the table saturates
      if constexpr (PH != kPhScore) stage[0][pod - pod0][pos] = static_cast<uint8_t>(status);
      if constexpr (PH != kPhFilter) stage[1][pod - pod0][pos] = static_cast<uint8_t>(score > 255 ? 255 : score);
    }
    hw = hnext;
  }
  if (a.out_raw != nullptr) return;
  __syncthreads();
  const int rows = static_cast<int>(pod1 - pod0);
  const int64_t col = base + lane * 4;
  if (col < a.row_stride) {
    for (int i = wave; i < 2 * rows; i += 4) {
      const int p = i >> 1, tbl = i & 1;
      if ((PH == kPhFilter && tbl == 1) || (PH == kPhScore && tbl == 0)) continue;
      uint8_t* out = (tbl ? a.out_score : a.out_status) + (pod0 + p) * a.row_stride + col;
      *reinterpret_cast<uint32_t*>(out) = *reinterpret_cast<const uint32_t*>(&stage[tbl][p][lane * 4]);
    }
  }
}

}  // namespace

bool launch_nrt_fast(const NrtArgs& a, hipStream_t s) {
  if (!a.fast) return false;
  const int n_tiles = static_cast<int>((a.n_nodes + kWindow - 1) / kWindow);  // windows of 256 nodes
  const int64_t chunks = (a.row_end - a.row_begin + kPodsPerUnit - 1) / kPodsPerUnit;
  const unsigned blocks = static_cast<unsigned>(chunks * (n_tiles >= kXcdMapWindows ? ((n_tiles + 7) / 8) * 8 : n_tiles));  // see the kernel's block map
  const int sg = a.strategy == SPX_NRT_LEAST_NUMA_NODES ? kSgLeastNuma
               : a.strategy == SPX_NRT_BALANCED_ALLOCATION ? kSgBalanced : (a.strategy == SPX_NRT_LEAST_ALLOCATED ? kSgLeast : kSgMost);
  const bool split = sg != kSgLeastNuma && a.out_raw == nullptr && !(a.opts & kOptNrtSingleLaunch);
#define SPX_NRTF_CASE(RMV, SGV)                                                                           \
  if ((a.n_res <= 4) == (RMV == 4) && sg == SGV) {                                                        \
    if (SGV != kSgLeastNuma && split) { /* the Filter half does not depend on the strategy */ \
      hipLaunchKernelGGL((k_nrt_fast<RMV, kSgLeast, kPhFilter>), dim3(blocks), dim3(256), 0, s, a, n_tiles); \
      hipLaunchKernelGGL((k_nrt_fast<RMV, (SGV == kSgLeastNuma ? kSgLeast : SGV), kPhScore>), dim3(blocks), dim3(256), 0, s, a, n_tiles);  \
    } else {                                                                                              \
      hipLaunchKernelGGL((k_nrt_fast<RMV, SGV, kPhBoth>), dim3(blocks), dim3(256), 0, s, a, n_tiles);     \
    }                                                                                                     \
    return true;                                                                                          \
  }
  SPX_NRTF_CASE(4, kSgLeast)
  SPX_NRTF_CASE(4, kSgMost)
  SPX_NRTF_CASE(4, kSgBalanced)
  SPX_NRTF_CASE(4, kSgLeastNuma)
  SPX_NRTF_CASE(8, kSgLeast)
  SPX_NRTF_CASE(8, kSgMost)
  SPX_NRTF_CASE(8, kSgBalanced)
  SPX_NRTF_CASE(8, kSgLeastNuma)
#undef SPX_NRTF_CASE
  return false;
}

}  // namespace spx
