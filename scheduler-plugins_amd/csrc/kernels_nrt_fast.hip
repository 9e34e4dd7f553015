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
#include <utility>

#include "spx_internal.h"
#include "nrt_fast_device.h"

namespace spx {

namespace {

using namespace nrtdev;

// ---------------------------------------------------------------- BalancedAllocation in float32 (Score launch)
//
// The float64 form above costs ~44 float64 instructions per zone and container and holds 144 VGPRs of node tables; float32
// instructions issue twice as fast on this part (tools/micro/valu_rate.hip).  The zone score is (1 - variance) * 100 truncated,
// and scoreForEachNUMANode takes the minimum of the zones' truncated scores — truncation is monotone, so that minimum is
// trunc(min over zones of the untruncated value): ONE truncation per container, and a float32 evaluation decides it
// whenever the float32 minimum is farther from an integer than its error bound.  Otherwise — and whenever float32 cannot
// tell "request > capacity" — the cell is marked kBalRedo in the score table; k_nrt_bal_scan collects the marked
// cells and k_nrt_bal_redo recomputes them with the float64 form (whole cell, all containers).
//   fraction   f = RN32(RN32(request) * RN32(RN64(1 / capacity)) + one)      |f - request/capacity| <= 1.9e-7 * f
//   variance   (sum f^2 - (sum f)^2 / n) / (n - 1)  — algebraically the reference's corrected two-pass value — with
//              f in [0, 1], n in 2..8:   |s32 - s| < 2.3e-4  (replayed in tests/test_exactness_arguments.py)
// A cell with capacity <= 0 contributes the reference's f = 1.0 through `one` (rcp = 0), and does not count for "over".
// Valid zone scores are >= 50 (unbiased variance of n values in [0,1] is at most 1/2), so 0 only ever means "no valid zone".
constexpr float kBalBand = 3e-4f;
constexpr float kBalTol = 2.4e-7f;  // 2^-22: two float32 roundings of nearly equal quantities
constexpr float kBalNoCap = 1e38f;
constexpr int kBalRedo = 255;  // score byte of a cell the fix-up launch recomputes (scores are <= 100)

// "request > capacity" (fractionOfCapacity > 1: the zone scores 0) is decided on the quantities themselves, not on the
// rounded quotient — a request that equals the capacity (one device wanted, one device free) is the common case, and its
// float32 quotient is 1.0 either way.  For integers, RN64(request / capacity) > 1.0 exactly when request > capacity; slots whose
// requests and capacities are all float32 values cluster-wide (NrtArgs.exact32_slots: whole cores, devices, GiB hugepages) compare
// exactly in float32, the others (bytes) are undecided when the two lie within 2^-22 of each other.
template <int RM>
struct BalNode {
  float rcp[kZ][RM];   // RN32(1 / Value(capacity)), 0 where the capacity is not positive
  float one[kZ][RM];   // 1 where the capacity is not positive, else 0
  float capf[kZ][RM];  // RN32(Value(capacity)), kBalNoCap where the capacity is not positive
};

template <int RM>
__device__ __forceinline__ int score_balanced_f32(const BalNode<RM>& bn, int nz, int cpu_slot, uint32_t exact32, const Item<RM>& it, bool* redo) {
  const uint32_t used = it.used;
  const int n_used = __builtin_popcount(used);
  if (n_used < 2) {
    // uniform.  One requested resource: the deviation from the mean is exactly 0, the variance 0 / (n - 1) = 0 / 0 is NaN, and the
    // float64 form's int(NaN) is 0 for every zone (v_cvt_i32_f64) — "no zone scores", 0, whatever the quantities are
    *redo = false;
    return 0;
  }
  const float rn = 1.0f / static_cast<float>(n_used), rm = 1.0f / static_cast<float>(n_used - 1);
  float value[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) value[r] = static_cast<float>(r == cpu_slot ? it.cpu_v : it.raw[r]);
  // resource outside, zone inside (the requested-slot test is a scalar branch; see score_each_fast): per-zone accumulators
  float sum[kZ], sq[kZ], mxd[kZ], nr[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) sum[z] = 0.0f, sq[z] = 0.0f, mxd[z] = -kBalNoCap, nr[z] = -1.0f;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((used >> r) & 1u)) continue;  // uniform
    SPX_KEEP_BRANCH();
    const bool inexact = !((exact32 >> r) & 1u);  // uniform
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      const float f = __builtin_fmaf(value[r], bn.rcp[z][r], bn.one[z][r]);
      const float d = value[r] - bn.capf[z][r];
      mxd[z] = __builtin_fmaxf(mxd[z], d);
      if (inexact) nr[z] = __builtin_fmaxf(nr[z], __builtin_fmaf(bn.capf[z][r], kBalTol, -__builtin_fabsf(d)));
      sum[z] += f;
      sq[z] = __builtin_fmaf(f, f, sq[z]);
    }
  }
  float best = __builtin_inff();
  bool undecided = false;
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    const float var = __builtin_fmaf(-(sum[z] * sum[z]), rn, sq[z]) * rm;
    const float sc = __builtin_fmaf(-var, 100.0f, 100.0f);
    const bool exists = z < nz;
    undecided |= exists && nr[z] >= 0.0f;
    best = (exists && !(mxd[z] > 0.0f)) ? __builtin_fminf(best, sc) : best;
  }
  const bool has = best < __builtin_inff();
  const float fl = __builtin_floorf(best), frac = best - fl;
  *redo = undecided || (has && (frac < kBalBand || frac > 1.0f - kBalBand));
  return has ? static_cast<int>(fl) : 0;
}

// ---------------------------------------------------------------- LeastNUMANodes (least_numa.go:35-233)
//
// numaNodesRequired + findSuitableCombination: the smallest subset size for which some subset of zones holds the request,
// and among the fitting subsets of that size the one the reference returns — the first in lexicographic order whose
// average distance equals the node's minimum for the size (is_min), else the first with the smallest distance.
//
// Every lane (node) evaluates ALL 255 subsets, branch-free, into a bit set (layout: LnLayout, spx_internal.h): the sums
// of a subset split into the part over zones 0..3 and the part over zones 4..7, so per requested resource 16 + 16 partial
// sums are built once and a subset costs one v_cmp_ge_f64 (lo[S & 15] >= request - hi[S >> 4], exact: all quantities are
// integers below 2^53) and one v_addc that shifts the verdict into the set.  The earlier form walked the subsets in a
// wave-uniform loop that stopped when all 64 lanes had their answer: some lane nearly always needs size 4 or 5, every
// iteration carried ballots, scalar table reads and a dependent distance load, and config #3 took 141 ms
// (3.3e10 scalar + 1.9e10 vector instructions per launch, 54 % of wave cycles waiting).  Here the selection needs no
// distance at all in the common case (the node's minimum-distance subsets are a register-resident bit set), and a
// 7-step bit-sliced minimum over per-node rank planes otherwise.
constexpr LnLayout kLn = make_ln_layout();

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int D, int Q>
__device__ __forceinline__ void ln_step(uint32_t& f, const double (&lo)[16], const double (&thr)[16]) {
  constexpr int S = kLn.subset[D][Q];
  asm("v_cmp_ge_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(f) : "v"(lo[S & 15]), "v"(thr[S >> 4]) : "vcc");
}
// four subsets per asm statement (the compiler pads every statement with an s_nop)
template <int D, int Q>
__device__ __forceinline__ void ln_step4(uint32_t& f, const double (&lo)[16], const double (&thr)[16]) {
  constexpr int S0 = kLn.subset[D][Q], S1 = kLn.subset[D][Q - 1], S2 = kLn.subset[D][Q - 2], S3 = kLn.subset[D][Q - 3];
  asm("v_cmp_ge_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_ge_f64 vcc, %3, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_ge_f64 vcc, %5, %6\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
      "v_cmp_ge_f64 vcc, %7, %8\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
      : "+v"(f)
      : "v"(lo[S0 & 15]), "v"(thr[S0 >> 4]), "v"(lo[S1 & 15]), "v"(thr[S1 >> 4]), "v"(lo[S2 & 15]), "v"(thr[S2 >> 4]), "v"(lo[S3 & 15]),
        "v"(thr[S3 >> 4])
      : "vcc");
}
template <int D>
__device__ __forceinline__ uint32_t ln_dword(const double (&lo)[16], const double (&thr)[16]) {
  uint32_t f = 0;
  constexpr int cnt = kLn.cnt[D];
  // highest used bit first: each step shifts the set left by one
  static_for<cnt / 4>([&](auto g) { ln_step4<D, cnt - 1 - 4 * decltype(g)::value>(f, lo, thr); });
  static_for<cnt % 4>([&](auto g) { ln_step<D, cnt % 4 - 1 - decltype(g)::value>(f, lo, thr); });
  return f;
}

// fall &= { S : sum over S of v[z] >= want }, for the dwords [D0, D1) of the layout.  The partial-sum tables are indexed by
// compile-time constants: a call that only covers sizes 1 and 2 (dwords 0 and 1) builds just the entries those subsets touch.
constexpr int kLnSmall = 2;  // dwords holding the subsets of sizes 1 and 2 (8 + 28)
static_assert(make_ln_layout().first[3] == kLnSmall, "sizes 1 and 2 occupy the first two dwords");
template <int D0, int D1>
__device__ __forceinline__ void ln_resource(uint32_t (&fall)[kLnDwords], const double (&v)[kZ], double want) {
  double lo[16], thr[16];
  lo[0] = 0.0;
  thr[0] = 0.0;
#pragma unroll
  for (int m = 1; m < 16; ++m) {
    lo[m] = lo[m & (m - 1)] + v[__builtin_ctz(m)];
    thr[m] = thr[m & (m - 1)] + v[4 + __builtin_ctz(m)];
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) thr[m] = want - thr[m];
  static_for<D1 - D0>([&](auto d) { fall[D0 + decltype(d)::value] &= ln_dword<D0 + decltype(d)::value>(lo, thr); });
}

// `choice`: the caller needs the reference's subset itself (a later container is charged against it); otherwise only
// its size and is_min are read and the distance ranks are not consulted.
//
// UNIFORM (the sweep): every lane holds the same item — the requested-resource sets are scalar, a resource's column is picked by a
// scalar branch.  !UNIFORM (k_nrt_ln_redo: a lane = one (pod, node) cell): sets and quantities are per lane; all RM resources
// are walked, a resource the lane does not compare gets the request -inf (every subset holds it), and both passes always run.
// DEFER (the sweep's batch Score launch): lanes the first pass leaves open are reported in *deferred instead of being worked
// through — 7 % of the (container, node) pairs, but spread so evenly that 87 % of the wave-level searches used to run the 219
// larger subsets for them; the launch lists those cells and k_nrt_ln_redo searches them with every lane of a wave at work.
template <int RM, bool UNIFORM = true, bool DEFER = false>
__device__ __forceinline__ uint32_t numa_required_fast(const FastNode<RM>& ns, const NrtArgs& a, const Item<RM>& it, int64_t n, bool active,
                                                       bool choice, const uint32_t (&mmin)[kLnDwords], const uint8_t* subset_lds,
                                                       const uint32_t* allow_lds, const double (&tot_all)[RM], bool* is_min,
                                                       bool* deferred = nullptr) {
  *is_min = false;
  if constexpr (DEFER) *deferred = false;
  if (__ballot(active) == 0) return 0;
  const uint32_t used = it.used, need = it.fit | it.always;
  // a valid subset lies inside V = the zones reporting every requested resource (isValidCombineResources): the bit set starts
  // as { S : S inside V } — a per-block LDS table indexed by V — and the sums use the table as it is (an unreported cell holds -1;
  // the subsets it would spoil are not in the set)
  uint32_t v_all = active ? 0xffu : 0u;
#pragma unroll
  for (int r = 0; r < RM; ++r)
    if ((used >> r) & 1u) v_all &= ns.repmask(r);
  const uint32_t* allow = allow_lds + v_all * kLnDwords;
  uint32_t fall[kLnDwords];
#pragma unroll
  for (int d = 0; d < kLnSmall; ++d) fall[d] = allow[d];
  // One pass per compared resource; a request of zero quantities compares nothing, and every subset of V "fits": that is the
  // pass over the pseudo resource RM (no quantity anywhere, nothing wanted).  The resource index is wave-uniform: its column is
  // picked by a scalar branch (as selects the eight doubles cost 64 v_cndmask per resource, more than the first pass's compares).
  //
  // Round 3: two passes.  71 % of config #3's (container, node) pairs fit one zone, 11 % two, 9 % no subset at all (the zones'
  // total is short) — so the first pass evaluates the 36 subsets of sizes 1 and 2, and only when some lane of the wave is
  // still open (the node's total suffices but no single zone or pair does) does the wave run the other 219 subsets.  The engine
  // orders the nodes of a window by how tight their largest zones are, so that such lanes share waves.
  auto column = [&](int r, double (&v)[kZ], double* want, double* total) {
    double w = 0.0, tot = 0.0;
#pragma unroll
    for (int z = 0; z < kZ; ++z) v[z] = 0.0;
#pragma unroll
    for (int q = 0; q < RM; ++q)
      if (r == q) {  // uniform
        SPX_KEEP_BRANCH();
        w = it.raw[q];
        tot = tot_all[q];
#pragma unroll
        for (int z = 0; z < kZ; ++z) v[z] = ns.av[z][q];
      }
    *want = w;
    *total = tot;
  };
  if constexpr (UNIFORM) {
  const uint32_t todo0 = need ? need : (1u << RM);
  // tot_all: what the node's zones held at the start of the pod (an upper bound once earlier containers were charged): a lane
  // it closes can hold no subset; a lane it fails to close merely takes the second pass
  bool feasible = active;
  for (uint32_t todo = todo0; todo;) {
    const int r = __builtin_ctz(todo);
    todo &= todo - 1;
    double v[kZ], want, total;
    column(r, v, &want, &total);
    feasible &= total >= want;
    ln_resource<0, kLnSmall>(fall, v, want);
  }
  const bool open = feasible && (fall[0] | fall[1]) == 0;
  if constexpr (DEFER) {
    *deferred = open;
#pragma unroll
    for (int d = kLnSmall; d < kLnDwords; ++d) fall[d] = 0u;  // open lanes: their cells are searched by k_nrt_ln_redo
  } else if (__ballot(open) != 0) {
#pragma unroll
    for (int d = kLnSmall; d < kLnDwords; ++d) fall[d] = allow[d];
    for (uint32_t todo = todo0; todo;) {
      const int r = __builtin_ctz(todo);
      todo &= todo - 1;
      double v[kZ], want, total;
      column(r, v, &want, &total);
      ln_resource<kLnSmall, kLnDwords>(fall, v, want);
    }
  } else {
#pragma unroll
    for (int d = kLnSmall; d < kLnDwords; ++d) fall[d] = 0u;  // nobody needs a larger subset: closed lanes hold sizes 1-2, the others nothing
  }
  } else {
    (void)column;
#pragma unroll
    for (int d = kLnSmall; d < kLnDwords; ++d) fall[d] = allow[d];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      if (__ballot(active && ((need >> r) & 1u)) == 0) continue;  // no lane of the wave compares this resource
      double v[kZ];
#pragma unroll
      for (int z = 0; z < kZ; ++z) v[z] = ns.av[z][r];
      const double want = ((need >> r) & 1u) ? it.raw[r] : -__builtin_inf();
      ln_resource<0, kLnDwords>(fall, v, want);
    }
  }
  // the smallest size with a fitting subset: its candidates c, those of them at the node's minimum distance h.  DEFER: only
  // sizes 1 and 2 can have been found (one dword each, rank planes of at most 5 bits)
  constexpr int KMAX = DEFER ? 2 : 8, JMAX = DEFER ? 1 : 3, BMAX = DEFER ? 5 : 7;
  static_assert(!DEFER || (kLn.nd[1] == 1 && kLn.nd[2] == 1 && kLn.bits[1] <= 5 && kLn.bits[2] <= 5), "sizes 1-2: one dword, <= 5 rank bits");
  uint32_t c[3] = {0, 0, 0}, h[3] = {0, 0, 0};
  int ksel = 0, fsel = 0;
  static_for<KMAX>([&](auto ki) {
    constexpr int k = KMAX - decltype(ki)::value, f = kLn.first[k], nd = kLn.nd[k];
    uint32_t t = fall[f];
    if constexpr (nd > 1) t |= fall[f + 1];
    if constexpr (nd > 2) t |= fall[f + 2];
    const bool sel = t != 0;
    static_for<JMAX>([&](auto ji) {
      constexpr int j = decltype(ji)::value;
      uint32_t cj = 0u, hj = 0u;
      if constexpr (j < nd) cj = fall[f + j], hj = fall[f + j] & mmin[f + j];
      c[j] = sel ? cj : c[j];
      h[j] = sel ? hj : h[j];
    });
    ksel = sel ? k : ksel;
    fsel = sel ? f : fsel;
  });
  const bool found = ksel != 0;
  const bool hit = (h[0] | h[1] | h[2]) != 0;
  if (hit) {
#pragma unroll
    for (int j = 0; j < JMAX; ++j) c[j] = h[j];
  } else if (choice && __ballot(found) != 0) {
    // no fitting subset at the minimum distance: the fitting subset with the smallest distance = the smallest rank;
    // bit-sliced minimum from the top plane down (a plane keeps the candidates whose rank has that bit clear, if any)
    if (found) {
      int bits = 0, nd = 1, prow = 0;
      static_for<KMAX>([&](auto ki) {
        constexpr int k = 1 + decltype(ki)::value, bk = kLn.bits[k], ndk = kLn.nd[k], pk = kLnDwords + kLn.pbase[k];
        const bool is = ksel == k;
        bits = is ? bk : bits;
        nd = is ? ndk : nd;
        prow = is ? pk : prow;
      });
      const uint32_t nn = static_cast<uint32_t>(a.n_nodes), n32 = opaque_lane(static_cast<uint32_t>(n));
      uint32_t pl[BMAX][3];
#pragma unroll
      for (int b = 0; b < BMAX; ++b)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          pl[b][j] = (j < JMAX && b < bits && j < nd) ? ld_off(a.ln_tab, (static_cast<uint32_t>(prow + b * nd + j) * nn + n32) * 4u) : 0u;
#pragma unroll
      for (int b = BMAX - 1; b >= 0; --b) {
        const uint32_t t0 = c[0] & ~pl[b][0], t1 = c[1] & ~pl[b][1], t2 = c[2] & ~pl[b][2];
        const bool any = (t0 | t1 | t2) != 0;
        c[0] = any ? t0 : c[0];
        c[1] = any ? t1 : c[1];
        c[2] = any ? t2 : c[2];
      }
    }
  }
  const int p = c[0] ? __builtin_ctz(c[0]) : (c[1] ? 32 + __builtin_ctz(c[1]) : 64 + __builtin_ctz(c[2] | 0x80000000u));
  *is_min = found && hit;
  return found ? subset_lds[fsel * 32 + p] : 0u;
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

// The LeastNUMANodes score of one pod on the wave's nodes (score.go:167-191): every node that has a fresh NRT is scored,
// whatever its topology-manager policy.  One loop serves both scopes — step -1 is the pod-level request for the pod-scope
// nodes (leastNUMAPodScopeScore), steps 0.. the containers for the others (leastNUMAContainerScopeScore) — so that the subset
// search exists once in the code; a step none of the wave's nodes takes part in is skipped.  `pit`: the pod's record (LDS),
// the same for every lane.  DEFER: a lane whose search needs more than sizes 1-2 stops and reports *listed (see
// numa_required_fast); its score is then meaningless.
template <int RM, bool DEFER, bool RESTORE = true>
__device__ __forceinline__ int ln_score_wave(FastNode<RM>& ns, const NrtArgs& a, const uint32_t* pit, int n_ctr, int64_t n, bool in, bool want_ln,
                                             bool pod_scope, int nns, const uint32_t (&mmin)[kLnDwords], const uint8_t* ln_subset,
                                             const uint32_t* ln_allow, const double (&ln_tot)[RM], int score, bool* listed) {
  const int R = a.n_res;
  int max_count = 0;
  bool all_min = true, failed = false, dirty = false, listed_cell = false;
  for (int c = -1; c < n_ctr; ++c) {
    const bool mine = want_ln && (c < 0 ? pod_scope : !pod_scope);
    if (__ballot(mine) == 0) continue;
    const Item<RM> it = decode_item<RM, true>(load_item<RM, true>(pit, c < 0 ? 1 : 2 + c));
    uint32_t any_rep = 0;
#pragma unroll
    for (int r = 0; r < RM; ++r)
      if ((it.used >> r) & 1u) any_rep |= ns.repmask(r);
    // any_rep == 0: onlyNonNUMAResources, the item is passed over.  A cell already listed for k_nrt_ln_redo stops here.
    const bool go = mine && !failed && !listed_cell && any_rep != 0;
    bool is_min, deferred = false;
    const uint32_t m = numa_required_fast<RM, true, DEFER>(ns, a, it, n, go, c >= 0 && c + 1 < n_ctr, mmin, ln_subset, ln_allow, ln_tot, &is_min, &deferred);
    listed_cell |= go && deferred;
    if (go && !deferred) {
      if (m == 0) {
        failed = true;
      } else {
        all_min &= is_min;
        const int cnt = __builtin_popcount(m);
        max_count = cnt > max_count ? cnt : max_count;
      }
    }
    if (c >= 0 && c + 1 < n_ctr && __ballot(go && m != 0) != 0) {  // the next container sees what this one took
      subtract_from_numas_fast(ns, it, go ? m : 0u);
      dirty |= go && m != 0;
    }
  }
  if (want_ln) score = failed ? 0 : (max_count == 0 ? 100 : 100 - max_count * nns + (all_min ? nns / 2 : 0));
  if (RESTORE && __ballot(dirty) != 0) {  // the reference scored on a private NUMANodeList: restore this lane's table
    const uint32_t n32 = opaque_lane(static_cast<uint32_t>(n));
#pragma unroll
    for (int z = 0; z < kZ; ++z)
#pragma unroll
      for (int r = 0; r < RM; ++r)
        ns.av[z][r] = (in && r < R) ? ld_off(a.f_av, (static_cast<uint32_t>(z * R + r) * static_cast<uint32_t>(a.n_nodes) + n32) * 8u) : -1.0;
  }
  *listed = listed_cell;
  return score;
}

// PH: 0 = Filter and Score in one launch; 1 = Filter only, 2 = Score only (LeastAllocated: its Score reads only b, the
// Filter only the mutable table, so each half keeps 64 instead of 128 state registers and runs at higher occupancy)
constexpr int kPhBoth = 0, kPhFilter = 1, kPhScore = 2;

// waves per SIMD the register allocation is bounded for (<= 4 resource slots).  A spilled VGPR is not cheap here: whatever the
// allocator parks in scratch is reloaded inside the pod loop behind an s_waitcnt vmcnt(0) (round 2's Filter launch, bounded to 5
// waves = 96 VGPRs, reloaded the lane's staging offset from scratch for EVERY pod: ~1300 cycles per pod and wave)
// (measured, tools/r3 + tools/variant.py builds: Filter 4, LeastAllocated Score 4, MostAllocated / BalancedAllocation Score 3 — more spills —
// LeastNUMANodes 2, one-launch form 3)
template <int RM, int SG, int PH>
constexpr int nrt_waves() {
  if (RM != 4) return 1;
  if (PH == kPhFilter) return 4;
  if (PH == kPhScore) return SG == kSgLeastNuma ? 2 : (SG == kSgMost || SG == kSgBalanced ? 3 : 4);
  return SG == kSgLeast ? 3 : 2;
}

// LNM (LeastNUMANodes, batch Score launch): kLnDefer = cells whose subset search needs more than sizes 1-2 are listed — per pod
// row, NrtArgs::redo_list — for k_nrt_ln_redo instead of searched here; kLnIfOverflow = the complete search, but the launch only
// acts when some row's list overflowed (every block leaves at once otherwise) — it then simply rewrites the whole table
constexpr int kLnFull = 0, kLnDefer = 1, kLnIfOverflow = 2;
constexpr int kLnRedo = 255;  // score byte of a listed cell (scores are <= 100)

template <int RM, int SG, int PH, int LNM = kLnFull, bool PK = false>
__global__ __launch_bounds__(256, (PK && RM == 4 ? 5 : nrt_waves<RM, SG, PH>())) void k_nrt_fast(NrtArgs a, int n_tiles) {
  static_assert(!PK || (PH == kPhScore && SG == kSgLeast), "the packed float32 zone totals belong to LeastAllocated's Score-only launch");
  SPX_RESOLVE_ROWS(a);
  if constexpr (LNM == kLnIfOverflow) {
    if (a.redo_list[0] == 0u) return;  // block-uniform: no row's list overflowed
  }
  constexpr bool FULL = PH != kPhFilter;  // only the Score reads the second half of a request item
  // A block owns a window of 256 consecutive nodes and a chunk of pod rows.  Inside the window the engine has
  // ordered the nodes by (aligned, scope) — perm[] — so that a wavefront's 64 nodes mostly share one code path
  // (measured before: 49 % of the VALU lanes active, pod-scope and container-scope nodes being interleaved).
  // Results are staged in LDS at the nodes' original positions and leave as whole 256-byte row segments.
  // (round 3) staged as one dword per node and group of four pod rows — a lane keeps its last four results in a register and
  // stores once per group: one LDS round trip per four pods instead of per pod — and regrouped into row segments on the way out
  constexpr int kScoreTab = PH == kPhBoth ? 1 : 0;  // a split launch stages one table
  __shared__ __align__(16) uint32_t stage[kScoreTab + 1][kPodsPerUnit / 4][kWindow];
  __shared__ __align__(16) uint32_t pod_lds[kPodsPerUnit * pod_words<RM>()];  // the chunk's pod records (20 KB for <= 4 slots)
  __shared__ __align__(4) uint8_t ln_subset[SG == kSgLeastNuma ? kLnDwords * 32 : 4];  // LeastNUMANodes: bit position -> zone mask
  __shared__ uint32_t ln_allow[SG == kSgLeastNuma ? 256 * kLnDwords : 1];  // ... zone set V -> the subsets inside V, in the bit layout
  // BalancedAllocation, float32 Score launch: the cells it leaves to the float64 form, collected per block and appended to the
  // global list with ONE atomic (round 2 found them again by scanning the 0.25 GB score table: 0.5 ms)
  constexpr bool kLnDeferred = SG == kSgLeastNuma && LNM == kLnDefer;
  constexpr int kRedoBuf = (SG == kSgBalanced && PH == kPhScore) ? 1024 : 1;
  __shared__ uint32_t redo_buf[kRedoBuf][2];
  __shared__ uint32_t redo_n, redo_base;
  __shared__ uint32_t pk_flagged;  // packed LeastAllocated Score: the chunk's pods with a request k_nrt_pk_tab_build lists for this window
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
  // the chunk's rows: 32 consecutive rows of the range, or 32 consecutive entries of the row list (pod equivalence classes)
  const bool listed = a.row_list != nullptr;
  const int64_t first = listed ? chunk * kPodsPerUnit : a.row_begin + chunk * kPodsPerUnit, last = listed ? a.n_list : a.row_end;
  if (first >= last) return;  // block-uniform
  const int rows = static_cast<int>(last - first < kPodsPerUnit ? last - first : kPodsPerUnit);
  auto row_of = [&](int p) -> int64_t { return listed ? static_cast<int64_t>(uload(a.row_list + first + p)) : first + p; };
  const int64_t base = static_cast<int64_t>(window) * kWindow;
  const int32_t pn = a.perm[base + threadIdx.x];
  const bool in = pn >= 0;
  const int64_t n = in ? pn : 0;
  const int pos = in ? static_cast<int>(n - base) : 0;
  const int R = a.n_res;

  FastNode<RM> ns;
  double cpu_v[kZ], braw[kZ];
  const uint32_t flags = in ? a.flags[n] : 0u;
  load_fast_node<RM, SG, PK>(a, n, in, ns, cpu_v, braw);
  // BalancedAllocation's Score launch works from float32 images of the reciprocals; the float64 tables die here
  constexpr bool kBalF32 = SG == kSgBalanced && PH == kPhScore;
  BalNode<RM> bn;
  if constexpr (kBalF32) {
#pragma unroll
    for (int z = 0; z < kZ; ++z)
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const bool cap = ns.av[z][r] > 0.0;
        bn.rcp[z][r] = cap ? static_cast<float>(ns.b[z][r]) : 0.0f;
        bn.one[z][r] = cap ? 0.0f : 1.0f;
        bn.capf[z][r] = cap ? static_cast<float>(r == a.cpu_slot ? cpu_v[z] : ns.av[z][r]) : kBalNoCap;
      }
  }
  const int nns = 100 / (in ? a.max_numa[n] : 8);  // normalizeScore's per-zone step, least_numa.go:90-100
  uint32_t mmin[kLnDwords];  // LeastNUMANodes: the node's minimum-distance subsets per size (LnLayout)
#pragma unroll
  for (int d = 0; d < kLnDwords; ++d) mmin[d] = (SG == kSgLeastNuma && in) ? a.ln_tab[static_cast<int64_t>(d) * a.n_nodes + n] : 0u;
  if constexpr (SG == kSgLeastNuma) {  // the block-constant tables, prepared by the engine: coalesced copies
    for (int i = threadIdx.x; i < 256 * kLnDwords; i += blockDim.x) ln_allow[i] = a.ln_const[i];
    for (int i = threadIdx.x; i < kLnDwords * 32 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(ln_subset)[i] = a.ln_const[256 * kLnDwords + i];
    __syncthreads();
  }
  double ln_tot[RM];  // LeastNUMANodes: what the node's zones hold per resource (first-pass bound of numa_required_fast)
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    ln_tot[r] = 0.0;
    if constexpr (SG == kSgLeastNuma) {
#pragma unroll
      for (int z = 0; z < kZ; ++z) ln_tot[r] += __builtin_fmax(ns.av[z][r], 0.0);
    }
  }
  const bool fresh = flags & SPX_NRT_F_FRESH;
  const bool has_nrt = flags & SPX_NRT_F_HAS_NRT;
  const bool single = flags & SPX_NRT_F_SINGLE_NUMA;
  const bool pod_scope = flags & SPX_NRT_F_POD_SCOPE;
  const bool aligned = fresh && has_nrt && single;  // the node's NUMA table decides Filter and Score

  {  // the chunk's pod records -> LDS: contiguous in memory, 16-byte pieces, one round trip
    constexpr int kPodQuads = pod_words<RM>() / 4;
    const int n_quads = rows * kPodQuads;
    uint4* dst = reinterpret_cast<uint4*>(pod_lds);
    if (!listed) {
      const uint4* src = reinterpret_cast<const uint4*>(a.pod_items + first * pod_words<RM>());
      for (int i = threadIdx.x; i < n_quads; i += 256) dst[i] = src[i];
    } else {  // a record per listed row
      for (int i = threadIdx.x; i < n_quads; i += 256) {
        const int p = i / kPodQuads, q = i - p * kPodQuads;
        dst[i] = reinterpret_cast<const uint4*>(a.pod_items + static_cast<int64_t>(a.row_list[first + p]) * pod_words<RM>())[q];
      }
    }
    if (threadIdx.x == 0) redo_n = 0, pk_flagged = 0;
    uint4* z = reinterpret_cast<uint4*>(&stage[0][0][0]) + threadIdx.x;  // empty slots of the window stay 0 (row padding)
#pragma unroll
    for (int i = 0; i < static_cast<int>(sizeof(stage) / 16 / 256); ++i) z[i * 256] = uint4{0, 0, 0, 0};
  }
  __syncthreads();
  if constexpr (PK) {
    // every staged request item into the packed loop's form, in place (PkRegs, nrt_fast_device.h): nv[r] = -float32(Value(request r));
    // and the pods whose table-slot request (any of their items) is listed for THIS node window: recomputed after the loop
    const int ts = a.pk_tab_slot;
    for (int i = threadIdx.x; i < rows * (kItemsPerPod - 1); i += 256) {
      const int p = i / (kItemsPerPod - 1);
      uint32_t* w = pod_lds + p * pod_words<RM>() + (1 + i % (kItemsPerPod - 1)) * item_words<RM>();
      const uint32_t used = w[2 * RM] & 0xffu;
      float nv[RM];
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const int at = r == a.cpu_slot ? 2 * RM + 2 : 2 * r;  // Value() of the cpu request / the request as written
        nv[r] = -static_cast<float>(__hiloint2double(static_cast<int>(w[at + 1]), static_cast<int>(w[at])));
      }
#pragma unroll
      for (int r = 0; r < RM; ++r) w[pk_nv_dword<RM>(r)] = __float_as_uint(nv[r]);
      if (ts < 0 || !((used >> ts) & 1u)) continue;
      const double k = __hiloint2double(static_cast<int>(w[2 * ts + 1]), static_cast<int>(w[2 * ts])) * a.pk_tab_inv_unit;
      // (the engine derived unit and kmax from this very batch: k is a whole number within the table; anything else is recomputed too)
      const bool inside = k >= 0.0 && k <= static_cast<double>(a.pk_tab_kmax) && k == __builtin_floor(k);
      const uint32_t word = inside ? a.pk_tab[static_cast<size_t>(static_cast<uint32_t>(k)) * a.pk_tab_words + (static_cast<uint32_t>(window) >> 5)] : ~0u;
      if ((word >> (static_cast<uint32_t>(window) & 31u)) & 1u) atomicOr(&pk_flagged, 1u << p);
    }
    __syncthreads();
  }
  // What the wave's 64 nodes have in common (wave-uniform): a pod the launch has nothing to compute for — not filtered,
  // not Guaranteed — skips the exec-masked regions with one scalar branch.  Round 2's loop ran every pod through them
  // (48 scalar instructions for a BestEffort pod; scalar and vector instructions issue at the same rate per SIMD).
  const bool w_pod = __ballot(aligned && pod_scope) != 0, w_ctr = __ballot(aligned && !pod_scope) != 0;
  const uint32_t st_stale = fresh ? 0u : static_cast<uint32_t>(SPX_NRT_ST_INVALID_TOPOLOGY);
  uint32_t acc_status = 0, acc_score = 0;  // this node's results of the current group of four pods
  int raw_score = 0;
  auto header = [&](int p) { return *reinterpret_cast<const u32x2*>(pod_lds + p * pod_words<RM>()); };
  u32x2 hv = header(0);
  for (int p = 0; p < rows; ++p) {
    // ---- wave-uniform pod record, from LDS; the next pod's header is requested before this pod's work
    const uint32_t* pit = pod_lds + p * pod_words<RM>();  // the pod's first item
    const uint32_t hw[2] = {static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(hv.x))),
                            static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(hv.y)))};
    hv = header(p + 1 < rows ? p + 1 : p);
    // the first container's item is requested now, ahead of the header decode (container-scope nodes nearly always need it)
    ItemRegs<RM, FULL> cur;
    PkRegs<RM> cur_pk;
    if constexpr (PK) {
      if (w_ctr) cur_pk = load_item_pk<RM>(pit, 2);
    } else {
      if (w_ctr) cur = load_item<RM, FULL>(pit, 2);
    }
    const int qos = hw[0] & 0xffu;
    const bool non_native = ((hw[0] >> 8) & 0xffu) != 0;
    const int n_ctr = (hw[0] >> 16) & 0xffu;
    const int last_app = static_cast<int>(hw[0] >> 24) == 0xff ? -1 : static_cast<int>(hw[0] >> 24);
    const uint32_t inv_n = hw[1];  // ceil(2^16 / n_ctr)
    const bool non_g = qos != SPX_QOS_GUARANTEED;
    const bool filtered = !(qos == SPX_QOS_BESTEFFORT && !non_native);  // filter.go:186-190
    const bool u_filter = PH != kPhScore && filtered;                            // wave-uniform: the pod has a Filter verdict to compute
    const bool u_score = SG != kSgLeastNuma && PH != kPhFilter && !non_g;        // ... a Score

    uint32_t status = filtered ? st_stale : 0u;
    int score = non_g ? 100 : 0;
    bool redo = false;  // BalancedAllocation, float32 form: some container's score could not be decided
    if (u_filter || u_score) {
    const bool want_filter = u_filter && aligned;
    const bool want_score = u_score && aligned;

    if constexpr (PK) {  // Score only, both scopes in the packed form
      if (w_pod && pod_scope && aligned && want_score) score = score_least_packed<RM>(ns, a, load_item_pk<RM>(pit, 1));
      if (w_ctr && !pod_scope && aligned) {
        int sum = 0;
        for (int c = 0; c < n_ctr; ++c) {
          const PkRegs<RM> nxt = load_item_pk<RM>(pit, 2 + (c + 1 < kC ? c + 1 : c));
          if (want_score) sum += score_least_packed<RM>(ns, a, cur_pk);
          cur_pk = nxt;
        }
        if (want_score) score = static_cast<int>((static_cast<uint32_t>(sum) * inv_n) >> 16);
      }
    }
    if (!PK && w_pod && pod_scope && aligned) {  // singleNUMAPodLevelHandler / podScopeScore
      const Item<RM> it = decode_item<RM, FULL>(load_item<RM, FULL>(pit, 1));
      if constexpr (PH != kPhScore) {
        if (want_filter) {
          uint32_t pos;
          if (!fits_fast(ns, it, &pos)) status = SPX_NRT_ST_POD;
        }
      }
      if constexpr (PH != kPhFilter) {
        if (want_score) {
          if constexpr (kBalF32) {
            bool rd;
            score = score_balanced_f32(bn, ns.nz, a.cpu_slot, a.exact32_slots, it, &rd);
            redo |= rd;
          } else {
            score = score_each_fast<RM, SG>(ns, a, it, cpu_v, braw);
          }
        }
      }
    }
    if (!PK && w_ctr && !pod_scope && aligned) {  // singleNUMAContainerLevelHandler / containerScopeScore
      // One pass in container order (init containers come first — checked at upload): an init container must fit
      // and is never subtracted; an app container is placed on the lowest fitting zone and subtracted from it.
      // Least/MostAllocated's zone scores read only b (never the mutable table), so they score in the same pass;
      // BalancedAllocation scores after the undo.  The next container's item is requested from LDS before this one is worked on.
      uint32_t chosen = 0;  // per app container: the zone it was subtracted from + 1 (0 = not placed), 4 bits each, for the undo
      int sum = 0;
      for (int c = 0; c < n_ctr; ++c) {
        const ItemRegs<RM, FULL> nxt = load_item<RM, FULL>(pit, 2 + (c + 1 < kC ? c + 1 : c));
        const Item<RM> it = decode_item<RM, FULL>(cur);
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
        cur = nxt;
      }
      if constexpr (PH != kPhScore) if (want_filter && last_app > 0) {  // undo: Filter works on a private copy in the reference
        cur = load_item<RM, FULL>(pit, 2);
        for (int c = 0; c < last_app; ++c) {
          const ItemRegs<RM, FULL> nxt = load_item<RM, FULL>(pit, 2 + c + 1);
          const Item<RM> it = decode_item<RM, FULL>(cur);
          if (it.kind == SPX_CTR_APP) adjust_fast(ns, it, ((chosen >> (4 * c)) & 0xfu) - 1u, ((chosen >> (4 * c)) & 0xfu) != 0, 1.0);
          cur = nxt;
        }
      }
      if constexpr (SG != kSgLeast && SG != kSgMost && PH != kPhFilter) {
        if (want_score) {
          for (int c = 0; c < n_ctr; ++c) {
            const Item<RM> it = decode_item<RM, FULL>(load_item<RM, FULL>(pit, 2 + c));
            if constexpr (kBalF32) {
              bool rd;
              sum += score_balanced_f32(bn, ns.nz, a.cpu_slot, a.exact32_slots, it, &rd);
              redo |= rd;
            } else {
              sum += score_each_fast<RM, SG>(ns, a, it, cpu_v, braw);
            }
          }
        }
      }
      if (want_score) score = static_cast<int>((static_cast<uint32_t>(sum) * inv_n) >> 16);  // int64(mean): sum / n_ctr, sum <= 800
    }
    if constexpr (kBalF32) {
      const bool mark = want_score && redo;
      score = mark ? kBalRedo : score;  // k_nrt_bal_redo recomputes the cell in float64
      const uint64_t mm = __ballot(mark);
      if (mm != 0) {  // wave-uniform: one LDS atomic per wave and pod
        const int leader = __builtin_ctzll(mm);
        uint32_t at = 0;
        if (lane == leader) at = atomicAdd(&redo_n, static_cast<uint32_t>(__builtin_popcountll(mm)));
        at = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(at), leader)) +
             static_cast<uint32_t>(__builtin_popcountll(mm & ((1ull << lane) - 1ull)));
        if (mark) {
          const uint32_t row = static_cast<uint32_t>(row_of(p));
          if (at < static_cast<uint32_t>(kRedoBuf)) {
            redo_buf[at][0] = row, redo_buf[at][1] = static_cast<uint32_t>(n);
          } else {  // the block's buffer is full (a window of exactly-integer scores): straight to the global list
            const uint32_t g = atomicAdd(a.redo_list, 1u);
            if (g < a.redo_cap) a.redo_list[2 + 2 * static_cast<size_t>(g)] = row, a.redo_list[3 + 2 * static_cast<size_t>(g)] = static_cast<uint32_t>(n);
          }
        }
      }
    }
    }  // the pod has something to compute

    if constexpr (SG == kSgLeastNuma) {
      if (!non_g) {
        bool listed_cell = false;
        score = ln_score_wave<RM, kLnDeferred>(ns, a, pit, n_ctr, n, in, fresh && has_nrt, pod_scope, nns, mmin, ln_subset, ln_allow, ln_tot, score,
                                               &listed_cell);
        if constexpr (kLnDeferred) {
          // the listed cells of this pod row: appended to the row's node lists — one for the pod-scope nodes (they take the
          // pod-level item), one for the container-scope nodes, so that a wave of k_nrt_ln_redo walks one kind of item — with one
          // atomic per wave, pod and scope (the engine groups a window's nodes by scope: a wave nearly always holds one)
          if (__ballot(listed_cell) != 0) {
            const int64_t slot = listed ? first + p : first + p - a.row_begin;
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
              const bool mine = listed_cell && (pod_scope ? 1 : 0) == sc;
              const uint64_t mm = __ballot(mine);
              if (mm == 0) continue;
              const int leader = __builtin_ctzll(mm);
              uint32_t at = 0;
              if (lane == leader) at = atomicAdd(a.redo_list + 2 + 2 * slot + sc, static_cast<uint32_t>(__builtin_popcountll(mm)));
              at = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(at), leader)) +
                   static_cast<uint32_t>(__builtin_popcountll(mm & ((1ull << lane) - 1ull)));
              if (mine) {
                if (at < a.ln_per_row) a.redo_list[2 + 2 * a.ln_rows + (2 * slot + sc) * a.ln_per_row + at] = static_cast<uint32_t>(n);
                else a.redo_list[0] = 1u;  // the list is full: the launch falls back to the complete sweep
              }
            }
          }
          score = listed_cell ? kLnRedo : score;
        }
      }
    }

    raw_score = score;  // (raw rows: the single row of the launch)
    if constexpr (SG == kSgLeastNuma) score = score < 0 ? 0 : score;  // 100 - count*(100/maxNUMA) can go negative; the table saturates
    const int sh = 8 * (p & 3);
    if constexpr (PH != kPhScore) acc_status |= status << sh;
    if constexpr (PH != kPhFilter) acc_score |= static_cast<uint32_t>(score > 255 ? 255 : score) << sh;
    if ((p & 3) == 3 || p + 1 == rows) {  // wave-uniform
      if (in) {
        if constexpr (PH != kPhScore) stage[0][p >> 2][pos] = acc_status;
        if constexpr (PH != kPhFilter) stage[kScoreTab][p >> 2][pos] = acc_score;
      }
      acc_status = acc_score = 0;
    }
  }
  if (a.out_raw != nullptr) {  // raw int64 scores of the launch's single row, no table writes
    if (in) a.out_raw[n] = raw_score;
    return;
  }
  if constexpr (LNM == kLnIfOverflow) {  // the fallback ran: its cells count as re-evaluated (spx_fetch_stats)
    if (a.stats && threadIdx.x == 0)
      atomicAdd(a.stats + (SPX_PLUGIN_NRT * kStatSlots + static_cast<int>(blockIdx.x & (kStatSlots - 1))) * kStatStride, static_cast<unsigned long long>(rows) * kWindow);
  }
  __syncthreads();
  if constexpr (PK) {
    const uint32_t flagged = pk_flagged;  // block-uniform (every atomicOr precedes the barrier above)
    if (flagged != 0) {
      // second pass: the flagged pods' cells of this window with the table slot in the float64 form — its eight multipliers read
      // again (the loop kept float32 pairs), the other slots packed as before.  Each lane rewrites its own byte of the staged dword.
      double bt[kZ];
#pragma unroll
      for (int z = 0; z < kZ; ++z) {
        const double b = in ? a.f_rc[(static_cast<int64_t>(z) * R + a.pk_tab_slot) * a.n_nodes + n] : kNrtNoCap;
        bt[z] = b == kNrtNoCap ? __builtin_inf() : b;
      }
      uint32_t n_redone = 0;
      for (uint32_t left = flagged; left != 0; left &= left - 1) {
        const int p = __builtin_ctz(left);
        const uint32_t* pit = pod_lds + p * pod_words<RM>();
        const uint32_t h0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(pit[0])));
        const uint32_t h1 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(pit[1])));
        if ((h0 & 0xffu) != SPX_QOS_GUARANTEED) continue;  // uniform: 100 everywhere, the loop wrote it
        ++n_redone;
        const int n_ctr = (h0 >> 16) & 0xffu;
        auto raw_of = [&](int slot) {  // the item's table-slot request as written
          const uint32_t* w = pit + slot * item_words<RM>() + 2 * a.pk_tab_slot;
          return __hiloint2double(static_cast<int>(w[1]), static_cast<int>(w[0]));
        };
        int score = 0;
        if (aligned) {
          if (pod_scope) {
            score = score_least_packed<RM, true>(ns, a, load_item_pk<RM>(pit, 1), bt, raw_of(1));
          } else {
            int sum = 0;
            for (int c = 0; c < n_ctr; ++c) sum += score_least_packed<RM, true>(ns, a, load_item_pk<RM>(pit, 2 + c), bt, raw_of(2 + c));
            score = static_cast<int>((static_cast<uint32_t>(sum) * h1) >> 16);
          }
        }
        if (in) {
          const int sh = 8 * (p & 3);
          uint32_t& cell = stage[kScoreTab][p >> 2][pos];
          cell = (cell & ~(0xffu << sh)) | (static_cast<uint32_t>(score) << sh);
        }
      }
      if (a.stats && threadIdx.x == 0 && n_redone)  // these cells count as re-evaluated (spx_fetch_stats)
        atomicAdd(a.stats + (SPX_PLUGIN_NRT * kStatSlots + static_cast<int>(blockIdx.x & (kStatSlots - 1))) * kStatStride, static_cast<unsigned long long>(n_redone) * kWindow);
      __syncthreads();
    }
  }
  if constexpr (kBalF32) {
    const uint32_t cnt = redo_n < static_cast<uint32_t>(kRedoBuf) ? redo_n : static_cast<uint32_t>(kRedoBuf);
    if (cnt != 0) {  // block-uniform
      if (threadIdx.x == 0) redo_base = atomicAdd(a.redo_list, cnt);
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
        const uint32_t g = redo_base + i;  // past the capacity: the count says so and k_nrt_bal_scan takes over
        if (g < a.redo_cap) a.redo_list[2 + 2 * static_cast<size_t>(g)] = redo_buf[i][0], a.redo_list[3 + 2 * static_cast<size_t>(g)] = redo_buf[i][1];
      }
    }
  }
  // rows leave as whole 256-byte segments: lane l gathers byte (row & 3) of the four dwords of nodes 4l .. 4l+3
  const int64_t col = base + lane * 4;
  if (col < a.row_stride) {
    for (int i = wave; i < rows; i += 4) {
      const int64_t row = row_of(i);
      const uint32_t b = static_cast<uint32_t>(i & 3);
      const uint32_t pick = 0x0c0c0000u | ((4u + b) << 8) | b;  // v_perm_b32: byte b of the low operand, byte b of the high one, 0, 0
#pragma unroll
      for (int tbl = 0; tbl < 2; ++tbl) {
        if ((PH == kPhFilter && tbl == 1) || (PH == kPhScore && tbl == 0)) continue;
        const u32x4 w = *reinterpret_cast<const u32x4*>(&stage[tbl ? kScoreTab : 0][i >> 2][lane * 4]);
        const uint32_t lo = __builtin_amdgcn_perm(w.y, w.x, pick), hi = __builtin_amdgcn_perm(w.w, w.z, pick);
        uint8_t* out = (tbl ? a.out_score : a.out_status) + row * a.row_stride + col;
        *reinterpret_cast<uint32_t*>(out) = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
    }
  }
}

// BalancedAllocation fix-up: the cells the float32 Score launch marked kBalRedo, recomputed with the float64 form — the node's
// tables loaded for that one cell, the pod's items read with ordinary (per-lane) loads.  One thread per 16 bytes of a score
// row; nearly all threads find nothing.
// the float64 form for one (pod, node) cell
template <int RM>
__device__ __forceinline__ int balanced_cell_exact(const NrtArgs& a, int64_t pod, int64_t n) {
  const uint32_t* pit = a.pod_items + pod * (kItemsPerPod * item_words<RM>());
  const uint32_t h0 = pit[0], inv_n = pit[1];
  const int n_ctr = (h0 >> 16) & 0xffu;
  FastNode<RM> ns;
  double cpu_v[kZ], braw[kZ];
  load_fast_node<RM, kSgBalanced>(a, n, true, ns, cpu_v, braw);
  int score;
  if (a.flags[n] & SPX_NRT_F_POD_SCOPE) {
    score = score_each_fast<RM, kSgBalanced>(ns, a, decode_item<RM, true, false>(load_item<RM, true>(pit, 1)), cpu_v, braw);
  } else {
    int sum = 0;
#pragma unroll 1
    for (int c = 0; c < n_ctr; ++c) sum += score_each_fast<RM, kSgBalanced>(ns, a, decode_item<RM, true, false>(load_item<RM, true>(pit, 2 + c)), cpu_v, braw);
    score = static_cast<int>((static_cast<uint32_t>(sum) * inv_n) >> 16);
  }
  return score > 254 ? 254 : score;
}

// Scan: one thread per 16 bytes of a score row.  The marked cells are rare and scattered (config #3: 0.7 % — exactly integer
// scores of small-integer quantities, mostly), and recomputing them where they are found leaves 1-7 lanes of a wave working
// through ~30 us of dependent loads each (10 ms for 1.8e6 cells).  So the scan only compacts them into a list — one atomic
// per block of 1024 threads — and k_nrt_bal_redo works through the list with full waves.  Cells that do not fit the list are
// recomputed in place.
// (5-8 resource slots: the in-place recomputation holds a node's 2 x 8 x 8 doubles — 256 threads, so that it has the registers)
template <int RM>
constexpr int scan_threads() { return RM == 4 ? 1024 : 256; }
template <int RM>
__global__ __launch_bounds__(scan_threads<RM>()) void k_nrt_bal_scan(NrtArgs a) {
  constexpr int kScanThreads = scan_threads<RM>();
  SPX_RESOLVE_ROWS(a);
  // the Score launch lists its undecided cells itself; this pass only runs when they did not all fit the list — it then finds
  // every cell still marked and recomputes it where it is (the list is full: `at` below is past the capacity)
  if (a.redo_list[0] <= a.redo_cap) return;
  __shared__ uint32_t wave_total[kScanThreads / 64];
  __shared__ uint32_t block_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t per_row = a.row_stride / 16;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool live = idx < (a.row_end - a.row_begin) * per_row;
  const int64_t pod = a.row_begin + (live ? idx / per_row : 0);
  const int64_t col = live ? (idx % per_row) * 16 : 0;
  uint8_t* cells = a.out_score + pod * a.row_stride + col;
  uint4 w = uint4{0, 0, 0, 0};
  if (live) w = *reinterpret_cast<const uint4*>(cells);
  // bit j of `marks`: byte j is kBalRedo (0xff)
  uint32_t marks = 0;
  {
    const uint32_t words[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) marks |= (((words[q] >> (8 * b)) & 0xffu) == static_cast<uint32_t>(kBalRedo) ? 1u : 0u) << (4 * q + b);
    // row padding past the last node window is never written by the sweep (and not zeroed at allocation): a stray 0xff there is
    // not a cell
    const int64_t left = a.n_nodes - col;
    marks = left >= 16 ? marks : (left <= 0 ? 0u : marks & ((1u << left) - 1u));
  }
  const uint32_t mine = static_cast<uint32_t>(__builtin_popcount(marks));
  if (__syncthreads_or(mine != 0) == 0) return;  // nothing marked in this block's 16 KB of table
  // exclusive prefix over the block: in the wave by shuffles, across waves through LDS
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), d));
    incl += lane >= d ? up : 0u;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int i = 0; i < kScanThreads / 64; ++i) {
      const uint32_t c = wave_total[i];
      wave_total[i] = t;
      t += c;
    }
    block_base = atomicAdd(a.redo_list, t);
  }
  __syncthreads();
  uint32_t at = block_base + wave_total[wave] + incl - mine;
  unsigned in_place = 0;
  while (marks) {
    const int j = __builtin_ctz(marks);
    marks &= marks - 1;
    if (at < a.redo_cap) {
      a.redo_list[2 + 2 * static_cast<size_t>(at)] = static_cast<uint32_t>(pod);
      a.redo_list[3 + 2 * static_cast<size_t>(at)] = static_cast<uint32_t>(col + j);
    } else {
      cells[j] = static_cast<uint8_t>(balanced_cell_exact<RM>(a, pod, col + j));
      ++in_place;
    }
    ++at;
  }
  if (a.stats && in_place)
    atomicAdd(a.stats + (SPX_PLUGIN_NRT * kStatSlots + static_cast<int>(idx & (kStatSlots - 1))) * kStatStride, static_cast<unsigned long long>(in_place));
}

template <int RM>
__global__ __launch_bounds__(256) void k_nrt_bal_redo(NrtArgs a) {
  SPX_RESOLVE_ROWS(a);
  const uint32_t count = a.redo_list[0] < a.redo_cap ? a.redo_list[0] : a.redo_cap;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t pod = a.redo_list[2 + 2 * static_cast<size_t>(i)];  // absolute row
  const int64_t n = a.redo_list[3 + 2 * static_cast<size_t>(i)];
  a.out_score[pod * a.row_stride + n] = static_cast<uint8_t>(balanced_cell_exact<RM>(a, pod, n));
  if (a.stats && (threadIdx.x & 63) == 0) {  // one update per wave: the live lanes of the wave
    const unsigned live = min(64u, count - i);
    atomicAdd(a.stats + (SPX_PLUGIN_NRT * kStatSlots + static_cast<int>((i >> 6) & (kStatSlots - 1))) * kStatStride, static_cast<unsigned long long>(live));
  }
}

// What k_nrt_ln_redo reads of a node, as ONE record: its lanes hold arbitrary nodes, and the column layout the sweep streams
// through (a cache line = 16 consecutive nodes of one column) cost every lane ~50 separate lines — 57 GB of L2 traffic for
// config #3's 1.8e7 listed cells, the kernel's bound.  Record: av[kZ][RM] doubles, then kLnDwords dwords of minimum-distance
// sets, then {rep masks (RM bytes, padded to 8), node_present, flags | n_zones << 8 | max_numa << 16}; 16-byte aligned.
template <int RM>
constexpr int ln_rec_words() { return kZ * RM * 2 + kLnDwords + 4; }
static_assert(ln_rec_words<4>() % 4 == 0 && ln_rec_words<8>() % 4 == 0, "records are read as uint4");

template <int RM>
__global__ __launch_bounds__(256) void k_nrt_ln_pack(NrtArgs a) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (n >= a.n_nodes) return;
  FastNode<RM> ns;
  double cpu_v[kZ], braw[kZ];
  load_fast_node<RM, kSgLeastNuma>(a, n, true, ns, cpu_v, braw);
  uint32_t* rec = a.ln_rec + n * ln_rec_words<RM>();
#pragma unroll
  for (int z = 0; z < kZ; ++z)
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      rec[2 * (z * RM + r)] = static_cast<uint32_t>(__double2loint(ns.av[z][r]));
      rec[2 * (z * RM + r) + 1] = static_cast<uint32_t>(__double2hiint(ns.av[z][r]));
    }
  uint32_t* t = rec + kZ * RM * 2;
#pragma unroll
  for (int d = 0; d < kLnDwords; ++d) t[d] = a.ln_tab[static_cast<int64_t>(d) * a.n_nodes + n];
  t[kLnDwords] = ns.rep[0];
  t[kLnDwords + 1] = RM > 4 ? ns.rep[RM > 4 ? 1 : 0] : 0u;
  t[kLnDwords + 2] = ns.node_present;
  t[kLnDwords + 3] = static_cast<uint32_t>(a.flags[n]) | (static_cast<uint32_t>(ns.nz) << 8) | (static_cast<uint32_t>(a.max_numa[n]) << 16);
}

// LeastNUMANodes fix-up: the cells the batch Score launch listed (kLnRedo) because some container's subset search needs more
// than sizes 1 and 2 — 13 % of config #3's cells, spread so evenly that nearly every wave of the sweep used to run the 219 larger
// subsets for a handful of its lanes.  The lists are per pod row: a wave here takes 64 listed nodes of ONE pod, so the pod's
// items are wave-uniform exactly as in the sweep (scalar resource sets, the same ln_score_wave), every lane has work, and the
// node tables are loaded per lane.  A workgroup = four consecutive 64-node segments of a row's list.
// redo_list[1] = the longest list: bounds k_nrt_ln_redo's walk (one workgroup; an atomicMax per wave and pod in the sweep, all on
// one address, cost 16 ms)
__global__ __launch_bounds__(1024) void k_nrt_ln_longest(NrtArgs a) {
  __shared__ uint32_t part[16];
  uint32_t m = 0;
  for (int64_t i = threadIdx.x; i < 2 * a.ln_rows; i += 1024) {
    const uint32_t c = a.redo_list[2 + i];
    m = c > m ? c : m;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = static_cast<uint32_t>(__shfl_xor(static_cast<int>(m), d, 64));
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = part[w] > m ? part[w] : m;
    a.redo_list[1] = m;
  }
}

template <int RM>
// (round 6: bounded to three waves per SIMD for <= 4 slots — 168 VGPRs, 18 of 196 spilled outside the search —: 3.32 -> 3.16 ms at config #3)
__global__ __launch_bounds__(256, RM == 4 ? 3 : 1) void k_nrt_ln_redo(NrtArgs a) {
  SPX_RESOLVE_ROWS(a);
  __shared__ __align__(4) uint8_t ln_subset[kLnDwords * 32];
  __shared__ uint32_t ln_allow[256 * kLnDwords];
  __shared__ __align__(16) uint32_t pod_recs[4][pod_words<RM>()];
  // Persistent waves: the 12 KB of block-constant tables are copied once per workgroup, then every wave walks its share of the
  // (list, 64-entry segment) units on its own — a workgroup per 256 entries spent as long on that copy as on the search.
  // Segment-major order: consecutive units are the same segment of consecutive lists, so that the lists' first (often only)
  // segments spread over all waves; redo_list[1] = the longest list bounds the walk.
  for (int i = threadIdx.x; i < 256 * kLnDwords; i += blockDim.x) ln_allow[i] = a.ln_const[i];
  for (int i = threadIdx.x; i < kLnDwords * 32 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(ln_subset)[i] = a.ln_const[256 * kLnDwords + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t* pod_rec = pod_recs[wave];
  const int64_t n_lists = 2 * a.ln_rows;
  const uint32_t longest = a.redo_list[1] < a.ln_per_row ? a.redo_list[1] : a.ln_per_row;
  const int64_t units = n_lists * ((longest + 63u) / 64u);
  for (int64_t u = static_cast<int64_t>(blockIdx.x) * 4 + wave; u < units; u += static_cast<int64_t>(gridDim.x) * 4) {
  const int64_t list = u % n_lists;  // 2 * row slot + scope
  const int64_t slot = list >> 1;
  const uint32_t seg0 = static_cast<uint32_t>(u / n_lists) * 64u;
  const uint32_t listed = uload(a.redo_list + 2 + list);
  const uint32_t count = listed < a.ln_per_row ? listed : a.ln_per_row;
  if (seg0 >= count) continue;  // wave-uniform
  const int64_t pod = a.row_list ? static_cast<int64_t>(uload(a.row_list + slot)) : a.row_begin + slot;
  __builtin_amdgcn_wave_barrier();  // the previous unit's reads of pod_rec are done (LDS is in order within a wave)
  for (int i = lane; i < pod_words<RM>(); i += 64) pod_rec[i] = a.pod_items[pod * pod_words<RM>() + i];
  __builtin_amdgcn_wave_barrier();
  const uint32_t at = seg0 + lane;
  const bool live = at < count;
  const int64_t n = a.redo_list[2 + 2 * a.ln_rows + list * a.ln_per_row + (live ? at : seg0)];
  const int n_ctr = static_cast<int>((static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(pod_rec[0]))) >> 16) & 0xffu);
  FastNode<RM> ns;
  uint32_t mmin[kLnDwords];
  uint32_t flags;
  int nns;
  {  // the node's record (k_nrt_ln_pack): a few adjacent cache lines per lane
    constexpr int kW = ln_rec_words<RM>();
    const u32x4* rec = reinterpret_cast<const u32x4*>(a.ln_rec + n * kW);
    uint32_t w[kW];
#pragma unroll
    for (int q = 0; q < kW / 4; ++q) {
      const u32x4 v = rec[q];
      w[4 * q] = v.x, w[4 * q + 1] = v.y, w[4 * q + 2] = v.z, w[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int z = 0; z < kZ; ++z)
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        ns.av[z][r] = __hiloint2double(static_cast<int>(w[2 * (z * RM + r) + 1]), static_cast<int>(w[2 * (z * RM + r)]));
        ns.b[z][r] = kNoCap;
      }
    const uint32_t* t = w + kZ * RM * 2;
#pragma unroll
    for (int d = 0; d < kLnDwords; ++d) mmin[d] = t[d];
    ns.rep[0] = t[kLnDwords];
    if constexpr (RM > 4) ns.rep[1] = t[kLnDwords + 1];
    ns.node_present = t[kLnDwords + 2];
    flags = t[kLnDwords + 3] & 0xffu;
    ns.nz = static_cast<int>((t[kLnDwords + 3] >> 8) & 0xffu);
    nns = 100 / static_cast<int>(t[kLnDwords + 3] >> 16);
#pragma unroll
    for (int i = 0; i < RM / 4; ++i) ns.fill[i] = 0;
#pragma unroll
    for (int r = 0; r < RM; ++r)
      if (r < a.n_res && (a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL) && ns.repmask(r) == 0) ns.fill[r >> 2] |= 0xffu << (8 * (r & 3));
  }
  double ln_tot[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    ln_tot[r] = 0.0;
#pragma unroll
    for (int z = 0; z < kZ; ++z) ln_tot[r] += __builtin_fmax(ns.av[z][r], 0.0);
  }
  bool unused;
  // a listed cell belongs to a Guaranteed pod and a node with a fresh NRT
  int score = ln_score_wave<RM, false, false>(ns, a, pod_rec, n_ctr, n, true, live, (flags & SPX_NRT_F_POD_SCOPE) != 0, nns, mmin, ln_subset, ln_allow, ln_tot, 0, &unused);
  if (live) {
    score = score < 0 ? 0 : (score > 254 ? 254 : score);
    a.out_score[pod * a.row_stride + n] = static_cast<uint8_t>(score);
  }
  if (a.stats && lane == 0)  // one update per wave: the live lanes of the wave
    atomicAdd(a.stats + (SPX_PLUGIN_NRT * kStatSlots + static_cast<int>(u & (kStatSlots - 1))) * kStatStride, static_cast<unsigned long long>(min(64u, count - at)));
  }  // units
}

}  // namespace

namespace {
// The requests of the table slot for which the packed float32 LeastAllocated score differs from the integer division
// (score_least_packed).  A difference needs x = 100 - 100 v / c within o + |e| < 1.8e-5 of an integer, i.e. v within 1.8e-7 c of
// j c / 100 for some j = 0..100: one thread per (node, zone, j) walks the multiples of the unit inside 2.7e-7 c of that point (mostly
// none: the unit is 2^20 for memory, 2.7e-7 c is 4.6 KB for a 16 GiB zone), REPLAYS the kernel's expression on each and compares with
// (c - v) * 100 / c.  Bit (window & 31) of word window / 32 of the value's row.
__global__ __launch_bounds__(256) void k_nrt_pk_tab_build(NrtArgs a) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int j = static_cast<int>(idx & 127), z = static_cast<int>((idx >> 7) % kZ);
  const int64_t slot = idx / (128 * kZ);  // position in the window order (perm: the block that owns the node is the one that looks it up)
  if (j > 100 || slot >= (a.n_nodes + kWindow - 1) / kWindow * kWindow || a.pk_tab_slot < 0) return;
  const int32_t pn = a.perm[slot];
  if (pn < 0) return;
  const int64_t at = (static_cast<int64_t>(z) * a.n_res + a.pk_tab_slot) * a.n_nodes + pn;
  const double c = a.f_av[at];  // available = the capacity LeastAllocated scores against; -1: not reported
  if (!(c > 0.0)) return;       // no capacity: 0 in both forms whatever the request
  const double b64 = a.f_rc[at];
  const float b32 = static_cast<float>(b64 == kNrtNoCap ? __builtin_inf() : b64);
  const double unit = 1.0 / a.pk_tab_inv_unit;
  const double mid = static_cast<double>(j) * c / 100.0, half = 2.7e-7 * c + 1e-9 * mid;
  double k0 = __builtin_ceil((mid - half) * a.pk_tab_inv_unit), k1 = __builtin_floor((mid + half) * a.pk_tab_inv_unit);
  k0 = k0 < 0.0 ? 0.0 : k0;
  k1 = k1 > static_cast<double>(a.pk_tab_kmax) ? static_cast<double>(a.pk_tab_kmax) : k1;
  const uint32_t window = static_cast<uint32_t>(slot / kWindow);
  const int64_t ci = static_cast<int64_t>(c);
  for (double k = k0; k <= k1; k += 1.0) {
    const double v = k * unit;
    const int64_t vi = static_cast<int64_t>(v);
    uint32_t want, got;
    if (a.strategy == SPX_NRT_MOST_ALLOCATED) {
      // mostAllocatedScore most_allocated.go:45-54; the fused walk's t = fma(+v, b32, -0.5 + o), taken only where the request fits (a compare
      // of its own there: the Filter's rank bits) — so requests above the capacity need no entry
      if (vi > ci) continue;
      want = static_cast<uint32_t>((vi * 100) / ci);
      got = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(static_cast<float>(v), b32, -0.5f + nrtdev::kPkOffsetTab), 0, 0u) & 0xffu;
    } else {
      want = vi > ci ? 0u : static_cast<uint32_t>(((ci - vi) * 100) / ci);  // leastAllocatedScore, least_allocated.go:45-55
      got = nrtdev::least_packed_one(static_cast<float>(v), b32, 99.5f + nrtdev::kPkOffsetTab);
    }
    if (got != want) atomicOr(a.pk_tab + static_cast<size_t>(k) * a.pk_tab_words + (window >> 5), 1u << (window & 31u));
  }
}
}  // namespace

void launch_nrt_pk_tab_build(const NrtArgs& a, int n_tiles, hipStream_t s) {
  (void)hipMemsetAsync(a.pk_tab, 0, (static_cast<size_t>(a.pk_tab_kmax) + 1) * a.pk_tab_words * 4, s);
  hipLaunchKernelGGL(k_nrt_pk_tab_build, dim3(static_cast<unsigned>(static_cast<int64_t>(n_tiles) * kWindow * kZ * 128 / 256)), dim3(256), 0, s, a);
}

// BalancedAllocation's fix-up launches: the scan pass only acts when the list overflowed (it then finds the marks in the table)
void launch_nrt_bal_fixups(const NrtArgs& a, hipStream_t s) {
  const int64_t units = (a.row_end - a.row_begin) * (a.row_stride / 16);
  if (a.n_res <= 4) {
    hipLaunchKernelGGL((k_nrt_bal_scan<4>), dim3(static_cast<unsigned>((units + scan_threads<4>() - 1) / scan_threads<4>())), dim3(scan_threads<4>()), 0, s, a);
    hipLaunchKernelGGL((k_nrt_bal_redo<4>), dim3((a.redo_cap + 255) / 256), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((k_nrt_bal_scan<8>), dim3(static_cast<unsigned>((units + scan_threads<8>() - 1) / scan_threads<8>())), dim3(scan_threads<8>()), 0, s, a);
    hipLaunchKernelGGL((k_nrt_bal_redo<8>), dim3((a.redo_cap + 255) / 256), dim3(256), 0, s, a);
  }
}

bool launch_nrt_fast(const NrtArgs& a, hipStream_t s) {
  if (!a.fast) return false;
  if (a.strategy == SPX_NRT_LEAST_NUMA_NODES && !a.ln_tab) return false;
  const int n_tiles = static_cast<int>((a.n_nodes + kWindow - 1) / kWindow);  // windows of 256 nodes
  const int64_t chunks = ((a.row_list ? a.n_list : a.row_end - a.row_begin) + kPodsPerUnit - 1) / kPodsPerUnit;
  const int64_t per_round = n_tiles >= kXcdMapWindows ? ((n_tiles + 7) / 8) * 8 : n_tiles;
  const unsigned blocks = static_cast<unsigned>(chunks * per_round);  // see the kernel's block map
  const int sg = a.strategy == SPX_NRT_LEAST_NUMA_NODES ? kSgLeastNuma
               : a.strategy == SPX_NRT_BALANCED_ALLOCATION ? kSgBalanced : (a.strategy == SPX_NRT_LEAST_ALLOCATED ? kSgLeast : kSgMost);
  // two launches (Filter, Score) for a batch; one for a single row — the sequential commit loop replays its per-pod launches from
  // a graph and is bound by their number, not by occupancy
  const bool split = a.out_raw == nullptr && !(a.opts & kOptNrtSingleLaunch) && a.row_ptr == nullptr;
#define SPX_NRTF_CASE(RMV, SGV)                                                                           \
  if ((a.n_res <= 4) == (RMV == 4) && sg == SGV) {                                                        \
    if (split) { /* the Filter half does not depend on the strategy */ \
      if (!launch_nrt_filter_fused(a, s) && !launch_nrt_filter_rank(a, n_tiles, s)) /* rank space when the engine built the chunk stream (kernels_nrt_rank.hip) */ \
        hipLaunchKernelGGL((k_nrt_fast<RMV, kSgLeast, kPhFilter>), dim3(blocks), dim3(256), 0, s, a, n_tiles); \
      const bool ln_lists = SGV == kSgLeastNuma && a.redo_list && a.ln_rec && a.ln_rows > 0; \
      if (SGV == kSgBalanced) (void)hipMemsetAsync(a.redo_list, 0, 8, s); /* the float32 Score launch lists the cells it could not decide */ \
      if (ln_lists) { /* sizes 1-2 here, the listed cells in k_nrt_ln_redo; the complete sweep if a row's list overflowed */ \
        (void)hipMemsetAsync(a.redo_list, 0, (2 + 2 * static_cast<size_t>(a.ln_rows)) * sizeof(uint32_t), s); \
        hipLaunchKernelGGL((k_nrt_fast<RMV, SGV, kPhScore, SGV == kSgLeastNuma ? kLnDefer : kLnFull>), dim3(blocks), dim3(256), 0, s, a, n_tiles);  \
        hipLaunchKernelGGL(k_nrt_ln_longest, dim3(1), dim3(1024), 0, s, a); \
        hipLaunchKernelGGL((k_nrt_ln_pack<RMV>), dim3(static_cast<unsigned>((a.n_nodes + 255) / 256)), dim3(256), 0, s, a); \
        hipLaunchKernelGGL((k_nrt_ln_redo<RMV>), dim3(RMV == 4 ? 8192 : 2048), dim3(256), 0, s, a); /* persistent waves; the units' costs differ by the pod: 768 / 2048 / 6144 / 12288 / 49152 workgroups 3.52 / 3.17 / 2.92 / 2.93 / 3.24 ms */ \
        hipLaunchKernelGGL((k_nrt_fast<RMV, SGV, kPhScore, SGV == kSgLeastNuma ? kLnIfOverflow : kLnFull>), dim3(blocks), dim3(256), 0, s, a, n_tiles);  \
      } else { \
        if (SGV == kSgLeast && a.pk_mode && a.pk_tab_slot >= 0 && !(a.pk_tab_built && *a.pk_tab_built)) { /* the table of the packed float32 Score */ \
          launch_nrt_pk_tab_build(a, n_tiles, s); \
          if (a.pk_tab_built) *a.pk_tab_built = true; \
        } \
        if (SGV == kSgLeast && a.pk_mode) \
          hipLaunchKernelGGL((k_nrt_fast<RMV, kSgLeast, kPhScore, kLnFull, true>), dim3(blocks), dim3(256), 0, s, a, n_tiles);  \
        else \
          hipLaunchKernelGGL((k_nrt_fast<RMV, SGV, kPhScore>), dim3(blocks), dim3(256), 0, s, a, n_tiles);  \
      } \
      if (SGV == kSgBalanced) launch_nrt_bal_fixups(a, s); /* ... recomputed in float64 from the list */ \
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
