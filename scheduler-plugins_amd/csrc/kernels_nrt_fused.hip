// kernels_nrt_fused.hip — NodeResourceTopologyMatch Filter + LeastAllocated Score of a whole batch in ONE launch (round 6).
//
// Reference: pkg/noderesourcetopology/filter.go:42-245 (singleNUMAContainerLevelHandler, resourcesAvailableInAnyNUMANodes,
// singleNUMAPodLevelHandler, Filter), numaresources.go:145-182 (subtractResourcesFromNUMANodeList), score.go:62-165 (Score,
// scoreForEachNUMANode, podScopeScore, containerScopeScore), least_allocated.go:25-55.
//
// Rounds 4-5 ran two launches per sweep: the Filter in rank space (kernels_nrt_rank.hip: a zone's available quantity as the COUNT of
// the chunk's distinct request values it reaches, so that "available >= request" is a subtraction under a guard bit) and LeastAllocated's
// Score in packed float32 (score_least_packed, nrt_fast_device.h).  Both were instruction-issue bound, each staged and decoded the
// same 32 pod records per block and read the node's zone table, and the Filter issued nearly as many scalar as vector instructions
// (0.91: one scalar unit per CU serves four SIMDs).  This kernel does both for a (256-node window, 32-row chunk) block:
//
//  * the rank image of the zone table costs 2-4 registers per resource (8-16 for four resources), the Score's float32 multipliers 32 —
//    together still 4 waves per SIMD, which the float64 one-launch form (128 registers of tables) never reached;
//  * the Filter's comparison vectors are evaluated BRANCH-FREE over all resource slots: the engine writes, per vector and slot, the
//    position of the request in the chunk's list, 1 ("any reporting zone suits") or 0 (slot not compared: every count passes), and the
//    block start turns "host-level resource no zone reports" into an all-ones count and "not reported at node level" into count 0 —
//    so a vector is 2 x (sub, and) per resource and layout dword with no per-resource scalar test;
//  * what a container means for the walk (status code of a misfit, whether its verdict merges the charged-zone vectors, whether its
//    zone is remembered) is ONE byte the engine prepared (kRkOp*, spx_internal.h) instead of a decode of kinds, positions and slot sets;
//  * nothing tracks "the pod already failed": a misfit only sets the status of a lane whose status is still 0, and an empty verdict
//    charges no zone by itself;
//  * LeastAllocated's zone totals for unit weights (the reference's default: every requested resource weighs 1) are a chain of TWO
//    full-rate float32 instructions per (zone, resource) instead of 4 per zone pair at the float64 rate:
//        s   = clamp01(fma(-v, b32 / 128, (99.5 + o) / 128))        the packed form's t = fma(-v, b32, 99.5 + o), scaled by 2^-7 (exact) and
//                                                                     clamped by the instruction's output modifier: t < 0, -inf, NaN -> 0
//        acc = fma(s, -128, acc)                                      acc starts at 1.5 * 2^23: ulp 1, so this rounds t to nearest-even
//    and acc = 1.5 * 2^23 - (sum of the resource scores), exactly what v_cvt_pk_u8_f32 + v_pk_mad_u16 produced: the same float32 t, the
//    same rounding (ties to even — a tie can only occur in the table slot, which is chained FIRST, onto the even start value; the other
//    slots' t are never within 1.5e-5 of a tie, nrt_fast_device.h), hence the same table of exceptions (k_nrt_pk_tab_build) and the same
//    second pass for the pods it lists.  "total - k" (k = requested weighted slots = sum of weights) as an unsigned integer drops the
//    zones whose score floor(total / k) is 0 out of the minimum, as before;
//  * the per-item constants of the Score (requested weighted slots, 2^15 / k rounded up, -float32(Value(request)), the table slot's raw
//    request) are packed by k_nrt_fused_pack into 8-12 dwords per item instead of rewritten in LDS by every block.
//
// The engine launches it when the sweep is a whole batch over a row list with a rank stream (pod classes, or every row), the strategy is
// LeastAllocated with the packed Score's preconditions (nrt_packed_score) and unit weights; every other case keeps the two launches.
// Output: the same two tables, byte for byte (tests/test_gpu_nrt.py::test_fused_sweep_equals_two_launches, the every-cell tests).
#include <hip/hip_runtime.h>

#include "spx_internal.h"
#include "nrt_fast_device.h"
#include "nrt_rank_device.h"

namespace spx {

namespace {

using namespace nrtdev;

// what a launch of the walk computes: the Filter alone, or the Filter and the Score of one of the two strategies whose zone totals chain
constexpr int kFzFilter = 0, kFzLeast = 1, kFzMost = 2, kFzBalanced = 3;
constexpr float kFzMagic = 12582912.0f;          // 1.5 * 2^23: integers m with |m| < 2^22 are exact around it, bits = kFzMagicBits + m
constexpr uint32_t kFzMagicBits = 0x4b400000u;
constexpr int kFzItems = 1 + kC;                  // the pod-level request, then the containers
constexpr float kFzBalBand = 3e-4f;               // BalancedAllocation: |float32 minimum - float64 value| < 2.3e-4 (kBalBand, kernels_nrt_fast.hip)
constexpr uint32_t kFzBalRedo = 255u;             // score byte of a cell k_nrt_bal_redo recomputes (kBalRedo)

// a packed Score item: nv[RM] (float32: -Value(request)); requested weighted slots | ceil(2^15 / their count k) << 8; the bits of
// 1.5 * 2^23 minus k; the table slot's request (float64)
template <int RM>
constexpr int fz_item_words() { return RM + 4; }
template <int RM>
constexpr int fz_pod_words() { return kFzItems * fz_item_words<RM>(); }

// one thread per (list position, item)
template <int RM>
__global__ __launch_bounds__(256) void k_nrt_fused_pack(NrtArgs a, uint32_t* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t i = idx / kFzItems;
  const int j = static_cast<int>(idx - i * kFzItems);
  if (i >= a.n_list) return;
  const int64_t row = a.row_list ? a.row_list[i] : i;
  const uint32_t* w = a.pod_items + row * pod_words<RM>() + (1 + j) * item_words<RM>();
  uint32_t wmask = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r)
    if (r < a.n_res && a.slot_weight[r] != 0) wmask |= 1u << r;
  auto f64 = [&](int at) { return __hiloint2double(static_cast<int>(w[at + 1]), static_cast<int>(w[at])); };
  uint32_t* o = out + idx * fz_item_words<RM>();
  const bool balanced = a.strategy == SPX_NRT_BALANCED_ALLOCATION;  // (weights do not enter: balanced_allocation.go:27-46 iterates the requested resources)
  const uint32_t used = w[2 * RM] & 0xffu & (balanced ? 0xffu : wmask);
  // -Value(request) (LeastAllocated: t = 99.5 + o - v b) or +Value(request) (MostAllocated: t = -0.5 + o + v b; BalancedAllocation: the fraction v / c); -inf for a slot that is
  // not requested or weighs nothing: its chain, run because another lane's item needs it or because the walk only knows the unweighted
  // slot set, adds clamp01(-inf) = 0 (b >= 0; MostAllocated's b is 0 without capacity: NaN, clamped to 0 as well)
  const float sign = a.strategy == SPX_NRT_LEAST_ALLOCATED ? -1.0f : 1.0f;
#pragma unroll
  for (int r = 0; r < RM; ++r)
    o[r] = ((used >> r) & 1u) ? __float_as_uint(sign * static_cast<float>(f64(r == a.cpu_slot ? 2 * RM + 2 : 2 * r))) : 0xff800000u;
  const uint32_t k = static_cast<uint32_t>(__builtin_popcount(used));
  if (balanced) {
    // 1 / n and 100 / (n - 1) (as RN32(100 * RN32(1 / (n - 1)))) for the variance over the item's n fractions; fewer than two: the
    // reference's variance is NaN and every zone scores 0 — NaN here, which leaves every zone out of the minimum.  A requested slot
    // whose Value() is 0 travels as 1e-30: its fraction is 0 on a zone with capacity (1e-30 * rcp, lost in every sum) and 1 on a zone
    // without (1e-30 * inf under the clamp; 0 * inf would be NaN)
#pragma unroll
    for (int r = 0; r < RM; ++r)
      if (((used >> r) & 1u) && f64(r == a.cpu_slot ? 2 * RM + 2 : 2 * r) == 0.0) o[r] = __float_as_uint(1e-30f);
    const float c100rm = k >= 2 ? 100.0f * (1.0f / static_cast<float>(k - 1)) : __builtin_nanf("");
    o[RM] = __float_as_uint(k ? 1.0f / static_cast<float>(k) : 0.0f);
    o[RM + 1] = __float_as_uint(c100rm);
    o[RM + 2] = o[RM + 3] = 0u;
    return;
  }
  o[RM] = used | ((k ? (32768u + k - 1u) / k : 0u) << 8);
  o[RM + 1] = kFzMagicBits - k;
  const double raw = a.pk_tab_slot >= 0 ? f64(2 * a.pk_tab_slot) : 0.0;
  o[RM + 2] = static_cast<uint32_t>(__double2loint(raw));
  o[RM + 3] = static_cast<uint32_t>(__double2hiint(raw));
}

// ---------------------------------------------------------------- per-window sorted cell values (the walk's block start, four slots)
// A block of the walk needs, for each of its node's 32 cells, how many entries of the chunk's list the cell's quantity reaches.  Rounds
// 4-6a searched the list per cell (7 dependent LDS steps x 32 cells per lane: 15 % of the walk's vector instructions).  The cells of a
// window do not change from chunk to chunk, so their ORDER is computed once per node table: S[w][r] = the window's 2 048 quantities of
// slot r, sorted; rank[node][z][r] = the cell's index in it.  A block then searches S once per LIST ENTRY (<= 128 per slot: two per
// thread), drops the results into a histogram over the 2 049 positions — one dword per position, a byte per slot: a list has at most
// 127 real entries, so no byte overflows and ONE prefix sum serves the four slots — and every cell reads its count at its rank:
//   entry L lands at T = #{cells < L};  cell of rank i (ties in any order) has  L <= cell  <=>  T <= i;  count = prefix[i].
constexpr int kWsCells = kWindow * kZ;  // 2 048 quantities per (window, slot)

__global__ __launch_bounds__(1024) void k_nrt_window_sort(NrtArgs a, double* __restrict__ wsort, uint16_t* __restrict__ wrank) {
  __shared__ double key[kWsCells];
  __shared__ uint16_t idx[kWsCells];
  const int R = a.n_res;
  const int w = static_cast<int>(blockIdx.x) / 4, r = static_cast<int>(blockIdx.x) & 3;
  for (int i = threadIdx.x; i < kWsCells; i += 1024) {
    const int32_t pn = a.perm[static_cast<int64_t>(w) * kWindow + (i >> 3)];
    // an empty slot of the window and a cell the zone does not report lie below every list entry (the lists start at 0): count 0, as before
    key[i] = (pn >= 0 && r < R) ? a.f_av[(static_cast<int64_t>(i & 7) * R + r) * a.n_nodes + pn] : -2.0;
    idx[i] = static_cast<uint16_t>(i);
  }
  __syncthreads();
  for (int k = 2; k <= kWsCells; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < kWsCells; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const double x = key[i], y = key[l];
          if ((x > y) == up) {
            key[i] = y, key[l] = x;
            const uint16_t t = idx[i];
            idx[i] = idx[l], idx[l] = t;
          }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < kWsCells; i += 1024) {
    wsort[(static_cast<int64_t>(w) * 4 + r) * kWsCells + i] = key[i];
    const int cell = idx[i];
    const int32_t pn = a.perm[static_cast<int64_t>(w) * kWindow + (cell >> 3)];
    if (pn >= 0) wrank[static_cast<int64_t>(pn) * 32 + (cell & 7) * 4 + r] = static_cast<uint16_t>(i);
  }
}

// a comparison vector's thresholds as fetched from LDS (one per slot, replicated into the layout's fields)
template <int RM>
struct FzThr {
  uint32_t t[RM];
};
template <int RM>
__device__ __forceinline__ FzThr<RM> fz_load_thr(const uint32_t* thr) {
  FzThr<RM> g;
  const u32x4 v = *reinterpret_cast<const u32x4*>(thr);
  g.t[0] = v.x, g.t[1] = v.y, g.t[2] = v.z, g.t[3] = v.w;
  if constexpr (RM == 8) {
    const u32x4 u = *reinterpret_cast<const u32x4*>(thr + 4);
    g.t[4] = u.x, g.t[5] = u.y, g.t[6] = u.z, g.t[7] = u.w;
  }
  return g;
}

// the 8 zones' verdicts for one comparison vector, all RM slots, no tests: guard bits of m[]
template <int RM, bool NARROW>
__device__ __forceinline__ void fz_mask(const uint32_t (&qa)[RM][RkLayout<NARROW>::W], const FzThr<RM>& g, uint32_t (&m)[RkLayout<NARROW>::W]) {
  using L = RkLayout<NARROW>;
#pragma unroll
  for (int j = 0; j < L::W; ++j) {
    uint32_t x = qa[0][j] - g.t[0];
#pragma unroll
    for (int r = 1; r < RM; ++r) x &= qa[r][j] - g.t[r];
    m[j] = x & L::G;
  }
}

// ... and the per-slot words with it: bit 8 (z & 3) + 7 of xs[r][z >> 2] says "zone z holds the item's request of slot r" — MostAllocated's
// `requested.Cmp(capacity) > 0 -> 0` (most_allocated.go:49-51) read off the Filter's subtraction instead of a compare of its own
template <int RM>
__device__ __forceinline__ void fz_mask_slots(const uint32_t (&qa)[RM][2], const FzThr<RM>& g, uint32_t (&m)[2], uint32_t (&xs)[RM][2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int r = 0; r < RM; ++r) xs[r][j] = qa[r][j] - g.t[r];
    uint32_t x = xs[0][j];
#pragma unroll
    for (int r = 1; r < RM; ++r) x &= xs[r][j];
    m[j] = x & RkLayout<true>::G;
  }
}

// the lowest zone of m (guard bits only: zones 0-3 in dword 0, 4-7 in dword 1) as a one-zone set; all zero when m is empty.  Seven
// full-rate instructions (lowest_zone's gather / spread multiplications are v_mul_lo_u32: quarter rate)
__device__ __forceinline__ void fz_lowest(const uint32_t (&m)[2], uint32_t (&z)[2]) {
  const uint32_t n0 = 0u - m[0], n1 = 0u - m[1];
  const uint32_t any0 = static_cast<uint32_t>(static_cast<int32_t>(m[0] | n0) >> 31);  // all ones when dword 0 holds a zone
  z[0] = m[0] & n0;
  z[1] = m[1] & n1 & ~any0;
}

template <int W>
__device__ __forceinline__ bool fz_none(const uint32_t (&m)[W]) {
  uint32_t o = m[0];
#pragma unroll
  for (int j = 1; j < W; ++j) o |= m[j];
  return o == 0u;
}

// s = clamp01(fma(nv, b, c0)) — the median with 0 and 1 becomes the fma's clamp output modifier (NaN -> 0: the code object runs with
// DX10_CLAMP set) — then acc = fma(s, -128, acc): `v_fma_f32 .. clamp` + `v_fmac_f32` per zone, both at the full rate
__device__ __forceinline__ void fz_chain(float (&acc)[kZ], float nv, const float (&b)[kZ], float c0) {
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    const float s = __builtin_amdgcn_fmed3f(__builtin_fmaf(nv, b[z], c0), 0.0f, 1.0f);
    acc[z] = __builtin_fmaf(s, -128.0f, acc[z]);
  }
}

// MostAllocated's chain: the same two instructions around a mask — the resource score counts only where the zone holds the request
// (fit: the slot's word of fz_mask_slots; v_bfe_i32 spreads the zone's bit over a dword, v_and clears s)
__device__ __forceinline__ void fz_chain_most(float (&acc)[kZ], float pv, const float (&b)[kZ], float c0, const uint32_t (&fit)[2]) {
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    const float s = __builtin_amdgcn_fmed3f(__builtin_fmaf(pv, b[z], c0), 0.0f, 1.0f);
    const uint32_t keep = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(fit[z >> 2]), 8 * (z & 3) + 7, 1));
    acc[z] = __builtin_fmaf(__uint_as_float(__float_as_uint(s) & keep), -128.0f, acc[z]);
  }
}

// an item's registers as fetched from LDS (requested at the top of a container's step, ahead of the Filter's work on it)
template <int RM>
struct FzItem {
  float nv[RM];
  uint32_t w0, w1;
};
template <int RM>
__device__ __forceinline__ FzItem<RM> fz_load_item(const uint32_t* it) {
  FzItem<RM> g;
  const u32x4 v = *reinterpret_cast<const u32x4*>(it);
  g.nv[0] = __uint_as_float(v.x), g.nv[1] = __uint_as_float(v.y), g.nv[2] = __uint_as_float(v.z), g.nv[3] = __uint_as_float(v.w);
  if constexpr (RM == 8) {
    const u32x4 u = *reinterpret_cast<const u32x4*>(it + 4);
    g.nv[4] = __uint_as_float(u.x), g.nv[5] = __uint_as_float(u.y), g.nv[6] = __uint_as_float(u.z), g.nv[7] = __uint_as_float(u.w);
  }
  const u32x2 hw = *reinterpret_cast<const u32x2*>(it + RM);
  g.w0 = hw.x, g.w1 = hw.y;
  return g;
}

// The score of one request item on the lane's node: floor(min over the zones that score of the total / k), 0 when none does
// (scoreForEachNUMANode score.go:110-124 over leastAllocatedScoreStrategy least_allocated.go:25-43, unit weights).  bs[r][z] =
// RN32(RN64(100 / capacity)) / 128 (+inf without capacity), c0s[r] = (99.5 + o_r) / 128.  MIXED (second pass): the table slot's resource
// scores come from the float64 form (bt[z] = RN64(100 / capacity), raw = the request as written), as score_least_packed<.., true>.
template <int RM, bool MIXED, bool FIRST1, bool MOST>
__device__ __forceinline__ uint32_t fz_score_item(const float (&bs)[RM][kZ], const float (&c0s)[RM], int ts, uint32_t slots, const FzItem<RM>& g,
                                                  const uint32_t (&xs)[RM][2], const uint32_t* it = nullptr, const double* __restrict__ bt = nullptr) {
  auto chain = [&](float (&acc)[kZ], int r) {
    if constexpr (MOST) fz_chain_most(acc, g.nv[r], bs[r], c0s[r], xs[r]);
    else fz_chain(acc, g.nv[r], bs[r], c0s[r]);
  };
  // `slots` (wave-uniform): the chains to run — every slot some lane at work requests (a lane's item holds -inf for the others).  What
  // depends on the item's OWN slot count k stays per lane: g.w1 = the bits of 1.5 * 2^23 minus k, g.w0 >> 8 = ceil(2^15 / k) (0: k = 0)
  float acc[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) acc[z] = kFzMagic;
  // One chain per slot, each behind a scalar test.  The table slot first: its t may tie, and only onto the even start value does
  // "round the sum to nearest even" equal "round t" — the table slot is slot 0 or 1 (launch_nrt_fused), and FIRST1, the launch's choice
  // of instantiation, says which of the two is chained first (slot 1 unless the table slot is slot 0)
  constexpr int A = FIRST1 ? 1 : 0, B = FIRST1 ? 0 : 1;
  if ((slots >> A) & 1u) {
    SPX_KEEP_BRANCH();
    if constexpr (MIXED) {
      if (ts == A) {  // uniform: the float64 form of this slot's resource scores (a lane that does not request it: raw = 0 against +inf... see below)
        const double raw = __hiloint2double(static_cast<int>(it[RM + 3]), static_cast<int>(it[RM + 2]));
        const bool mine = __float_as_uint(g.nv[A]) != 0xff800000u;
#pragma unroll
        for (int z = 0; z < kZ; ++z) {
          uint32_t rs;  // as score_each_fast's float64 forms (nrt_fast_device.h); bt = RN64(100 / capacity), +inf (Least) or 0 (Most) without capacity
          if constexpr (MOST) rs = ((xs[A][z >> 2] >> (8 * (z & 3) + 7)) & 1u) ? static_cast<uint32_t>((raw * (1.0 + 0x1p-49)) * bt[z]) : 0u;
          else rs = static_cast<uint32_t>(__builtin_fma(-raw, bt[z], 100.0 + 0x1p-43));
          acc[z] = mine ? kFzMagic - static_cast<float>(rs) : kFzMagic;
        }
      } else {
        chain(acc, A);
      }
    } else {
      chain(acc, A);
    }
  }
  if ((slots >> B) & 1u) {
    SPX_KEEP_BRANCH();
    chain(acc, B);
  }
#pragma unroll
  for (int r = 2; r < RM; ++r) {
    if (!((slots >> r) & 1u)) continue;  // uniform
    SPX_KEEP_BRANCH();
    chain(acc, r);
  }
  // u = total - k as an unsigned integer: a zone whose total is below k (score 0) wraps to the top and leaves the minimum
  const uint32_t top = g.w1;
  uint32_t u[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) u[z] = top - __float_as_uint(acc[z]);
  auto min3 = [](uint32_t x, uint32_t y, uint32_t w) { return min(min(x, y), w); };
  const uint32_t mm = min3(min3(u[0], u[1], u[2]), min3(u[3], u[4], u[5]), min(u[6], u[7]));
  // (no zone scores: mm + k wraps to the smallest total, below k, and the quotient is the reference's 0; k = 0: the multiplier is 0)
  return __umul24(mm + (kFzMagicBits - top), g.w0 >> 8) >> 15;  // floor(total / k): total <= 800, k <= 8, ceil(2^15 / k)
}

// BalancedAllocation's per-node constants next to the reciprocals (bs), four registers:
//   qc   the cpu slot's counts once more, of 1000 x Value(capacity) — "request > capacity" is decided on rounded-up cores for cpu
//        (fractionOfCapacity divides Value() by Value()), and ceil(request / 1000) <= cores  <=>  request <= 1000 cores: the same
//        subtraction of the request's rank, on another count.  All ones where the capacity is not positive (fraction 1, never
//        exceeded), 0 past the node's zones;
//   zc   the node's zones whose capacity of slot r is not positive (fraction 1: the zone stays in whatever is asked): bit 7 - r of the
//        zone's byte in the count layout, so that `zc << r` lines slot r's bits up with the guard bits.
struct FzBal {
  uint32_t qc[2];
  uint32_t zc[2];
  int cpu_slot;
};

// BalancedAllocation (balanced_allocation.go:27-54; gonum stat.Variance) of one request item on the lane's node, as score_balanced_f32
// (kernels_nrt_fast.hip) computes it — the same float32 operations on the fractions and the variance, hence the same error bound
// (2.3e-4 against the band of 3e-4; tests/test_exactness_arguments.py) — except for what decides "request > capacity":
//   * every slot but cpu: Value() is the quantity itself, so fraction > 1 is the Filter's own comparison — the guard bit of the item's
//     subtraction (xs, as MostAllocated reads it), exact, where the float32 form had to leave near-equal byte counts undecided;
//   * cpu: the same subtraction on the counts of 1000 x whole cores (FzBal::qc; the walk puts it in xs[cpu slot]).
// A zone without capacity for a slot contributes fraction 1 (rcp = +inf under the clamp; zc keeps the zone) and the fraction of a
// request that fits is clamped to 1 (RN32(v) * RN32(1 / c) can exceed it by an ulp when v == c).  Returns the truncated minimum over the
// zones that score (0: none); *redo |= the minimum is within the band of an integer.
template <int RM>
__device__ __forceinline__ uint32_t fz_score_item_bal(const float (&rcp)[RM][kZ], const FzBal& bal, uint32_t slots, const FzItem<RM>& g,
                                                      const uint32_t (&xs)[RM][2], uint32_t* redo) {
  float sum[kZ], sq[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) sum[z] = 0.0f, sq[z] = 0.0f;
  uint32_t ok[2] = {~0u, ~0u};  // (only the guard bits are read)
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((slots >> r) & 1u)) continue;  // uniform: a slot some lane at work requests (the others hold -inf: fraction 0)
    SPX_KEEP_BRANCH();
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      const float f = __builtin_amdgcn_fmed3f(g.nv[r] * rcp[r][z], 0.0f, 1.0f);
      sum[z] += f;
      sq[z] = __builtin_fmaf(f, f, sq[z]);
    }
    ok[0] &= xs[r][0] | (bal.zc[0] << r);
    ok[1] &= xs[r][1] | (bal.zc[1] << r);
  }
  // 100 (1 - var), var = (sq - sum^2 / n) / (n - 1): g.w0 = 1 / n, g.w1 = 100 / (n - 1) (NaN for n < 2: no zone scores)
  const float rn = __uint_as_float(g.w0), c100rm = -__uint_as_float(g.w1);
  const uint32_t out[2] = {~ok[0], ~ok[1]};
  // The minimum over the zones that are in, on the values' bit patterns: a zone's score is in [50, 100] (a positive float32 orders as
  // its bits), a zone that is out — some slot's guard bit cleared — becomes all ones, and a NaN
  // (n < 2) lies above +inf either way: v_min3_u32, no canonicalisation of NaN operands as the float minimum needs
  uint32_t u[kZ];
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    const float sc = __builtin_fmaf(__builtin_fmaf(-(sum[z] * sum[z]), rn, sq[z]), c100rm, 100.0f);
    u[z] = __float_as_uint(sc) | static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(out[z >> 2]), 8 * (z & 3) + 7, 1));
  }
  auto min3 = [](uint32_t x, uint32_t y, uint32_t w) { return min(min(x, y), w); };
  const uint32_t bu = min3(min3(u[0], u[1], u[2]), min3(u[3], u[4], u[5]), min(u[6], u[7]));
  const bool has = bu < 0x7f800000u;
  const float best = __uint_as_float(bu);
  const float fl = __builtin_floorf(best), frac = best - fl;
  *redo |= (has && (frac < kFzBalBand || frac > 1.0f - kFzBalBand)) ? 1u : 0u;
  return has ? static_cast<uint32_t>(static_cast<int>(fl)) : 0u;
}

// The pod loop.  One count layout: four zones per register (every chunk of the stream is narrow — the engine splits a chunk whose lists
// would pass 127 entries — or the fused sweep is not launched).
//
// One walk for both scopes: a pod-scope node (singleNUMAPodLevelHandler filter.go:165-184, podScopeScore score.go:142-150) is a
// container-scope node whose pod has ONE container — the pod-level request (comparison vector 0, Score item 0), misfit status "cannot
// align pod", mean over one.  Step c of the loop serves the container-scope lanes' container c and, at c = 0, the pod-scope lanes' pod-level
// item: a lane reads ITS vector and item (two addresses per wave at most), so a wave that holds both kinds of node — one in four: the
// engine orders a window's nodes by scope, and the boundary falls inside a wave — pays the longer path instead of the sum of the two.
// What steers scalar branches — which slots' Score chains run — is the union of the slot sets of the items at work (from the pod's head);
// an item holds -inf for a slot it does not request, whose chain then adds nothing, and the constants that follow from its own slot
// count (2^15 / k, 1.5 * 2^23 - k) are read per lane.
template <int RM, bool FIRST1, int MODE, bool NARROW = true>
__device__ __forceinline__ void fz_walk(const uint32_t (&q4)[RM][2], const float (&bs)[RM][kZ], const float (&c0s)[RM], int ts, const uint32_t* pods,
                                        const uint32_t* sitems, int rows, int lane, bool w_pod, bool w_ctr, bool aligned, bool pod_scope, uint32_t st_stale,
                                        bool in, int pos, uint32_t absent_bits, uint32_t* stage_status, uint32_t* stage_score, const FzBal& bal) {
  using L = RkLayout<NARROW>;
  constexpr int W = L::W;
  constexpr int PWR = kRkPodHead + kRkVectors * RM;
  constexpr bool SCORE = MODE != kFzFilter, MOST = MODE == kFzMost, BAL = MODE == kFzBalanced;
  constexpr bool FITS = MOST || BAL;  // the Score reads per-slot fit bits off the Filter's subtractions
  uint32_t qa[RM][W];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int j = 0; j < W; ++j) qa[r][j] = q4[r][j];
  // a lane's first step: vector / item 0 (the pod-level request) for a pod-scope node, 1 (container 0) otherwise — byte offsets into the record
  const uint32_t thr0 = (kRkPodHead + (pod_scope ? 0 : RM)) * 4u, item0 = (pod_scope ? 0 : fz_item_words<RM>()) * 4u;
  uint32_t acc_status = 0, acc_score = 0;
  uint32_t hv = pods[lane & 15];  // a pod's head: lane l holds dword l & 15; the fields become scalars as they are needed
  for (int p = 0; p < rows; ++p) {
    const uint32_t* rec = pods + p * PWR;
    const uint32_t* sit = sitems + p * fz_pod_words<RM>();
    const uint32_t hcur = hv;
    hv = pods[(p + 1 < rows ? p + 1 : p) * PWR + (lane & 15)];  // the next pod's head, requested before this pod's work
    auto head = [&](int i) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(hcur), i)); };
    const uint32_t w0 = head(0);
    const uint32_t qos = w0 & 0xffu;
    const bool filtered = !(qos == SPX_QOS_BESTEFFORT && ((w0 >> 8) & 0xffu) == 0);  // filter.go:186-190
    const bool scored = qos == SPX_QOS_GUARANTEED;                                  // score.go:72-76: every other pod scores 100
    uint32_t status = filtered ? st_stale : 0u;
    uint32_t score = scored ? 0u : 100u;
    const int n_ctr = static_cast<int>((w0 >> 16) & 0xffu);
    const int n_steps = w_ctr && n_ctr > 0 ? n_ctr : (w_pod ? 1 : 0);
    if (filtered && aligned && n_steps > 0) {  // (filtered, n_steps: uniform; a Guaranteed pod is always filtered)
      uint32_t ops = head(12);
      const uint32_t ops_hi = head(13);
      uint32_t z0[W], z1[W];
#pragma unroll
      for (int j = 0; j < W; ++j) z0[j] = z1[j] = 0u;  // the zones app containers a0 / a1 were charged to (packed one-zone sets)
      uint32_t sum = 0, redo = 0;  // (redo, BalancedAllocation: some item's minimum is too close to an integer for float32)
      // the next step's thresholds and Score item are in flight while this one is worked on
      FzThr<RM> t = fz_load_thr<RM>(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(rec) + thr0));
      FzItem<RM> g{};
      if constexpr (SCORE) g = fz_load_item<RM>(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(sit) + item0));
      for (int c = 0; c < n_steps; ++c) {  // (left to the compiler: it peels the first steps, 5 % faster than `unroll 1`)
        const uint32_t op = ops & 0xffu;
        ops = c == 3 ? ops_hi : ops >> 8;
        const int cn = c + 1 < kC ? c + 1 : c;
        const FzThr<RM> tn = fz_load_thr<RM>(rec + kRkPodHead + (1 + cn) * RM);
        FzItem<RM> gn{};
        if constexpr (SCORE) gn = fz_load_item<RM>(sit + (1 + cn) * fz_item_words<RM>());
        // who takes this step: container-scope lanes while the pod has containers, pod-scope lanes the first one; the Score chains of
        // every slot either kind of item requests (head dwords 2 / 3 + c: the items' slot sets)
        const bool mine = pod_scope ? c == 0 : c < n_ctr;
        uint32_t slots = 0;
        if constexpr (SCORE) {
          slots = (w_ctr && c < n_ctr) ? head(3 + c) & 0xffu : 0u;
          if (w_pod && c == 0) slots |= head(2) & 0xffu;
        }
        if (mine) {
          uint32_t m[W], xs[RM][2];
          if constexpr (FITS) {
            fz_mask_slots<RM>(qa, t, m, xs);
            if constexpr (BAL) {  // the Score's cpu comparison is on whole cores (FzBal::qc); the Filter's verdict m keeps the quantities'
#pragma unroll
              for (int r = 0; r < RM; ++r)
                if (r == bal.cpu_slot) xs[r][0] = bal.qc[0] - t.t[r], xs[r][1] = bal.qc[1] - t.t[r];  // uniform
            }
          } else {
            fz_mask<RM, NARROW>(qa, t, m);
#pragma unroll
            for (int r = 0; r < RM; ++r) xs[r][0] = xs[r][1] = 0u;
          }
          if (op & (kRkOpMerge1 | kRkOpMerge3)) {  // uniform: an earlier app container may have been charged to a zone (c >= 1)
            if (op & kRkOpMerge1) {
              uint32_t ms[W];
              fz_mask<RM, NARROW>(qa, fz_load_thr<RM>(rec + kRkPodHead + 9 * RM), ms);
#pragma unroll
              for (int j = 0; j < W; ++j) m[j] = (m[j] & ~z0[j]) | (ms[j] & z0[j]);
            } else {  // the third app container: its own vector, + a0, + a1, + both
              uint32_t m0[W], m1[W], mb[W];
              fz_mask<RM, NARROW>(qa, fz_load_thr<RM>(rec + kRkPodHead + 10 * RM), m0);
              fz_mask<RM, NARROW>(qa, fz_load_thr<RM>(rec + kRkPodHead + 11 * RM), m1);
              fz_mask<RM, NARROW>(qa, fz_load_thr<RM>(rec + kRkPodHead + 12 * RM), mb);
#pragma unroll
              for (int j = 0; j < W; ++j) {
                const uint32_t both = z0[j] & z1[j];
                uint32_t x = (m[j] & ~z1[j]) | (m1[j] & z1[j]);
                x = (x & ~z0[j]) | (m0[j] & z0[j]);
                m[j] = (x & ~both) | (mb[j] & both);
              }
            }
          }
          if constexpr (FITS) {
            // MostAllocated / BalancedAllocation read "the zone holds the request" off the counts, so they stay the zones' own: a compared slot the node does not
            // report at node level (filter.go:101-104: the Filter fails, the Score does not care) is tested here instead of zeroing its counts
            const uint32_t sc = head(3 + c), sp = head(2);
            const uint32_t need_c = ((sc >> 8) | (sc >> 16)) & 0xffu, need_p = ((sp >> 8) | (sp >> 16)) & 0xffu;
            const uint32_t need = (w_pod && c == 0 && pod_scope) ? need_p : need_c;
            if ((need & absent_bits) != 0u) m[0] = m[1] = 0u;
          }
          // the first misfit names the status (a later one finds it set); an empty verdict has no lowest zone: nothing is charged
          const uint32_t code = (w_pod && c == 0 && pod_scope) ? static_cast<uint32_t>(SPX_NRT_ST_POD) : (op & 7u);
          if (fz_none(m) && status == 0u) status = code;
          if (op & (kRkOpCharge0 | kRkOpCharge1)) {  // uniform (a pod-scope lane's copy is never read)
            uint32_t z[W];
            fz_lowest(m, z);
#pragma unroll
            for (int j = 0; j < W; ++j) {
              if (op & kRkOpCharge0) z0[j] = z[j];
              else z1[j] = z[j];
            }
          }
          if constexpr (BAL) {
            if (scored) sum += fz_score_item_bal<RM>(bs, bal, slots, g, xs, &redo);
          } else if constexpr (SCORE) {
            if (scored) sum += fz_score_item<RM, false, FIRST1, MOST>(bs, c0s, ts, slots, g, xs);
          }
        }
        t = tn;
        g = gn;
      }
      // int64(mean): sum / n_ctr, sum <= 800; head 1 = ceil(2^16 / n_ctr); a pod-scope lane's one item divides by one
      if (scored) score = (sum * (pod_scope ? 65536u : head(1))) >> 16;
      if constexpr (BAL) score = (scored && redo) ? kFzBalRedo : score;  // k_nrt_bal_redo recomputes the cell in float64
    }
    const int sh = 8 * (p & 3);
    acc_status |= status << sh;
    acc_score |= score << sh;
    if ((p & 3) == 3 || p + 1 == rows) {  // uniform: one LDS write per table and four pods
      if (in) {
        stage_status[(p >> 2) * kWindow + pos] = acc_status;
        if constexpr (SCORE) stage_score[(p >> 2) * kWindow + pos] = acc_score;
      }
      acc_status = acc_score = 0;
    }
  }
}

// dynamic LDS: the chunk block of the rank stream (header, lists, pod records), the chunk's packed Score items, then the staged
// status and score dwords [2][kPodsPerUnit / 4][kWindow]
// SCORE = false: the Filter alone (status table only) — what the two-launch forms (another strategy, other weights) run before their Score
// launch: the same walk without the Score's tables, items and chains
template <int RM, bool FIRST1, int MODE>
__global__ __launch_bounds__(256, RM == 4 ? 4 : 2) void k_nrt_fused(NrtArgs a, const uint32_t* __restrict__ fz_items, int n_tiles) {
  constexpr bool SCORE = MODE != kFzFilter, MOST = MODE == kFzMost, BAL = MODE == kFzBalanced, FITS = MOST || BAL;
  extern __shared__ __align__(16) uint32_t lds[];
  __shared__ uint32_t pk_flagged;  // the chunk's pods with a table-slot request k_nrt_pk_tab_build lists for this window
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_windows = n_tiles;
  int window;
  int64_t chunk;
  if (n_windows >= kXcdMapWindows) {  // as k_nrt_fast: every XCD walks its own node windows
    const int wpx = (n_windows + 7) >> 3;
    const int64_t seq = blockIdx.x >> 3;
    window = static_cast<int>(blockIdx.x & 7u) + 8 * static_cast<int>(seq % wpx);
    chunk = seq / wpx;
    if (window >= n_windows) return;
  } else {
    // Fewer windows (the node tables fit every XCD's L2): the blocks of one chunk — all its windows — run on ONE XCD, one after the
    // other, so that the chunk's rank block and Score items (20 KB) come into that L2 once instead of into all eight (counter traffic
    // of config #3: 151 MB of fetch for 273 MB written before, most of it these streams eight times over)
    const int64_t seq = blockIdx.x >> 3;
    window = static_cast<int>(seq % n_windows);
    chunk = (seq / n_windows) * 8 + static_cast<int64_t>(blockIdx.x & 7u);
  }
  if (chunk >= a.rk_chunks) return;
  const int64_t first = a.rk_first[chunk];  // a chunk: up to 32 consecutive rows of the list
  const int rows = static_cast<int>(a.rk_first[chunk + 1] - first);
  auto row_of = [&](int p) -> int64_t { return a.row_list ? static_cast<int64_t>(uload(a.row_list + first + p)) : first + p; };
  const int64_t base = static_cast<int64_t>(window) * kWindow;
  const int32_t pn = a.perm[base + threadIdx.x];
  const bool in = pn >= 0;
  const int64_t n = in ? pn : 0;
  const int pos = in ? static_cast<int>(n - base) : 0;
  const int R = a.n_res;
  const int ts = a.pk_tab_slot;

  // ---- the chunk block and the chunk's Score items -> LDS (coalesced 16-byte pieces), the stage zeroed
  const uint32_t c0 = a.rk_off[chunk], c1 = a.rk_off[chunk + 1];
  uint32_t* const sitems = lds + a.rk_max_dwords;
  uint32_t* const stage = sitems + (SCORE ? kPodsPerUnit * fz_pod_words<RM>() : 0);  // [2][kPodsPerUnit / 4][kWindow] (SCORE: else one table)
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.rk_stream + c0);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    const int n_quads = static_cast<int>((c1 - c0) >> 2);
    for (int i = threadIdx.x; i < n_quads; i += 256) dst[i] = src[i];
    if constexpr (SCORE) {
      const uint4* isrc = reinterpret_cast<const uint4*>(fz_items + first * fz_pod_words<RM>());
      uint4* idst = reinterpret_cast<uint4*>(sitems);
      const int i_quads = rows * fz_pod_words<RM>() / 4;
      for (int i = threadIdx.x; i < i_quads; i += 256) idst[i] = isrc[i];
    }
    uint4* z = reinterpret_cast<uint4*>(stage) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < (SCORE ? 2 : 1) * kPodsPerUnit / 4 * kWindow / 4 / 256; ++i) z[i * 256] = uint4{0, 0, 0, 0};
    if (threadIdx.x == 0) pk_flagged = 0;
  }
  const uint32_t flags = in ? a.flags[n] : 0u;
  const uint32_t node_present = in ? a.node_present[n] : 0u;
  const uint32_t nn = static_cast<uint32_t>(a.n_nodes), n32 = static_cast<uint32_t>(n);
  // host-level resources no zone of the node reports are not checked (filter.go:110-116): their counts become all ones
  uint32_t fill_bits = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    const uint32_t rep = (in && r < R) ? ld_off(a.f_rep, static_cast<uint32_t>(r) * nn + n32) : 0u;
    fill_bits |= (r < R && (a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL) && rep == 0) ? 1u << r : 0u;
  }
  // the Score's multipliers, float32 / 128 (the zone table's other image)
  float bs[RM][kZ];
  float c0s[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    c0s[r] = ((MOST ? -0.5f : 99.5f) + (r == ts ? kPkOffsetTab : kPkOffsetSmall)) * 0x1p-7f;
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      bs[r][z] = 0.0f;
      if constexpr (BAL) {
        // RN32(RN64(1 / Value(capacity))); +inf where the capacity is not positive: the clamped fraction of any request is then 1
        const uint32_t at = (static_cast<uint32_t>(z * R + r) * nn + n32) * 8u;
        const bool cap = in && r < R && ld_off(a.f_av, at) > 0.0;
        bs[r][z] = cap ? static_cast<float>(ld_off(a.f_rcv, at)) : __builtin_inff();
      } else if constexpr (SCORE) {
        const double b = (in && r < R) ? ld_off(a.f_rc, (static_cast<uint32_t>(z * R + r) * nn + n32) * 8u) : kNoCap;
        bs[r][z] = b == kNoCap ? (MOST ? 0.0f : __builtin_inff()) : static_cast<float>(b) * 0x1p-7f;  // no capacity: the resource scores 0
      }
    }
  }
  FzBal bal{};
  bal.cpu_slot = a.cpu_slot;
  double cpu_q[kZ];  // BalancedAllocation: 1000 x Value() of the zones' cpu capacity (1e300: not positive; -1: past the node's zones)
  if constexpr (BAL) {
    const int nz = in ? a.n_zones[n] : 0;
    const uint32_t cs = static_cast<uint32_t>(a.cpu_slot >= 0 ? a.cpu_slot : 0);
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      const bool cap = in && a.cpu_slot >= 0 && ld_off(a.f_av, (static_cast<uint32_t>(z * R) + cs) * nn * 8u + n32 * 8u) > 0.0;
      cpu_q[z] = z < nz ? (cap ? 1000.0 * ld_off(a.f_cpu, (static_cast<uint32_t>(z) * nn + n32) * 8u) : 1e300) : -1.0;  // (1e300: above every entry, below the lists' +inf padding)
    }
    bal.zc[0] = bal.zc[1] = 0u;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
#pragma unroll
      for (int z = 0; z < kZ; ++z) {
        const bool none = in && r < R && r != a.cpu_slot && z < nz && !(ld_off(a.f_av, (static_cast<uint32_t>(z * R + r) * nn + n32) * 8u) > 0.0);
        bal.zc[z >> 2] |= none ? 0x80u >> r << (8 * (z & 3)) : 0u;
      }
    }
  }
  __syncthreads();
  // the pods whose table-slot request (any of their items) is listed for THIS node window: recomputed after the loop
  if (SCORE && ts >= 0) {
    for (int i = threadIdx.x; i < rows * kFzItems; i += 256) {
      const uint32_t* it = sitems + i * fz_item_words<RM>();
      if (!((it[RM] >> ts) & 1u)) continue;
      const double k = __hiloint2double(static_cast<int>(it[RM + 3]), static_cast<int>(it[RM + 2])) * a.pk_tab_inv_unit;
      // (the engine derived unit and kmax from this very batch: k is a whole number within the table; anything else is recomputed too)
      const bool inside = k >= 0.0 && k <= static_cast<double>(a.pk_tab_kmax) && k == __builtin_floor(k);
      const uint32_t word = inside ? a.pk_tab[static_cast<size_t>(static_cast<uint32_t>(k)) * a.pk_tab_words + (static_cast<uint32_t>(window) >> 5)] : ~0u;
      if ((word >> (static_cast<uint32_t>(window) & 31u)) & 1u) atomicOr(&pk_flagged, 1u << (i / kFzItems));
    }
  }
  // ---- the node's cells as counts: for each resource the eight zones' quantities, ranked against the chunk's list
  uint32_t q4[RM][2];
  const double* lists = reinterpret_cast<const double*>(lds + 16);
  uint32_t list_doubles = 0;
  bool counted = false;
  if constexpr (RM == 4) {
    if (a.wsort != nullptr) {  // uniform: the window's cells are sorted (k_nrt_window_sort): one search per list entry, one lookup per cell
      counted = true;
      uint32_t* const hist = stage;  // [kWsCells + 1] dwords, a byte per slot; the stage is not in use yet (zeroed above, and again below)
      uint32_t lo_of[RM], len_of[RM];
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const uint32_t hw = r < R ? lds[r] : 0u;
        lo_of[r] = hw >> 8, len_of[r] = r < R ? 1u << (hw & 0xffu) : 0u;
        if (r < R) list_doubles = lo_of[r] + len_of[r];
      }
      {  // thread t: entries (t & 63) and (t & 63) + 64 of slot t >> 6
        const int r = static_cast<int>(threadIdx.x >> 6);
        uint32_t lo_r = 0, len_r = 0;
#pragma unroll
        for (int q = 0; q < RM; ++q)
          if (q == r) lo_r = lo_of[q], len_r = len_of[q];
        const double* S = a.wsort + (static_cast<int64_t>(window) * 4 + r) * kWsCells;
        const uint32_t k0 = threadIdx.x & 63u, k1 = k0 + 64u;
        const bool on0 = k0 < len_r, on1 = k1 < len_r;
        const double l0 = on0 ? lists[lo_r + k0] : 0.0, l1 = on1 ? lists[lo_r + k1] : 0.0;
        // lower bound — how many of the window's quantities lie below the entry — as one binary step and five quaternary ones: six dependent
        // round trips to the L2 instead of eleven (the block start is latency, not instructions)
        static_assert(kWsCells == 2048, "1024 + 3 * (256 + 64 + 16 + 4 + 1) = 2047");
        uint32_t p0 = S[1023] < l0 ? 1024u : 0u, p1 = S[1023] < l1 ? 1024u : 0u;
#pragma unroll
        for (uint32_t q = 256; q > 0; q >>= 2) {
          const double a0 = S[p0 + q - 1], b0 = S[p0 + 2 * q - 1], c0v = S[p0 + 3 * q - 1];
          const double a1 = S[p1 + q - 1], b1 = S[p1 + 2 * q - 1], c1v = S[p1 + 3 * q - 1];
          p0 += ((a0 < l0 ? 1u : 0u) + (b0 < l0 ? 1u : 0u) + (c0v < l0 ? 1u : 0u)) * q;
          p1 += ((a1 < l1 ? 1u : 0u) + (b1 < l1 ? 1u : 0u) + (c1v < l1 ? 1u : 0u)) * q;
        }
        // (the search covers 2 047 positions; one more compare for the last)
        p0 = (p0 == kWsCells - 1 && S[kWsCells - 1] < l0) ? kWsCells : p0;
        p1 = (p1 == kWsCells - 1 && S[kWsCells - 1] < l1) ? kWsCells : p1;
        if (on0) atomicAdd(&hist[p0], 1u << (8 * r));
        if (on1) atomicAdd(&hist[p1], 1u << (8 * r));
      }
      __syncthreads();
      {  // inclusive prefix over the positions: eight per thread, then the threads' totals (no byte overflows: <= 127 real entries per slot)
        u32x4 h0 = *reinterpret_cast<const u32x4*>(hist + threadIdx.x * 8), h1 = *reinterpret_cast<const u32x4*>(hist + threadIdx.x * 8 + 4);
        h0.y += h0.x, h0.z += h0.y, h0.w += h0.z;
        h1.x += h0.w, h1.y += h1.x, h1.z += h1.y, h1.w += h1.z;
        uint32_t incl = h1.w;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t up = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), d));
          incl += lane >= d ? up : 0u;
        }
        __shared__ uint32_t wave_sum[4];
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t before = incl - h1.w;
#pragma unroll
        for (int wv = 0; wv < 3; ++wv) before += wv < wave ? wave_sum[wv] : 0u;
        h0.x += before, h0.y += before, h0.z += before, h0.w += before;
        h1.x += before, h1.y += before, h1.z += before, h1.w += before;
        *reinterpret_cast<u32x4*>(hist + threadIdx.x * 8) = h0;
        *reinterpret_cast<u32x4*>(hist + threadIdx.x * 8 + 4) = h1;
      }
      __syncthreads();
      uint32_t cnt4[kZ];  // per zone: the four slots' counts, a byte each
      {
        const u32x4* rk = reinterpret_cast<const u32x4*>(a.wrank + n * 32);
        const u32x4 r0 = in ? rk[0] : u32x4{0, 0, 0, 0}, r1 = in ? rk[1] : u32x4{0, 0, 0, 0}, r2 = in ? rk[2] : u32x4{0, 0, 0, 0},
                    r3 = in ? rk[3] : u32x4{0, 0, 0, 0};
        const uint32_t rw[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
#pragma unroll
        for (int z = 0; z < kZ; ++z) {
          // rank of (z, r): halfword z * 4 + r;  the count of slot r sits in byte r of the prefix dword at that rank
          const uint32_t i0 = rw[2 * z] & 0xffffu, i1 = rw[2 * z] >> 16, i2 = rw[2 * z + 1] & 0xffffu, i3 = rw[2 * z + 1] >> 16;
          const uint32_t c0v = hist[i0], c1v = hist[i1], c2v = hist[i2], c3v = hist[i3];
          cnt4[z] = in ? (c0v & 0xffu) | (c1v & 0xff00u) | (c2v & 0xff0000u) | (c3v & 0xff000000u) : 0u;
        }
      }
      __syncthreads();  // every lookup is done: the histogram's memory becomes the stage again
      {
        uint4* zz = reinterpret_cast<uint4*>(stage) + threadIdx.x;
#pragma unroll
        for (int i = 0; i < (kWsCells + 8) / 4 / 256 + 1; ++i)
          if (i * 256 + static_cast<int>(threadIdx.x) < (SCORE ? 2 : 1) * kPodsPerUnit / 4 * kWindow / 4) zz[i * 256] = uint4{0, 0, 0, 0};
      }
      __syncthreads();  // (a lane's first staged dword must not meet another lane's zeroing)
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const bool absent = !FITS && !((node_present >> r) & 1u), fill = (fill_bits >> r) & 1u;
        uint32_t lo4 = RkLayout<true>::G, hi4 = RkLayout<true>::G;
#pragma unroll
        for (int z = 0; z < 4; ++z) lo4 |= ((cnt4[z] >> (8 * r)) & 0xffu) << (8 * z), hi4 |= ((cnt4[z + 4] >> (8 * r)) & 0xffu) << (8 * z);
        q4[r][0] = (r >= R || absent) ? RkLayout<true>::G : (fill ? ~0u : lo4);
        q4[r][1] = (r >= R || absent) ? RkLayout<true>::G : (fill ? ~0u : hi4);
      }
    }
  }
  if (!counted) {
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    q4[r][0] = q4[r][1] = RkLayout<true>::G;  // (slots past the table: never requested)
    if (r >= R) continue;  // uniform
    const uint32_t hw = lds[r];
    const int steps = static_cast<int>(hw & 0xffu);
    const uint32_t lo = hw >> 8;
    list_doubles = lo + (1u << steps);
    double av[kZ];
#pragma unroll
    for (int z = 0; z < kZ; ++z) av[z] = in ? ld_off(a.f_av, (static_cast<uint32_t>(z * R + r) * nn + n32) * 8u) : -1.0;
    uint32_t cnt[kZ];
#pragma unroll
    for (int z = 0; z < kZ; ++z) cnt[z] = 0;
    for (int b = 1 << (steps - 1); b > 0; b >>= 1) {  // uniform trip count; the eight searches advance together
#pragma unroll
      for (int z = 0; z < kZ; ++z) {
        const double v = lists[lo + cnt[z] + static_cast<uint32_t>(b) - 1u];
        cnt[z] = v <= av[z] ? cnt[z] + static_cast<uint32_t>(b) : cnt[z];
      }
    }
    // a compared resource the node does not report at node level fails whatever the zones say (filter.go:101-104): count 0 (a slot that
    // is not compared subtracts 0 and passes); a host-level resource no zone reports passes whatever is asked: all ones
    const bool absent = !FITS && !((node_present >> r) & 1u), fill = (fill_bits >> r) & 1u;  // (Most / BalancedAllocation: fz_walk tests node-level absence itself)
    const uint32_t lo4 = RkLayout<true>::G | cnt[0] | (cnt[1] << 8) | (cnt[2] << 16) | (cnt[3] << 24);
    const uint32_t hi4 = RkLayout<true>::G | cnt[4] | (cnt[5] << 8) | (cnt[6] << 16) | (cnt[7] << 24);
    q4[r][0] = absent ? RkLayout<true>::G : (fill ? ~0u : lo4);
    q4[r][1] = absent ? RkLayout<true>::G : (fill ? ~0u : hi4);
  }
  }
  if constexpr (BAL) {
    bal.qc[0] = bal.qc[1] = ~0u;  // (no cpu slot: nothing is subtracted from it)
    if (a.cpu_slot >= 0) {  // uniform: the eight quantities ranked against the chunk's cpu list, as the block start without sorted windows does
      const uint32_t hw = lds[a.cpu_slot];
      const int steps = static_cast<int>(hw & 0xffu);
      const uint32_t lo = hw >> 8;
      uint32_t cnt[kZ];
#pragma unroll
      for (int z = 0; z < kZ; ++z) cnt[z] = 0;
      for (int b = 1 << (steps - 1); b > 0; b >>= 1) {
#pragma unroll
        for (int z = 0; z < kZ; ++z) {
          const double v = lists[lo + cnt[z] + static_cast<uint32_t>(b) - 1u];
          cnt[z] = v <= cpu_q[z] ? cnt[z] + static_cast<uint32_t>(b) : cnt[z];
        }
      }
      bal.qc[0] = RkLayout<true>::G | cnt[0] | (cnt[1] << 8) | (cnt[2] << 16) | (cnt[3] << 24);
      bal.qc[1] = RkLayout<true>::G | cnt[4] | (cnt[5] << 8) | (cnt[6] << 16) | (cnt[7] << 24);
    }
  }
  const uint32_t* const pods = lds + 16 + 2 * list_doubles;
  const bool fresh = flags & SPX_NRT_F_FRESH;
  const bool has_nrt = flags & SPX_NRT_F_HAS_NRT;
  const bool single = flags & SPX_NRT_F_SINGLE_NUMA;
  const bool pod_scope = flags & SPX_NRT_F_POD_SCOPE;
  const bool aligned = fresh && has_nrt && single;
  const bool w_pod = __ballot(aligned && pod_scope) != 0, w_ctr = __ballot(aligned && !pod_scope) != 0;
  const uint32_t st_stale = fresh ? 0u : static_cast<uint32_t>(SPX_NRT_ST_INVALID_TOPOLOGY);
  uint32_t* const stage_status = stage;
  uint32_t* const stage_score = stage + kPodsPerUnit / 4 * kWindow;
  uint32_t absent_bits = 0;  // slots the node does not report at node level
#pragma unroll
  for (int r = 0; r < RM; ++r) absent_bits |= (r < R && !((node_present >> r) & 1u)) ? 1u << r : 0u;
  fz_walk<RM, FIRST1, MODE>(q4, bs, c0s, ts, pods, sitems, rows, lane, w_pod, w_ctr, aligned, pod_scope, st_stale, in, pos, absent_bits, stage_status, stage_score, bal);
  __syncthreads();
  if constexpr (BAL) {
    // the cells left to the float64 form (score byte 255), listed for k_nrt_bal_redo: ~1 % of config #3's cells.  A thread counts the
    // marks of its node's staged dwords, the block takes ONE range of the global list (an atomic per marked cell on the one counter
    // made the launch 9.7 ms); past the list's capacity the count says so and k_nrt_bal_scan finds the marks in the table
    __shared__ uint32_t redo_n, redo_base;
    if (threadIdx.x == 0) redo_n = 0;
    __syncthreads();
    const int quads = (rows + 3) >> 2;
    uint32_t mine = 0;
    if (in)
      for (int q = 0; q < quads; ++q) {
        const uint32_t w4 = stage_score[q * kWindow + pos];
        // bytes equal to 0xff (scores are <= 100: only the mark has bit 7 set)
        mine += static_cast<uint32_t>(__builtin_popcount(w4 & 0x80808080u));
      }
    uint32_t at = mine ? atomicAdd(&redo_n, mine) : 0u;
    __syncthreads();
    if (redo_n != 0u) {  // block-uniform
      if (threadIdx.x == 0) redo_base = atomicAdd(a.redo_list, redo_n);
      __syncthreads();
      at += redo_base;
      if (mine)
        for (int q = 0; q < quads; ++q) {
          uint32_t w4 = stage_score[q * kWindow + pos] & 0x80808080u;
          for (; w4 != 0u; w4 &= w4 - 1u, ++at) {
            if (at >= a.redo_cap) continue;
            a.redo_list[2 + 2 * static_cast<size_t>(at)] = static_cast<uint32_t>(row_of(4 * q + (__builtin_ctz(w4) >> 3)));
            a.redo_list[3 + 2 * static_cast<size_t>(at)] = n32;
          }
        }
    }
  }
  if constexpr (SCORE && !BAL) {
    const uint32_t flagged = pk_flagged;  // block-uniform (every atomicOr precedes the barrier above)
    if (flagged != 0) {
      // second pass: the flagged pods' cells of this window with the table slot in the float64 form — its eight multipliers read
      // again, the other slots chained as before.  Each lane rewrites its own byte of the staged dword.
      double bt[kZ];
#pragma unroll
      for (int z = 0; z < kZ; ++z) {
        const double b = in ? a.f_rc[(static_cast<int64_t>(z) * R + ts) * a.n_nodes + n] : kNrtNoCap;
        bt[z] = b == kNrtNoCap ? (MOST ? 0.0 : __builtin_inf()) : b;
      }
      constexpr int PWR = kRkPodHead + kRkVectors * RM;
      uint32_t n_redone = 0;
      for (uint32_t left = flagged; left != 0; left &= left - 1) {
        const int p = __builtin_ctz(left);
        const uint32_t* rec = pods + p * PWR;
        const uint32_t h0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rec[0])));
        const uint32_t h1 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(rec[1])));
        if ((h0 & 0xffu) != SPX_QOS_GUARANTEED) continue;  // uniform: 100 everywhere, the loop wrote it
        ++n_redone;
        const int n_ctr = (h0 >> 16) & 0xffu;
        const uint32_t* sit = sitems + p * fz_pod_words<RM>();
        auto fit_words = [&](int vec, uint32_t (&xs)[RM][2]) {  // MostAllocated: which zones hold the item's requests (its own comparison vector)
          if constexpr (MOST) {
            uint32_t m[2];
            fz_mask_slots<RM>(q4, fz_load_thr<RM>(rec + kRkPodHead + vec * RM), m, xs);
          } else {
#pragma unroll
            for (int r = 0; r < RM; ++r) xs[r][0] = xs[r][1] = 0u;
          }
        };
        uint32_t score = 0;
        if (aligned) {
          if (pod_scope) {
            uint32_t xs[RM][2];
            fit_words(0, xs);
            score = fz_score_item<RM, true, FIRST1, MOST>(bs, c0s, ts, sit[RM] & 0xffu, fz_load_item<RM>(sit), xs, sit, bt);
          } else {
            uint32_t sum = 0;
#pragma unroll 1
            for (int c = 0; c < n_ctr; ++c) {
              const uint32_t* it = sit + (1 + c) * fz_item_words<RM>();
              uint32_t xs[RM][2];
              fit_words(1 + c, xs);
              sum += fz_score_item<RM, true, FIRST1, MOST>(bs, c0s, ts, it[RM] & 0xffu, fz_load_item<RM>(it), xs, it, bt);
            }
            score = (sum * h1) >> 16;
          }
        }
        if (in) {
          const int sh = 8 * (p & 3);
          uint32_t& cell = stage_score[(p >> 2) * kWindow + pos];
          cell = (cell & ~(0xffu << sh)) | (score << sh);
        }
      }
      if (a.stats && threadIdx.x == 0 && n_redone)  // these cells count as re-evaluated (spx_fetch_stats)
        atomicAdd(a.stats + (SPX_PLUGIN_NRT * kStatSlots + static_cast<int>(blockIdx.x & (kStatSlots - 1))) * kStatStride, static_cast<unsigned long long>(n_redone) * kWindow);
      __syncthreads();
    }
  }
  // rows leave as whole 256-byte segments (as k_nrt_fast): lane l gathers byte (row & 3) of the four dwords of nodes 4l .. 4l+3
  const int64_t col = base + lane * 4;
  if (col < a.row_stride) {
    for (int i = wave; i < rows; i += 4) {
      const int64_t row = row_of(i);
      const uint32_t b = static_cast<uint32_t>(i & 3);
      const uint32_t pick = 0x0c0c0000u | ((4u + b) << 8) | b;
#pragma unroll
      for (int tbl = 0; tbl < (SCORE ? 2 : 1); ++tbl) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(&(tbl ? stage_score : stage_status)[(i >> 2) * kWindow + lane * 4]);
        const uint32_t lo = __builtin_amdgcn_perm(w.y, w.x, pick), hi = __builtin_amdgcn_perm(w.w, w.z, pick);
        uint8_t* out = (tbl ? a.out_score : a.out_status) + row * a.row_stride + col;
        *reinterpret_cast<uint32_t*>(out) = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
    }
  }
}

}  // namespace

// The Filter launch of a two-launch sweep (another strategy, weighted slots) as the fused walk without its Score: false = not launched
// (no narrow rank stream, or the chunk block does not fit next to the stage)
bool launch_nrt_filter_fused(const NrtArgs& a, hipStream_t s) {
  if (!a.fast || !a.rk_all_narrow || !a.rk_stream || !a.rk_off || !a.rk_first || a.rk_max_dwords == 0 || !a.out_status || a.out_raw || a.row_ptr) return false;
  const int n_tiles = static_cast<int>((a.n_nodes + kWindow - 1) / kWindow);
  const int64_t chunks = a.rk_chunks;
  const unsigned blocks = static_cast<unsigned>(n_tiles >= kXcdMapWindows ? chunks * (((n_tiles + 7) / 8) * 8) : ((chunks + 7) / 8) * 8 * n_tiles);
  const size_t lds = static_cast<size_t>(a.rk_max_dwords) * 4 + static_cast<size_t>(kPodsPerUnit / 4) * kWindow * 4 + 64;  // (+ the histogram's last position)
  if (lds > 64 * 1024) return false;
  if (a.n_res <= 4) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nrt_fused<4, true, kFzFilter>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL((k_nrt_fused<4, true, kFzFilter>), dim3(blocks), dim3(256), lds, s, a, static_cast<const uint32_t*>(nullptr), n_tiles);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nrt_fused<8, true, kFzFilter>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL((k_nrt_fused<8, true, kFzFilter>), dim3(blocks), dim3(256), lds, s, a, static_cast<const uint32_t*>(nullptr), n_tiles);
  }
  return true;
}

// the per-window sorted quantities and the cells' ranks (k_nrt_window_sort): wsort [windows][4][2048] doubles, wrank [nodes][32] halfwords
size_t nrt_window_sort_bytes(int64_t n_nodes, size_t* rank_bytes) {
  *rank_bytes = static_cast<size_t>(n_nodes) * 32 * sizeof(uint16_t);
  return static_cast<size_t>((n_nodes + kWindow - 1) / kWindow) * 4 * kWsCells * sizeof(double);
}
void launch_nrt_window_sort(const NrtArgs& a, double* wsort, uint16_t* wrank, hipStream_t s) {
  const unsigned windows = static_cast<unsigned>((a.n_nodes + kWindow - 1) / kWindow);
  hipLaunchKernelGGL(k_nrt_window_sort, dim3(windows * 4), dim3(1024), 0, s, a, wsort, wrank);
}

// words of scratch the packed Score items of `n_list` rows take (NrtArgs::fz_items)
size_t nrt_fused_item_words(int n_res, int64_t n_list) {
  return static_cast<size_t>(n_list) * static_cast<size_t>(n_res <= 4 ? fz_pod_words<4>() : fz_pod_words<8>());
}

// The whole-batch sweep in one launch; false = not launched (the caller runs the Filter and Score launches): no rank stream, a
// strategy or weights the chain does not cover, or a chunk block that does not fit in LDS next to the items and the stage.
bool launch_nrt_fused(const NrtArgs& a, hipStream_t s) {
  if (!a.fast || !a.fz_items || !a.rk_stream || !a.rk_off || !a.rk_first || a.rk_max_dwords == 0 || !a.out_status || !a.out_score || a.out_raw || a.row_ptr) return false;
  const bool most = a.strategy == SPX_NRT_MOST_ALLOCATED, balanced = a.strategy == SPX_NRT_BALANCED_ALLOCATION;
  if (balanced) {
    // the float32 form needs the cpu slot's whole cores exact in float32 and somewhere to list the cells it leaves to float64
    if (!a.redo_list || a.redo_cap == 0 || !a.f_rcv || !a.f_cpu || (a.cpu_slot >= 0 && !((a.exact32_slots >> a.cpu_slot) & 1u))) return false;
  } else if ((a.strategy != SPX_NRT_LEAST_ALLOCATED && !most) || !a.pk_mode || a.pk_tab_slot > 1) {
    return false;  // (the table slot is chained first: slot 0 or 1)
  }
  // five to eight resource slots: the Score's multipliers alone are 64 registers — 192 with the rest, two waves per SIMD, 2.25 ms for the
  // six-slot config #3 against 1.76 for the Filter-only walk + the packed Score launch (measured): those tables take the two launches
  // (BalancedAllocation's six-slot Score launch is the slow one — 6.6 ms at two waves per SIMD —: there the one-launch form at two waves wins)
  if (a.n_res > 4 && !balanced) return false;
  const bool wide = a.n_res > 4;
  for (int r = 0; r < a.n_res && !balanced; ++r)
    if (a.slot_weight[r] != 0 && a.slot_weight[r] != 1) return false;
  const int n_tiles = static_cast<int>((a.n_nodes + kWindow - 1) / kWindow);
  const int64_t chunks = a.rk_chunks;
  // the kernel's block map: 8 XCDs x their windows per chunk, or 8 chunks (one per XCD) x all windows
  const unsigned blocks = static_cast<unsigned>(n_tiles >= kXcdMapWindows ? chunks * (((n_tiles + 7) / 8) * 8) : ((chunks + 7) / 8) * 8 * n_tiles);
  const size_t lds = static_cast<size_t>(a.rk_max_dwords) * 4 + static_cast<size_t>(kPodsPerUnit) * (wide ? fz_pod_words<8>() : fz_pod_words<4>()) * 4 +
                     static_cast<size_t>(2) * (kPodsPerUnit / 4) * kWindow * 4 + 64;
  if (lds > 64 * 1024) return false;
  if (!balanced && a.pk_tab_slot >= 0 && !(a.pk_tab_built && *a.pk_tab_built)) {  // the table of the packed float32 Score (kernels_nrt_fast.hip)
    launch_nrt_pk_tab_build(a, n_tiles, s);
    if (a.pk_tab_built) *a.pk_tab_built = true;
  }
  const unsigned pack_blocks = static_cast<unsigned>((a.n_list * kFzItems + 255) / 256);
  // FIRST1: the two-slot fast path chains slot 1 before slot 0 — right whenever the table slot is not slot 0 (it is memory, or there is none)
  const bool first1 = a.pk_tab_slot != 0;
#define SPX_FZ_LAUNCH(RMV, F1V, MODEV)                                                                                                     \
  do {                                                                                                                                     \
    if (a.fz_pack) hipLaunchKernelGGL((k_nrt_fused_pack<RMV>), dim3(pack_blocks), dim3(256), 0, s, a, a.fz_items);                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nrt_fused<RMV, F1V, MODEV>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
    hipLaunchKernelGGL((k_nrt_fused<RMV, F1V, MODEV>), dim3(blocks), dim3(256), lds, s, a, a.fz_items, n_tiles);                          \
  } while (0)
  if (balanced) {
    NrtArgs b = a;
    b.pk_tab_slot = -1;  // (no table slot: the walk's second pass and its flags stay idle)
    (void)hipMemsetAsync(a.redo_list, 0, 8, s);
    if (wide) {
      if (a.fz_pack) hipLaunchKernelGGL((k_nrt_fused_pack<8>), dim3(pack_blocks), dim3(256), 0, s, b, a.fz_items);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nrt_fused<8, true, kFzBalanced>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      hipLaunchKernelGGL((k_nrt_fused<8, true, kFzBalanced>), dim3(blocks), dim3(256), lds, s, b, a.fz_items, n_tiles);
    } else {
      if (a.fz_pack) hipLaunchKernelGGL((k_nrt_fused_pack<4>), dim3(pack_blocks), dim3(256), 0, s, b, a.fz_items);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nrt_fused<4, true, kFzBalanced>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      hipLaunchKernelGGL((k_nrt_fused<4, true, kFzBalanced>), dim3(blocks), dim3(256), lds, s, b, a.fz_items, n_tiles);
    }
    launch_nrt_bal_fixups(a, s);  // the listed cells in float64 (kernels_nrt_fast.hip)
  } else if (most) {
    if (first1) SPX_FZ_LAUNCH(4, true, kFzMost);
    else SPX_FZ_LAUNCH(4, false, kFzMost);
  } else {
    if (first1) SPX_FZ_LAUNCH(4, true, kFzLeast);
    else SPX_FZ_LAUNCH(4, false, kFzLeast);
  }
#undef SPX_FZ_LAUNCH
  return true;
}

}  // namespace spx
