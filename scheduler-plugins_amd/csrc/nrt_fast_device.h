// nrt_fast_device.h — device-side pieces of the float64 NodeResourceTopologyMatch formulation shared by the sweep kernels
// (kernels_nrt_fast.hip) and the cooperative sequential-commit kernel (kernels_commit_coop.hip): a node's NUMA tables in
// registers, the pod record stream's items, the single-NUMA fit test, the container subtraction and the Least / Most /
// BalancedAllocation zone scores.  Derivations and exactness arguments: the header of kernels_nrt_fast.hip.
#pragma once

#include <cstdlib>
#include <utility>

#include "spx_internal.h"

namespace spx {
namespace nrtdev {

constexpr int kZ = SPX_NRT_MAX_ZONES;
constexpr int kC = SPX_NRT_MAX_CTRS;
constexpr int kSgLeast = 0;
constexpr int kSgMost = 1;
constexpr int kSgBalanced = 2;
constexpr int kSgLeastNuma = 3;
constexpr double kNoCap = kNrtNoCap;  // b[][] of a cell whose capacity is not positive
constexpr int kPodsPerUnit = 32;    // pod rows per block
constexpr int kWindow = 256;        // nodes per block (4 wavefronts)
constexpr int kXcdMapWindows = 32;  // from this many node windows on (8k nodes), blocks are mapped XCD-aware (see k_nrt_fast)

// Placed at the top of a block guarded by a wave-uniform condition (a requested-resource bit of the pod record): an empty
// volatile asm cannot be speculated, so the backend keeps the scalar branch and the wave skips the block.  Without it the
// optimiser may if-convert the short per-resource blocks — compute all RM resources and select — depending on code that has
// nothing to do with them: LeastAllocated's Score launch went from 1.5 to 2.1 ms that way when the BalancedAllocation
// fix-up kernel was added to this file.
#define SPX_KEEP_BRANCH() asm volatile("")

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

// A load at (wave-uniform base) + (per-lane 32-bit byte offset): the form the backend turns into saddr + voffset.  With
// 64-bit per-lane pointers the loop-invariant address arithmetic of the loads inside the pod loop was hoisted out of it and
// held 70 VGPRs of addresses for the whole kernel (the LeastNUMANodes table restore: 32 columns); callers there also make the node index opaque
// (opaque_lane) so that not even the 32-bit offsets are precomputed and parked in scratch.
__device__ __forceinline__ uint32_t opaque_lane(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}
template <typename T>
__device__ __forceinline__ T ld_off(const T* base, uint32_t byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// The pod record stream (built by the engine at upload, spx_engine.hip: nrt_pod_items): per pod 10 items of IW dwords
// (16 for <= 4 resource slots, else 32) — item 0 the header (2 dwords used), item 1 the pod-level request, items 2..9 the
// containers in order (init containers first).  A request item: doubles raw[RM] (dwords 0..2RM-1), the slot-set dword
// (2RM), a pad, then the three doubles only the Score reads: Value() of the cpu request, the sum of the weights of the
// requested slots and its biased reciprocal.
//
// Round 3: a block copies the records of its 32 pods into LDS once (coalesced 16-byte loads: ONE memory round trip per
// chunk) and every wave reads them from there with broadcast ds_reads.  Rounds 1-2 fetched them with scalar loads, one
// pod ahead: 32 MB of records (50k pods) do not stay in the 16 KB scalar cache nor in an XCD's L2, so every pod iteration
// of every wave waited ~650 ns for its s_loads — a batch of BestEffort pods, which the sweep has nothing to compute for,
// took 1.0 of the mixed batch's 2.8 ms (tools/r3/exp_qos.py).  The quantities now sit in VGPRs (the same value in every
// lane, read as VGPR operands); the slot-set dword and the header are made scalar (v_readfirstlane) because they steer
// wave-uniform branches.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kItemsPerPod = 2 + kC;
template <int RM>
constexpr int item_words() { return RM == 4 ? 16 : 32; }
template <int RM>
constexpr int pod_words() { return kItemsPerPod * item_words<RM>(); }

// an item's dwords as fetched: w[0 .. 2RM] always, the Score's tail (2RM+2 .. 2RM+7) when FULL
template <int RM, bool FULL>
struct ItemRegs {
  uint32_t w[FULL ? 2 * RM + 8 : 2 * RM + 1];
};

// `pod_rec`: the pod's record (in LDS for the sweep, in global memory for the per-cell fix-up); `slot`: 0 header, 1 pod-level
// request, 2.. containers
template <int RM, bool FULL>
__device__ __forceinline__ ItemRegs<RM, FULL> load_item(const uint32_t* pod_rec, int slot) {
  const u32x4* p = reinterpret_cast<const u32x4*>(pod_rec + slot * item_words<RM>());
  ItemRegs<RM, FULL> r;
  constexpr int kQuads = (FULL ? 2 * RM + 8 : 2 * RM) / 4;
#pragma unroll
  for (int q = 0; q < kQuads; ++q) {
    const u32x4 v = p[q];
    r.w[4 * q] = v.x, r.w[4 * q + 1] = v.y, r.w[4 * q + 2] = v.z, r.w[4 * q + 3] = v.w;
  }
  if constexpr (!FULL) r.w[2 * RM] = pod_rec[slot * item_words<RM>() + 2 * RM];
  return r;
}

template <int RM>
struct Item {
  double raw[RM];  // requests as written (cpu in millicores); 0 for absent slots
  double cpu_v;    // Quantity.Value() of the cpu request (whole cores, rounded up)
  double wsum;     // sum of the weights of the requested slots
  double wrc;      // its biased reciprocal
  uint32_t wsum_i; // the same sum as an integer (Least/MostAllocated accumulate integer zone totals)
  uint32_t used;   // requested slots (Score iterates these)
  uint32_t fit;    // non-zero requests compared per zone: available >= quantity
  uint32_t always; // non-zero requests of a non-Guaranteed pod for a NUMA-affine resource: any reporting zone suits
  uint32_t kind;   // SPX_CTR_*
};

// UNIFORM: every lane holds the same item (the sweep): the slot sets are made scalar so that the per-resource tests stay
// scalar branches.  The per-cell fix-up decodes a different item per lane.
template <int RM, bool FULL, bool UNIFORM = true>
__device__ __forceinline__ Item<RM> decode_item(const ItemRegs<RM, FULL>& g) {
  Item<RM> it;
  auto f64 = [&](int i) { return __hiloint2double(static_cast<int>(g.w[i + 1]), static_cast<int>(g.w[i])); };
#pragma unroll
  for (int r = 0; r < RM; ++r) it.raw[r] = f64(2 * r);
  const uint32_t s = UNIFORM ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(g.w[2 * RM]))) : g.w[2 * RM];
  it.used = s & 0xffu;
  it.fit = (s >> 8) & 0xffu;
  it.always = (s >> 16) & 0xffu;
  it.kind = s >> 24;
  if constexpr (FULL) {
    it.cpu_v = f64(2 * RM + 2);
    it.wsum = f64(2 * RM + 4);
    it.wrc = f64(2 * RM + 6);
    it.wsum_i = UNIFORM ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(g.w[2 * RM + 1]))) : g.w[2 * RM + 1];
  } else {
    it.cpu_v = it.wsum = it.wrc = 0.0;  // Score-only fields
    it.wsum_i = 0;
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
      rb = 0;  // zone 7 first: each compare's verdict is shifted in from the right (v_cmp + v_addc, no select/or)
      static_assert(kZ == 8, "one asm statement for the eight zones");
      asm("v_cmp_le_f64 vcc, %1, %9\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
          "v_cmp_le_f64 vcc, %1, %8\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
          "v_cmp_le_f64 vcc, %1, %7\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
          "v_cmp_le_f64 vcc, %1, %6\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
          "v_cmp_le_f64 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
          "v_cmp_le_f64 vcc, %1, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
          "v_cmp_le_f64 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
          "v_cmp_le_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
          : "+v"(rb)
          : "v"(it.raw[r]), "v"(ns.av[0][r]), "v"(ns.av[1][r]), "v"(ns.av[2][r]), "v"(ns.av[3][r]), "v"(ns.av[4][r]), "v"(ns.av[5][r]),
            "v"(ns.av[6][r]), "v"(ns.av[7][r])
          : "vcc");
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
    SPX_KEEP_BRANCH();
#pragma unroll
    for (int z = 0; z < kZ; ++z) ns.av[z][r] = __builtin_fma(sel[z], it.raw[r], ns.av[z][r]);
  }
}

// LeastAllocated's zone totals in PACKED float32 (round 5; the Score-only launch of a whole-batch sweep, NrtArgs::pk_mode).
// The Score launch issues one vector instruction per 4 cycles whatever its kind (VALU slots 95 % used, profiles/r04), so what counts
// is the NUMBER of instructions: the float64 form spends 3 per (zone, resource) — v_fma_f64, v_cvt_u32_f64, v_mad_u32_u24; this one
// 4 per zone PAIR — v_pk_fma_f32, 2 x v_cvt_pk_u8_f32, v_pk_mad_u16 — with the totals of two zones in the halves of one register.
//   t = fma(-v, b32, 99.5 + o),  b32 = RN32(RN64(100 / c));  rs = v_cvt_pk_u8_f32(t) = RNE(t) clamped to [0, 255], NaN -> 0
// (tools/micro/cvt_pk_u8.hip).  x = 100 - 100 v / c; t = x - 0.5 + o + e with |e| <= 5.96e-6 (b32's relative error 2^-24 on
// 100 v / c <= 100) + 3.8e-6 (the fma's rounding below 128) < 9.8e-6.  With x = n + f: RNE(t) = n whenever 0 < f + o + e < 1.
// The engine turns the form on only when EVERY weighted slot is of one of two kinds (nrt_packed_score, spx_engine.hip):
//   * small: with 2^s the largest power of two dividing every zone capacity and every request of the slot, c / 2^s <= 32768 and
//     v / 2^s < 2^24 (cpu in whole cores, devices, hugepages in pages, memory on clusters that report whole MiB): v is a float32 value
//     and x a multiple of 2^s / c, so f is 0 or lies in [3.05e-5, 1 - 3.05e-5]; with o = 2^-16 the condition always holds;
//   * the table slot (one at most: memory in bytes): o = 2^-17, and k_nrt_pk_tab_build REPLAYS this very formula for every request
//     value k * unit that lands within 2.7e-5 of an integer score for some zone, against the integer division; the values it gets
//     wrong (about 8 in 10^6 per zone) are listed per node window.  The block that stages a pod with a listed request for ITS window
//     recomputes that pod's 256 cells in the float64 form after the sweep of the chunk (k_nrt_fast's second pass, about 3 % of the
//     (pod, window) pairs of config #3) — the loop itself never branches on it.
// x < 0 (request above capacity) and cells without capacity (b32 = +inf: -inf, or NaN for an explicit zero request) convert to 0 —
// the reference's zeros.  Zone totals are u16: start at -sum(weights) mod 2^16 so that a zone whose score is 0 stays "negative" = at
// least 2^15 (a total below sum(weights): exactly the zones whose score total / sum(weights) is 0) and drops out of the unsigned minimum
// (100 * sum(weights) < 2^15: kNrtPkMaxWeightSum).  The float32 multipliers live as
// zone pairs in ns.b[0..3][r] (load_fast_node<.., PK>); ns.b[4..7][r] are unused.
constexpr float kPkOffsetSmall = 0x1p-16f, kPkOffsetTab = 0x1p-17f;

// one resource of one zone, exactly as the packed loop computes it (k_nrt_pk_tab_build replays the table slot with it)
__device__ __forceinline__ uint32_t least_packed_one(float v, float b32, float c0) {
  return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(-v, b32, c0), 0, 0u) & 0xffu;
}

// The request item as the packed loop reads it.  The block that staged the chunk's records rewrote every item IN PLACE
// (k_nrt_fast): nv[r] = -float32(Value(request r)) into the dwords the float64 Score keeps Value(cpu) and the float64 weight sum in
// (<= 4 slots: dwords 2RM+2 .. 2RM+5) or into the unused tail (8 slots: dwords 24 .. 31) — the raw float64 requests (dwords 0 .. 2RM-1)
// stay for the second pass.  The loop fetches dwords 2RM .. : slot sets, integer weight sum, nv[], the biased reciprocal.
template <int RM>
struct PkRegs {
  uint32_t w[RM == 4 ? 8 : 16];
};
template <int RM>
__device__ __forceinline__ constexpr int pk_nv_dword(int r) { return RM == 4 ? 2 * RM + 2 + r : 24 + r; }
template <int RM>
__device__ __forceinline__ PkRegs<RM> load_item_pk(const uint32_t* pod_rec, int slot) {
  const u32x4* p = reinterpret_cast<const u32x4*>(pod_rec + slot * item_words<RM>() + 2 * RM);
  PkRegs<RM> r;
#pragma unroll
  for (int q = 0; q < (RM == 4 ? 2 : 4); ++q) {
    const u32x4 v = p[q];
    r.w[4 * q] = v.x, r.w[4 * q + 1] = v.y, r.w[4 * q + 2] = v.z, r.w[4 * q + 3] = v.w;
  }
  return r;
}

// MIXED (k_nrt_fast's second pass over the pods the table lists for the block's window): the table slot in the float64 form from
// bt[zone] = RN64(100 / c) (+inf without capacity) and the raw request, every other slot packed as in the loop
template <int RM, bool MIXED = false>
__device__ __forceinline__ int score_least_packed(const FastNode<RM>& ns, const NrtArgs& a, const PkRegs<RM>& g, const double* __restrict__ bt = nullptr,
                                                  double raw_tab = 0.0) {
  typedef float F32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned short U16x2 __attribute__((ext_vector_type(2)));
  static_assert(kZ == 8, "four zone pairs");
  const uint32_t used = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(g.w[0]))) & 0xffu;
  const uint32_t wsum = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(g.w[1])));
  if (wsum == 0) return 0;  // no weighted slot requested: wave-uniform
  const double wrc = __hiloint2double(static_cast<int>(g.w[7]), static_cast<int>(g.w[6]));
  const unsigned short a0 = static_cast<unsigned short>(0u - wsum);
  U16x2 accp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) accp[k] = U16x2{a0, a0};
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (!((used >> r) & 1u)) continue;
    SPX_KEEP_BRANCH();
    const unsigned short w16 = static_cast<unsigned short>(a.slot_weight[r]);
    const U16x2 wv{w16, w16};
    if constexpr (MIXED) {
      if (r == a.pk_tab_slot) {  // uniform
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t r0 = static_cast<uint32_t>(__builtin_fma(-raw_tab, bt[2 * k], 100.0 + 0x1p-43));  // (as score_each_fast's float64 form)
          const uint32_t r1 = static_cast<uint32_t>(__builtin_fma(-raw_tab, bt[2 * k + 1], 100.0 + 0x1p-43));
          const uint32_t pk = r0 | (r1 << 16);
          U16x2 pv;
          __builtin_memcpy(&pv, &pk, 4);
          accp[k] = pv * wv + accp[k];
        }
        continue;
      }
    }
    const float nvf = __uint_as_float(g.w[pk_nv_dword<RM>(r) - 2 * RM]);
    const float c0f = 99.5f + (r == a.pk_tab_slot ? kPkOffsetTab : kPkOffsetSmall);  // scalar select
    const F32x2 nv{nvf, nvf}, c0{c0f, c0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      F32x2 bp;
      __builtin_memcpy(&bp, &ns.b[k][r], 8);
      const F32x2 t = __builtin_elementwise_fma(nv, bp, c0);
      uint32_t pk = __builtin_amdgcn_cvt_pk_u8_f32(t.x, 0, 0u);
      pk = __builtin_amdgcn_cvt_pk_u8_f32(t.y, 2, pk);
      U16x2 pv;
      __builtin_memcpy(&pv, &pk, 4);
      accp[k] = pv * wv + accp[k];
    }
  }
  const U16x2 mm = __builtin_elementwise_min(__builtin_elementwise_min(accp[0], accp[1]), __builtin_elementwise_min(accp[2], accp[3]));
  const unsigned short m16 = mm.x < mm.y ? mm.x : mm.y;
  // (m16 >= 2^15: no zone scores — every total is below sum(weights), m16 + sum(weights) wraps to the smallest of them and the division gives 0)
  return static_cast<int>(static_cast<double>(static_cast<unsigned short>(m16 + static_cast<unsigned short>(wsum))) * wrc);
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
        SPX_KEEP_BRANCH();
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
    // Zone totals as integers (round 3): a resource score t >= 0 is truncated AND clamped by ONE v_cvt_u32_f64 (it saturates
    // negatives, -inf and NaN to 0: the reference's "request exceeds capacity" / "no capacity" zeros), the weighted sum is a
    // v_mad_u32_u24 per resource (scores <= 100, weights below 2^20 — checked at upload), and the total starts at -sum(weights):
    // a zone whose score floor(total / sum(weights)) is 0 ends negative, i.e. huge as unsigned, and drops out of the unsigned
    // minimum — scoreForEachNUMANode's "minimum of the non-zero zone scores" is floor(min valid total / sum(weights)), one
    // division per item instead of one per zone.  3 instructions per (zone, resource) + 1 per zone; the float64 form took 4 + 4.
    const uint32_t wsum = it.wsum_i;
    if (wsum == 0) return 0;  // no weighted slot requested: wave-uniform
    double vq[RM];  // MostAllocated: the request pre-multiplied for the "request <= capacity" product test
#pragma unroll
    for (int r = 0; r < RM; ++r) vq[r] = SG == kSgMost ? value[r] * (1.0 + 0x1p-49) : 0.0;
    double rawq = 0.0;  // the cpu slot's request in millicores (a dynamic index would move the array to scratch)
    if constexpr (SG == kSgMost) {
#pragma unroll
      for (int r = 0; r < RM; ++r) rawq = r == a.cpu_slot ? it.raw[r] * (1.0 + 0x1p-49) : rawq;
    }
    // v_cvt_u32_f64 (what the conversion compiles to; an asm statement would cost an s_nop each) saturates: < 0, -inf, NaN -> 0
    auto cvt_u32 = [](double t) { return static_cast<uint32_t>(t); };
    // resource outside, zone inside: the requested-slot test is a scalar branch, and the CU's ONE scalar unit serves all four
    // SIMDs (a scalar instruction costs a SIMD the same issue slot as a vector one) — zone outside paid it 8 x RM times per item
    uint32_t acc[kZ];
#pragma unroll
    for (int z = 0; z < kZ; ++z) acc[z] = 0u - wsum;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      if (!((used >> r) & 1u)) continue;
      SPX_KEEP_BRANCH();
      const uint32_t w = static_cast<uint32_t>(a.slot_weight[r]);
#pragma unroll
      for (int z = 0; z < kZ; ++z) {
        uint32_t rs;
        if constexpr (SG == kSgLeast) {
          // (cap_v - req_v) * 100 / cap_v == 100 - req_v * (100 / cap_v); see the header for why the floor is exact.
          // Cells without capacity hold b = +inf: -v * inf is -inf (v > 0) or NaN (explicit zero request) -> 0
          rs = cvt_u32(__builtin_fma(-value[r], ns.b[z][r], 100.0 + 0x1p-43));
        } else {
          // req_v * 100 / cap_v, zero when the request exceeds the capacity.  "request <= capacity" is read off the same
          // kind of product instead of the mutable table (so MostAllocated, like LeastAllocated, scores from b alone):
          // t = (q * (1 + 2^-49)) * RN(100 / c) <= 100 * (1 + 2^-48)  <=>  q <= c   for integers q, c < 2^42
          // (q <= c gives t <= 100 * (1 + 1.2 * 2^-49); q >= c + 1 gives t >= 100 * (1 + 2^-42)).  The cpu slot compares the
          // raw millicore quantities (braw), the score uses whole cores (b).
          const double tp = vq[r] * ns.b[z][r];
          const double chk = r == a.cpu_slot ? rawq * braw[z] : tp;
          rs = chk <= 100.0 * (1.0 + 0x1p-48) ? cvt_u32(tp) : 0u;
        }
        acc[z] = __umul24(rs, w) + acc[z];
      }
    }
#pragma unroll
    for (int z = 0; z < kZ; ++z) m = acc[z] < m ? acc[z] : m;
    if (m >= 0x80000000u) return 0;  // no zone scores
    return static_cast<int>(static_cast<double>(m + wsum) * it.wrc);  // floor(total / sum(weights)) in 1..100
  }
  return static_cast<int>(m + 1u);
}

// the node's zone tables into registers (prologue of the sweep; the BalancedAllocation fix-up loads single nodes with it)
template <int RM, int SG, bool PK = false>
__device__ __forceinline__ void load_fast_node(const NrtArgs& a, int64_t n, bool in, FastNode<RM>& ns, double (&cpu_v)[kZ], double (&braw)[kZ]) {
  const int R = a.n_res;
  ns.nz = in ? a.n_zones[n] : 0;
  ns.node_present = in ? a.node_present[n] : 0u;
#pragma unroll
  for (int i = 0; i < RM / 4; ++i) ns.rep[i] = ns.fill[i] = 0;
  // element offsets fit 32 bits (Z * R * N < 2^28 for the node counts a device holds): one scalar multiply per column instead of a
  // 64-bit multiply-add chain — the prologue runs once per block of 32 pods, and scalar instructions are not free (one scalar unit per CU)
  const uint32_t nn = static_cast<uint32_t>(a.n_nodes), n32 = static_cast<uint32_t>(n);
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= R) continue;
    const uint32_t rep = in ? ld_off(a.f_rep, static_cast<uint32_t>(r) * nn + n32) : 0u;
    ns.rep[r >> 2] |= rep << (8 * (r & 3));
    if ((a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL) && rep == 0) ns.fill[r >> 2] |= 0xffu << (8 * (r & 3));
  }
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    const uint32_t zo = (static_cast<uint32_t>(z) * nn + n32) * 8u;
    cpu_v[z] = (SG == kSgBalanced && in && a.cpu_slot >= 0) ? ld_off(a.f_cpu, zo) : 0.0;
    braw[z] = (SG == kSgMost && in && a.cpu_slot >= 0) ? ld_off(a.f_braw, zo) : kNoCap;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const uint32_t i = (static_cast<uint32_t>(z * R + r) * nn + n32) * 8u;
      ns.av[z][r] = (in && r < R) ? ld_off(a.f_av, i) : -1.0;
      const double b = (SG != kSgBalanced && SG != kSgLeastNuma && in && r < R) ? ld_off(a.f_rc, i) : kNoCap;
      ns.b[z][r] = (SG == kSgLeast && b == kNoCap) ? __builtin_inf() : b;
      if constexpr (SG == kSgBalanced) ns.b[z][r] = (in && r < R) ? ld_off(a.f_rcv, i) : 1.0;  // RN(1 / Value(capacity)) for div_rn
    }
  }
  if constexpr (SG == kSgLeast && PK) {
    // packed float32 Score: every slot's multipliers as zone pairs in b[0..3][r] (score_least_packed; 1e200 "no capacity" -> +inf)
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      float f[kZ];
#pragma unroll
      for (int z = 0; z < kZ; ++z) f[z] = static_cast<float>(ns.b[z][r]);
#pragma unroll
      for (int k = 0; k < 4; ++k) ns.b[k][r] = __hiloint2double(__float_as_int(f[2 * k + 1]), __float_as_int(f[2 * k]));
#pragma unroll
      for (int k = 4; k < kZ; ++k) ns.b[k][r] = 0.0;
    }
  }
}

}  // namespace nrtdev
}  // namespace spx
