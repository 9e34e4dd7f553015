// kernels_nrt.hip — gfx950 kernel for NodeResourceTopologyMatch Filter + Score (NUMA-zone fit).
//
// Decomposition: lane = node, pod record = wave-uniform.  One wavefront owns 64 consecutive nodes and a
// chunk of pod rows.  A node's NUMA table (up to 8 zones x RM resources of int64 "available", presence
// bitmasks, NUMA ids) is loaded once into VGPRs — node columns are stored zone/resource-major
// ([z][r][node]) so that those loads are coalesced — and every pod of the chunk is then evaluated
// against it with the pod's container requests arriving through scalar (wave-uniform) loads, so the
// container/resource loops and the QoS branches are uniform; only per-node properties (scope, policy,
// zone count) diverge.  Container-scope Filter mutates the table exactly like the reference
// (subtractResourcesFromNUMANodeList) and undoes it before scoring; LeastNUMANodes' greedy subtraction
// is undone by re-reading the lane's table.
//
// This path is VALU-bound integer work (C x R x Z compares, subtractions, small divisions per cell),
// not HBM-bound: per (pod,node) it writes 2 bytes and reads nothing from HBM.
//
// Reference: pkg/noderesourcetopology/filter.go:42-258, score.go:62-191, least_numa.go:35-233,
// least_allocated.go, most_allocated.go, balanced_allocation.go, numaresources.go:105-215.
#include <cstdlib>

#include "spx_internal.h"

namespace spx {

namespace {

constexpr int kZ = SPX_NRT_MAX_ZONES;
constexpr int kC = SPX_NRT_MAX_CTRS;
constexpr int kPodsPerUnit = 16;
// strategy groups (one kernel instantiation each, so that a launch carries only the code it runs)
constexpr int kSgAlloc = 0;     // LeastAllocated / MostAllocated
constexpr int kSgBalanced = 1;  // BalancedAllocation
constexpr int kSgLeastNuma = 2; // LeastNUMANodes

// combin.Combinations(n, k) for n <= 8 as bitmasks over list positions, size-major then lexicographic
struct ComboTable {
  uint8_t mask[kZ][256];
  uint8_t start[kZ][kZ + 2];  // start[n-1][k-1] .. start[n-1][k]: subsets of size k
};

constexpr ComboTable make_combos() {
  ComboTable t{};
  for (int n = 1; n <= kZ; ++n) {
    int idx = 0;
    for (int k = 1; k <= n; ++k) {
      t.start[n - 1][k - 1] = static_cast<uint8_t>(idx);
      int c[kZ] = {};
      for (int i = 0; i < k; ++i) c[i] = i;
      while (true) {
        int m = 0;
        for (int i = 0; i < k; ++i) m |= 1 << c[i];
        t.mask[n - 1][idx++] = static_cast<uint8_t>(m);
        int i = k - 1;
        while (i >= 0 && c[i] == n - k + i) --i;
        if (i < 0) break;
        ++c[i];
        for (int j = i + 1; j < k; ++j) c[j] = c[j - 1] + 1;
      }
    }
    t.start[n - 1][n] = static_cast<uint8_t>(idx);  // 255 for n == 8
  }
  return t;
}

__constant__ ComboTable kCombo = make_combos();

template <int RM>
struct NodeState {
  int64_t avail[kZ][RM];
  uint32_t id_lo, id_hi;  // NUMA id per list position, 8 bits each
  uint32_t zp_lo, zp_hi;  // per-zone resource-presence bitmask, 8 bits each
  int nz;
  uint32_t flags;
  uint32_t node_present;
  int max_numa;
  __device__ __forceinline__ uint32_t id(int z) const { return ((z < 4 ? id_lo >> (8 * z) : id_hi >> (8 * (z - 4))) & 0xffu); }
  __device__ __forceinline__ uint32_t zp(int z) const { return ((z < 4 ? zp_lo >> (8 * z) : zp_hi >> (8 * (z - 4))) & 0xffu); }
};

template <int RM>
__device__ __forceinline__ void load_avail(NodeState<RM>& ns, const NrtArgs& a, int64_t n, bool in) {
#pragma unroll
  for (int z = 0; z < kZ; ++z)
#pragma unroll
    for (int r = 0; r < RM; ++r)
      ns.avail[z][r] = (in && r < a.n_res) ? a.zone_avail[(static_cast<int64_t>(z) * a.n_res + r) * a.n_nodes + n] : 0;
}

// resourcesAvailableInAnyNUMANodes filter.go:93-163.  req/present are wave-uniform.
template <int RM>
__device__ __forceinline__ bool fits_any(const NodeState<RM>& ns, const NrtArgs& a, bool non_guaranteed, uint32_t present,
                                         const int64_t* __restrict__ req, uint32_t* numa_id) {
  uint64_t bitmask = ~0ull;
  bool ok = true;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= a.n_res || !((present >> r) & 1u)) continue;  // uniform
    const int64_t q = req[r];
    if (q == 0) continue;                                   // uniform: "ignoring zero-qty resource request"
    const bool always = non_guaranteed && (a.slot_flags[r] & SPX_NRT_SLOT_AFFINE);  // isResourceSetSuitable, uniform
    const bool host_level = a.slot_flags[r] & SPX_NRT_SLOT_HOST_LEVEL;
    if (!((ns.node_present >> r) & 1u)) ok = false;  // not reported at node level -> cannot meet request
    bool has_affinity = false;
    uint64_t rb = 0;
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      const bool rep = z < ns.nz && ((ns.zp(z) >> r) & 1u);
      has_affinity |= rep;
      if (rep && (always || ns.avail[z][r] >= q)) rb |= 1ull << ns.id(z);
    }
    if (!(!has_affinity && host_level)) bitmask &= rb;
  }
  *numa_id = bitmask ? static_cast<uint32_t>(__builtin_ctzll(bitmask)) : 0u;
  return ok && bitmask != 0;
}

// subtractResourcesFromNUMANodeList numaresources.go:145-182 (sign = -1) and its exact inverse (+1)
template <int RM>
__device__ __forceinline__ void adjust_numa(NodeState<RM>& ns, const NrtArgs& a, bool non_guaranteed, uint32_t present,
                                            const int64_t* __restrict__ req, uint32_t numa_id, bool apply, int sign) {
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= a.n_res || !((present >> r) & 1u)) continue;
    if (non_guaranteed && (a.slot_flags[r] & SPX_NRT_SLOT_AFFINE)) continue;
    const int64_t q = req[r];
    if (q == 0) continue;
#pragma unroll
    for (int z = 0; z < kZ; ++z) {
      const bool hit = apply && z < ns.nz && ns.id(z) == numa_id && ((ns.zp(z) >> r) & 1u);
      ns.avail[z][r] += hit ? sign * q : 0;
    }
  }
}

// floor(num / den) for 0 <= num <= 101 * den (quotient <= 101): float estimate + exact fix-up
__device__ __forceinline__ int64_t div_le100(uint64_t num, uint64_t den) {
  const float qf = static_cast<float>(num) * __frcp_rn(static_cast<float>(den));
  uint64_t q = static_cast<uint64_t>(static_cast<uint32_t>(qf));
  const uint64_t prod = q * den;
  if (prod > num) --q;
  else if (num - prod >= den) ++q;
  return static_cast<int64_t>(q);
}

__device__ __forceinline__ int64_t value_of(bool is_cpu, int64_t q) {  // Quantity.Value(): cpu is in millicores
  return is_cpu ? (q + 999) / 1000 : q;
}

// one NUMA zone's strategy score (least/most: least_allocated.go:25-55, most_allocated.go:25-54;
// balanced: balanced_allocation.go:27-54); zero when the request set is empty (the reference panics)
template <int RM, int SG>
__device__ __forceinline__ int64_t zone_score(const NodeState<RM>& ns, const NrtArgs& a, int z, uint32_t present,
                                              const int64_t* __restrict__ req, uint64_t weight_sum) {
  if constexpr (SG == kSgBalanced) {
    double fr[RM];
    bool over = false;
    int n = 0;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      fr[r] = 0.0;
      if (r >= a.n_res || !((present >> r) & 1u)) continue;
      const bool is_cpu = a.slot_flags[r] & SPX_NRT_SLOT_CPU;
      const int64_t cap = ((ns.zp(z) >> r) & 1u) ? ns.avail[z][r] : 0;
      const int64_t cap_v = value_of(is_cpu, cap);
      const double f = cap_v == 0 ? 1.0 : static_cast<double>(value_of(is_cpu, req[r])) / static_cast<double>(cap_v);
      over |= f > 1.0;
      fr[r] = f;
      ++n;
    }
    if (over) return 0;
    // gonum stat.Variance (corrected two-pass, unbiased), fractions taken in ascending resource id
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < RM; ++r) sum += fr[r];  // absent slots hold +0.0: x + 0.0 == x
    const double mean = sum / static_cast<double>(n);
    double ss = 0.0, comp = 0.0;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const bool used = r < a.n_res && ((present >> r) & 1u);
      const double d = used ? fr[r] - mean : 0.0;
      ss += d * d;
      comp += d;
    }
    const double variance = (ss - comp * comp / static_cast<double>(n)) / (static_cast<double>(n) - 1.0);
    return static_cast<int64_t>((1.0 - variance) * 100.0);
  } else {
  const bool least = a.strategy == SPX_NRT_LEAST_ALLOCATED;
  uint64_t acc = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= a.n_res || !((present >> r) & 1u)) continue;
    const bool is_cpu = a.slot_flags[r] & SPX_NRT_SLOT_CPU;
    const int64_t q = req[r];
    const int64_t cap = ((ns.zp(z) >> r) & 1u) ? ns.avail[z][r] : 0;
    int64_t rs = 0;
    if (cap != 0 && q <= cap) {
      const uint64_t cap_v = static_cast<uint64_t>(value_of(is_cpu, cap));
      const uint64_t req_v = static_cast<uint64_t>(value_of(is_cpu, q));
      rs = div_le100((least ? cap_v - req_v : req_v) * 100u, cap_v);
    }
    acc += static_cast<uint64_t>(rs) * static_cast<uint64_t>(a.slot_weight[r]);
  }
  if (weight_sum == 0) return 0;
  return div_le100(acc, weight_sum);
  }
}

// scoreForEachNUMANode score.go:110-124
template <int RM, int SG>
__device__ __forceinline__ int64_t score_each_numa(const NodeState<RM>& ns, const NrtArgs& a, uint32_t present,
                                                   const int64_t* __restrict__ req) {
  uint64_t weight_sum = 0;
#pragma unroll
  for (int r = 0; r < RM; ++r)
    if (r < a.n_res && ((present >> r) & 1u)) weight_sum += static_cast<uint64_t>(a.slot_weight[r]);
  int64_t min_score = 0;
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    if (z < ns.nz) {
      const int64_t s = zone_score<RM, SG>(ns, a, z, present, req, weight_sum);
      if (min_score == 0 || (s != 0 && s < min_score)) min_score = s;
    }
  }
  return min_score;
}

// ---------------------------------------------------------------- LeastNUMANodes (least_numa.go)

// onlyNonNUMAResources pluginhelpers.go:163-173
template <int RM>
__device__ __forceinline__ bool only_non_numa(const NodeState<RM>& ns, uint32_t present) {
  uint32_t any = 0;
#pragma unroll
  for (int z = 0; z < kZ; ++z) any |= z < ns.nz ? ns.zp(z) : 0u;
  return (any & present) == 0;
}

// numaNodesRequired + findSuitableCombination (least_numa.go:156-208): returns the chosen subset as a
// bitmask over LIST POSITIONS (0 = nil) and whether it has the minimal average distance for its size
template <int RM>
__device__ uint32_t numa_nodes_required(const NodeState<RM>& ns, const NrtArgs& a, int64_t n, uint32_t present,
                                        const int64_t* __restrict__ req, bool* is_min) {
  *is_min = false;
  if (ns.nz == 0) return 0;
  const uint8_t* masks = kCombo.mask[ns.nz - 1];
  const uint8_t* start = kCombo.start[ns.nz - 1];
  for (int k = 1; k <= ns.nz; ++k) {
    const float min_avg = a.min_avg[static_cast<int64_t>(k - 1) * a.n_nodes + n];
    uint32_t best = 0;
    float min_distance = 256.0f;
    for (int ci = start[k - 1]; ci < start[k]; ++ci) {
      const uint32_t m = masks[ci];
      // isValidCombineResources: every member reports every requested name
      uint32_t all_present = 0xffu;
#pragma unroll
      for (int z = 0; z < kZ; ++z) all_present &= ((m >> z) & 1u) ? ns.zp(z) : 0xffu;
      if ((all_present & present) != present) continue;
      // combineResources + checkResourcesFit (Guaranteed only reaches here: isResourceSetSuitable = sum >= qty)
      bool fit = true;
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        if (r >= a.n_res || !((present >> r) & 1u)) continue;
        const int64_t q = req[r];
        if (q == 0) continue;
        int64_t sum = 0;
#pragma unroll
        for (int z = 0; z < kZ; ++z) sum += ((m >> z) & 1u) ? ns.avail[z][r] : 0;
        fit &= sum >= q;
      }
      if (!fit) continue;
      // nodesAvgDistance (float32)
      int accu = 0;
      for (int i = 0; i < ns.nz; ++i)
        if ((m >> i) & 1u)
          for (int j = 0; j < ns.nz; ++j)
            if ((m >> j) & 1u) accu += a.zone_cost[(static_cast<int64_t>(i) * kZ + j) * a.n_nodes + n];
      const float distance = static_cast<float>(accu) / static_cast<float>(k * k);
      if (distance == min_avg) {
        *is_min = true;
        return m;
      }
      if (distance < min_distance) {
        min_distance = distance;
        best = m;
      }
    }
    if (best) return best;
  }
  return 0;
}

// subtractFromNUMAs numaresources.go:184-215: the bitmask holds NUMA ids but indexes list positions (appendix B.1)
template <int RM>
__device__ __forceinline__ void subtract_from_numas(NodeState<RM>& ns, const NrtArgs& a, uint32_t present,
                                                    const int64_t* __restrict__ req, uint64_t id_bits) {
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    if (r >= a.n_res || !((present >> r) & 1u)) continue;
    int64_t quantity = req[r];
#pragma unroll
    for (int z = 0; z < kZ; ++z) {  // positions >= nz cannot hold resources; ids >= 8 would index out of range in the reference
      const bool member = ((id_bits >> z) & 1ull) && z < ns.nz && ((ns.zp(z) >> r) & 1u) && quantity != 0;
      const int64_t available = ns.avail[z][r];
      const int64_t take = quantity >= available ? available : quantity;
      ns.avail[z][r] = member ? available - take : available;
      quantity = member ? quantity - take : quantity;
    }
  }
}

__device__ __forceinline__ int64_t normalize_score(int count, bool is_min, int max_numa) {  // least_numa.go:90-100
  const int64_t numa_node_score = 100 / static_cast<int64_t>(max_numa);
  const int64_t score = 100 - static_cast<int64_t>(count) * numa_node_score;
  return is_min ? score + numa_node_score / 2 : score;
}

template <int RM>
__device__ __forceinline__ uint64_t ids_of(const NodeState<RM>& ns, uint32_t pos_mask) {
  uint64_t bits = 0;
#pragma unroll
  for (int z = 0; z < kZ; ++z)
    if ((pos_mask >> z) & 1u) bits |= 1ull << ns.id(z);
  return bits;
}

// ---------------------------------------------------------------- the sweep

template <int RM, int SG>
__global__ __launch_bounds__(256, 2) void k_nrt(NrtArgs a, int n_tiles) {
  SPX_RESOLVE_ROWS(a);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const int tile = static_cast<int>(unit % n_tiles);
  const int64_t chunk = unit / n_tiles;
  const int64_t pod0 = a.row_begin + chunk * kPodsPerUnit;
  if (pod0 >= a.row_end) return;
  const int64_t pod1 = pod0 + kPodsPerUnit < a.row_end ? pod0 + kPodsPerUnit : a.row_end;
  const int64_t n = static_cast<int64_t>(tile) * 64 + lane;
  const bool in = n < a.n_nodes;
  const int R = a.n_res;

  NodeState<RM> ns;
  ns.flags = in ? a.flags[n] : 0u;
  ns.nz = in ? a.n_zones[n] : 0;
  ns.node_present = in ? a.node_present[n] : 0u;
  ns.max_numa = in ? a.max_numa[n] : 8;
  ns.id_lo = ns.id_hi = ns.zp_lo = ns.zp_hi = 0;
#pragma unroll
  for (int z = 0; z < kZ; ++z) {
    const uint32_t idv = in ? a.zone_id[static_cast<int64_t>(z) * a.n_nodes + n] : 0u;
    const uint32_t zpv = in ? a.zone_present[static_cast<int64_t>(z) * a.n_nodes + n] : 0u;
    if (z < 4) {
      ns.id_lo |= idv << (8 * z);
      ns.zp_lo |= zpv << (8 * z);
    } else {
      ns.id_hi |= idv << (8 * (z - 4));
      ns.zp_hi |= zpv << (8 * (z - 4));
    }
  }
  load_avail(ns, a, n, in);
  const bool fresh = ns.flags & SPX_NRT_F_FRESH;
  const bool has_nrt = ns.flags & SPX_NRT_F_HAS_NRT;
  const bool single = ns.flags & SPX_NRT_F_SINGLE_NUMA;
  const bool pod_scope = ns.flags & SPX_NRT_F_POD_SCOPE;

  for (int64_t pod = pod0; pod < pod1; ++pod) {
    // ---- wave-uniform pod record
    const int qos = a.qos[pod];
    const bool non_native = a.non_native[pod] != 0;
    const int n_ctr = a.n_ctr[pod];
    const uint32_t pod_present = a.pod_present[pod];
    const int64_t* __restrict__ preq = a.pod_req + pod * R;
    const uint8_t* __restrict__ ckind = a.ctr_kind + pod * kC;
    const uint8_t* __restrict__ cpres = a.ctr_present + pod * kC;
    const int64_t* __restrict__ creq = a.ctr_req + pod * kC * R;
    const bool non_g = qos != SPX_QOS_GUARANTEED;

    // ================= Filter (filter.go:179-245)
    uint32_t status = 0;
    if (!(qos == SPX_QOS_BESTEFFORT && !non_native)) {  // uniform
      if (!fresh) {
        status = SPX_NRT_ST_INVALID_TOPOLOGY;
      } else if (has_nrt && single) {
        if (pod_scope) {  // singleNUMAPodLevelHandler
          uint32_t id;
          if (!fits_any(ns, a, non_g, pod_present, preq, &id)) status = SPX_NRT_ST_POD;
        } else {  // singleNUMAContainerLevelHandler
          for (int c = 0; c < n_ctr; ++c) {  // init containers: must fit, never subtracted
            if (ckind[c] == SPX_CTR_APP) continue;
            uint32_t id;
            const bool ok = fits_any(ns, a, non_g, cpres[c], creq + c * R, &id);
            if (status == 0 && !ok) status = ckind[c] == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER;
          }
          uint64_t chosen = 0;   // NUMA id picked per app container (for the undo), 8 bits each
          uint32_t placed = 0;   // bit c: container c was subtracted on this lane
          for (int c = 0; c < n_ctr; ++c) {
            if (ckind[c] != SPX_CTR_APP) continue;
            uint32_t id;
            const bool ok = fits_any(ns, a, non_g, cpres[c], creq + c * R, &id);
            const bool live = status == 0;
            if (live && !ok) status = SPX_NRT_ST_CONTAINER;
            const bool apply = live && ok;
            adjust_numa(ns, a, non_g, cpres[c], creq + c * R, id, apply, -1);
            chosen |= static_cast<uint64_t>(apply ? id : 0u) << (8 * c);
            placed |= (apply ? 1u : 0u) << c;
          }
          for (int c = 0; c < n_ctr; ++c) {  // undo: Filter works on a private copy in the reference
            if (ckind[c] != SPX_CTR_APP) continue;
            adjust_numa(ns, a, non_g, cpres[c], creq + c * R, static_cast<uint32_t>((chosen >> (8 * c)) & 0xffu),
                        (placed >> c) & 1u, +1);
          }
        }
      }
    }

    // ================= Score (score.go:62-102)
    int64_t score;
    if (non_g) {
      score = 100;
    } else if (!fresh || !has_nrt) {
      score = 0;
    } else if constexpr (SG == kSgLeastNuma) {
      if (pod_scope) {  // leastNUMAPodScopeScore
        if (only_non_numa(ns, pod_present)) {
          score = 100;
        } else {
          bool is_min;
          const uint32_t m = numa_nodes_required(ns, a, n, pod_present, preq, &is_min);
          score = m ? normalize_score(__builtin_popcount(m), is_min, ns.max_numa) : 0;
        }
      } else {  // leastNUMAContainerScopeScore
        int max_count = 0;
        bool all_min = true, failed = false, dirty = false;
        for (int c = 0; c < n_ctr; ++c) {
          if (failed || only_non_numa(ns, cpres[c])) continue;
          bool is_min;
          const uint32_t m = numa_nodes_required(ns, a, n, cpres[c], creq + c * R, &is_min);
          if (!m) {
            failed = true;
            continue;
          }
          all_min &= is_min;
          const int cnt = __builtin_popcount(m);
          max_count = cnt > max_count ? cnt : max_count;
          subtract_from_numas(ns, a, cpres[c], creq + c * R, ids_of(ns, m));
          dirty = true;
        }
        score = failed ? 0 : (max_count == 0 ? 100 : normalize_score(max_count, all_min, ns.max_numa));
        if (dirty) load_avail(ns, a, n, in);  // the reference scored on a private NUMANodeList
      }
    } else if (!single) {
      score = 0;
    } else if (pod_scope) {
      score = score_each_numa<RM, SG>(ns, a, pod_present, preq);
    } else {  // containerScopeScore: int64(mean) over init + app containers
      int64_t sum = 0;
      for (int c = 0; c < n_ctr; ++c) sum += score_each_numa<RM, SG>(ns, a, cpres[c], creq + c * R);
      score = n_ctr > 0 ? sum / n_ctr : 0;
    }

    if (in && a.out_raw != nullptr) {  // parity harness: the int64 Score() value, one row
      a.out_raw[n] = score;
    } else if (in) {
      const int64_t cell = pod * a.row_stride + n;
      a.out_status[cell] = static_cast<uint8_t>(status);
      score = score < 0 ? 0 : (score > 255 ? 255 : score);
      a.out_score[cell] = static_cast<uint8_t>(score);
    }
  }
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void k_nrt_creq_from_items(const uint32_t* __restrict__ items, int n_res, int64_t n_pods, int64_t* __restrict__ out) {
  const int iw = n_res <= 4 ? 16 : 32;  // dwords per item; items 2.. of a pod's 10 are its containers (spx_engine.hip: nrt_pod_items)
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;  // (pod, container, resource)
  if (i >= n_pods * SPX_NRT_MAX_CTRS * n_res) return;
  const int r = static_cast<int>(i % n_res);
  const int64_t pc = i / n_res;
  const int c = static_cast<int>(pc % SPX_NRT_MAX_CTRS);
  const int64_t pod = pc / SPX_NRT_MAX_CTRS;
  const uint32_t* w = items + (pod * 10 + 2 + c) * iw + 2 * r;
  out[i] = static_cast<int64_t>(__hiloint2double(static_cast<int>(w[1]), static_cast<int>(w[0])));
}
}  // namespace

void launch_nrt_creq_from_items(const uint32_t* pod_items, int n_res, int64_t n_pods, int64_t* ctr_req, hipStream_t s) {
  const int64_t n = n_pods * SPX_NRT_MAX_CTRS * n_res;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_nrt_creq_from_items, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, pod_items, n_res, n_pods, ctr_req);
}

// true = the sweep ran as the fused Filter + Score launch (kernels_nrt_fused.hip)
bool launch_nrt(const NrtArgs& a, hipStream_t s) {
  if (a.row_end <= a.row_begin) return false;
  const bool generic_only = (a.opts & kOptNrtGeneric) != 0;  // SPX_OPT_REFERENCE_KERNELS
  if (!generic_only && launch_nrt_fused(a, s)) return true;
  if (!generic_only && launch_nrt_fast(a, s)) return false;
  const int n_tiles = static_cast<int>((a.n_nodes + 63) / 64);
  const int64_t chunks = (a.row_end - a.row_begin + kPodsPerUnit - 1) / kPodsPerUnit;
  const unsigned blocks = static_cast<unsigned>((chunks * n_tiles + 3) / 4);
  const int sg = a.strategy == SPX_NRT_LEAST_NUMA_NODES ? kSgLeastNuma : (a.strategy == SPX_NRT_BALANCED_ALLOCATION ? kSgBalanced : kSgAlloc);
#define SPX_NRT_CASE(RMV, SGV)                                                               \
  if ((a.n_res <= 4) == (RMV == 4) && sg == SGV) {                                           \
    hipLaunchKernelGGL((k_nrt<RMV, SGV>), dim3(blocks), dim3(256), 0, s, a, n_tiles);       \
    return false;                                                                            \
  }
  SPX_NRT_CASE(4, kSgAlloc)
  SPX_NRT_CASE(4, kSgBalanced)
  SPX_NRT_CASE(4, kSgLeastNuma)
  SPX_NRT_CASE(8, kSgAlloc)
  SPX_NRT_CASE(8, kSgBalanced)
  SPX_NRT_CASE(8, kSgLeastNuma)
#undef SPX_NRT_CASE
  return false;
}

}  // namespace spx
