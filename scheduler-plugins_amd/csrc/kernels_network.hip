// kernels_network.hip — gfx950 kernel for networkaware NetworkOverhead: PreFilter + Filter + Score +
// NormalizeScore for every (pod, node) of a frozen snapshot.
//
// The reference's PreFilter walks, per pod and per node, every (scheduled AppGroup pod x dependency)
// pair (networkoverhead.go:500-638).  Two observations remove almost all of that work:
//   1. the pair list depends only on the pod's (AppGroup, selector) "workload key" (flattened on the host);
//   2. a pair's contribution to a node depends on the node only through its (region, zone) labels —
//      unless the node is the pair's own host.  Nodes with equal labels form a "topology class".
// One wavefront handles one pod row:
//   phase 1  lanes = topology classes: accumulate (satisfied, violated, cost) over the pod's pairs into LDS;
//   phase 2  mark the (<= pairs) host nodes in an LDS bitmap;
//   phase 3  lanes = nodes, 4 consecutive nodes per lane: class lookup from LDS (host nodes and class-less
//            snapshots take the exact per-pair path), Filter = violated > satisfied, wave min/max of the
//            cost over feasible nodes;
//   phase 4  same sweep again, now normalising (100 - 100*(s-min)/(max-min)) and storing one dword of
//            status bytes and one dword of score bytes per lane (256 contiguous bytes per wave per table).
// Output-write bound: 2 B per (pod,node); inputs are 8 B per node, a few bytes per pod, cost matrices in L2.
#include <cstdlib>

#include "spx_internal.h"

namespace spx {

namespace {

constexpr int kNpl = 4;  // nodes per lane
constexpr int kSameZone = SPX_NET_SAME_ZONE;
constexpr int kMaxCost = SPX_NET_MAX_COST;

struct Acc {
  int sat, vio, cost;
};

// contribution of one (scheduled pod on `host`, dependency with `max_cost`) pair to a node with labels
// (region, zone) that is NOT the host — checkMaxNetworkCostRequirements :536-567 + getAccumulatedCost :605-633
__device__ __forceinline__ void add_pair(Acc& a, const NetArgs& g, int region, int zone, int host_region, int host_zone,
                                         int64_t max_cost) {
  if (host_region < 0 && host_zone < 0) {  // placed node carries neither label
    a.vio += 1;
    a.cost += kMaxCost;
  } else if (region == host_region) {
    if (zone == host_zone) {
      a.sat += 1;
      a.cost += kSameZone;
    } else {
      const int c = (zone >= 0 && host_zone >= 0) ? g.zone_cost[static_cast<int64_t>(zone) * g.n_zones + host_zone] : -1;
      if (c >= 0) {
        if (c <= max_cost) a.sat += 1;
        else a.vio += 1;
        a.cost += c;
      } else {
        a.cost += kMaxCost;  // missing entry: not counted, but charged MaxCost
      }
    }
  } else {
    const int c = (region >= 0 && host_region >= 0) ? g.region_cost[static_cast<int64_t>(region) * g.n_regions + host_region] : -1;
    if (c >= 0) {
      if (c <= max_cost) a.sat += 1;
      else a.vio += 1;
      a.cost += c;
    } else {
      a.cost += kMaxCost;
    }
  }
}

// exact per-pair evaluation of one node (host nodes; snapshots without a class table)
__device__ Acc direct_eval(const NetArgs& g, int64_t node, int lo, int hi) {
  Acc a{0, 0, 0};
  const int region = g.region[node], zone = g.zone[node];
  for (int i = lo; i < hi; ++i) {
    const int host = g.pair_node[i];
    if (host == node) {
      a.sat += 1;  // same hostname: satisfied, cost 0
      continue;
    }
    add_pair(a, g, region, zone, g.region[host], g.zone[host], g.pair_max[i]);
  }
  return a;
}

// the same over a pair list staged in LDS (host, its region and zone, MaxNetworkCost): the single-row launch of the sequential
// commit loop has one workgroup and nothing to hide the two dependent global loads per pair behind (25 us per pod at 20k nodes)
struct StagedPairs {
  const int* host;
  const int* region;
  const int* zone;
  const long long* max_cost;
  int n;
};
__device__ Acc direct_eval_staged(const NetArgs& g, int64_t node, const StagedPairs& sp) {
  Acc a{0, 0, 0};
  const int region = g.region[node], zone = g.zone[node];
  for (int i = 0; i < sp.n; ++i) {
    if (sp.host[i] == node) {
      a.sat += 1;
      continue;
    }
    add_pair(a, g, region, zone, sp.region[i], sp.zone[i], sp.max_cost[i]);
  }
  return a;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(64) void k_net(NetArgs g) {
  SPX_RESOLVE_ROWS(g);
  extern __shared__ __align__(16) int lds[];
  int* cls_sat = lds;
  int* cls_vio = lds + g.n_classes;
  int* cls_cost = lds + 2 * g.n_classes;
  unsigned* host_bits = reinterpret_cast<unsigned*>(lds + 3 * g.n_classes);
  const int lane = threadIdx.x;
  const int64_t pod = g.row_begin + blockIdx.x;
  if (pod >= g.row_end) return;
  const int key = g.pod_key[pod];
  const int flag = g.key_flag[key];
  const int lo = g.pair_ptr[key], hi = g.pair_end ? g.pair_end[key] : g.pair_ptr[key + 1];  // pair_end: lists that grow (commit loop)
  const int64_t n_words = (g.n_nodes + 31) / 32;
  const bool use_cls = g.n_classes > 0;
  const uint8_t* other0 = g.other_status[0] ? g.other_status[0] + pod * g.row_stride : nullptr;
  const uint8_t* other1 = g.other_status[1] ? g.other_status[1] + pod * g.row_stride : nullptr;
  const int64_t tiles = (g.row_stride + 64 * kNpl - 1) / (64 * kNpl);

  if (flag != 0) {
    // scoreEqually: Filter passes, Score = MinNodeScore, NormalizeScore leaves all-zero rows alone
    // (networkoverhead.go:342-345, :376-379, :400-402); flag 2 = PreFilter returned Error
    const uint32_t st = flag == 2 ? 0xffffffffu : 0u;
    for (int64_t t = 0; t < tiles; ++t) {
      const int64_t n0 = (t * 64 + lane) * kNpl;
      if (n0 >= g.row_stride) continue;
      if (g.out_raw) {
        for (int j = 0; j < kNpl; ++j)
          if (n0 + j < g.n_nodes) g.out_raw[n0 + j] = 0;
      } else {
        *reinterpret_cast<uint32_t*>(g.out_status + pod * g.row_stride + n0) = st;
        *reinterpret_cast<uint32_t*>(g.out_score + pod * g.row_stride + n0) = 0u;
      }
    }
    return;
  }

  // ---- phase 1: per-class accumulation
  if (use_cls) {
    for (int c = lane; c < g.n_classes; c += 64) {
      Acc a{0, 0, 0};
      const int region = g.cls_region[c], zone = g.cls_zone[c];
      for (int i = lo; i < hi; ++i) {
        const int host = g.pair_node[i];  // wave-uniform
        add_pair(a, g, region, zone, g.region[host], g.zone[host], g.pair_max[i]);
      }
      cls_sat[c] = a.sat;
      cls_vio[c] = a.vio;
      cls_cost[c] = a.cost;
    }
    // ---- phase 2: host bitmap
    for (int64_t w = lane; w < n_words; w += 64) host_bits[w] = 0u;
    __syncthreads();
    for (int i = lo + lane; i < hi; i += 64) {
      const int host = g.pair_node[i];
      atomicOr(&host_bits[host >> 5], 1u << (host & 31));
    }
    __syncthreads();
  }

  auto eval = [&](int64_t n) -> Acc {
    if (!use_cls || ((host_bits[n >> 5] >> (n & 31)) & 1u)) return direct_eval(g, n, lo, hi);
    const int c = g.node_class[n];
    return Acc{cls_sat[c], cls_vio[c], cls_cost[c]};
  };

  // ---- phase 3: Filter + min/max of the cost over feasible nodes (upstream scores feasible nodes only)
  int mn = INT32_MAX, mx = INT32_MIN;
  for (int64_t t = 0; t < tiles; ++t) {
    const int64_t n0 = (t * 64 + lane) * kNpl;
#pragma unroll
    for (int j = 0; j < kNpl; ++j) {
      const int64_t n = n0 + j;
      if (n >= g.n_nodes) continue;
      const Acc a = eval(n);
      const bool feasible = !(a.vio > a.sat) && (!other0 || other0[n] == 0) && (!other1 || other1[n] == 0);
      if (feasible) {
        mn = a.cost < mn ? a.cost : mn;
        mx = a.cost > mx ? a.cost : mx;
      }
    }
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  const int range = mx - mn;

  // ---- phase 4: NormalizeScore (networkoverhead.go:389-418) + stores
  for (int64_t t = 0; t < tiles; ++t) {
    const int64_t n0 = (t * 64 + lane) * kNpl;
    if (n0 >= g.row_stride) continue;
    uint32_t st_w = 0, sc_w = 0;
#pragma unroll
    for (int j = 0; j < kNpl; ++j) {
      const int64_t n = n0 + j;
      if (n >= g.n_nodes) continue;
      const Acc a = eval(n);
      const bool pass = !(a.vio > a.sat);
      const bool feasible = pass && (!other0 || other0[n] == 0) && (!other1 || other1[n] == 0);
      int score = 0;
      if (feasible) {
        if (mn == 0 && mx == 0) score = a.cost;                            // all minimum: untouched (== 0)
        else if (range != 0) score = 100 - (100 * (a.cost - mn)) / range;  // == 100 - int64(100.0*d/r): 100*d/r is never within 1e-6 of an integer from below
        else score = 100 - (a.cost - mn);                                  // max == min != 0
      }
      if (g.out_raw) {
        g.out_raw[n] = g.raw_which == SPX_NET_RAW_SATISFIED ? a.sat : (g.raw_which == SPX_NET_RAW_VIOLATED ? a.vio : a.cost);
      } else {
        score = score < 0 ? 0 : (score > 255 ? 255 : score);
        st_w |= (pass ? 0u : static_cast<uint32_t>(SPX_NET_ST_UNSCHEDULABLE)) << (8 * j);
        sc_w |= static_cast<uint32_t>(score) << (8 * j);
      }
    }
    if (!g.out_raw) {
      *reinterpret_cast<uint32_t*>(g.out_status + pod * g.row_stride + n0) = st_w;
      *reinterpret_cast<uint32_t*>(g.out_score + pod * g.row_stride + n0) = sc_w;
    }
  }
}


// ---------------------------------------------------------------- table sweep through the class table
//
// k_net above evaluates every node on its own (class lookup + host test + three LDS reads, twice).  For whole
// tables the per-node work shrinks to one 16-bit class id and one LDS read of a finished byte pair:
//   phase 1  lanes = classes: (cost, Filter verdict) per class into LDS                       [as above]
//   phase 2  host bitmap; distinct hosts counted per class
//   phase 3  min/max of the cost over the pod's feasible nodes:
//              no other Filter plugin in the evaluation: over classes that keep at least one non-host node, plus
//              the (<= pairs) host nodes evaluated exactly — no pass over the nodes at all;
//              otherwise one pass over the nodes with dword loads of the other plugins' status bytes
//   phase 3b lanes = classes: NormalizeScore per class -> (status << 8 | score) in LDS
//   phase 4  lanes = 4 consecutive nodes: one 8-byte load of class ids, 4 LDS reads, host nodes (rare) patched
//            exactly, one dword store per table.
__device__ __forceinline__ int norm_cost(int cost, int mn, int mx) {
  // networkoverhead.go:389-418; 100*d/r is never within 1e-6 of an integer from below, so int64(float) == integer division
  if (mn == 0 && mx == 0) return cost;
  const int range = mx - mn;
  return range != 0 ? 100 - (100 * (cost - mn)) / range : 100 - (cost - mn);
}

// One wavefront per pod row in a batch launch; a single-row launch (the sequential commit loop) puts kRowThreads threads on
// its row — every loop below strides by the block size, the one reduction combines the waves through LDS.
constexpr int kRowThreads = 1024;
constexpr int kNetStagePairs = 512;  // pairs of a key staged in LDS by a single-row launch
__global__ __launch_bounds__(kRowThreads) void k_net_cls(NetArgs g) {
  SPX_RESOLVE_ROWS(g);
  extern __shared__ __align__(16) int lds[];
  const int C = g.n_classes;
  int* cls_word = lds;                                            // cost | (Filter fails) << 31
  int* cls_hosts = lds + C;                                       // distinct host nodes of the class
  uint32_t* cls_fin = reinterpret_cast<uint32_t*>(lds + 2 * C);   // status << 8 | normalised score
  unsigned* host_bits = reinterpret_cast<unsigned*>(lds + 3 * C);
  const int lane = threadIdx.x, nthr = blockDim.x;
  const int64_t pod = g.row_begin + blockIdx.x;
  if (pod >= g.row_end) return;
  const int key = g.pod_key[pod];
  const int flag = g.key_flag[key];
  const int lo = g.pair_ptr[key], hi = g.pair_end ? g.pair_end[key] : g.pair_ptr[key + 1];  // pair_end: lists that grow (commit loop)
  const int64_t n_words = (g.n_nodes + 31) / 32;
  const uint8_t* other0 = g.other_status[0] ? g.other_status[0] + pod * g.row_stride : nullptr;
  const uint8_t* other1 = g.other_status[1] ? g.other_status[1] + pod * g.row_stride : nullptr;
  uint8_t* out_st = g.out_status + pod * g.row_stride;
  uint8_t* out_sc = g.out_score + pod * g.row_stride;

  constexpr int kG = 16;
  const int64_t groups = g.row_stride / kG;  // rows are padded to a multiple of 16 bytes
  uint16_t* scored = reinterpret_cast<uint16_t*>(host_bits + n_words);  // [groups]
  const bool fuse = g.out_alloc != nullptr && nthr == 64;  // uniform (the engine asks for it in batch launches only)
  uint8_t* out_al = fuse ? g.out_alloc + pod * g.row_stride : nullptr;
  auto load16 = [](const void* p) { return *reinterpret_cast<const uint4*>(p); };
  auto byte_of = [](const uint32_t (&w)[4], int j) { return (w[j >> 2] >> (8 * (j & 3))) & 0xffu; };
  auto half_of = [](const uint32_t (&w)[8], int j) { return (w[j >> 1] >> (16 * (j & 1))) & 0xffffu; };
  auto rel16 = [&](int64_t n0, uint32_t (&r)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = load16(g.alloc_rel + n0 + 4 * q);
      r[4 * q] = v.x, r[4 * q + 1] = v.y, r[4 * q + 2] = v.z, r[4 * q + 3] = v.w;
    }
  };
  auto others16 = [&](int64_t n0, uint32_t (&oth)[4]) {  // non-zero byte: another Filter plugin rejected the node
    uint4 o4 = uint4{0, 0, 0, 0};
    if (other0) o4 = load16(other0 + n0);
    if (other1) {
      const uint4 v = load16(other1 + n0);
      o4.x |= v.x, o4.y |= v.y, o4.z |= v.z, o4.w |= v.w;
    }
    oth[0] = o4.x, oth[1] = o4.y, oth[2] = o4.z, oth[3] = o4.w;
  };
  auto open16 = [&](int64_t n0, const uint32_t (&oth)[4]) {  // passed the other plugins, inside the table
    uint32_t open = 0;
#pragma unroll
    for (int j = 0; j < kG; ++j) open |= (n0 + j < g.n_nodes && byte_of(oth, j) == 0u) ? 1u << j : 0u;
    return open;
  };
  auto wave_range = [&](uint32_t& lo32, uint32_t& hi32) {
    lo32 = static_cast<uint32_t>(wave_min(static_cast<int>(lo32 ^ 0x80000000u))) ^ 0x80000000u;
    hi32 = static_cast<uint32_t>(wave_max(static_cast<int>(hi32 ^ 0x80000000u))) ^ 0x80000000u;
  };
  auto alloc16 = [&](int64_t n0, uint32_t ok, uint32_t alo, double ab, uint32_t (&al_w)[4]) {  // floor((rel - lo) * 100 / range); infeasible cells hold 0
    uint32_t rel[16];
    rel16(n0, rel);
#pragma unroll
    for (int j = 0; j < kG; ++j)
      if ((ok >> j) & 1u) al_w[j >> 2] |= static_cast<uint32_t>(static_cast<double>(rel[j] - alo) * ab) << (8 * (j & 3));
  };

  if (flag != 0) {  // scoreEqually / PreFilter error, as in k_net: status 0 / 0xff everywhere, score 0
    const uint32_t st = flag == 2 ? 0xffffffffu : 0u;
    uint32_t alo = 0xffffffffu, ahi = 0u;
    const bool fz = fuse && flag != 2;  // Allocatable: scored = passed the other plugins (a PreFilter error leaves no feasible node: zeros)
    if (fz) {
      for (int64_t q = lane; q < groups; q += nthr) {
        const int64_t n0 = q * kG;
        uint32_t open = 0;
        if (n0 < g.n_nodes) {
          uint32_t oth[4], rel[16];
          others16(n0, oth);
          rel16(n0, rel);
          open = open16(n0, oth);
#pragma unroll
          for (int j = 0; j < kG; ++j)
            if ((open >> j) & 1u) alo = rel[j] < alo ? rel[j] : alo, ahi = rel[j] > ahi ? rel[j] : ahi;
        }
        scored[q] = static_cast<uint16_t>(open);
      }
      wave_range(alo, ahi);
    }
    const uint32_t arange = (fz && ahi >= alo) ? ahi - alo : 0u;
    const double ab = arange ? (100.0 / static_cast<double>(arange)) * (1.0 + 0x1p-49) : 0.0;
    for (int64_t q = lane; q < groups; q += nthr) {
      const int64_t n0 = q * kG;
      *reinterpret_cast<uint4*>(out_st + n0) = uint4{st, st, st, st};
      *reinterpret_cast<uint4*>(out_sc + n0) = uint4{0u, 0u, 0u, 0u};
      if (fuse) {
        uint32_t al_w[4] = {0, 0, 0, 0};
        const uint32_t ok = fz ? scored[q] : 0u;
        if (ok != 0 && arange != 0) alloc16(n0, ok, alo, ab, al_w);
        *reinterpret_cast<uint4*>(out_al + n0) = uint4{al_w[0], al_w[1], al_w[2], al_w[3]};
      }
    }
    return;
  }

  // ---- the pair list into LDS when it is short enough (it nearly always is: the placed pods of one AppGroup's dependencies)
  // (dynamic LDS behind the host bitmap, present only in a single-row launch: a batch launch runs one wave per row and keeps its
  // LDS footprint — its occupancy — as it was)
  constexpr int kStage = kNetStagePairs;
  const int scored_words = static_cast<int>((g.row_stride / 16 + 1) / 2);  // the "scored" bits of phases 3-4: 16 per group of 16 nodes
  long long* sp_max = reinterpret_cast<long long*>(lds + ((3 * C + static_cast<int>(n_words) + scored_words + 1) & ~1));
  int* sp_host = reinterpret_cast<int*>(sp_max + kStage);
  int* sp_region = sp_host + kStage;
  int* sp_zone = sp_region + kStage;
  const bool staged = nthr == kRowThreads && hi - lo <= kStage;
  const StagedPairs sp{sp_host, sp_region, sp_zone, sp_max, hi - lo};
  if (staged) {
    for (int i = lane; i < hi - lo; i += nthr) {
      const int host = g.pair_node[lo + i];
      sp_host[i] = host, sp_region[i] = g.region[host], sp_zone[i] = g.zone[host], sp_max[i] = g.pair_max[lo + i];
    }
    __syncthreads();
  }
  auto direct = [&](int64_t node) { return staged ? direct_eval_staged(g, node, sp) : direct_eval(g, node, lo, hi); };
  // ---- phase 1 + 2
  for (int c = lane; c < C; c += nthr) {
    Acc a{0, 0, 0};
    const int region = g.cls_region[c], zone = g.cls_zone[c];
    if (staged) {
      for (int i = 0; i < sp.n; ++i) add_pair(a, g, region, zone, sp_region[i], sp_zone[i], sp_max[i]);
    } else {
      for (int i = lo; i < hi; ++i) {
        const int host = g.pair_node[i];  // wave-uniform
        add_pair(a, g, region, zone, g.region[host], g.zone[host], g.pair_max[i]);
      }
    }
    cls_word[c] = a.cost | (a.vio > a.sat ? static_cast<int>(0x80000000u) : 0);
    cls_hosts[c] = 0;
  }
  for (int64_t w = lane; w < n_words; w += nthr) host_bits[w] = 0u;
  __syncthreads();
  for (int i = lo + lane; i < hi; i += nthr) {
    const int host = g.pair_node[i];
    const unsigned bit = 1u << (host & 31);
    if (!(atomicOr(&host_bits[host >> 5], bit) & bit)) atomicAdd(&cls_hosts[g.node_class16[host]], 1);
  }
  __syncthreads();

  // ---- phases 3 and 4 walk the row in groups of 16 consecutive nodes per lane (round 5): one 16-byte load per status table, two of
  // class ids, one 16-byte store per output table (dword accesses — 256 contiguous bytes per wave and table — reached 2.6 TB/s).
  // The first walk leaves each group's 16 "scored" bits (passed every other Filter plugin and this one) in LDS, so the other plugins'
  // status rows are read from HBM once.  With NetArgs::out_alloc set the same two walks also carry NodeResourcesAllocatable's
  // feasibility-aware NormalizeScore (what k_alloc_masked's compact path computes, kernels_profile.hip: min/max of the raw-score
  // offsets over the scored nodes, then floor((rel - lo) * 100 / range) as one float64 multiply) — the feasible set is the same, and
  // the separate launch would read both status tables again (2.5 GB at config #5's share).
  auto classes16 = [&](int64_t n0, uint32_t (&cw)[8]) {
    const uint4 c0 = load16(g.node_class16 + n0), c1 = load16(g.node_class16 + n0 + 8);
    cw[0] = c0.x, cw[1] = c0.y, cw[2] = c0.z, cw[3] = c0.w, cw[4] = c1.x, cw[5] = c1.y, cw[6] = c1.z, cw[7] = c1.w;
  };
  auto hosts16 = [&](int64_t n0) -> uint32_t { return (host_bits[n0 >> 5] >> (n0 & 31)) & 0xffffu; };  // n0 is a multiple of 16

  // ---- phase 3
  int mn = INT32_MAX, mx = INT32_MIN;
  uint32_t alo = 0xffffffffu, ahi = 0u;  // Allocatable: offsets of the scored nodes
  const bool walk = other0 || other1 || fuse;
  if (!walk) {
    for (int c = lane; c < C; c += nthr) {
      const int w = cls_word[c];
      if (w >= 0 && g.cls_size[c] - cls_hosts[c] > 0) {
        mn = w < mn ? w : mn;
        mx = w > mx ? w : mx;
      }
    }
    for (int i = lo + lane; i < hi; i += nthr) {
      const Acc a = direct(g.pair_node[i]);
      if (!(a.vio > a.sat)) {
        mn = a.cost < mn ? a.cost : mn;
        mx = a.cost > mx ? a.cost : mx;
      }
    }
  } else {
    for (int64_t q = lane; q < groups; q += nthr) {
      const int64_t n0 = q * kG;
      uint32_t ok = 0;
      if (n0 < g.n_nodes) {
        uint32_t oth[4], cw[8], rel[16];
        others16(n0, oth);
        classes16(n0, cw);
        if (fuse) rel16(n0, rel);
        const uint32_t hb = hosts16(n0);
        const uint32_t open = open16(n0, oth);
#pragma unroll
        for (int j = 0; j < kG; ++j) {
          const int w = cls_word[half_of(cw, j)];
          if (((open & ~hb) >> j) & 1u && w >= 0) {
            mn = w < mn ? w : mn;
            mx = w > mx ? w : mx;
            ok |= 1u << j;
            if (fuse) alo = rel[j] < alo ? rel[j] : alo, ahi = rel[j] > ahi ? rel[j] : ahi;
          }
        }
        for (uint32_t hs = open & hb; hs != 0; hs &= hs - 1) {  // hosts of the pod's pairs (rare): exact
          const int j = __builtin_ctz(hs);
          const Acc a = direct(n0 + j);
          if (!(a.vio > a.sat)) {
            mn = a.cost < mn ? a.cost : mn;
            mx = a.cost > mx ? a.cost : mx;
            ok |= 1u << j;
            if (fuse) {
              const uint32_t r = g.alloc_rel[n0 + j];
              alo = r < alo ? r : alo, ahi = r > ahi ? r : ahi;
            }
          }
        }
      }
      scored[q] = static_cast<uint16_t>(ok);
    }
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  if (fuse) wave_range(alo, ahi);
  if (nthr > 64) {
    __shared__ int s_mn[kRowThreads / 64], s_mx[kRowThreads / 64];
    if ((lane & 63) == 0) s_mn[lane >> 6] = mn, s_mx[lane >> 6] = mx;
    __syncthreads();
    mn = INT32_MAX, mx = INT32_MIN;
    for (int w = 0; w < (nthr >> 6); ++w) {
      mn = s_mn[w] < mn ? s_mn[w] : mn;
      mx = s_mx[w] > mx ? s_mx[w] : mx;
    }
  }

  // ---- phase 3b
  for (int c = lane; c < C; c += nthr) {
    const int w = cls_word[c];
    int score = w >= 0 ? norm_cost(w, mn, mx) : 0;
    score = score < 0 ? 0 : (score > 255 ? 255 : score);
    cls_fin[c] = static_cast<uint32_t>(score) | (w < 0 ? static_cast<uint32_t>(SPX_NET_ST_UNSCHEDULABLE) << 8 : 0u);
  }
  __syncthreads();
  // Allocatable's row constants (a fused launch is a batch launch: one wave per row, alo / ahi are the row's)
  const uint32_t arange = (fuse && ahi >= alo) ? ahi - alo : 0u;
  const double ab = arange ? (100.0 / static_cast<double>(arange)) * (1.0 + 0x1p-49) : 0.0;

  // ---- phase 4
  for (int64_t q = lane; q < groups; q += nthr) {
    const int64_t n0 = q * kG;
    uint32_t st_w[4] = {0, 0, 0, 0}, sc_w[4] = {0, 0, 0, 0}, al_w[4] = {0, 0, 0, 0};
    if (n0 < g.n_nodes) {
      const uint32_t ok = walk ? scored[q] : 0xffffu;  // (written by this thread) rejected elsewhere: not scored
      uint32_t cw[8];
      classes16(n0, cw);
      const uint32_t hb = hosts16(n0);
#pragma unroll
      for (int j = 0; j < kG; ++j) {
        uint32_t fin = cls_fin[half_of(cw, j)];
        if (n0 + j >= g.n_nodes) fin = 0;
        st_w[j >> 2] |= (fin >> 8) << (8 * (j & 3));
        if ((ok >> j) & 1u) sc_w[j >> 2] |= (fin & 0xffu) << (8 * (j & 3));
      }
      for (uint32_t hs = hb; hs != 0; hs &= hs - 1) {  // a host of one of the pod's pairs: exact
        const int j = __builtin_ctz(hs);
        if (n0 + j >= g.n_nodes) continue;
        const Acc a = direct(n0 + j);
        const bool pass = !(a.vio > a.sat);
        int score = pass ? norm_cost(a.cost, mn, mx) : 0;
        score = score < 0 ? 0 : (score > 255 ? 255 : score);
        const uint32_t sh = 8 * (j & 3), keep = ~(0xffu << sh);
        st_w[j >> 2] = (st_w[j >> 2] & keep) | ((pass ? 0u : static_cast<uint32_t>(SPX_NET_ST_UNSCHEDULABLE)) << sh);
        sc_w[j >> 2] = (sc_w[j >> 2] & keep) | ((((ok >> j) & 1u) ? static_cast<uint32_t>(score) : 0u) << sh);
      }
      if (fuse && ok != 0 && arange != 0) alloc16(n0, ok, alo, ab, al_w);
    }
    *reinterpret_cast<uint4*>(out_st + n0) = uint4{st_w[0], st_w[1], st_w[2], st_w[3]};
    *reinterpret_cast<uint4*>(out_sc + n0) = uint4{sc_w[0], sc_w[1], sc_w[2], sc_w[3]};
    if (fuse) *reinterpret_cast<uint4*>(out_al + n0) = uint4{al_w[0], al_w[1], al_w[2], al_w[3]};
  }
}

}  // namespace

size_t net_lds_bytes(int n_classes, int64_t n_nodes) {
  // class tables, host bitmap, the 16 "scored" bits per group of 16 nodes of the padded row
  return static_cast<size_t>(3 * n_classes) * sizeof(int) + static_cast<size_t>((n_nodes + 31) / 32) * sizeof(unsigned) +
         static_cast<size_t>((n_nodes + 4096) / 16 + 2) * sizeof(uint16_t) + 16;  // (row stride < n_nodes + 4096: SPX_OPT_ROW_ALIGN's limit; + the roundings of the kernel's own layout)
}

bool launch_net(const NetArgs& g, hipStream_t s) {
  if (g.row_end <= g.row_begin) return false;
  const unsigned blocks = static_cast<unsigned>(g.row_end - g.row_begin);
  size_t lds = g.n_classes > 0 ? net_lds_bytes(g.n_classes, g.n_nodes) : 16;
  if (blocks == 1) lds = ((lds + 7) & ~static_cast<size_t>(7)) + kNetStagePairs * (sizeof(long long) + 3 * sizeof(int));  // the staged pair list
  const bool generic_only = (g.opts & kOptNetGeneric) != 0;  // SPX_OPT_REFERENCE_KERNELS
  if (!generic_only && !g.out_raw && g.n_classes > 0 && g.n_classes <= 65535 && g.node_class16 && g.row_stride % 16 == 0) {
    NetArgs h = g;
    if (blocks == 1 || !h.alloc_rel) h.out_alloc = nullptr;
    hipLaunchKernelGGL(k_net_cls, dim3(blocks), dim3(blocks == 1 ? kRowThreads : 64), lds, s, h);
    return h.out_alloc != nullptr;
  }
  hipLaunchKernelGGL(k_net, dim3(blocks), dim3(64), lds, s, g);
  return false;
}

}  // namespace spx
