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
  const int lo = g.pair_ptr[key], hi = g.pair_ptr[key + 1];
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

}  // namespace

size_t net_lds_bytes(int n_classes, int64_t n_nodes) {
  return static_cast<size_t>(3 * n_classes) * sizeof(int) + static_cast<size_t>((n_nodes + 31) / 32) * sizeof(unsigned);
}

void launch_net(const NetArgs& g, hipStream_t s) {
  if (g.row_end <= g.row_begin) return;
  const unsigned blocks = static_cast<unsigned>(g.row_end - g.row_begin);
  const size_t lds = g.n_classes > 0 ? net_lds_bytes(g.n_classes, g.n_nodes) : 16;
  hipLaunchKernelGGL(k_net, dim3(blocks), dim3(64), lds, s, g);
}

}  // namespace spx
