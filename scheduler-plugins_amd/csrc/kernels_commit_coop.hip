// kernels_commit_coop.hip — the sequential commit loop for profiles WITH Filter plugins (second half)
// as ONE cooperative persistent launch.
//
// Upstream schedules one pod at a time; between two cycles the plugins' Reserve / event hooks change what the next pod sees
// (kernels_commit.hip lists them with their reference lines).  Rounds 2-3 replayed a dozen single-row launches per pod from a graph:
// 75 us per pod, kernel-time-bound — single workgroups walking dependent loads through tables in memory.  Here:
//   * a workgroup owns a window of 256 nodes (the NRT node order's window) and keeps their state in REGISTERS for the whole batch:
//     the NUMA zone tables of the float64 NRT formulation (lane = node, 128 VGPRs), TargetLoadPacking's node columns,
//     Allocatable's raw offset, the NetworkOverhead labels.  Nothing per node is re-read per pod;
//   * per pod the workgroups need to agree twice — on the minima / maxima NormalizeScore runs over each pod's feasible nodes
//     (Allocatable, NetworkOverhead), then on the weighted argmax.  Each exchange is an all-to-all of self-tagged 8-byte
//     granules (value | pod sequence << 32), written and polled with agent-scope relaxed atomics (sc1): the granule IS the
//     message, so no fence, no flag and no other shared mutable memory exists between workgroups
//     (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility": valid forms, R2);
//   * what a commit changes is applied where the state lives: the winning node's owner lane updates its NRT zones
//     (overreserve.go:170-186, store.go:315-356) and trimaran's missing utilisation (handler.go:131-139); quota usage
//     (elasticquota.go:89-98) and the AppGroup's scheduled list (networkoverhead.go:174-298) are small and every workgroup replays
//     them identically on its own copy (quota in LDS; the growing pair lists in a private global array, their ends in LDS).
// Every spin is bounded; a workgroup that gives up sets *err and all workgroups leave (the host then reports SPX_ERR_HIP).
// Results: node / weighted score / tie count / feasible count per pod, equal to the per-pod loop's (tests/test_gpu_commit.py
// compares both with the oracle's one-pod-at-a-time evaluation).
#include <cstdlib>

#include "nrt_fast_device.h"
#include "spx_internal.h"
#include "trimaran_math.h"

namespace spx {

namespace {

using namespace nrtdev;
using namespace trimath;

constexpr int S = SPX_QUOTA_SLOTS;
constexpr int RM = 4;
constexpr int kT = kCoopWindow;  // threads per workgroup = nodes per window
constexpr int kSameZone = SPX_NET_SAME_ZONE;
constexpr int kMaxCost = SPX_NET_MAX_COST;
constexpr uint32_t kSpinLimit = 4u << 20;  // polls of one granule before a workgroup gives up (seconds; a poll is ~1 us)

// ---- what one pod brings: copied into LDS two pods ahead
struct PodBlk {
  uint32_t rec[pod_words<RM>()];  // NRT record stream
  int64_t tlp_pod;                // predicted millicores
  int64_t nrt_req[RM];            // GetPodEffectiveRequest per slot
  int64_t q_req[S];
  int64_t eff_cost[kCoopMaxEffects];
  int32_t eff_key[kCoopMaxEffects];
  uint32_t nrt_present;
  uint32_t q_reqp;
  int32_t q_ns, q_prio;
  int32_t net_key, net_lo;        // workload key and the start of its pair list
  int32_t n_eff;
  int32_t pad[3];
};
static_assert(sizeof(PodBlk) % 16 == 0, "pod blocks are copied as 16-byte pieces");

struct Layout {
  uint32_t pod, pairs_host, pairs_region, pairs_zone, pairs_max, cls, hostcnt, pos, zcost, rcost, cls_rz, dyn_end, key_flag;
  uint32_t q_used, q_other, q_nomreq, q_max, q_min, q_small, total;
};
__host__ __device__ inline uint32_t up16(uint32_t v) { return (v + 15u) & ~15u; }
__host__ __device__ inline Layout make_layout(const CoopArgs& c) {
  Layout l{};
  uint32_t at = 0;
  auto take = [&](uint32_t bytes) {
    const uint32_t o = at;
    at = up16(at + bytes);
    return o;
  };
  const bool W = (c.use >> SPX_PLUGIN_NETOVERHEAD) & 1u, Q = (c.use >> SPX_PLUGIN_CAPACITY) & 1u;
  l.pod = take(3 * sizeof(PodBlk));
  if (W) {
    l.pairs_host = take(kCoopMaxPairs * 4);
    l.pairs_region = take(kCoopMaxPairs * 4);
    l.pairs_zone = take(kCoopMaxPairs * 4);
    l.pairs_max = take(kCoopMaxPairs * 8);
    l.cls = take(3u * static_cast<uint32_t>(c.net.n_classes) * 4);
    l.hostcnt = take(kT * 4);
    l.pos = take(kT * 2);
    l.zcost = take(static_cast<uint32_t>(c.net.n_zones) * static_cast<uint32_t>(c.net.n_zones) * 4);
    l.rcost = take(static_cast<uint32_t>(c.net.n_regions) * static_cast<uint32_t>(c.net.n_regions) * 4);
    l.cls_rz = take(2u * static_cast<uint32_t>(c.net.n_classes) * 4);
    l.dyn_end = take(static_cast<uint32_t>(c.net_n_keys) * 4);
    l.key_flag = take(static_cast<uint32_t>(c.net_n_keys));
  }
  if (Q) {
    const uint32_t ns = static_cast<uint32_t>(c.q_ns), nn = static_cast<uint32_t>(c.q_n_nom);
    l.q_used = take(ns * S * 8);
    l.q_other = take(ns * S * 8);
    l.q_nomreq = take((nn ? nn : 1) * S * 8);
    l.q_max = take(ns * S * 8);
    l.q_min = take(ns * S * 8);
    // small arrays: agg[9] int64 | nom_ptr[ns+1] | nom_prio[nn] | nom_pending[nn] (int32) | bytes: has, usedp, maxp, minp, otherp [ns], nom_reqp [nn]
    l.q_small = take(16 * 8 + (ns + 1 + 2 * nn) * 4 + 5 * ns + nn + 64);
  }
  l.total = at;
  return l;
}

// ---- cross-lane helpers (every lane live).  Reductions run on DPP row permutations (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror: every lane of a 16-lane row ends with the row's result) and combine the four rows on the scalar unit — a fraction of the
// latency of six ds_bpermute round trips per value, and an exchange reduces eight values twice.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_perm(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), CTRL, 0xf, 0xf, false));
}
template <typename F>
__device__ __forceinline__ uint32_t wave_reduce(uint32_t v, F f) {
  v = f(v, dpp_perm<0xB1>(v));
  v = f(v, dpp_perm<0x4E>(v));
  v = f(v, dpp_perm<0x141>(v));
  v = f(v, dpp_perm<0x140>(v));
  const uint32_t r0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 0));
  const uint32_t r1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 16));
  const uint32_t r2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 32));
  const uint32_t r3 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
  return f(f(r0, r1), f(r2, r3));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  return wave_reduce(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  return wave_reduce(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
}
__device__ __forceinline__ int wave_min_i32(int v) {
  return static_cast<int>(wave_reduce(static_cast<uint32_t>(v), [](uint32_t a, uint32_t b) { return static_cast<int>(a) < static_cast<int>(b) ? a : b; }));
}
__device__ __forceinline__ int wave_max_i32(int v) {
  return static_cast<int>(wave_reduce(static_cast<uint32_t>(v), [](uint32_t a, uint32_t b) { return static_cast<int>(a) > static_cast<int>(b) ? a : b; }));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  return static_cast<int>(wave_reduce(static_cast<uint32_t>(v), [](uint32_t a, uint32_t b) { return a + b; }));
}

// (best total, lowest node reaching it, how many nodes tie) merged the way k_best_fast merges lanes; node < 0 = none
struct Best {
  int total, node, ties;
};
__device__ __forceinline__ void merge_best(Best& a, const Best& b) {
  if (b.node < 0) return;
  if (a.node < 0 || b.total > a.total) {
    a = b;
  } else if (b.total == a.total) {
    a.ties += b.ties;
    a.node = b.node < a.node ? b.node : a.node;
  }
}
// the same over a wavefront, as three dependent reductions: the best total, then the lowest node and the tie count among its holders
__device__ __forceinline__ Best wave_best(Best v) {
  const int has = v.node >= 0;
  const int best = wave_max_i32(has ? v.total : INT32_MIN);
  const bool mine = has && v.total == best;
  Best r;
  r.total = best;
  r.node = wave_min_i32(mine ? v.node : INT32_MAX);
  r.ties = wave_sum_i32(mine ? v.ties : 0);
  if (r.node == INT32_MAX) r = Best{0, -1, 0};
  return r;
}

// ---- the granule exchange: every workgroup publishes four 32-bit values, every workgroup reads everybody's.
// granule = value | (tag << 32), tag = pod sequence + 1; slots are double-buffered by the sequence's parity (a workgroup may
// be one exchange ahead of the slowest reader of its previous granules, never two: it needs everybody's next granules first).
__device__ __forceinline__ void publish(unsigned long long* sync, int par, int kind0, int wg, uint32_t tag, int k, uint32_t value) {
  unsigned long long* slot = sync + (static_cast<size_t>(par) * kCoopKinds + static_cast<size_t>(kind0 + k)) * kCoopMaxWg + wg;
  __hip_atomic_store(slot, static_cast<unsigned long long>(value) | (static_cast<unsigned long long>(tag) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the four values workgroup `wg` published for this tag; false when it never came (bounded spin).  The four loads are in flight
// together (one round trip when everybody has published; polled one after the other they cost four)
__device__ __forceinline__ bool collect(const unsigned long long* sync, int par, int kind0, int wg, uint32_t tag, uint32_t (&v)[4]) {
  const unsigned long long* slot = sync + (static_cast<size_t>(par) * kCoopKinds + static_cast<size_t>(kind0)) * kCoopMaxWg + wg;
  uint32_t spins = 0;
  while (true) {
    const unsigned long long g0 = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long g1 = __hip_atomic_load(slot + kCoopMaxWg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long g2 = __hip_atomic_load(slot + 2 * kCoopMaxWg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long g3 = __hip_atomic_load(slot + 3 * kCoopMaxWg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (static_cast<uint32_t>(g0 >> 32) == tag && static_cast<uint32_t>(g1 >> 32) == tag && static_cast<uint32_t>(g2 >> 32) == tag &&
        static_cast<uint32_t>(g3 >> 32) == tag) {
      v[0] = static_cast<uint32_t>(g0), v[1] = static_cast<uint32_t>(g1), v[2] = static_cast<uint32_t>(g2), v[3] = static_cast<uint32_t>(g3);
      return true;
    }
    if (++spins > kSpinLimit) return false;
    __builtin_amdgcn_s_sleep(1);
  }
}

// contribution of one (scheduled pod on a host with labels hr / hz, dependency with max_cost) pair to a node with labels (region, zone)
// that is not the host — checkMaxNetworkCostRequirements networkoverhead.go:536-567 + getAccumulatedCost :605-633 (as k_net_cls)
struct Acc {
  int sat, vio, cost;
};
__device__ __forceinline__ void add_pair(Acc& a, int region, int zone, int hr, int hz, long long max_cost, const int* zcost, int n_zones, const int* rcost,
                                         int n_regions) {
  if (hr < 0 && hz < 0) {
    a.vio += 1;
    a.cost += kMaxCost;
  } else if (region == hr) {
    if (zone == hz) {
      a.sat += 1;
      a.cost += kSameZone;
    } else {
      const int cst = (zone >= 0 && hz >= 0) ? zcost[zone * n_zones + hz] : -1;
      if (cst >= 0) {
        if (cst <= max_cost) a.sat += 1;
        else a.vio += 1;
        a.cost += cst;
      } else {
        a.cost += kMaxCost;
      }
    }
  } else {
    const int cst = (region >= 0 && hr >= 0) ? rcost[region * n_regions + hr] : -1;
    if (cst >= 0) {
      if (cst <= max_cost) a.sat += 1;
      else a.vio += 1;
      a.cost += cst;
    } else {
      a.cost += kMaxCost;
    }
  }
}
__device__ __forceinline__ int norm_cost(int cost, int mn, int mx) {  // networkoverhead.go:389-418 (as k_net_cls)
  if (mn == 0 && mx == 0) return cost;
  const int range = mx - mn;
  return range != 0 ? 100 - (100 * (cost - mn)) / range : 100 - (cost - mn);
}

__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) { return static_cast<int64_t>(static_cast<uint64_t>(a) + static_cast<uint64_t>(b)); }

template <int SG>
__global__ __launch_bounds__(kT) void k_commit_coop(CoopArgs c) {
  extern __shared__ __align__(16) unsigned char lds[];
  __shared__ uint32_t s_red[4][kT / 64];                    // a workgroup's own partials on their way to its granules
  __shared__ uint32_t s_x1[4][kT / 64], s_x2[4][kT / 64];  // the waves' shares of the two exchanges' results (separate arrays: no barrier between
                                                           // a slow reader of one exchange and the writers of the next)
  __shared__ uint32_t s_all[8];
  __shared__ int s_flag_misc[8];  // [0] pod_ok, [1] q_other needs a rebuild, [2] this pod's network flag, [3] staged pair count, [4] give up
  const Layout L = make_layout(c);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const bool A = (c.use >> SPX_PLUGIN_ALLOCATABLE) & 1u, Tl = (c.use >> SPX_PLUGIN_TLP) & 1u, Lv = (c.use >> SPX_PLUGIN_LVRB) & 1u;
  const bool Nr = (c.use >> SPX_PLUGIN_NRT) & 1u, W = (c.use >> SPX_PLUGIN_NETOVERHEAD) & 1u, Q = (c.use >> SPX_PLUGIN_CAPACITY) & 1u;
  const NrtArgs& a = c.nrt;
  const int64_t base = static_cast<int64_t>(wg) * kT;

  // ---- this lane's node
  const int32_t pn = Nr ? a.perm[base + tid] : (base + tid < c.n_nodes ? static_cast<int32_t>(base + tid) : -1);
  const bool in = pn >= 0;
  const int64_t n = in ? pn : 0;
  FastNode<RM> ns;
  double cpu_v[kZ], braw[kZ];
  uint32_t nflags = 0;
  if (Nr) {
    nflags = in ? a.flags[n] : 0u;
    load_fast_node<RM, SG>(a, n, in, ns, cpu_v, braw);
  }
  const bool fresh = nflags & SPX_NRT_F_FRESH, has_nrt = nflags & SPX_NRT_F_HAS_NRT, single = nflags & SPX_NRT_F_SINGLE_NUMA;
  const bool pod_scope = nflags & SPX_NRT_F_POD_SCOPE;
  const bool aligned = fresh && has_nrt && single;
  const bool w_pod = __ballot(aligned && pod_scope) != 0, w_ctr = __ballot(aligned && !pod_scope) != 0;
  const uint32_t st_stale = fresh ? 0u : static_cast<uint32_t>(SPX_NRT_ST_INVALID_TOPOLOGY);
  // TargetLoadPacking: the node's columns as the reference reads them (one cell per lane and pod: the float64 sequence itself)
  TlpNode tn{0.0, 0.0, 0.0, false};
  int64_t missing_i = 0;
  if (Tl && in) {
    tn.cap = static_cast<double>(c.t.cap_cpu_milli[n]);
    tn.util_millis = (c.t.tlp_cpu_util[n] / 100.0) * tn.cap;
    missing_i = c.t.tlp_missing_milli[n];
    tn.missing = static_cast<double>(missing_i);
    tn.valid = c.t.tlp_valid[n] != 0;
  }
  const uint32_t rel = (A && in) ? c.alloc_rel[n] : 0u;
  const int my_region = (W && in) ? c.net.region[n] : -1, my_zone = (W && in) ? c.net.zone[n] : -1;
  const int my_class = (W && in) ? c.net.node_class[n] : 0;

  // ---- LDS areas
  PodBlk* blk = reinterpret_cast<PodBlk*>(lds + L.pod);
  int* p_host = reinterpret_cast<int*>(lds + L.pairs_host);
  int* p_region = reinterpret_cast<int*>(lds + L.pairs_region);
  int* p_zone = reinterpret_cast<int*>(lds + L.pairs_zone);
  long long* p_max = reinterpret_cast<long long*>(lds + L.pairs_max);
  const int C = W ? c.net.n_classes : 0;
  int* cls_sat = reinterpret_cast<int*>(lds + L.cls);
  int* cls_vio = cls_sat + C;
  int* cls_cost = cls_vio + C;
  int* hostcnt = reinterpret_cast<int*>(lds + L.hostcnt);
  int16_t* pos_of = reinterpret_cast<int16_t*>(lds + L.pos);
  int* zcost = reinterpret_cast<int*>(lds + L.zcost);
  int* rcost = reinterpret_cast<int*>(lds + L.rcost);
  int* cls_region = reinterpret_cast<int*>(lds + L.cls_rz);
  int* cls_zone = cls_region + C;
  int* dyn_end = reinterpret_cast<int*>(lds + L.dyn_end);
  uint8_t* key_flag = lds + L.key_flag;
  int64_t* q_used = reinterpret_cast<int64_t*>(lds + L.q_used);
  int64_t* q_other = reinterpret_cast<int64_t*>(lds + L.q_other);
  int64_t* q_nomreq = reinterpret_cast<int64_t*>(lds + L.q_nomreq);
  int64_t* q_max = reinterpret_cast<int64_t*>(lds + L.q_max);
  int64_t* q_min = reinterpret_cast<int64_t*>(lds + L.q_min);
  const int NS = Q ? c.q_ns : 0, NN = Q ? c.q_n_nom : 0;
  int64_t* q_agg = reinterpret_cast<int64_t*>(lds + L.q_small);  // [9], then room to 16
  int32_t* q_nom_ptr = reinterpret_cast<int32_t*>(q_agg + 16);
  int32_t* q_nom_prio = q_nom_ptr + NS + 1;
  int32_t* q_nom_pending = q_nom_prio + NN;
  uint8_t* q_has = reinterpret_cast<uint8_t*>(q_nom_pending + NN);
  uint8_t* q_usedp = q_has + NS;
  uint8_t* q_maxp = q_usedp + NS;
  uint8_t* q_minp = q_maxp + NS;
  uint8_t* q_otherp = q_minp + NS;
  uint8_t* q_nomreqp = q_otherp + NS;

  int32_t* priv_node = W ? c.net_priv_node + static_cast<int64_t>(wg) * c.net_cap : nullptr;
  int64_t* priv_max = W ? c.net_priv_max + static_cast<int64_t>(wg) * c.net_cap : nullptr;

  // ---- prologue: the replicated state
  if (W) {
    for (int i = tid; i < c.net.n_zones * c.net.n_zones; i += kT) zcost[i] = c.net.zone_cost[i];
    for (int i = tid; i < c.net.n_regions * c.net.n_regions; i += kT) rcost[i] = c.net.region_cost[i];
    for (int i = tid; i < C; i += kT) cls_region[i] = c.net.cls_region[i], cls_zone[i] = c.net.cls_zone[i];
    for (int i = tid; i < c.net_n_keys; i += kT) dyn_end[i] = c.net_init_end[i], key_flag[i] = c.net_init_flag[i];
    for (int64_t i = tid; i < c.net_cap; i += kT) priv_node[i] = c.net_init_node[i], priv_max[i] = c.net_init_max[i];
    pos_of[tid] = -1;
  }
  if (Q) {
    for (int i = tid; i < NS * S; i += kT) q_used[i] = c.q_used[i], q_other[i] = c.q_other[i], q_max[i] = c.q_max[i], q_min[i] = c.q_min[i];
    for (int i = tid; i < NN * S; i += kT) q_nomreq[i] = c.q_nom_req[i];
    for (int i = tid; i < S + 1; i += kT) q_agg[i] = c.q_agg[i];
    for (int i = tid; i < NS + 1; i += kT) q_nom_ptr[i] = c.q_nom_ptr[i];
    for (int i = tid; i < NN; i += kT) {
      q_nom_prio[i] = c.q_nom_prio[i];
      const int64_t pend = c.q_nom_pending[i];
      q_nom_pending[i] = (pend < 0 || pend > INT32_MAX) ? -1 : static_cast<int32_t>(pend);
      q_nomreqp[i] = c.q_nom_reqp[i];
    }
    for (int i = tid; i < NS; i += kT)
      q_has[i] = c.q_has[i], q_usedp[i] = c.q_usedp[i], q_maxp[i] = c.q_maxp[i], q_minp[i] = c.q_minp[i], q_otherp[i] = c.q_otherp[i];
  }
  if (tid < 8) s_flag_misc[tid] = 0;
  __syncthreads();
  if (W && in && n >= base && n < base + kT) pos_of[n - base] = static_cast<int16_t>(tid);  // node of the window -> its lane (the NRT order permutes inside the window)

  // the pod block of row `row` into slot row % 3: 16-byte pieces spread over the threads
  auto issue_block = [&](int64_t row, uint4 (&r)[2]) {
    // what a thread fetches: piece tid of the record (pod_words / 4 = 40 pieces), the small fields by the threads behind them
    r[0] = uint4{0, 0, 0, 0};
    r[1] = uint4{0, 0, 0, 0};
    if (row >= c.row_end) return;
    constexpr int kRecQuads = pod_words<RM>() / 4;
    if (Nr && tid < kRecQuads) r[0] = reinterpret_cast<const uint4*>(a.pod_items + row * pod_words<RM>())[tid];
    if (tid == 64) {
      const int64_t v = Tl ? c.t.tlp_pod_milli[row] : 0;
      r[0].x = static_cast<uint32_t>(v), r[0].y = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
      if (Nr) r[0].z = a.pod_present[row];
      if (Q) r[0].w = c.q_pod_reqp[row];
      if (Q) r[1].x = static_cast<uint32_t>(c.q_pod_ns[row]), r[1].y = static_cast<uint32_t>(c.q_pod_prio[row]);
      if (W) {
        const int key = c.net.pod_key[row];
        r[1].z = static_cast<uint32_t>(key);
        r[1].w = static_cast<uint32_t>(c.net.pair_ptr[key]);
      }
    }
    if (Nr && tid >= 65 && tid < 65 + RM && tid - 65 < a.n_res) {
      const int64_t v = a.pod_req[row * a.n_res + (tid - 65)];
      r[0].x = static_cast<uint32_t>(v), r[0].y = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
    }
    if (Q && tid >= 72 && tid < 72 + S) {
      const int64_t v = c.q_pod_req[row * S + (tid - 72)];
      r[0].x = static_cast<uint32_t>(v), r[0].y = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
    }
    if (W && tid >= 96 && tid < 96 + kCoopMaxEffects) {
      const int lo = c.eff_ptr[row], hi = c.eff_ptr[row + 1];
      const int e = lo + (tid - 96);
      r[1].x = static_cast<uint32_t>(hi - lo);
      if (e < hi) {
        const int64_t v = c.eff_cost[e];
        r[0].x = static_cast<uint32_t>(v), r[0].y = static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32);
        r[0].z = static_cast<uint32_t>(c.eff_key[e]);
      }
    }
  };
  auto land_block = [&](int64_t row, const uint4 (&r)[2]) {
    if (row >= c.row_end) return;
    PodBlk& b = blk[row % 3];
    constexpr int kRecQuads = pod_words<RM>() / 4;
    auto i64 = [&](const uint4& q) { return static_cast<int64_t>(static_cast<uint64_t>(q.x) | (static_cast<uint64_t>(q.y) << 32)); };
    if (Nr && tid < kRecQuads) reinterpret_cast<uint4*>(b.rec)[tid] = r[0];
    if (tid == 64) {
      b.tlp_pod = i64(r[0]);
      b.nrt_present = r[0].z;
      b.q_reqp = r[0].w;
      b.q_ns = static_cast<int32_t>(r[1].x), b.q_prio = static_cast<int32_t>(r[1].y);
      b.net_key = static_cast<int32_t>(r[1].z), b.net_lo = static_cast<int32_t>(r[1].w);
      if (!W) b.n_eff = 0;
    }
    if (tid >= 65 && tid < 65 + RM) b.nrt_req[tid - 65] = (Nr && tid - 65 < a.n_res) ? i64(r[0]) : 0;
    if (Q && tid >= 72 && tid < 72 + S) b.q_req[tid - 72] = i64(r[0]);
    if (W && tid >= 96 && tid < 96 + kCoopMaxEffects) {
      if (tid == 96) b.n_eff = static_cast<int32_t>(r[1].x);
      b.eff_cost[tid - 96] = i64(r[0]);
      b.eff_key[tid - 96] = static_cast<int32_t>(r[0].z);
    }
  };
  // NetworkOverhead: the pod's pair list -> LDS (host, its labels, MaxNetworkCost), the class table over it, and how many pairs sit
  // on each node of this window (a node that hosts a pair counts that pair differently: networkoverhead.go:536-544).  In three
  // steps so that the two dependent rounds of global loads (the list, then the hosts' labels) fly while the quota verdict and the
  // NRT evaluation run: stage_issue -> stage_labels -> stage_land.
  constexpr int kSlots = kCoopMaxPairs / kT;  // staged pairs per thread
  int st_host[kSlots], st_region[kSlots], st_zone[kSlots], st_np = 0;
  long long st_max[kSlots];
  auto stage_issue = [&](const PodBlk& b) {
    const int lo = b.net_lo;
    st_np = dyn_end[b.net_key] - lo;  // (<= kCoopMaxPairs: checked by the launcher over the batch's final lists)
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      const int i = tid + q * kT;
      st_host[q] = 0, st_max[q] = 0;
      if (i < st_np) st_host[q] = priv_node[lo + i], st_max[q] = priv_max[lo + i];
    }
    hostcnt[tid] = 0;
    for (int i = tid; i < 3 * C; i += kT) cls_sat[i] = 0;  // (sat | vio | cost are one array)
  };
  auto stage_labels = [&]() {
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      st_region[q] = -1, st_zone[q] = -1;
      if (tid + q * kT < st_np) st_region[q] = c.net.region[st_host[q]], st_zone[q] = c.net.zone[st_host[q]];
    }
  };
  // classes x pairs spread over the workgroup: slice q of the threads takes every n_slices-th pair of its class
  const int n_slices = C > 0 ? (kT / C > 0 ? kT / C : 1) : 1;
  auto stage_land = [&]() {
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      const int i = tid + q * kT;
      if (i < st_np) {
        const int host = st_host[q];
        p_host[i] = host, p_region[i] = st_region[q], p_zone[i] = st_zone[q], p_max[i] = st_max[q];
        if (host >= base && host < base + kT && pos_of[host - base] >= 0) atomicAdd(&hostcnt[pos_of[host - base]], 1);
      }
    }
    __syncthreads();
    for (int u = tid; u < C * n_slices; u += kT) {
      const int cl = u % C, q = u / C;
      Acc acc{0, 0, 0};
      const int region = cls_region[cl], zone = cls_zone[cl];
      for (int i = q; i < st_np; i += n_slices) add_pair(acc, region, zone, p_region[i], p_zone[i], p_max[i], zcost, c.net.n_zones, rcost, c.net.n_regions);
      if (acc.sat) atomicAdd(&cls_sat[cl], acc.sat);
      if (acc.vio) atomicAdd(&cls_vio[cl], acc.vio);
      if (acc.cost) atomicAdd(&cls_cost[cl], acc.cost);
    }
    __syncthreads();
  };

  uint4 pre[2];
  issue_block(c.row_begin, pre);
  land_block(c.row_begin, pre);
  issue_block(c.row_begin + 1, pre);
  land_block(c.row_begin + 1, pre);
  __syncthreads();

#ifdef SPX_COOP_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = wall_clock64();
#define SPX_MARK(i) do { const unsigned long long t_now = wall_clock64(); prof[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define SPX_MARK(i) do { } while (0)
#endif
  uint8_t lv_next = (Lv && in) ? c.lv_table[c.row_begin * c.row_stride + n] : 0;
  const double t_tlp = c.t.tlp_target;
  bool dead = false;

  for (int64_t pod = c.row_begin; pod < c.row_end; ++pod) {
    const uint32_t seq = static_cast<uint32_t>(pod - c.row_begin);
    const int par = static_cast<int>(seq & 1u);
    const uint32_t tag = seq + 1u;
    const PodBlk& b = blk[pod % 3];
    issue_block(pod + 2, pre);  // lands at the end of this pod
    const uint32_t lv_byte = lv_next;
    if (Lv && in && pod + 1 < c.row_end) lv_next = c.lv_table[(pod + 1) * c.row_stride + n];

    if (W) stage_issue(b);
    // ---- CapacityScheduling.PreFilter on the replicated quota state (capacity_scheduling.go:208-283; as k_quota), one lane per
    // resource slot (a single thread walking ~100 dependent LDS reads cost 1.9 us per pod)
    if (Q && wave == 3 && lane < S) {
      const int s = lane;
      const int nsid = b.q_ns;
      int status = 0;
      if (nsid >= 0 && nsid < NS && q_has[nsid]) {
        int64_t in_eq = b.q_req[s];
        uint32_t in_p = b.q_reqp;
        for (int j = q_nom_ptr[nsid]; j < q_nom_ptr[nsid + 1]; ++j) {
          if (q_nom_pending[j] == static_cast<int32_t>(pod) || q_nom_prio[j] < b.q_prio) continue;
          in_eq = wadd(in_eq, q_nomreq[j * S + s]);
          in_p |= q_nomreqp[j];
        }
        const int64_t with_used = wadd(in_eq, q_used[nsid * S + s]);
        const int64_t ymax = (s < 4 || ((q_maxp[nsid] >> s) & 1u)) ? q_max[nsid * S + s] : INT64_MAX;
        const bool over = (s < 4 || ((in_p >> s) & 1u)) && with_used > ymax;
        if (__ballot(over) != 0) {
          status = SPX_QUOTA_ST_OVER_MAX;
        } else {
          const uint32_t agg_p = static_cast<uint32_t>(q_agg[S]) | in_p | q_otherp[nsid];
          const int64_t agg = wadd(wadd(q_agg[s], in_eq), q_other[nsid * S + s]);
          const int64_t ymin = (s < 4 || ((c.q_agg_min_present >> s) & 1u)) ? c.q_agg_min[s] : 0;
          const bool over_min = (s < 4 || ((agg_p >> s) & 1u)) && agg > ymin;
          if (__ballot(over_min) != 0) status = SPX_QUOTA_ST_OVER_MIN;
        }
      }
      if (s == 0) s_flag_misc[0] = status;
    }
    if (W) stage_labels();
    SPX_MARK(0);
    __syncthreads();  // the verdict above; the staging areas zeroed by stage_issue
    SPX_MARK(1);
    const bool pod_ok = !Q || s_flag_misc[0] == 0;
    const int net_flag = W ? key_flag[b.net_key] : 0;

    Best gbest{0, -1, 0};
    int gfeas = 0;
    if (pod_ok) {
      // ---- NodeResourceTopologyMatch Filter + Score for this node (filter.go:42-245, score.go:62-191; the pod loop body of k_nrt_fast)
      uint32_t nrt_status = 0;
      int nrt_score = 0;
      if (Nr) {
        const uint32_t* pit = b.rec;
        const uint32_t h0 = pit[0], inv_n = pit[1];
        const int qos = h0 & 0xffu;
        const bool non_native = ((h0 >> 8) & 0xffu) != 0;
        const int n_ctr = (h0 >> 16) & 0xffu;
        const int last_app = static_cast<int>(h0 >> 24) == 0xff ? -1 : static_cast<int>(h0 >> 24);
        const bool non_g = qos != SPX_QOS_GUARANTEED;
        const bool filtered = !(qos == SPX_QOS_BESTEFFORT && !non_native);
        nrt_status = filtered ? st_stale : 0u;
        nrt_score = non_g ? 100 : 0;
        const bool u_filter = filtered, u_score = !non_g;
        if (u_filter || u_score) {
          const bool want_filter = u_filter && aligned, want_score = u_score && aligned;
          if (w_pod && pod_scope && aligned) {
            const Item<RM> it = decode_item<RM, true>(load_item<RM, true>(pit, 1));
            if (want_filter) {
              uint32_t pos;
              if (!fits_fast(ns, it, &pos)) nrt_status = SPX_NRT_ST_POD;
            }
            if (want_score) nrt_score = score_each_fast<RM, SG>(ns, a, it, cpu_v, braw);
          }
          if (w_ctr && !pod_scope && aligned) {
            uint32_t chosen = 0;
            int sum = 0;
            for (int ct = 0; ct < n_ctr; ++ct) {
              const Item<RM> it = decode_item<RM, true>(load_item<RM, true>(pit, 2 + ct));
              if (want_filter) {
                uint32_t pos;
                const bool ok = fits_fast(ns, it, &pos);
                const bool live = nrt_status == 0;
                if (it.kind != SPX_CTR_APP) {
                  if (live && !ok) nrt_status = it.kind == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER;
                } else {
                  if (live && !ok) nrt_status = SPX_NRT_ST_CONTAINER;
                  if (ct != last_app) {
                    const bool apply = live && ok;
                    adjust_fast(ns, it, pos, apply, -1.0);
                    chosen |= (apply ? pos + 1u : 0u) << (4 * ct);
                  }
                }
              }
              if (want_score) sum += score_each_fast<RM, SG>(ns, a, it, cpu_v, braw);
            }
            if (want_filter && last_app > 0) {  // undo: Filter works on a private copy in the reference
              for (int ct = 0; ct < last_app; ++ct) {
                const Item<RM> it = decode_item<RM, true>(load_item<RM, true>(pit, 2 + ct));
                if (it.kind == SPX_CTR_APP) adjust_fast(ns, it, ((chosen >> (4 * ct)) & 0xfu) - 1u, ((chosen >> (4 * ct)) & 0xfu) != 0, 1.0);
              }
            }
            if (want_score) nrt_score = static_cast<int>((static_cast<uint32_t>(sum) * inv_n) >> 16);
          }
        }
      }
      // ---- NetworkOverhead: accumulated cost / satisfied / violated of this node from the class table
      if (W) stage_land();
      Acc na{0, 0, 0};
      bool net_pass = true;
      if (W) {
        if (net_flag == 0) {
          if (in) {
            na = Acc{cls_sat[my_class], cls_vio[my_class], cls_cost[my_class]};
            const int own = hostcnt[tid];
            if (own != 0) {  // pairs hosted here: same hostname -> satisfied at cost 0 (the class table counted them as a neighbour's)
              if (my_region < 0 && my_zone < 0) na.vio -= own, na.sat += own, na.cost -= own * kMaxCost;
              else na.cost -= own * kSameZone;
            }
          }
          net_pass = !(na.vio > na.sat);
        } else {
          net_pass = net_flag != 2;  // PreFilter error: every node rejected; scoreEqually: all pass with score 0
        }
      }
      const bool feasible = in && nrt_status == 0 && net_pass;
      SPX_MARK(2);
      // ---- first exchange: what NormalizeScore runs over (allocatable.go:143-168, networkoverhead.go:389-418)
      {
        uint32_t lo = feasible ? rel : 0xffffffffu, hi = feasible ? rel : 0u;
        int mn = (feasible && net_flag == 0) ? na.cost : INT32_MAX, mx = (feasible && net_flag == 0) ? na.cost : INT32_MIN;
        lo = wave_min_u32(lo), hi = wave_max_u32(hi), mn = wave_min_i32(mn), mx = wave_max_i32(mx);
        if (c.n_wg == 1) {  // one window: the waves' partials ARE the result (a granule hop costs 1.6 us even to oneself)
          if (lane == 0) s_x1[0][wave] = lo, s_x1[1][wave] = hi, s_x1[2][wave] = static_cast<uint32_t>(mn), s_x1[3][wave] = static_cast<uint32_t>(mx);
          __syncthreads();
        } else {
        if (lane == 0) s_red[0][wave] = lo, s_red[1][wave] = hi, s_red[2][wave] = static_cast<uint32_t>(mn), s_red[3][wave] = static_cast<uint32_t>(mx);
        __syncthreads();
        if (tid < 4) {
          uint32_t v = s_red[tid][0];
          for (int wv = 1; wv < kT / 64; ++wv) {
            const uint32_t o = s_red[tid][wv];
            if (tid == 0) v = o < v ? o : v;
            else if (tid == 1) v = o > v ? o : v;
            else if (tid == 2) v = static_cast<int>(o) < static_cast<int>(v) ? o : v;
            else v = static_cast<int>(o) > static_cast<int>(v) ? o : v;
          }
          publish(c.sync, par, 0, wg, tag, tid, v);
        }
        uint32_t v4[4] = {0xffffffffu, 0u, static_cast<uint32_t>(INT32_MAX), static_cast<uint32_t>(INT32_MIN)};
        bool ok = true;
        if (tid < c.n_wg) ok = collect(c.sync, par, 0, tid, tag, v4);
        lo = wave_min_u32(v4[0]), hi = wave_max_u32(v4[1]), mn = wave_min_i32(static_cast<int>(v4[2])), mx = wave_max_i32(static_cast<int>(v4[3]));
        const bool wok = __ballot(!ok) == 0;
        if (lane == 0) s_x1[0][wave] = lo, s_x1[1][wave] = hi, s_x1[2][wave] = static_cast<uint32_t>(mn), s_x1[3][wave] = static_cast<uint32_t>(mx);
        if (lane == 0 && !wok) s_flag_misc[4] = 1;
        __syncthreads();
        }
      }
      if (s_flag_misc[4]) {
        dead = true;
        break;
      }
      SPX_MARK(3);
      uint32_t g_lo = s_x1[0][0], g_hi = s_x1[1][0];
      int g_mn = static_cast<int>(s_x1[2][0]), g_mx = static_cast<int>(s_x1[3][0]);
#pragma unroll
      for (int wv = 1; wv < kT / 64; ++wv) {
        g_lo = s_x1[0][wv] < g_lo ? s_x1[0][wv] : g_lo;
        g_hi = s_x1[1][wv] > g_hi ? s_x1[1][wv] : g_hi;
        g_mn = static_cast<int>(s_x1[2][wv]) < g_mn ? static_cast<int>(s_x1[2][wv]) : g_mn;
        g_mx = static_cast<int>(s_x1[3][wv]) > g_mx ? static_cast<int>(s_x1[3][wv]) : g_mx;
      }
      // ---- this node's weighted total
      int total = 0;
      if (feasible) {
        if (A) {
          const uint32_t range = g_hi >= g_lo ? g_hi - g_lo : 0u;
          const double bq = range ? (100.0 / static_cast<double>(range)) * (1.0 + 0x1p-49) : 0.0;  // as k_decide_masked
          total += range ? c.w[SPX_PLUGIN_ALLOCATABLE] * static_cast<int>(static_cast<uint32_t>(static_cast<double>(rel - g_lo) * bq)) : 0;
        }
        if (Tl) {
          bool zero;
          const double x = tlp_unrounded(tn, static_cast<double>(b.tlp_pod), t_tlp, &zero);
          total += c.w[SPX_PLUGIN_TLP] * static_cast<int>(zero ? 0u : to_u8(x));
        }
        if (Lv) total += c.w[SPX_PLUGIN_LVRB] * static_cast<int>(lv_byte);
        if (Nr) total += c.w[SPX_PLUGIN_NRT] * (nrt_score > 255 ? 255 : nrt_score);
        if (W && net_flag == 0) {
          int sc = norm_cost(na.cost, g_mn, g_mx);
          sc = sc < 0 ? 0 : (sc > 255 ? 255 : sc);
          total += c.w[SPX_PLUGIN_NETOVERHEAD] * sc;
        }
      }
      // ---- second exchange: the argmax (lowest node among equals, tie count, feasible count)
      {
        Best mine{total, feasible ? static_cast<int>(n) : -1, feasible ? 1 : 0};
        mine = wave_best(mine);
        const int feas = wave_sum_i32(feasible ? 1 : 0);
        if (c.n_wg == 1) {
          if (lane == 0) s_x2[0][wave] = static_cast<uint32_t>(mine.total), s_x2[1][wave] = static_cast<uint32_t>(mine.node), s_x2[2][wave] = static_cast<uint32_t>(mine.ties),
                         s_x2[3][wave] = static_cast<uint32_t>(feas);
          __syncthreads();
        } else {
        if (lane == 0) s_red[0][wave] = static_cast<uint32_t>(mine.total), s_red[1][wave] = static_cast<uint32_t>(mine.node), s_red[2][wave] = static_cast<uint32_t>(mine.ties),
                       s_red[3][wave] = static_cast<uint32_t>(feas);
        __syncthreads();
        if (tid == 0) {
          Best bb{static_cast<int>(s_red[0][0]), static_cast<int>(s_red[1][0]), static_cast<int>(s_red[2][0])};
          int f = static_cast<int>(s_red[3][0]);
          for (int wv = 1; wv < kT / 64; ++wv) {
            merge_best(bb, Best{static_cast<int>(s_red[0][wv]), static_cast<int>(s_red[1][wv]), static_cast<int>(s_red[2][wv])});
            f += static_cast<int>(s_red[3][wv]);
          }
          publish(c.sync, par, 4, wg, tag, 0, static_cast<uint32_t>(bb.total));
          publish(c.sync, par, 4, wg, tag, 1, static_cast<uint32_t>(bb.node));
          publish(c.sync, par, 4, wg, tag, 2, static_cast<uint32_t>(bb.ties));
          publish(c.sync, par, 4, wg, tag, 3, static_cast<uint32_t>(f));
        }
        uint32_t v4[4] = {0u, 0xffffffffu, 0u, 0u};
        bool ok = true;
        if (tid < c.n_wg) ok = collect(c.sync, par, 4, tid, tag, v4);
        Best theirs{static_cast<int>(v4[0]), static_cast<int>(v4[1]), static_cast<int>(v4[2])};
        theirs = wave_best(theirs);
        const int feas_all = wave_sum_i32(static_cast<int>(v4[3]));
        const bool wok = __ballot(!ok) == 0;
        if (lane == 0) s_x2[0][wave] = static_cast<uint32_t>(theirs.total), s_x2[1][wave] = static_cast<uint32_t>(theirs.node), s_x2[2][wave] = static_cast<uint32_t>(theirs.ties),
                       s_x2[3][wave] = static_cast<uint32_t>(feas_all);
        if (lane == 0 && !wok) s_flag_misc[4] = 1;
        __syncthreads();
        }
      }
      if (s_flag_misc[4]) {
        dead = true;
        break;
      }
      SPX_MARK(4);
      gbest = Best{static_cast<int>(s_x2[0][0]), static_cast<int>(s_x2[1][0]), static_cast<int>(s_x2[2][0])};
      gfeas = static_cast<int>(s_x2[3][0]);
#pragma unroll
      for (int wv = 1; wv < kT / 64; ++wv) {
        merge_best(gbest, Best{static_cast<int>(s_x2[0][wv]), static_cast<int>(s_x2[1][wv]), static_cast<int>(s_x2[2][wv])});
        gfeas += static_cast<int>(s_x2[3][wv]);
      }
    }
    const int win = gbest.node;
    if (wg == 0 && tid == 0) {
      c.best_node[pod] = win;
      c.best_score[pod] = win >= 0 ? gbest.total : 0;
      c.best_ties[pod] = win >= 0 ? gbest.ties : 0;
      c.best_feasible[pod] = gfeas;
    }

    // ---- Reserve: what binding the pod to `win` changes
    if (win >= 0) {
      if (in && n == win) {  // the owner lane
        if (Tl) {
          missing_i += b.tlp_pod;  // handler.go:131-139 -> targetloadpacking.go:151-168
          tn.missing = static_cast<double>(missing_i);
        }
        if (Nr && has_nrt) {  // OverReserve.ReserveNodeResources -> resourceStore.UpdateNRT: every zone that reports the resource
#pragma unroll
          for (int r = 0; r < RM; ++r) {
            if (!((b.nrt_present >> r) & 1u) || r >= a.n_res) continue;
            const int64_t qty = b.nrt_req[r];
            if (qty == 0) continue;  // nothing to subtract: the derived cells stay as they are
            const bool is_cpu = r == a.cpu_slot;
#pragma unroll
            for (int z = 0; z < kZ; ++z) {
              if (!(ns.av[z][r] >= 0.0)) continue;  // not reported by the zone
              const int64_t have = static_cast<int64_t>(ns.av[z][r]);
              const int64_t av = have < qty ? 0 : have - qty;  // store.go:335-351
              const double cap_v = static_cast<double>(is_cpu ? (av + 999) / 1000 : av);
              ns.av[z][r] = static_cast<double>(av);
              const double rc = cap_v > 0.0 ? 100.0 / cap_v : kNoCap;
              ns.b[z][r] = (SG == kSgLeast && rc == kNoCap) ? __builtin_inf() : rc;
              if (is_cpu && SG == kSgMost) braw[z] = av > 0 ? 100.0 / static_cast<double>(av) : kNoCap;
            }
          }
        }
      }
      if (Q) {  // reserveResource elasticquota.go:89-98; a bound pod that was nominated leaves the nominator
        const int nsid = b.q_ns;
        const bool counted = nsid >= 0 && nsid < NS && q_has[nsid];
        if (wave == 3 && lane < S) {  // one lane per resource slot
          const int s = lane;
          int rebuild = 0;
          if (counted) {
            auto slot_over_min = [&]() {
              const int64_t ymin = (s < 4 || ((q_minp[nsid] >> s) & 1u)) ? q_min[nsid * S + s] : 0;
              return (s < 4 || ((q_usedp[nsid] >> s) & 1u)) && q_used[nsid * S + s] > ymin;
            };
            const bool before = __ballot(slot_over_min()) != 0;
            q_used[nsid * S + s] = wadd(q_used[nsid * S + s], b.q_req[s]);
            q_agg[s] = wadd(q_agg[s], b.q_req[s]);
            if (s == 0) {
              q_usedp[nsid] = static_cast<uint8_t>(q_usedp[nsid] | b.q_reqp);
              q_agg[S] |= static_cast<int64_t>(b.q_reqp);
            }
            bool was_nominated = false;
            for (int j = q_nom_ptr[nsid]; j < q_nom_ptr[nsid + 1]; ++j)
              if (q_nom_pending[j] == static_cast<int32_t>(pod)) {
                q_nomreq[j * S + s] = 0;
                if (s == 0) q_nomreqp[j] = 0;
                was_nominated = true;
              }
            // "nominated requests of the other namespaces whose quota is not over min" changes only when this namespace's own
            // contribution does: its over-min status flipped, or one of its nominated pods just left
            const bool after = __ballot(slot_over_min()) != 0;  // (reads what the eight lanes just wrote: one wave, LDS in order)
            rebuild = (was_nominated || before != after) ? 1 : 0;
          }
          if (s == 0) s_flag_misc[1] = rebuild;
        }
        __syncthreads();
        if (s_flag_misc[1]) {  // as k_commit_apply: own[m] per namespace, their total, then total - own[k]
          __shared__ int64_t q_total[S];
          __shared__ int q_holders[S];
          if (tid < S) q_total[tid] = 0, q_holders[tid] = 0;
          __syncthreads();
          for (int m = tid; m < NS; m += kT) {
            int64_t own[S] = {0};
            uint32_t ownp = 0;
            bool over = false;
            for (int s = 0; s < 4; ++s) over |= q_used[m * S + s] > q_min[m * S + s];
            for (int s = 4; s < S; ++s) {
              const int64_t yq = ((q_minp[m] >> s) & 1u) ? q_min[m * S + s] : 0;
              over |= ((q_usedp[m] >> s) & 1u) && q_used[m * S + s] > yq;
            }
            if (q_has[m] && !over)
              for (int j = q_nom_ptr[m]; j < q_nom_ptr[m + 1]; ++j) {
                for (int s = 0; s < S; ++s) own[s] = wadd(own[s], q_nomreq[j * S + s]);
                ownp |= q_nomreqp[j];
              }
            for (int s = 0; s < S; ++s) {
              q_other[m * S + s] = own[s];
              if (own[s]) atomicAdd(reinterpret_cast<unsigned long long*>(&q_total[s]), static_cast<unsigned long long>(own[s]));
              if ((ownp >> s) & 1u) atomicAdd(&q_holders[s], 1);
            }
            q_otherp[m] = static_cast<uint8_t>(ownp);
          }
          __syncthreads();
          for (int m = tid; m < NS; m += kT) {
            const uint32_t ownp = q_otherp[m];
            uint32_t others = 0;
            for (int s = 0; s < S; ++s) {
              const int64_t own = q_other[m * S + s];
              q_other[m * S + s] = static_cast<int64_t>(static_cast<uint64_t>(q_total[s]) - static_cast<uint64_t>(own));
              if (q_holders[s] - static_cast<int>((ownp >> s) & 1u) > 0) others |= 1u << s;
            }
            q_otherp[m] = static_cast<uint8_t>(others);
          }
        }
      }
      if (W && tid < b.n_eff) {  // the pod joins its AppGroup's scheduled list (as k_commit_apply, on this workgroup's lists)
        const int k = b.eff_key[tid];
        if (key_flag[k] == 1) key_flag[k] = 0;  // (several effects may name one key: they all write 0)
        const int64_t cost = b.eff_cost[tid];
        if (cost >= 0) {
          const int at = atomicAdd(&dyn_end[k], 1);
          priv_node[at] = win;
          priv_max[at] = cost;
        }
      }
    }
    SPX_MARK(5);
    land_block(pod + 2, pre);
    __syncthreads();  // the pod block two ahead, the quota state and the grown lists are in place
    SPX_MARK(6);
  }
#ifdef SPX_COOP_PROF
  if (tid == 0 && (wg == 0 || wg == c.n_wg - 1))
    printf("wg %d: per pod, 10 ns units: quota %llu | stage %llu | eval %llu | exchange1 %llu | total+exchange2 %llu | reserve %llu | land %llu\n", wg,
           prof[0] / (c.row_end - c.row_begin), prof[1] / (c.row_end - c.row_begin), prof[2] / (c.row_end - c.row_begin), prof[3] / (c.row_end - c.row_begin),
           prof[4] / (c.row_end - c.row_begin), prof[5] / (c.row_end - c.row_begin), prof[6] / (c.row_end - c.row_begin));
#endif

  if (dead) {
    if (tid == 0) atomicExch(c.err, 1);
    return;
  }
  if (c.missing_out && Tl && in) c.missing_out[n] = missing_i;
}

__global__ void k_spread_pairs(int32_t n_keys, const int32_t* src_ptr, const int32_t* dst_ptr, const int32_t* src_node, const int64_t* src_max, int32_t* dst_node,
                               int64_t* dst_max) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_keys) return;
  const int lo = src_ptr[k], hi = src_ptr[k + 1], to = dst_ptr[k];
  for (int i = lo; i < hi; ++i) dst_node[to + (i - lo)] = src_node[i], dst_max[to + (i - lo)] = src_max[i];
}

}  // namespace

void launch_spread_pairs(int32_t n_keys, const int32_t* src_ptr, const int32_t* dst_ptr, const int32_t* src_node, const int64_t* src_max, int32_t* dst_node,
                         int64_t* dst_max, hipStream_t s) {
  if (n_keys <= 0) return;
  hipLaunchKernelGGL(k_spread_pairs, dim3(static_cast<unsigned>((n_keys + 255) / 256)), dim3(256), 0, s, n_keys, src_ptr, dst_ptr, src_node, src_max, dst_node, dst_max);
}

size_t commit_coop_lds_bytes(const CoopArgs& c) {
  const Layout l = make_layout(c);
  return l.total;
}

// How many workgroups of this launch the device can hold at once: every workgroup spins on every other one's granules, so the
// launch is only valid when all n_wg of them are resident (a smaller part, a CU mask, or a device shared with another process's
// kernels holds fewer).  <= 0: the runtime could not tell.
int commit_coop_max_resident(const CoopArgs& c, int device) {
  const size_t lds = commit_coop_lds_bytes(c);
  int per_cu = 0, cus = 0;
  if (c.nrt_sg == kSgMost)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_commit_coop<kSgMost>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  else
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_commit_coop<kSgLeast>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipError_t rc = c.nrt_sg == kSgMost
                      ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_commit_coop<kSgMost>), kT, lds)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_commit_coop<kSgLeast>), kT, lds);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return per_cu * cus;
}

void launch_commit_coop(const CoopArgs& c, hipStream_t s) {
  const size_t lds = commit_coop_lds_bytes(c);
  if (c.nrt_sg == kSgMost) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_commit_coop<kSgMost>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL((k_commit_coop<kSgMost>), dim3(static_cast<unsigned>(c.n_wg)), dim3(kT), lds, s, c);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_commit_coop<kSgLeast>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL((k_commit_coop<kSgLeast>), dim3(static_cast<unsigned>(c.n_wg)), dim3(kT), lds, s, c);
  }
}

}  // namespace spx
