// kernels_commit.hip — what binding ONE pod changes for the pods scheduled after it, applied to the
// engine's device tables between two single-row evaluations of the sequential commit loop.
//
// Upstream schedules one pod at a time.  Between two cycles the plugins' Reserve / event hooks run:
//   trimaran            the pod enters the ScheduledPodsCache and counts as "missing utilisation" of its node until the
//                       metrics catch up                                   pkg/trimaran/handler.go:131-139, targetloadpacking.go:151-168
//   NodeResourceTopology TopologyMatch.Reserve -> OverReserve.ReserveNodeResources stores GetPodEffectiveRequest(pod) under the
//                       node; GetCachedNRTCopy then subtracts it from EVERY zone of that node that reports the resource
//                       ("pessimistic overallocation"; a zone with less than the quantity drops to zero)
//                                                                          reserve.go:28-46, cache/overreserve.go:170-186, cache/store.go:315-356
//   CapacityScheduling  Reserve -> addPodIfNotPresent -> reserveResource: the namespace's Used grows by the pod's request;
//                       the pod leaves the nominator, so it no longer counts as a nominated pod
//                                                                          capacity_scheduling.go:350-364, elasticquota.go:89-98
//   NetworkOverhead     the pod is in the pod lister with a node name: it joins its AppGroup's scheduled list, so workloads that
//                       depend on its workload see one more (host, MaxNetworkCost) pair, and workloads of the group that have
//                       dependencies stop "scoring equally"                networkoverhead.go:174-298, util.go GetScheduledList
// One workgroup; the work is a few hundred scalar updates.  Every table touched is one the single-row sweeps read next.
#include <hip/hip_runtime.h>

#include "spx_internal.h"

namespace spx {
namespace {

constexpr int S = SPX_QUOTA_SLOTS;
constexpr int kZ = SPX_NRT_MAX_ZONES;

__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) { return static_cast<int64_t>(static_cast<uint64_t>(a) + static_cast<uint64_t>(b)); }

// cmp(Used, Min, LowerBoundOfMin) elasticquota.go:166-191 via cmp2 with a zero second operand (as the host flattener does)
__device__ bool used_over_min(const CommitApplyArgs& a, int ns) {
  const int64_t* x = a.q_used + static_cast<int64_t>(ns) * S;
  const int64_t* y = a.q_min + static_cast<int64_t>(ns) * S;
  const uint32_t xp = a.q_used_present[ns], yp = a.q_min_present[ns];
  bool over = false;
  for (int s = 0; s < 4; ++s) over |= x[s] > y[s];
  for (int s = 4; s < S; ++s) {
    const int64_t yq = ((yp >> s) & 1u) ? y[s] : 0;
    over |= ((xp >> s) & 1u) && x[s] > yq;
  }
  return over;
}

__global__ __launch_bounds__(256) void k_commit_apply(CommitApplyArgs a) {
  const int64_t pod = a.row_counter ? *a.row_counter : a.pod;
  const int t = threadIdx.x;
  __syncthreads();  // every thread has read the counter before thread 0 advances it
  if (a.row_counter && t == 0) *a.row_counter = pod + 1;
  const int32_t node = a.best_node[pod];
  if (node < 0) return;  // unschedulable this cycle (PreFilter rejection or no feasible node): nothing is reserved
  const int64_t N = a.n_nodes;

  if (a.tlp_missing && t == 0) a.tlp_missing[node] += a.tlp_pod_milli[pod];

  if (a.nrt_avail && (a.nrt_flags[node] & SPX_NRT_F_HAS_NRT)) {  // "ignoring reserve": the cache holds no NRT for the node
    const int R = a.nrt_n_res;
    const uint32_t present = a.nrt_pod_present[pod];
    for (int c = t; c < kZ * R; c += blockDim.x) {
      const int z = c / R, r = c % R;
      if (!((a.nrt_zone_present[static_cast<int64_t>(z) * N + node] >> r) & 1u) || !((present >> r) & 1u)) continue;
      const int64_t i = (static_cast<int64_t>(z) * R + r) * N + node;
      const int64_t qty = a.nrt_pod_req[pod * R + r];
      const int64_t av = a.nrt_avail[i] < qty ? 0 : a.nrt_avail[i] - qty;  // store.go:335-351
      a.nrt_avail[i] = av;
      // the float64 sweep's derived cells (spx_upload_nrt_nodes computes the same expressions on the host)
      const bool is_cpu = r == a.nrt_cpu_slot;
      const double cap_v = static_cast<double>(is_cpu ? (av + 999) / 1000 : av);
      a.f_av[i] = static_cast<double>(av);
      a.f_rc[i] = cap_v > 0.0 ? 100.0 / cap_v : kNrtNoCap;
      a.f_rcv[i] = cap_v > 0.0 ? 1.0 / cap_v : 1.0;
      if (is_cpu) {
        a.f_cpu[static_cast<int64_t>(z) * N + node] = cap_v;
        a.f_braw[static_cast<int64_t>(z) * N + node] = av > 0 ? 100.0 / static_cast<double>(av) : kNrtNoCap;
      }
    }
  }

  if (a.q_used) {
    const int ns = a.q_pod_ns[pod];
    const int NS = a.q_n_namespaces;
    const bool counted = ns >= 0 && ns < NS && a.q_has[ns];
    if (t == 0 && counted) {
      uint32_t up = a.q_used_present[ns];
      const uint32_t rp = a.q_pod_reqp[pod];
      for (int s = 0; s < S; ++s) {
        const int64_t q = a.q_pod_req[pod * S + s];
        a.q_used[static_cast<int64_t>(ns) * S + s] = wadd(a.q_used[static_cast<int64_t>(ns) * S + s], q);
        a.q_agg_used[s] = wadd(a.q_agg_used[s], q);
      }
      a.q_used_present[ns] = static_cast<uint8_t>(up | rp);  // SetScalar creates the key
      a.q_agg_used[S] |= static_cast<int64_t>(rp);
      for (int j = a.q_nom_ptr[ns]; j < a.q_nom_ptr[ns + 1]; ++j)
        if (a.q_nom_pending[j] == pod) {  // the bound pod was itself nominated: it leaves the nominator
          for (int s = 0; s < S; ++s) a.q_nom_req[static_cast<int64_t>(j) * S + s] = 0;
          a.q_nom_reqp[j] = 0;
        }
    }
    __syncthreads();
    // nominated requests of the OTHER namespaces whose quota is not over min (capacity_scheduling.go:248-250): both the set of
    // nominated pods and "over min" may just have changed.  own[m] per namespace, their total, then total - own[k].
    __shared__ int64_t total[S];
    __shared__ int holders[S];  // per presence bit: how many namespaces' own sets carry it
    if (t < S) total[t] = 0, holders[t] = 0;
    __syncthreads();
    for (int m = t; m < NS; m += blockDim.x) {
      int64_t own[S] = {0};
      uint32_t ownp = 0;
      if (a.q_has[m] && !used_over_min(a, m))
        for (int j = a.q_nom_ptr[m]; j < a.q_nom_ptr[m + 1]; ++j) {
          for (int s = 0; s < S; ++s) own[s] = wadd(own[s], a.q_nom_req[static_cast<int64_t>(j) * S + s]);
          ownp |= a.q_nom_reqp[j];
        }
      for (int s = 0; s < S; ++s) {
        a.q_other[static_cast<int64_t>(m) * S + s] = own[s];  // parked; becomes total - own below
        if (own[s]) atomicAdd(reinterpret_cast<unsigned long long*>(&total[s]), static_cast<unsigned long long>(own[s]));
        if ((ownp >> s) & 1u) atomicAdd(&holders[s], 1);
      }
      a.q_otherp[m] = static_cast<uint8_t>(ownp);  // parked likewise
    }
    __syncthreads();
    for (int m = t; m < NS; m += blockDim.x) {
      const uint32_t ownp = a.q_otherp[m];
      uint32_t others = 0;  // union of the OTHER namespaces' scalar keys
      for (int s = 0; s < S; ++s) {
        const int64_t own = a.q_other[static_cast<int64_t>(m) * S + s];
        a.q_other[static_cast<int64_t>(m) * S + s] = static_cast<int64_t>(static_cast<uint64_t>(total[s]) - static_cast<uint64_t>(own));
        if (holders[s] - static_cast<int>((ownp >> s) & 1u) > 0) others |= 1u << s;
      }
      a.q_otherp[m] = static_cast<uint8_t>(others);
    }
  }

  if (a.net_eff_ptr && t == 0) {
    for (int e = a.net_eff_ptr[pod]; e < a.net_eff_ptr[pod + 1]; ++e) {
      const int k = a.net_eff_key[e];
      if (a.net_key_flag[k] == 1) a.net_key_flag[k] = 0;  // dependencies and a non-empty scheduled list: PreFilter evaluates
      const int64_t cost = a.net_eff_cost[e];
      if (cost >= 0) {
        const int at = a.net_pair_end[k]++;
        a.net_pair_node[at] = node;
        a.net_pair_max[at] = cost;
      }
    }
  }
}

}  // namespace

void launch_commit_apply(const CommitApplyArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_commit_apply, dim3(1), dim3(256), 0, s, a); }

}  // namespace spx
