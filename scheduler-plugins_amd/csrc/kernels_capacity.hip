// kernels_capacity.hip — CapacityScheduling.PreFilter for every pending pod (one thread per pod).
//
// Per pod only two things are not hoistable: which same-namespace nominated pods outrank it
// (priority >= the pod's, and not the pod itself) and the two cmp2 gates.  Everything else (Σ Used, Σ Min,
// the other namespaces' nominated requests, request vectors) arrives precomputed from the host flattener.
// Reference: pkg/capacityscheduling/capacity_scheduling.go:208-283, elasticquota.go:48-59, :117-131, :193-221.
// This is O(P * nominated-in-namespace) integer work on a few MB — latency-, not bandwidth-bound.
#include "spx_internal.h"

namespace spx {

namespace {

constexpr int S = SPX_QUOTA_SLOTS;

__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) {
  return static_cast<int64_t>(static_cast<uint64_t>(a) + static_cast<uint64_t>(b));
}

// cmp2 elasticquota.go:193-221
__device__ bool cmp2(const int64_t* x1, uint32_t x1p, const int64_t* x2, const int64_t* y, uint32_t yp, int64_t bound) {
  bool over = false;
#pragma unroll
  for (int s = 0; s < 4; ++s) over |= wadd(x1[s], x2 ? x2[s] : 0) > y[s];
#pragma unroll
  for (int s = 4; s < S; ++s) {
    const int64_t yq = ((yp >> s) & 1u) ? y[s] : bound;
    over |= ((x1p >> s) & 1u) && wadd(x1[s], x2 ? x2[s] : 0) > yq;
  }
  return over;
}

__global__ void k_quota(QuotaArgs a) {
  SPX_RESOLVE_ROWS(a);
  const int64_t pod = a.row_begin + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (pod >= a.row_end) return;
  const int ns = a.pod_ns[pod];
  uint8_t status = 0;
  if (ns >= 0 && ns < a.n_namespaces && a.has_quota[ns]) {
    const int prio = a.pod_priority[pod];
    int64_t in_eq[S];
    uint32_t in_p = a.pod_req_present[pod];
#pragma unroll
    for (int s = 0; s < S; ++s) in_eq[s] = a.pod_req[pod * S + s];
    for (int j = a.nom_ptr[ns]; j < a.nom_ptr[ns + 1]; ++j) {  // same quota, more important than the preemptor
      if (a.nom_pending_index[j] == pod || a.nom_priority[j] < prio) continue;
#pragma unroll
      for (int s = 0; s < S; ++s) in_eq[s] = wadd(in_eq[s], a.nom_req[static_cast<int64_t>(j) * S + s]);
      in_p |= a.nom_req_present[j];
    }
    if (cmp2(in_eq, in_p, a.used + static_cast<int64_t>(ns) * S, a.max + static_cast<int64_t>(ns) * S, a.max_present[ns], INT64_MAX)) {
      status = SPX_QUOTA_ST_OVER_MAX;
    } else {
      int64_t agg[S];
#pragma unroll
      for (int s = 0; s < S; ++s)
        agg[s] = wadd(wadd(a.agg_used_dyn ? a.agg_used_dyn[s] : a.agg_used[s], in_eq[s]), a.other_nominated[static_cast<int64_t>(ns) * S + s]);
      const uint32_t agg_p = (a.agg_used_dyn ? static_cast<uint32_t>(a.agg_used_dyn[S]) : a.agg_used_present) | in_p | a.other_nominated_present[ns];
      if (cmp2(agg, agg_p, nullptr, a.agg_min, a.agg_min_present, 0)) status = SPX_QUOTA_ST_OVER_MIN;
    }
  }
  a.out_status[pod] = status;
}

}  // namespace

void launch_quota(const QuotaArgs& a, hipStream_t s) {
  if (a.row_end <= a.row_begin) return;
  const unsigned blocks = static_cast<unsigned>((a.row_end - a.row_begin + 255) / 256);
  hipLaunchKernelGGL(k_quota, dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace spx
