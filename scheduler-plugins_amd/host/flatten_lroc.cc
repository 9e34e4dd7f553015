// flatten_lroc.cc — object tables -> SoA columns for trimaran LowRiskOverCommitment.  Host-side product code
// (once per snapshot, O(N + pods on nodes + P)).
//
// What is hoisted out of the per-(pod,node) path, and where the reference does it per call:
//   pod  : CreatePodResourcesStateData (PreScore): requests, limits raised to them   lowriskovercommitment.go:259-268
//   node : the running sums of GetNodeRequestsAndLimits over the pods already on the node — everything that loop
//          accumulates before it reaches the pending pod                              resourcestats.go:184-206
// The capacity caps (:208-211) and the pending pod's own share stay in the kernel: they depend on the pair.
#include <cstdint>

#include "../../include/spx.h"
#include "parallel.hpp"

namespace {

inline bool find_qty(const int32_t* res, const int64_t* qty, int32_t lo, int32_t hi, int32_t want, int64_t* out) {
  for (int32_t i = lo; i < hi; ++i) {
    if (res[i] == want) {
      *out = qty[i];
      return true;
    }
  }
  return false;
}

struct CpuMem {
  int64_t cpu = 0, mem = 0;
};

// GetEffectiveResource (resourcestats.go:123-146) over one of the two per-container lists: sum of the app
// containers, raised to any (restartable or not) init container, plus the pod overhead.
CpuMem effective(const spx_pod_objects* pods, int64_t i, const int32_t* ptr, const int32_t* res, const int64_t* qty) {
  CpuMem r;
  int64_t q;
  for (int32_t c = pods->ctr_ptr[i]; c < pods->ctr_ptr[i + 1]; ++c) {
    if (pods->ctr_kind[c] != SPX_CTR_APP) continue;
    if (find_qty(res, qty, ptr[c], ptr[c + 1], SPX_RES_CPU, &q)) r.cpu += q;
    if (find_qty(res, qty, ptr[c], ptr[c + 1], SPX_RES_MEMORY, &q)) r.mem += q;
  }
  for (int32_t c = pods->ctr_ptr[i]; c < pods->ctr_ptr[i + 1]; ++c) {
    if (pods->ctr_kind[c] == SPX_CTR_APP) continue;
    if (find_qty(res, qty, ptr[c], ptr[c + 1], SPX_RES_CPU, &q) && q > r.cpu) r.cpu = q;
    if (find_qty(res, qty, ptr[c], ptr[c + 1], SPX_RES_MEMORY, &q) && q > r.mem) r.mem = q;
  }
  if (pods->ovh_ptr != nullptr) {
    if (find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[i], pods->ovh_ptr[i + 1], SPX_RES_CPU, &q)) r.cpu += q;
    if (find_qty(pods->ovh_res, pods->ovh_qty, pods->ovh_ptr[i], pods->ovh_ptr[i + 1], SPX_RES_MEMORY, &q)) r.mem += q;
  }
  return r;
}

// requests and SetMaxLimits(requests, limits) of one pod (resourcestats.go:229-232)
inline void requests_limits(const spx_pod_objects* pods, int64_t i, CpuMem* req, CpuMem* lim) {
  *req = effective(pods, i, pods->req_ptr, pods->req_res, pods->req_qty);
  *lim = effective(pods, i, pods->lim_ptr, pods->lim_res, pods->lim_qty);
  if (lim->cpu < req->cpu) lim->cpu = req->cpu;
  if (lim->mem < req->mem) lim->mem = req->mem;
}

}  // namespace

extern "C" int spx_flatten_lroc_pods(const spx_pod_objects* pods, int64_t* req_cpu_milli, int64_t* req_mem, int64_t* lim_cpu_milli,
                                     int64_t* lim_mem) {
  if (!pods || !req_cpu_milli || !req_mem || !lim_cpu_milli || !lim_mem) return SPX_ERR_ARG;
  spx_host::parallel_rows(pods->n_pods, [&](int64_t row0, int64_t row1) {
    for (int64_t i = row0; i < row1; ++i) {
      CpuMem req, lim;
      requests_limits(pods, i, &req, &lim);
      req_cpu_milli[i] = req.cpu;
      req_mem[i] = req.mem;
      lim_cpu_milli[i] = lim.cpu;
      lim_mem[i] = lim.mem;
    }
  });
  return SPX_OK;
}

extern "C" int spx_flatten_lroc_nodes(const spx_node_objects* nodes, const spx_node_pods_objects* node_pods, int64_t* req_cpu_milli,
                                      int64_t* req_mem, int64_t* lim_cpu_milli, int64_t* lim_mem) {
  if (!nodes || !req_cpu_milli || !req_mem || !lim_cpu_milli || !lim_mem) return SPX_ERR_ARG;
  const bool any = node_pods != nullptr && node_pods->p_ptr != nullptr;
  if (any && (node_pods->pods == nullptr || (node_pods->p_ptr[nodes->n_nodes] > 0 && node_pods->p_pod == nullptr))) return SPX_ERR_ARG;
  spx_host::parallel_rows(nodes->n_nodes, [&](int64_t row0, int64_t row1) {
    for (int64_t n = row0; n < row1; ++n) {
      CpuMem rs, ls;
      if (any) {
        for (int32_t k = node_pods->p_ptr[n]; k < node_pods->p_ptr[n + 1]; ++k) {
          CpuMem req, lim;
          requests_limits(node_pods->pods, node_pods->p_pod[k], &req, &lim);
          rs.cpu += req.cpu, rs.mem += req.mem, ls.cpu += lim.cpu, ls.mem += lim.mem;
        }
      }
      req_cpu_milli[n] = rs.cpu;
      req_mem[n] = rs.mem;
      lim_cpu_milli[n] = ls.cpu;
      lim_mem[n] = ls.mem;
    }
  });
  return SPX_OK;
}
