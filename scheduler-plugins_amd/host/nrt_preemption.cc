// nrt_preemption.cc — the NRT side of a preemption dry-run.  Host-side product code.
//
// Reference: TopologyMatch.Filter, when the cycle state carries victims (filter.go:205-220), replaces the node's
// NodeResourceTopology by preemption.GetNRTPostPodsEviction(nrt, victims, numaPlacementInfo) (preemption.go:39-157) and then
// runs the ordinary single-NUMA handlers on it.  The transformation only touches the per-zone `available` quantities, so on
// this engine a dry-run over many candidate nodes is: spx_nrt_post_eviction per candidate (this file, O(victim containers)),
// spx_flatten_nrt_nodes / spx_upload_nrt_nodes with the returned availabilities, then the same Filter sweep.
//
// Rules restated: only app containers of the victims count; a victim that is neither Guaranteed nor requests a non-native
// resource is skipped outright; a container contributes to the NUMA node the placement info pins it to (none / unknown:
// skipped); per resource only what was exclusively assigned is given back — extended resources the NRT reports, whole cpus
// and memory / hugepages of Guaranteed pods (resourcerequests/exclusive.go:78-102); a release that would lift `available`
// above `allocatable` voids the whole simulation.
#include <cstdint>
#include <vector>

#include "../../include/spx.h"

namespace {

struct Give {
  int32_t numa;
  int32_t res;
  int64_t qty;
};

inline bool native(const spx_resource_classes* rc, int32_t res) {
  if (res == SPX_RES_CPU || res == SPX_RES_MEMORY || res == SPX_RES_EPHEMERAL || res == SPX_RES_PODS || res == SPX_RES_STORAGE) return true;
  return rc != nullptr && res >= 0 && res < rc->n_res && (rc->flags[res] & SPX_RC_NATIVE) != 0;
}
inline bool hugepage(const spx_resource_classes* rc, int32_t res) {
  return rc != nullptr && res >= SPX_RES_FIRST_DYNAMIC && res < rc->n_res && (rc->flags[res] & SPX_RC_HUGEPAGE) != 0;
}

}  // namespace

extern "C" int spx_nrt_post_eviction(const spx_nrt_objects* nrt, const spx_resource_classes* rc, int64_t node, const spx_pod_objects* victims,
                                     const uint8_t* victim_qos, const int32_t* ctr_numa, int32_t placement_present, int32_t placement_containers,
                                     int64_t* zres_avail_out, int32_t* code_out) {
  if (!nrt || !zres_avail_out || !code_out || node < 0 || node >= nrt->n_nodes) return SPX_ERR_ARG;
  const int32_t z0 = nrt->zone_ptr[node], z1 = nrt->zone_ptr[node + 1];
  const int32_t e0 = nrt->zres_ptr[z0], e1 = nrt->zres_ptr[z1];
  auto restore = [&] {
    for (int32_t e = e0; e < e1; ++e) zres_avail_out[e - e0] = nrt->zres_avail[e];
  };
  restore();
  auto done = [&](int32_t code) {
    *code_out = code;
    return SPX_OK;
  };
  if (!nrt->has_nrt[node]) return done(SPX_EVICT_NO_NRT);
  if (!victims || victims->n_pods == 0) return done(SPX_EVICT_NO_VICTIMS);
  if (!placement_present) return done(SPX_EVICT_NO_PLACEMENT);
  if (placement_containers == 0) return done(SPX_EVICT_NO_CONTAINERS);
  if (!victim_qos || !ctr_numa || !nrt->zres_allocatable) return SPX_ERR_ARG;

  auto reported = [&](int32_t res) {  // cache.ResourceNamesFromNRT
    for (int32_t e = e0; e < e1; ++e)
      if (nrt->zres_res[e] == res) return true;
    return false;
  };
  std::vector<Give> gives;  // one entry per (numa, resource), quantities summed
  for (int64_t v = 0; v < victims->n_pods; ++v) {
    const int qos = victim_qos[v];
    bool non_native = false;  // resourcerequests.IncludeNonNative: any container, init ones included
    for (int32_t c = victims->ctr_ptr[v]; c < victims->ctr_ptr[v + 1] && !non_native; ++c)
      for (int32_t i = victims->req_ptr[c]; i < victims->req_ptr[c + 1]; ++i)
        if (!native(rc, victims->req_res[i])) {
          non_native = true;
          break;
        }
    if (qos != SPX_QOS_GUARANTEED && !non_native) continue;
    for (int32_t c = victims->ctr_ptr[v]; c < victims->ctr_ptr[v + 1]; ++c) {
      if (victims->ctr_kind[c] != SPX_CTR_APP || ctr_numa[c] < 0) continue;
      for (int32_t i = victims->req_ptr[c]; i < victims->req_ptr[c + 1]; ++i) {
        const int32_t res = victims->req_res[i];
        const int64_t q = victims->req_qty[i];
        bool exclusive;
        if (!native(rc, res)) exclusive = reported(res);
        else if (qos != SPX_QOS_GUARANTEED) exclusive = false;
        else if (res == SPX_RES_CPU) exclusive = q > 0 && q % 1000 == 0;
        else exclusive = (res == SPX_RES_MEMORY || hugepage(rc, res)) && q > 0;
        if (!exclusive) continue;
        bool merged = false;
        for (Give& g : gives)
          if (g.numa == ctr_numa[c] && g.res == res) {
            g.qty += q;
            merged = true;
            break;
          }
        if (!merged) gives.push_back(Give{ctr_numa[c], res, q});
      }
    }
  }
  if (gives.empty()) return done(SPX_EVICT_NOTHING_TO_ADD);
  for (int32_t z = z0; z < z1; ++z) {
    const int32_t id = nrt->zone_numa_id[z];
    if (id < 0) continue;
    for (const Give& g : gives) {
      if (g.numa != id) continue;
      for (int32_t e = nrt->zres_ptr[z]; e < nrt->zres_ptr[z + 1]; ++e) {
        if (nrt->zres_res[e] != g.res) continue;
        const int64_t after = nrt->zres_avail[e] + g.qty;
        if (after > nrt->zres_allocatable[e]) {
          restore();
          return done(SPX_EVICT_EXCEEDS_ALLOCATABLE);
        }
        zres_avail_out[e - e0] = after;
        break;  // the first ResourceInfo of that name
      }
    }
  }
  return done(SPX_EVICT_OK);
}
