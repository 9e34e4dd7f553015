/*
 * orc_network.c — restatement of networkaware NetworkOverhead (PreFilter/Filter/Score/NormalizeScore) and
 * TopologicalSort.Less (TEST INFRASTRUCTURE, see spx_oracle.h).
 *
 * Follows pkg/networkaware/networkoverhead/networkoverhead.go:174-298 (PreFilter), :326-359 (Filter),
 * :362-386 (Score), :389-435 (NormalizeScore), :448-497 (populateCostMap), :500-573
 * (checkMaxNetworkCostRequirements), :576-638 (getAccumulatedCost); pkg/networkaware/util/util.go:138-232;
 * pkg/networkaware/topologicalsort/topologicalsort.go:102-132.  queuesort.PrioritySort.Less is upstream
 * (k8s.io/kubernetes v1.35.7): priority descending, then queue timestamp ascending.
 *
 * String keys are interned ids in the object tables (include/spx.h): regions and zones are separate id
 * spaces, so a CostKey{Origin, Destination} is (space, origin id, destination id).
 */
#include <stdlib.h>

#include "spx_oracle.h"

enum { SAME_HOSTNAME = 0, SAME_ZONE = 1, MAX_COST = 100 }; /* networkoverhead.go:57-63 */

/* costMap lookup for the node being evaluated: populateCostMap only inserts rows whose origin is this node's
 * own region / zone, and only when that label is non-empty (:458, :477). */
static int cost_lookup(const int32_t* ptr, const int32_t* dest, const int64_t* cost, int32_t origin, int32_t n_origins,
                       int32_t destination, int64_t* out) {
  if (origin < 0 || origin >= n_origins) return 0;
  int found = 0;
  for (int32_t i = ptr[origin]; i < ptr[origin + 1]; ++i)
    if (dest[i] == destination) { /* later entries override: map assignment */
      *out = cost[i];
      found = 1;
    }
  return found;
}

/* GetDependencyList util.go:194-212: dependencies of every workload whose selector equals the pod's */
static int dependency_list(const spx_appgroup_objects* ag, int32_t g, int32_t selector, int32_t* dep_sel, int64_t* dep_max, int cap) {
  int n = 0;
  for (int32_t w = ag->wl_ptr[g]; w < ag->wl_ptr[g + 1]; ++w)
    if (ag->wl_selector[w] == selector)
      for (int32_t d = ag->dep_ptr[w]; d < ag->dep_ptr[w + 1] && n < cap; ++d) {
        dep_sel[n] = ag->dep_selector[d];
        dep_max[n] = ag->dep_max_cost[d];
        ++n;
      }
  return n;
}

/* NetworkOverhead.PreFilter for one pod: fills sat/vio/cost[n_nodes]; returns 1 when scoreEqually, -1 on the
 * Error paths ("pod hostname not found"), 0 otherwise */
int orc_net_prefilter(const spx_node_objects* nodes, const spx_pod_objects* pods, const spx_appgroup_objects* ag,
                      const spx_nettopo_objects* nt, int64_t pod, int64_t* sat, int64_t* vio, int64_t* cost) {
  return orc_net_prefilter_range(nodes, pods, ag, nt, pod, 0, nodes->n_nodes, sat, vio, cost);
}

/* the same for the nodes [node_begin, node_end) of PreFilter's node loop (:243-280; the iterations are independent): what one
 * worker of a chunked parallel-for computes.  The early exits do not depend on the range. */
int orc_net_prefilter_range(const spx_node_objects* nodes, const spx_pod_objects* pods, const spx_appgroup_objects* ag,
                            const spx_nettopo_objects* nt, int64_t pod, int64_t node_begin, int64_t node_end,
                            int64_t* sat, int64_t* vio, int64_t* cost) {
  const int64_t n = nodes->n_nodes;
  for (int64_t i = node_begin; i < node_end; ++i) sat[i] = vio[i] = cost[i] = 0;
  int32_t g = pods->appgroup[pod];
  if (g < 0 || g >= ag->n_groups) return 1; /* "Pod does not belong to an AppGroup" :187-190 */
  int32_t dep_sel[256];
  int64_t dep_max[256];
  int nd = dependency_list(ag, g, pods->selector[pod], dep_sel, dep_max, 256);
  if (nd == 0) return 1; /* "Pod has no dependencies" :205-207 */
  int32_t s0 = ag->placed_ptr[g], s1 = ag->placed_ptr[g + 1];
  if (s1 == s0) return 1; /* no pods listed / scheduled list empty :217-228 */

  for (int64_t node = node_begin; node < node_end; ++node) { /* :243-280 */
    int32_t region = nodes->region[node], zone = nodes->zone[node];
    int64_t satisfied = 0, violated = 0, acc = 0;
    for (int32_t s = s0; s < s1; ++s) {
      for (int d = 0; d < nd; ++d) {
        if (ag->placed_selector[s] != dep_sel[d]) continue;
        int32_t host = ag->placed_node[s];
        if (host == node) { /* same hostname */
          satisfied += 1;
          acc += SAME_HOSTNAME;
          continue;
        }
        if (host < 0 || host >= n) return -1; /* NodeInfos().Get(hostname) failed */
        int32_t region_p = nodes->region[host], zone_p = nodes->zone[host];
        int64_t c;
        if (region_p < 0 && zone_p < 0) { /* placed node has no region and no zone */
          violated += 1;
          acc += MAX_COST;
        } else if (region == region_p) {
          if (zone == zone_p) {
            satisfied += 1;
            acc += SAME_ZONE;
          } else if (cost_lookup(nt->zc_ptr, nt->zc_dest, nt->zc_cost, zone, nt->n_zones, zone_p, &c)) {
            if (c <= dep_max[d]) satisfied += 1;
            else violated += 1;
            acc += c;
          } else {
            acc += MAX_COST; /* missing entry: ignored by the counter (:548-557), charged by the accumulator (:617-622) */
          }
        } else if (cost_lookup(nt->rc_ptr, nt->rc_dest, nt->rc_cost, region, nt->n_regions, region_p, &c)) {
          if (c <= dep_max[d]) satisfied += 1;
          else violated += 1;
          acc += c;
        } else {
          acc += MAX_COST;
        }
      }
    }
    sat[node] = satisfied;
    vio[node] = violated;
    cost[node] = acc;
  }
  return 0;
}

/* NetworkOverhead.NormalizeScore networkoverhead.go:389-418 */
void orc_net_normalize(int64_t* scores, int64_t n) {
  int64_t max = INT64_MIN, min = INT64_MAX; /* getMinMaxScores :421-435 */
  for (int64_t i = 0; i < n; ++i) {
    if (scores[i] > max) max = scores[i];
    if (scores[i] < min) min = scores[i];
  }
  if (min == 0 && max == 0) return;
  for (int64_t i = 0; i < n; ++i) {
    double norm;
    if (max != min) {
      norm = 100.0 * (double)(scores[i] - min) / (double)(max - min);
      scores[i] = 100 - (int64_t)norm;
    } else {
      norm = (double)(scores[i] - min);
      scores[i] = 100 - (int64_t)norm;
    }
  }
}

/* FindPodOrder util.go:138-153: binary search of Status.TopologyOrder by selector */
int32_t orc_find_pod_order(const spx_appgroup_objects* ag, int32_t g, int32_t selector) {
  const int32_t base = ag->topo_ptr[g];
  int low = 0, high = ag->topo_ptr[g + 1] - base - 1;
  while (low <= high) {
    int mid = (low + high) / 2;
    if (ag->topo_selector[base + mid] == selector) return ag->topo_index[base + mid];
    else if (ag->topo_selector[base + mid] < selector) low = mid + 1;
    else high = mid - 1;
  }
  return -1;
}

/* TopologicalSort.Less topologicalsort.go:102-132 */
int orc_toposort_less(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int64_t p1, int64_t p2) {
  int32_t g1 = pods->appgroup[p1], g2 = pods->appgroup[p2];
  if (g1 != g2 || g1 < 0) { /* different AppGroups, or none: queuesort.PrioritySort */
    int32_t pr1 = pods->priority[p1], pr2 = pods->priority[p2];
    return (pr1 > pr2) || (pr1 == pr2 && pods->queue_ts[p1] < pods->queue_ts[p2]);
  }
  int32_t o1 = orc_find_pod_order(ag, g1, pods->selector[p1]);
  int32_t o2 = orc_find_pod_order(ag, g1, pods->selector[p2]);
  return o1 <= o2;
}

/* The parity statement for a batched TopologicalSort (hard part 6): Less is not a strict weak order, so
 * "the sorted queue" is heap-implementation dependent; what can be checked is that every adjacent pair (x, y) of a proposed
 * order satisfies Less(x, y) — or is a PrioritySort tie (different/no AppGroup, equal priority and timestamp), which the
 * heap may pop either way.  Returns the number of adjacent pairs that do neither; -1 when perm is not a permutation. */
int64_t orc_toposort_order_violations(const spx_pod_objects* pods, const spx_appgroup_objects* ag, const int32_t* perm, int64_t n) {
  if (!pods || !ag || !perm || n != pods->n_pods) return -1;
  unsigned char* seen = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  if (!seen) return -1;
  for (int64_t i = 0; i < n; ++i) {
    if (perm[i] < 0 || perm[i] >= n || seen[perm[i]]) {
      free(seen);
      return -1;
    }
    seen[perm[i]] = 1;
  }
  free(seen);
  int64_t bad = 0;
  for (int64_t i = 0; i + 1 < n; ++i) {
    const int64_t x = perm[i], y = perm[i + 1];
    if (orc_toposort_less(pods, ag, x, y)) continue;
    const int32_t gx = pods->appgroup[x], gy = pods->appgroup[y];
    if ((gx != gy || gx < 0) && pods->priority[x] == pods->priority[y] && pods->queue_ts[x] == pods->queue_ts[y]) continue;
    ++bad;
  }
  return bad;
}
