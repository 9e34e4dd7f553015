/*
 * orc_nrt.c — restatement of NodeResourceTopologyMatch Filter + Score (TEST INFRASTRUCTURE, see spx_oracle.h).
 *
 * Follows pkg/noderesourcetopology/{filter.go:42-258, score.go:62-191, least_numa.go:35-233,
 * least_allocated.go:25-55, most_allocated.go:25-54, balanced_allocation.go:27-54, numaresources.go:105-215,
 * pluginhelpers.go:105-173}, nodeconfig/topologymanager.go:78-162, cache/store.go:315-356 (UpdateNRT),
 * resourcerequests/exclusive.go:28-44 and pkg/util/resource.go:28-85.
 *
 * Helpers that live outside the reference tree are restated from their published behaviour at the pinned
 * versions: v1qos.GetPodQOS (k8s.io/kubernetes v1.35.7), bitmask.BitMask (one uint64),
 * gonum stat.Mean/Variance and combin.Combinations v0.12.0, resource.Quantity Value()/Cmp.
 * Go map iteration order is random; every loop over a map below is order-independent in its result, except
 * where noted (stat.Variance over >= 3 fractions).
 */
#include <math.h>
#include <string.h>

#include "spx_oracle.h"

#define ORC_MAXR 32
#define ORC_MAXZ 64

typedef struct rlist { /* v1.ResourceList: map[name]Quantity with key presence */
  int n;
  int32_t res[ORC_MAXR];
  int64_t qty[ORC_MAXR];
} rlist;

typedef struct numa_node { /* NUMANode numaresources.go:32-36 */
  int id;
  rlist resources;
  int n_cost;
  int cost_id[ORC_MAXZ];
  int64_t cost_val[ORC_MAXZ];
} numa_node;

typedef struct numa_list {
  int n;
  numa_node z[ORC_MAXZ];
} numa_list;

typedef struct tm_conf { /* nodeconfig.TopologyManager */
  int scope;  /* 0 container, 1 pod */
  int policy; /* 0 none, 1 best-effort, 2 restricted, 3 single-numa-node */
  int max_numa;
} tm_conf;

/* ---------------------------------------------------------------- ResourceList helpers */

static int rl_find(const rlist* l, int32_t res) {
  for (int i = 0; i < l->n; ++i)
    if (l->res[i] == res) return i;
  return -1;
}
static void rl_set(rlist* l, int32_t res, int64_t q) {
  int i = rl_find(l, res);
  if (i < 0) {
    i = l->n++;
    l->res[i] = res;
  }
  l->qty[i] = q;
}
static void rl_from_csr(rlist* l, const int32_t* res, const int64_t* qty, int32_t lo, int32_t hi) {
  l->n = 0;
  for (int32_t i = lo; i < hi; ++i) rl_set(l, res[i], qty[i]);
}
static void ctr_requests(const spx_pod_objects* p, int32_t c, rlist* out) {
  rl_from_csr(out, p->req_res, p->req_qty, p->req_ptr[c], p->req_ptr[c + 1]);
}

/* Quantity.Value(): cpu is kept in millicores -> ceil(milli / 1000); everything else already Value() */
static int64_t q_value(int32_t res, int64_t q) {
  if (res != SPX_RES_CPU) return q;
  return q >= 0 ? (q + 999) / 1000 : -((-q) / 1000);
}

static int rc_flag(const spx_resource_classes* rc, int32_t res, int flag) {
  if (res == SPX_RES_CPU || res == SPX_RES_MEMORY || res == SPX_RES_EPHEMERAL || res == SPX_RES_PODS || res == SPX_RES_STORAGE)
    return flag == SPX_RC_NATIVE;
  if (!rc || res < 0 || res >= rc->n_res) return 0;
  return (rc->flags[res] & flag) != 0;
}

/* isHostLevelResource numaresources.go:105-118 */
static int is_host_level(const spx_resource_classes* rc, int32_t res) {
  if (res == SPX_RES_EPHEMERAL) return 1;
  if (res == SPX_RES_STORAGE) return 1;
  if (!rc_flag(rc, res, SPX_RC_NATIVE)) return 1;
  return 0;
}
/* isNUMAAffineResource numaresources.go:120-135 */
static int is_numa_affine(const spx_resource_classes* rc, int32_t res) {
  if (res == SPX_RES_CPU) return 1;
  if (res == SPX_RES_MEMORY) return 1;
  if (rc_flag(rc, res, SPX_RC_HUGEPAGE)) return 1;
  return 0;
}
/* isResourceSetSuitable numaresources.go:137-142 */
static int is_suitable(const spx_resource_classes* rc, int qos, int32_t res, int64_t quantity, int64_t numa_quantity) {
  if (qos != SPX_QOS_GUARANTEED && is_numa_affine(rc, res)) return 1;
  return numa_quantity >= quantity; /* numaQuantity.Cmp(quantity) >= 0 (same canonical unit) */
}

/* ---------------------------------------------------------------- pod-side helpers */

/* v1qos.ComputePodQOS (k8s.io/kubernetes/pkg/apis/core/v1/helper/qos/qos.go): cpu and memory only, over
 * app + init containers; zero quantities are ignored; Guaranteed needs cpu AND memory limits on every
 * container and summed requests == summed limits with identical key sets. */
int orc_pod_qos(const spx_pod_objects* p, int64_t pod) {
  int64_t req[2] = {0, 0}, lim[2] = {0, 0};
  int has_req[2] = {0, 0}, has_lim[2] = {0, 0};
  int is_guaranteed = 1;
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c) {
    for (int32_t i = p->req_ptr[c]; i < p->req_ptr[c + 1]; ++i) {
      int32_t r = p->req_res[i];
      if (r != SPX_RES_CPU && r != SPX_RES_MEMORY) continue;
      if (p->req_qty[i] > 0) {
        req[r] += p->req_qty[i];
        has_req[r] = 1;
      }
    }
    int found[2] = {0, 0};
    for (int32_t i = p->lim_ptr[c]; i < p->lim_ptr[c + 1]; ++i) {
      int32_t r = p->lim_res[i];
      if (r != SPX_RES_CPU && r != SPX_RES_MEMORY) continue;
      if (p->lim_qty[i] > 0) {
        found[r] = 1;
        lim[r] += p->lim_qty[i];
        has_lim[r] = 1;
      }
    }
    if (!(found[0] && found[1])) is_guaranteed = 0;
  }
  if (!has_req[0] && !has_req[1] && !has_lim[0] && !has_lim[1]) return SPX_QOS_BESTEFFORT;
  if (is_guaranteed) {
    for (int r = 0; r < 2; ++r)
      if (has_req[r] && (!has_lim[r] || lim[r] != req[r])) is_guaranteed = 0;
  }
  if (is_guaranteed && (has_req[0] + has_req[1]) == (has_lim[0] + has_lim[1])) return SPX_QOS_GUARANTEED;
  return SPX_QOS_BURSTABLE;
}

/* resourcerequests.IncludeNonNative exclusive.go:28-44 */
int orc_include_non_native(const spx_pod_objects* p, const spx_resource_classes* rc, int64_t pod) {
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c)
    for (int32_t i = p->req_ptr[c]; i < p->req_ptr[c + 1]; ++i)
      if (!rc_flag(rc, p->req_res[i], SPX_RC_NATIVE)) return 1;
  return 0;
}

/* util.GetPodEffectiveRequest pkg/util/resource.go:51-85 */
static void effective_request(const spx_pod_objects* p, int64_t pod, rlist* resources) {
  rlist init_resources;
  init_resources.n = 0;
  resources->n = 0;
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c) {
    if (p->ctr_kind[c] == SPX_CTR_APP) continue;
    for (int32_t i = p->req_ptr[c]; i < p->req_ptr[c + 1]; ++i) {
      int k = rl_find(&init_resources, p->req_res[i]);
      if (k >= 0 && p->req_qty[i] <= init_resources.qty[k]) continue; /* quantity.Cmp(q) <= 0 */
      rl_set(&init_resources, p->req_res[i], p->req_qty[i]);
    }
  }
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c) {
    if (p->ctr_kind[c] != SPX_CTR_APP) continue;
    for (int32_t i = p->req_ptr[c]; i < p->req_ptr[c + 1]; ++i) {
      int k = rl_find(resources, p->req_res[i]);
      rl_set(resources, p->req_res[i], p->req_qty[i] + (k >= 0 ? resources->qty[k] : 0));
    }
  }
  for (int i = 0; i < init_resources.n; ++i) {
    int k = rl_find(resources, init_resources.res[i]);
    if (k >= 0 && init_resources.qty[i] <= resources->qty[k]) continue;
    rl_set(resources, init_resources.res[i], init_resources.qty[i]);
  }
  if (p->ovh_ptr) {
    for (int32_t i = p->ovh_ptr[pod]; i < p->ovh_ptr[pod + 1]; ++i) {
      int k = rl_find(resources, p->ovh_res[i]);
      rl_set(resources, p->ovh_res[i], p->ovh_qty[i] + (k >= 0 ? resources->qty[k] : 0));
    }
  }
}

/* ---------------------------------------------------------------- node-side helpers */

/* TopologyManagerFromNodeResourceTopology nodeconfig/topologymanager.go:78-162 */
static tm_conf conf_of(const spx_nrt_objects* nrt, int64_t node) {
  tm_conf c = {0 /* container */, 0 /* none */, 8 /* DefaultMaxNUMANodes */};
  int lp = nrt->legacy_policy ? nrt->legacy_policy[node] : -1;
  if (lp >= 0) { /* updateFromPolicies */
    c.policy = lp >> 1;
    c.scope = lp & 1;
  }
  if (nrt->attr_scope && nrt->attr_scope[node] >= 0) c.scope = nrt->attr_scope[node];
  if (nrt->attr_policy && nrt->attr_policy[node] >= 0) c.policy = nrt->attr_policy[node];
  if (nrt->attr_max_numa && nrt->attr_max_numa[node] > 1) {
    int v = nrt->attr_max_numa[node];
    c.max_numa = v > 1024 ? 1024 : v; /* clampMaxNUMANodes */
  }
  return c;
}

/* GetCachedNRTCopy (OverReserve.UpdateNRT cache/store.go:315-356) + createNUMANodeList pluginhelpers.go:105-134 */
static void numa_list_of(const spx_nrt_objects* nrt, int64_t node, numa_list* out) {
  out->n = 0;
  for (int32_t z = nrt->zone_ptr[node]; z < nrt->zone_ptr[node + 1]; ++z) {
    if (!nrt->zone_is_node[z]) continue; /* zone.Type != "Node" */
    int id = nrt->zone_numa_id[z];
    if (id < 0 || id > 64) continue; /* NameToID error or numaID > maxNUMAId */
    if (out->n >= ORC_MAXZ) break;
    numa_node* nn = &out->z[out->n++];
    nn->id = id;
    nn->resources.n = 0;
    for (int32_t i = nrt->zres_ptr[z]; i < nrt->zres_ptr[z + 1]; ++i) { /* extractResources: Available */
      int64_t avail = nrt->zres_avail[i];
      if (nrt->assumed_ptr) { /* subtract every assumed pod from every zone, floor at zero */
        for (int32_t a = nrt->assumed_ptr[node]; a < nrt->assumed_ptr[node + 1]; ++a)
          for (int32_t k = nrt->arl_ptr[a]; k < nrt->arl_ptr[a + 1]; ++k)
            if (nrt->arl_res[k] == nrt->zres_res[i]) {
              if (avail < nrt->arl_qty[k]) avail = 0; /* zr.Available.Cmp(qty) < 0 -> Quantity{} */
              else avail -= nrt->arl_qty[k];
            }
      }
      rl_set(&nn->resources, nrt->zres_res[i], avail);
    }
    nn->n_cost = 0;
    if (nrt->zcost_ptr) { /* extractCosts pluginhelpers.go:136-153 */
      for (int32_t i = nrt->zcost_ptr[z]; i < nrt->zcost_ptr[z + 1]; ++i) {
        int cid = nrt->zcost_numa_id[i];
        if (cid < 0 || cid > 64) continue;
        int k;
        for (k = 0; k < nn->n_cost; ++k)
          if (nn->cost_id[k] == cid) break;
        if (k == nn->n_cost) nn->n_cost++;
        nn->cost_id[k] = cid;
        nn->cost_val[k] = nrt->zcost_value[i];
      }
    }
  }
}

/* key exists in util.ResourceList(nodeInfo.Allocatable) pkg/util/resource.go:30-44: cpu, memory, pods,
 * ephemeral-storage always; scalar resources when present in Allocatable.ScalarResources */
static int node_has_resource(const spx_node_objects* nodes, int64_t node, int32_t res) {
  if (res == SPX_RES_CPU || res == SPX_RES_MEMORY || res == SPX_RES_PODS || res == SPX_RES_EPHEMERAL) return 1;
  for (int32_t i = nodes->scalar_ptr[node]; i < nodes->scalar_ptr[node + 1]; ++i)
    if (nodes->scalar_res[i] == res) return 1;
  return 0;
}

/* ---------------------------------------------------------------- Filter */

/* resourcesAvailableInAnyNUMANodes filter.go:93-163; returns match, *numa_id = chosen NUMA id */
static int resources_available(const spx_node_objects* nodes, int64_t node, const spx_resource_classes* rc, const numa_list* nl,
                               const tm_conf* conf, int qos, const rlist* resources, int* numa_id) {
  uint64_t bitmask = ~0ull; /* bm.NewEmptyBitMask(); Fill() */
  *numa_id = conf->max_numa;
  for (int i = 0; i < resources->n; ++i) {
    int32_t res = resources->res[i];
    int64_t quantity = resources->qty[i];
    if (quantity == 0) continue;                         /* :104-108 */
    if (!node_has_resource(nodes, node, res)) {          /* :110-116 */
      *numa_id = -1;
      return 0;
    }
    int has_numa_affinity = 0;
    uint64_t resource_bitmask = 0;
    for (int z = 0; z < nl->n; ++z) {                    /* :122-138 */
      int k = rl_find(&nl->z[z].resources, res);
      if (k < 0) continue;
      has_numa_affinity = 1;
      if (!is_suitable(rc, qos, res, quantity, nl->z[z].resources.qty[k])) continue;
      if (nl->z[z].id < 64) resource_bitmask |= 1ull << nl->z[z].id; /* BitMask.Add rejects bits >= 64 */
    }
    if (!has_numa_affinity && is_host_level(rc, res)) continue; /* :142-145 */
    bitmask &= resource_bitmask;
    if (bitmask == 0) return 0;                          /* :148-151 */
  }
  *numa_id = __builtin_ctzll(bitmask);                   /* bitmask.GetBits()[0] */
  return bitmask != 0;
}

/* subtractResourcesFromNUMANodeList numaresources.go:145-182; returns 0 ok, -1 "inconsistent resource accounting" */
static int subtract_from_numa(const spx_resource_classes* rc, numa_list* nl, int numa_id, int qos, const rlist* ctr) {
  for (int z = 0; z < nl->n; ++z) {
    if (nl->z[z].id != numa_id) continue;
    for (int i = 0; i < ctr->n; ++i) {
      if (qos != SPX_QOS_GUARANTEED && is_numa_affine(rc, ctr->res[i])) continue;
      if (ctr->qty[i] == 0) continue;
      int k = rl_find(&nl->z[z].resources, ctr->res[i]);
      if (k < 0) continue;
      int64_t left = nl->z[z].resources.qty[k] - ctr->qty[i];
      if (left < 0) return -1;
      nl->z[z].resources.qty[k] = left;
    }
  }
  return 0;
}

/* TopologyMatch.Filter filter.go:179-245 -> 0 pass, SPX_NRT_ST_* reason (Unschedulable), -1 Error */
int orc_nrt_filter(const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_resource_classes* rc,
                   const spx_pod_objects* pods, int64_t pod, int64_t node) {
  int qos = orc_pod_qos(pods, pod);
  if (qos == SPX_QOS_BESTEFFORT && !orc_include_non_native(pods, rc, pod)) return 0; /* :183-186 */
  if (!nrt->fresh[node]) return SPX_NRT_ST_INVALID_TOPOLOGY;                         /* :196-200 */
  if (!nrt->has_nrt[node]) return 0;                                                 /* :201-203 */
  tm_conf conf = conf_of(nrt, node);
  if (conf.policy != 3) return 0; /* filterHandlerFromTopologyManager :247-258 */
  numa_list nl;
  numa_list_of(nrt, node, &nl);
  if (conf.scope == 1) { /* singleNUMAPodLevelHandler :165-176 */
    rlist resources;
    effective_request(pods, pod, &resources);
    int numa_id;
    if (!resources_available(nodes, node, rc, &nl, &conf, qos, &resources, &numa_id)) return SPX_NRT_ST_POD;
    return 0;
  }
  /* singleNUMAContainerLevelHandler :42-81 */
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) {
    if (pods->ctr_kind[c] == SPX_CTR_APP) continue;
    rlist req;
    ctr_requests(pods, c, &req);
    int numa_id;
    if (!resources_available(nodes, node, rc, &nl, &conf, qos, &req, &numa_id))
      return pods->ctr_kind[c] == SPX_CTR_SIDECAR ? SPX_NRT_ST_SIDECAR_CONTAINER : SPX_NRT_ST_INIT_CONTAINER;
  }
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) {
    if (pods->ctr_kind[c] != SPX_CTR_APP) continue;
    rlist req;
    ctr_requests(pods, c, &req);
    int numa_id;
    if (!resources_available(nodes, node, rc, &nl, &conf, qos, &req, &numa_id)) return SPX_NRT_ST_CONTAINER;
    if (subtract_from_numa(rc, &nl, numa_id, qos, &req) != 0) return -1;
  }
  return 0;
}

/* ---------------------------------------------------------------- Score strategies */

static int64_t weight_of(const spx_nrt_params* p, int32_t res) { /* resourceToWeightMap.weight score.go:49-60 */
  for (int i = 0; i < p->n_weights; ++i)
    if (p->weight_res[i] == res) return p->weight[i] < 1 ? 1 : p->weight[i];
  return 1;
}

/* leastAllocatedScore / mostAllocatedScore least_allocated.go:44-55, most_allocated.go:44-54 */
static int64_t alloc_score(int least, int32_t res, int64_t requested, int64_t capacity) {
  if (capacity == 0) return 0;         /* numaCapacity.CmpInt64(0) == 0 (also a missing key) */
  if (requested > capacity) return 0;  /* requested.Cmp(numaCapacity) > 0 */
  int64_t cap_v = q_value(res, capacity), req_v = q_value(res, requested);
  if (least) return (cap_v - req_v) * 100 / cap_v;
  return req_v * 100 / cap_v;
}

static int64_t strategy_score(const spx_nrt_params* p, const rlist* requested, const rlist* allocatable) {
  if (p->strategy == SPX_NRT_BALANCED_ALLOCATION) { /* balanced_allocation.go:27-54 */
    double fr[ORC_MAXR];
    int n = 0;
    /* Go ranges the map in random order; this restatement fixes ascending resource id so that the
     * float64 sums below are reproducible (order only matters in the last ulp, and only for n >= 3) */
    int order[ORC_MAXR];
    for (int i = 0; i < requested->n; ++i) order[i] = i;
    for (int i = 1; i < requested->n; ++i)
      for (int j = i; j > 0 && requested->res[order[j - 1]] > requested->res[order[j]]; --j) {
        int t = order[j];
        order[j] = order[j - 1];
        order[j - 1] = t;
      }
    for (int oi = 0; oi < requested->n; ++oi) {
      int i = order[oi];
      int k = rl_find(allocatable, requested->res[i]);
      int64_t cap_v = k >= 0 ? q_value(requested->res[i], allocatable->qty[k]) : 0;
      double f = cap_v == 0 ? 1.0 : (double)q_value(requested->res[i], requested->qty[i]) / (double)cap_v;
      if (f > 1) return 0;
      fr[n++] = f;
    }
    /* gonum stat.Variance(x, nil): corrected two-pass, unbiased (n-1).  With >= 3 fractions the result
     * depends on Go's (random) map order in the last ulp — parity-unpinned there; n == 1 is 0/0 = NaN. */
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += fr[i];
    double mean = sum / (double)n;
    double ss = 0, comp = 0;
    for (int i = 0; i < n; ++i) {
      double d = fr[i] - mean;
      ss += d * d;
      comp += d;
    }
    double variance = (ss - comp * comp / (double)n) / ((double)n - 1);
    return (int64_t)((1 - variance) * 100.0);
  }
  int least = p->strategy == SPX_NRT_LEAST_ALLOCATED;
  int64_t numa_node_score = 0, weight_sum = 0;
  for (int i = 0; i < requested->n; ++i) { /* least_allocated.go:29-38 */
    int k = rl_find(allocatable, requested->res[i]);
    int64_t rs = alloc_score(least, requested->res[i], requested->qty[i], k >= 0 ? allocatable->qty[k] : 0);
    int64_t w = weight_of(p, requested->res[i]);
    numa_node_score += rs * w;
    weight_sum += w;
  }
  if (weight_sum == 0) return 0; /* empty request list: the reference divides by zero and panics (appendix B.2) */
  return numa_node_score / weight_sum;
}

/* scoreForEachNUMANode score.go:110-124 */
static int64_t score_for_each_numa(const spx_nrt_params* p, const rlist* requested, const numa_list* nl) {
  int64_t min_score = 0;
  for (int z = 0; z < nl->n; ++z) {
    int64_t s = strategy_score(p, requested, &nl->z[z].resources);
    if (min_score == 0 || (s != 0 && s < min_score)) min_score = s;
  }
  return min_score;
}

/* ---------------------------------------------------------------- LeastNUMANodes */

/* nodesAvgDistance least_numa.go:115-138 (float32) */
static float nodes_avg_distance(const numa_list* nl, const int* combo, int k) {
  if (k == 0) return 255.0f;
  int accu = 0;
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) {
      const numa_node* n1 = &nl->z[combo[a]];
      int want = nl->z[combo[b]].id;
      int cost = 255; /* maxDistanceValue when Costs has no entry */
      for (int c = 0; c < n1->n_cost; ++c)
        if (n1->cost_id[c] == want) cost = (int)n1->cost_val[c];
      accu += cost;
    }
  return (float)accu / (float)(k * k);
}

/* combin.Combinations(n, k): lexicographic successor; returns 0 when exhausted */
static int next_combination(int* c, int n, int k) {
  int i = k - 1;
  while (i >= 0 && c[i] == n - k + i) --i;
  if (i < 0) return 0;
  ++c[i];
  for (int j = i + 1; j < k; ++j) c[j] = c[j - 1] + 1;
  return 1;
}

/* numaNodesRequired + findSuitableCombination least_numa.go:156-208; returns 1 and the id bitmask, or 0 (nil) */
static int numa_nodes_required(const spx_resource_classes* rc, int qos, const numa_list* nl, const rlist* resources,
                               uint64_t* bm_out, int* is_min_distance) {
  for (int k = 1; k <= nl->n; ++k) {
    int combo[ORC_MAXZ];
    /* minAvgDistanceInCombinations over ALL combinations of this size (:102-113) */
    float min_avg = 255.0f;
    for (int i = 0; i < k; ++i) combo[i] = i;
    do {
      float d = nodes_avg_distance(nl, combo, k);
      if (d < min_avg) min_avg = d;
    } while (next_combination(combo, nl->n, k));

    int best[ORC_MAXZ], have_best = 0;
    float min_distance = 256.0f;
    for (int i = 0; i < k; ++i) combo[i] = i;
    do {
      /* isValidCombineResources :224-233: every node of the combination reports every requested name */
      int valid = 1;
      for (int a = 0; a < k && valid; ++a)
        for (int i = 0; i < resources->n; ++i)
          if (rl_find(&nl->z[combo[a]].resources, resources->res[i]) < 0) {
            valid = 0;
            break;
          }
      if (!valid) continue;
      /* combineResources :140-154 + checkResourcesFit :210-222 */
      int fit = 1;
      for (int i = 0; i < resources->n && fit; ++i) {
        if (resources->qty[i] == 0) continue;
        int64_t sum = 0;
        for (int a = 0; a < k; ++a) {
          int kk = rl_find(&nl->z[combo[a]].resources, resources->res[i]);
          if (kk >= 0) sum += nl->z[combo[a]].resources.qty[kk];
        }
        if (!is_suitable(rc, qos, resources->res[i], resources->qty[i], sum)) fit = 0;
      }
      if (!fit) continue;
      float distance = nodes_avg_distance(nl, combo, k);
      if (distance == min_avg) { /* :195-198 */
        uint64_t bm = 0;
        for (int a = 0; a < k; ++a) bm |= 1ull << nl->z[combo[a]].id;
        *bm_out = bm;
        *is_min_distance = 1;
        return 1;
      }
      if (distance < min_distance) { /* :200-203 */
        min_distance = distance;
        memcpy(best, combo, sizeof(int) * (size_t)k);
        have_best = 1;
      }
    } while (next_combination(combo, nl->n, k));
    if (have_best) {
      uint64_t bm = 0;
      for (int a = 0; a < k; ++a) bm |= 1ull << nl->z[best[a]].id;
      *bm_out = bm;
      *is_min_distance = 0;
      return 1;
    }
  }
  return 0;
}

/* onlyNonNUMAResources pluginhelpers.go:163-173 */
static int only_non_numa(const numa_list* nl, const rlist* resources) {
  for (int i = 0; i < resources->n; ++i)
    for (int z = 0; z < nl->n; ++z)
      if (rl_find(&nl->z[z].resources, resources->res[i]) >= 0) return 0;
  return 1;
}

/* subtractFromNUMAs numaresources.go:184-215 — `nodes` are NUMA ids used as LIST POSITIONS (appendix B.1) */
static void subtract_from_numas(const rlist* resources, numa_list* nl, uint64_t bits) {
  for (int i = 0; i < resources->n; ++i) {
    int64_t quantity = resources->qty[i];
    for (int pos = 0; pos < 64; ++pos) {
      if (!(bits >> pos & 1)) continue;
      if (quantity == 0) break;
      if (pos >= nl->n) continue; /* the reference would index out of range; the generator keeps id == position */
      int k = rl_find(&nl->z[pos].resources, resources->res[i]);
      if (k < 0) continue;
      int64_t available = nl->z[pos].resources.qty[k];
      if (quantity >= available) {
        quantity -= available;
        nl->z[pos].resources.qty[k] = 0;
      } else {
        nl->z[pos].resources.qty[k] = available - quantity;
        quantity = 0;
      }
    }
  }
}

/* normalizeScore least_numa.go:90-100 */
int64_t orc_nrt_normalize_score(int numa_nodes_count, int is_min_avg_distance, int highest_numa_id) {
  int64_t numa_node_score = 100 / (int64_t)highest_numa_id;
  int64_t score = 100 - (int64_t)numa_nodes_count * numa_node_score;
  if (is_min_avg_distance) return score + numa_node_score / 2;
  return score;
}

/* numaNodesRequired on the pod's effective request against the node's NUMA list (least_numa_test.go:35) */
int orc_nrt_numa_nodes_required(const spx_nrt_objects* nrt, const spx_resource_classes* rc, const spx_pod_objects* pods,
                                int64_t pod, int64_t node, int qos, uint64_t* bitmask, int* is_min_distance) {
  numa_list nl;
  numa_list_of(nrt, node, &nl);
  rlist resources;
  effective_request(pods, pod, &resources);
  *bitmask = 0;
  *is_min_distance = 0;
  return numa_nodes_required(rc, qos, &nl, &resources, bitmask, is_min_distance);
}

/* ---------------------------------------------------------------- Score */

/* TopologyMatch.Score score.go:62-102 (+ handlers :142-191, least_numa.go:35-88) */
int64_t orc_nrt_score(const spx_nrt_objects* nrt, const spx_resource_classes* rc, const spx_pod_objects* pods,
                      const spx_nrt_params* p, int64_t pod, int64_t node) {
  int qos = orc_pod_qos(pods, pod);
  if (qos != SPX_QOS_GUARANTEED) return 100;   /* :71-75 */
  if (!nrt->fresh[node]) return 0;             /* :79-82 */
  if (!nrt->has_nrt[node]) return 0;           /* :83-86 */
  tm_conf conf = conf_of(nrt, node);
  numa_list nl;
  if (p->strategy == SPX_NRT_LEAST_NUMA_NODES) { /* scoringHandlerFromTopologyManagerConfig :167-176 */
    numa_list_of(nrt, node, &nl);
    if (conf.scope == 1) { /* leastNUMAPodScopeScore least_numa.go:73-88 */
      rlist resources;
      effective_request(pods, pod, &resources);
      if (only_non_numa(&nl, &resources)) return 100;
      uint64_t bm;
      int is_min;
      if (!numa_nodes_required(rc, qos, &nl, &resources, &bm, &is_min)) return 0;
      return orc_nrt_normalize_score(__builtin_popcountll(bm), is_min, conf.max_numa);
    }
    /* leastNUMAContainerScopeScore least_numa.go:35-71 */
    int max_count = 0, all_min = 1;
    for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) { /* init containers first, then app */
      rlist req;
      ctr_requests(pods, c, &req);
      if (only_non_numa(&nl, &req)) continue;
      uint64_t bm;
      int is_min;
      if (!numa_nodes_required(rc, qos, &nl, &req, &bm, &is_min)) return 0;
      if (!is_min) all_min = 0;
      int cnt = __builtin_popcountll(bm);
      if (cnt > max_count) max_count = cnt;
      subtract_from_numas(&req, &nl, bm);
    }
    if (max_count == 0) return 100;
    return orc_nrt_normalize_score(max_count, all_min, conf.max_numa);
  }
  if (conf.policy != 3) return 0; /* :177-179 */
  numa_list_of(nrt, node, &nl);
  if (conf.scope == 1) { /* podScopeScore :142-150 */
    rlist resources;
    effective_request(pods, pod, &resources);
    return score_for_each_numa(p, &resources, &nl);
  }
  /* containerScopeScore :152-165: int64(stat.Mean(per-container scores)) over init + app containers */
  double sum = 0;
  int n = 0;
  for (int32_t c = pods->ctr_ptr[pod]; c < pods->ctr_ptr[pod + 1]; ++c) {
    rlist req;
    ctr_requests(pods, c, &req);
    sum += (double)score_for_each_numa(p, &req, &nl);
    ++n;
  }
  if (n == 0) return 0; /* a pod without containers cannot exist; stat.Mean of nothing is NaN in the reference */
  return (int64_t)(sum / (double)n);
}

/* ---------------------------------------------------------------- hooks for the reference's helper-level tables
 * (numaresources_test.go, pluginhelpers_test.go, nodeconfig/topologymanager_test.go, pkg/util/resource_test.go):
 * thin exported views of the static restatements above, so that tests/ can pin them one by one. */

int orc_nrt_is_host_level(const spx_resource_classes* rc, int32_t res) { return is_host_level(rc, res); }
int orc_nrt_is_numa_affine(const spx_resource_classes* rc, int32_t res) { return is_numa_affine(rc, res); }

/* util.GetPodEffectiveRequest: writes up to cap (resource id, quantity) pairs, returns their number */
int orc_pod_effective_request(const spx_pod_objects* pods, int64_t pod, int32_t* res_out, int64_t* qty_out, int cap) {
  rlist r;
  effective_request(pods, pod, &r);
  int n = r.n < cap ? r.n : cap;
  for (int i = 0; i < n; ++i) {
    res_out[i] = r.res[i];
    qty_out[i] = r.qty[i];
  }
  return r.n;
}

/* TopologyManagerFromNodeResourceTopology: policy 0 none / 1 best-effort / 2 restricted / 3 single-numa-node,
 * scope 0 container / 1 pod */
void orc_nrt_conf(const spx_nrt_objects* nrt, int64_t node, int* policy, int* scope, int* max_numa) {
  tm_conf c = conf_of(nrt, node);
  *policy = c.policy;
  *scope = c.scope;
  *max_numa = c.max_numa;
}

/* onlyNonNUMAResources(node's NUMANodeList, requests of the pod's first container) */
int orc_nrt_only_non_numa(const spx_nrt_objects* nrt, int64_t node, const spx_pod_objects* pods, int64_t pod) {
  numa_list nl;
  numa_list_of(nrt, node, &nl);
  rlist r;
  ctr_requests(pods, pods->ctr_ptr[pod], &r);
  return only_non_numa(&nl, &r);
}

static void dump_zones(const numa_list* nl, const int32_t* q_res, int n_q, int64_t* out) {
  for (int z = 0; z < nl->n; ++z)
    for (int i = 0; i < n_q; ++i) {
      int k = rl_find(&nl->z[z].resources, q_res[i]);
      out[z * n_q + i] = k >= 0 ? nl->z[z].resources.qty[k] : -1;
    }
}

/* subtractResourcesFromNUMANodeList(nodes, numa_id, qos, requests of the pod's first container); out[z*n_q+i] =
 * zone z's quantity of q_res[i] afterwards (-1: not reported); returns 0, or -1 for the reference's error */
int orc_nrt_test_subtract_numa(const spx_nrt_objects* nrt, const spx_resource_classes* rc, int64_t node, int numa_id, int qos,
                               const spx_pod_objects* pods, int64_t pod, const int32_t* q_res, int n_q, int64_t* out) {
  numa_list nl;
  numa_list_of(nrt, node, &nl);
  rlist r;
  ctr_requests(pods, pods->ctr_ptr[pod], &r);
  int rc_ = subtract_from_numa(rc, &nl, numa_id, qos, &r);
  dump_zones(&nl, q_res, n_q, out);
  return rc_;
}

/* subtractFromNUMAs(requests of the pod's first container, nodes, ids in `bits`...) */
void orc_nrt_test_subtract_numas(const spx_nrt_objects* nrt, int64_t node, const spx_pod_objects* pods, int64_t pod, uint64_t bits,
                                 const int32_t* q_res, int n_q, int64_t* out) {
  numa_list nl;
  numa_list_of(nrt, node, &nl);
  rlist r;
  ctr_requests(pods, pods->ctr_ptr[pod], &r);
  subtract_from_numas(&r, &nl, bits);
  dump_zones(&nl, q_res, n_q, out);
}

/* ---------------------------------------------------------------- preemption flow (preemption/preemption.go) */

/* resourcerequests.IsExclusive exclusive.go:78-102; nrt_resources = cache.ResourceNamesFromNRT of the node */
static int is_exclusive(const spx_resource_classes* rc, int qos, int32_t res, int64_t qty, const rlist* nrt_resources) {
  if (!rc_flag(rc, res, SPX_RC_NATIVE)) return rl_find(nrt_resources, res) >= 0; /* :83-85 */
  if (qos != SPX_QOS_GUARANTEED) return 0;                                        /* :86-89 */
  if (res == SPX_RES_CPU && qty % 1000 == 0 && qty > 0) return 1;                 /* :90-95: Value()*1000 == MilliValue(), > 0 */
  if ((res == SPX_RES_MEMORY || rc_flag(rc, res, SPX_RC_HUGEPAGE)) && qty > 0) return 1; /* :96-100 */
  return 0;
}

int orc_nrt_post_eviction(const spx_nrt_objects* nrt, const spx_resource_classes* rc, int64_t node, const spx_pod_objects* victims,
                          const uint8_t* victim_qos, const int32_t* ctr_numa, int32_t placement_present, int32_t placement_containers,
                          int64_t* out) {
  const int32_t z0 = nrt->zone_ptr[node], z1 = nrt->zone_ptr[node + 1];
  const int32_t e0 = nrt->zres_ptr[z0], e1 = nrt->zres_ptr[z1];
  for (int32_t e = e0; e < e1; ++e) out[e - e0] = nrt->zres_avail[e];
  if (!nrt->has_nrt[node]) return SPX_EVICT_NO_NRT;                         /* :40-42 */
  if (!victims || victims->n_pods == 0) return SPX_EVICT_NO_VICTIMS;        /* :44-46 */
  if (!placement_present) return SPX_EVICT_NO_PLACEMENT;                    /* :48-50 */
  if (placement_containers == 0) return SPX_EVICT_NO_CONTAINERS;            /* :52-54 */
  rlist names; /* ResourceNamesFromNRT: every resource any zone reports */
  names.n = 0;
  for (int32_t e = e0; e < e1; ++e) rl_set(&names, nrt->zres_res[e], 0);
  /* accumulateResourcesToAddPerNUMA :66-112 */
  static __thread rlist to_add[64];
  int used[64] = {0};
  int any = 0;
  for (int64_t v = 0; v < victims->n_pods; ++v) {
    int qos = victim_qos[v];
    if (qos != SPX_QOS_GUARANTEED && !orc_include_non_native(victims, rc, v)) continue; /* :70-73 */
    for (int32_t c = victims->ctr_ptr[v]; c < victims->ctr_ptr[v + 1]; ++c) {
      if (victims->ctr_kind[c] != SPX_CTR_APP) continue; /* victim.Spec.Containers */
      int numa = ctr_numa[c];
      if (numa == SPX_EVICT_CTR_UNKNOWN || numa == -1 || numa >= 64) continue; /* :81-85 */
      for (int32_t i = victims->req_ptr[c]; i < victims->req_ptr[c + 1]; ++i) {
        if (!is_exclusive(rc, qos, victims->req_res[i], victims->req_qty[i], &names)) continue; /* :88-90 */
        if (!used[numa]) {
          used[numa] = 1;
          to_add[numa].n = 0;
        }
        int k = rl_find(&to_add[numa], victims->req_res[i]);
        rl_set(&to_add[numa], victims->req_res[i], (k >= 0 ? to_add[numa].qty[k] : 0) + victims->req_qty[i]);
        any = 1;
      }
    }
  }
  if (!any) return SPX_EVICT_NOTHING_TO_ADD; /* :60-62 */
  /* addResourcesToNodeResourcesTopology :114-157 */
  for (int32_t z = z0; z < z1; ++z) {
    int id = nrt->zone_numa_id[z];
    if (id < 0 || id >= 64 || !used[id]) continue; /* NameToID error / nothing to add for this zone */
    for (int a = 0; a < to_add[id].n; ++a) {
      for (int32_t e = nrt->zres_ptr[z]; e < nrt->zres_ptr[z + 1]; ++e) {
        if (nrt->zres_res[e] != to_add[id].res[a]) continue;
        int64_t tmp = nrt->zres_avail[e] + to_add[id].qty[a];
        if (tmp > nrt->zres_allocatable[e]) { /* :136-148 one mistake voids the whole update */
          for (int32_t r = e0; r < e1; ++r) out[r - e0] = nrt->zres_avail[r];
          return SPX_EVICT_EXCEEDS_ALLOCATABLE;
        }
        out[e - e0] = tmp;
        break;
      }
    }
  }
  return SPX_EVICT_OK;
}
