/*
 * orc_allocatable.c — restatement of noderesources.Allocatable (TEST INFRASTRUCTURE, see spx_oracle.h).
 *
 * Follows pkg/noderesources/allocatable.go:63-168 and pkg/noderesources/resource_allocation.go:49-131.
 */
#include "spx_oracle.h"

/* Go int64 arithmetic wraps; C signed overflow is undefined, so wrap through uint64. */
static int64_t wrap_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
static int64_t wrap_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }

/* calculateResourceAllocatableRequest (resource_allocation.go:79-100), allocatable half.
 * The `requested` half (calculatePodResourceRequest, :105-131) is computed by the reference and
 * never read by the scorer (allocatable.go:118-126 indexes `allocable` only) — SURVEY appendix B.3 —
 * so the raw score does not depend on the pod and it is not restated. */
static int64_t allocatable_of(const spx_node_objects* nodes, const spx_resource_classes* rc, int32_t res, int64_t node) {
  switch (res) {
    case SPX_RES_CPU: return nodes->alloc_cpu_milli[node];  /* GetAllocatable().GetMilliCPU() :84 */
    case SPX_RES_MEMORY: return nodes->alloc_mem[node];     /* :86 */
    case SPX_RES_EPHEMERAL: return nodes->alloc_eph[node];  /* :89 */
    default: break;
  }
  /* default: schedutil.IsScalarResourceName(resource) -> ScalarResources[resource] (:91-93) */
  if (rc && res >= 0 && res < rc->n_res && (rc->flags[res] & SPX_RC_SCALAR)) {
    for (int32_t i = nodes->scalar_ptr[node]; i < nodes->scalar_ptr[node + 1]; ++i)
      if (nodes->scalar_res[i] == res) return nodes->scalar_qty[i];
    return 0; /* missing map key */
  }
  return 0; /* "Requested resource not considered for node score calculation" :95-99 */
}

/* score(capacity, mode) allocatable.go:130-140 */
static int64_t mode_score(int64_t capacity, int32_t mode) {
  if (mode == SPX_MODE_LEAST) return wrap_mul(-1, capacity);
  if (mode == SPX_MODE_MOST) return capacity;
  return 0;
}

/* Allocatable.Score -> resourceAllocationScorer.score -> resourceScorer closure
 * (allocatable.go:63-71, resource_allocation.go:49-76, allocatable.go:117-128) */
int64_t orc_allocatable_score(const spx_node_objects* nodes, const spx_resource_classes* rc,
                              const spx_allocatable_params* p, int64_t node) {
  int64_t node_score = 0, weight_sum = 0;
  for (int32_t r = 0; r < p->n_res; ++r) { /* map iteration: order-free, + and * commute under wrap */
    int64_t resource_score = mode_score(allocatable_of(nodes, rc, p->res[r], node), p->mode);
    node_score = wrap_add(node_score, wrap_mul(resource_score, p->weight[r]));
    weight_sum = wrap_add(weight_sum, p->weight[r]);
  }
  return node_score / weight_sum; /* Go `/`: truncates toward zero, as C99 */
}

/* Allocatable.NormalizeScore allocatable.go:143-168 */
void orc_allocatable_normalize(int64_t* scores, int64_t n) {
  int64_t highest = -INT64_MAX; /* -math.MaxInt64 */
  int64_t lowest = INT64_MAX;
  for (int64_t i = 0; i < n; ++i) {
    if (scores[i] > highest) highest = scores[i];
    if (scores[i] < lowest) lowest = scores[i];
  }
  int64_t old_range = (int64_t)((uint64_t)highest - (uint64_t)lowest);
  const int64_t new_range = 100 - 0; /* fwk.MaxNodeScore - fwk.MinNodeScore */
  for (int64_t i = 0; i < n; ++i) {
    if (old_range == 0) {
      scores[i] = 0;
    } else {
      int64_t d = (int64_t)((uint64_t)scores[i] - (uint64_t)lowest);
      scores[i] = wrap_mul(d, new_range) / old_range + 0;
    }
  }
}
