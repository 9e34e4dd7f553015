/*
 * orc_capacity.c — restatement of CapacityScheduling.PreFilter (TEST INFRASTRUCTURE, see spx_oracle.h).
 *
 * Follows pkg/capacityscheduling/capacity_scheduling.go:208-283 (PreFilter), :865-883
 * (computePodResourceRequest) and pkg/capacityscheduling/elasticquota.go:48-59, :117-131, :189-221
 * (aggregatedUsedOverMinWith, usedOverMaxWith, usedOverMin, cmp, cmp2).  framework.Resource.Add /
 * SetMaxResource are upstream (k8s.io/kubernetes v1.35.7 pkg/scheduler/framework/types.go): cpu in millis,
 * memory/ephemeral-storage/pods by Value(), every other name only if schedutil.IsScalarResourceName.
 */
#include <string.h>

#include "spx_oracle.h"

typedef struct fres { /* framework.Resource over the quota slot vector */
  int64_t v[SPX_QUOTA_SLOTS];
  uint8_t present; /* scalar keys that exist in ScalarResources (bits 4..7) */
} fres;

static int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }

static int slot_of(const spx_quota_objects* q, const spx_resource_classes* rc, int32_t res) {
  if (res == SPX_RES_CPU) return 0;
  if (res == SPX_RES_MEMORY) return 1;
  if (res == SPX_RES_EPHEMERAL) return 2;
  if (res == SPX_RES_PODS) return 3;
  if (!rc || res < 0 || res >= rc->n_res || !(rc->flags[res] & SPX_RC_SCALAR)) return -1; /* not a scalar name: dropped by Add */
  for (int s = 0; s < q->n_scalar_slots; ++s)
    if (q->scalar_res[s] == res) return 4 + s;
  return -2; /* scalar resource without a slot: the caller's slot table is incomplete */
}

/* Resource.Add(ResourceList) */
static void fres_add_list(fres* r, const spx_quota_objects* q, const spx_resource_classes* rc, const int32_t* res, const int64_t* qty, int32_t lo, int32_t hi) {
  for (int32_t i = lo; i < hi; ++i) {
    int s = slot_of(q, rc, res[i]);
    if (s < 0) continue;
    r->v[s] = wadd(r->v[s], qty[i]);
    if (s >= 4) r->present |= (uint8_t)(1u << s);
  }
}

/* Resource.SetMaxResource(ResourceList): cpu, memory, ephemeral-storage, scalars (not pods) */
static void fres_set_max_list(fres* r, const spx_quota_objects* q, const spx_resource_classes* rc, const int32_t* res, const int64_t* qty, int32_t lo, int32_t hi) {
  for (int32_t i = lo; i < hi; ++i) {
    int s = slot_of(q, rc, res[i]);
    if (s < 0 || s == 3) continue;
    if (qty[i] > r->v[s]) r->v[s] = qty[i];
    if (s >= 4) r->present |= (uint8_t)(1u << s);
  }
}

/* computePodResourceRequest capacity_scheduling.go:865-883 */
static void pod_request(const spx_pod_objects* p, const spx_quota_objects* q, const spx_resource_classes* rc, int64_t pod, fres* out) {
  memset(out, 0, sizeof *out);
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c)
    if (p->ctr_kind[c] == SPX_CTR_APP) fres_add_list(out, q, rc, p->req_res, p->req_qty, p->req_ptr[c], p->req_ptr[c + 1]);
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c)
    if (p->ctr_kind[c] != SPX_CTR_APP) fres_set_max_list(out, q, rc, p->req_res, p->req_qty, p->req_ptr[c], p->req_ptr[c + 1]);
  if (p->ovh_ptr) fres_add_list(out, q, rc, p->ovh_res, p->ovh_qty, p->ovh_ptr[pod], p->ovh_ptr[pod + 1]);
}

/* Resource.Add(util.ResourceList(x)): every field, scalar keys carried over */
static void fres_add(fres* r, const fres* x) {
  for (int s = 0; s < SPX_QUOTA_SLOTS; ++s) r->v[s] = wadd(r->v[s], x->v[s]);
  r->present |= x->present;
}

static void fres_load(fres* r, const int64_t* v, uint8_t present) {
  memcpy(r->v, v, sizeof r->v);
  r->present = present;
}

/* cmp2 elasticquota.go:193-221 */
int orc_quota_cmp2(const int64_t* x1, uint8_t x1_present, const int64_t* x2, const int64_t* y, uint8_t y_present, int64_t bound) {
  for (int s = 0; s < 4; ++s)
    if (wadd(x1[s], x2[s]) > y[s]) return 1;
  for (int s = 4; s < SPX_QUOTA_SLOTS; ++s) {
    if (!(x1_present >> s & 1)) continue; /* ranges over x1.ScalarResources */
    int64_t y_quant = (y_present >> s & 1) ? y[s] : bound;
    if (wadd(x1[s], x2[s]) > y_quant) return 1;
  }
  return 0;
}

/* CapacityScheduling.PreFilter -> 0 Success, SPX_QUOTA_ST_OVER_MAX, SPX_QUOTA_ST_OVER_MIN */
int orc_capacity_prefilter(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* q, int64_t pod) {
  static const int64_t zero[SPX_QUOTA_SLOTS] = {0};
  const int32_t ns = pods->ns[pod];
  if (ns < 0 || ns >= q->n_namespaces || !q->has_quota[ns]) return 0; /* eq == nil :216-222 */
  fres pod_req, nom_in_eq, nom_total;
  pod_request(pods, q, rc, pod, &pod_req);
  memset(&nom_in_eq, 0, sizeof nom_in_eq);
  memset(&nom_total, 0, sizeof nom_total);
  for (int64_t j = 0; j < q->n_nominated; ++j) { /* :236-253 (node order is irrelevant to the sums) */
    if (q->nom_pending_index[j] == pod) continue; /* p.UID == pod.UID */
    const int32_t pns = q->nom_ns[j];
    if (pns < 0 || pns >= q->n_namespaces || !q->has_quota[pns]) continue; /* info == nil */
    fres preq;
    pod_request(q->nom_pods, q, rc, j, &preq);
    if (pns == ns && q->nom_priority[j] >= pods->priority[pod]) {
      fres_add(&nom_in_eq, &preq);
      fres_add(&nom_total, &preq);
    } else if (pns != ns) {
      /* !info.usedOverMin(): cmp(Used, Min, LowerBoundOfMin) */
      const int over = orc_quota_cmp2(q->used + (size_t)pns * SPX_QUOTA_SLOTS, q->used_present[pns], zero,
                                      q->min + (size_t)pns * SPX_QUOTA_SLOTS, q->min_present[pns], 0);
      if (!over) fres_add(&nom_total, &preq);
    }
  }
  fres_add(&nom_in_eq, &pod_req);
  fres_add(&nom_total, &pod_req);
  /* eq.usedOverMaxWith(nominatedPodsReqInEQWithPodReq) :275-277 */
  if (orc_quota_cmp2(nom_in_eq.v, nom_in_eq.present, q->used + (size_t)ns * SPX_QUOTA_SLOTS, q->max + (size_t)ns * SPX_QUOTA_SLOTS,
                     q->max_present[ns], INT64_MAX))
    return SPX_QUOTA_ST_OVER_MAX;
  /* elasticQuotaInfos.aggregatedUsedOverMinWith(nominatedPodsReqWithPodReq) :279-281 */
  fres used, min, t;
  memset(&used, 0, sizeof used);
  memset(&min, 0, sizeof min);
  for (int32_t k = 0; k < q->n_namespaces; ++k) {
    if (!q->has_quota[k]) continue;
    fres_load(&t, q->used + (size_t)k * SPX_QUOTA_SLOTS, q->used_present[k]);
    fres_add(&used, &t);
    fres_load(&t, q->min + (size_t)k * SPX_QUOTA_SLOTS, q->min_present[k]);
    fres_add(&min, &t);
  }
  fres_add(&used, &nom_total);
  if (orc_quota_cmp2(used.v, used.present, zero, min.v, min.present, 0)) return SPX_QUOTA_ST_OVER_MIN;
  return 0;
}

/* CapacityScheduling.Reserve capacity_scheduling.go:350-364 -> ElasticQuotaInfo.addPodIfNotPresent elasticquota.go:153-166 ->
 * reserveResource :89-98, on a caller-owned Used table ([n_namespaces][SPX_QUOTA_SLOTS] + the scalar-key bits): the namespace's Used
 * grows by computePodResourceRequest(pod); SetScalar creates a scalar key the request carries.  A namespace without an ElasticQuota
 * reserves nothing.  (The `pods` set of addPodIfNotPresent only guards against adding one pod twice.) */
void orc_capacity_reserve(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* q, int64_t pod,
                          int64_t* used, uint8_t* used_present) {
  const int32_t ns = pods->ns[pod];
  if (ns < 0 || ns >= q->n_namespaces || !q->has_quota[ns]) return;
  fres req;
  pod_request(pods, q, rc, pod, &req);
  int64_t* u = used + (size_t)ns * SPX_QUOTA_SLOTS;
  for (int s = 0; s < 4; ++s) u[s] = wadd(u[s], req.v[s]);
  for (int s = 4; s < SPX_QUOTA_SLOTS; ++s)
    if (req.present >> s & 1) { /* ranges over request.ScalarResources */
      u[s] = wadd((used_present[ns] >> s & 1) ? u[s] : 0, req.v[s]);
      used_present[ns] |= (uint8_t)(1u << s);
    }
}
