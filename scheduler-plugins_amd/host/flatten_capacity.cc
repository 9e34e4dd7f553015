// flatten_capacity.cc — object tables -> SoA for CapacityScheduling.PreFilter (host side, once per snapshot).
//
// The reference clones every ElasticQuotaInfo per pod (capacity_scheduling.go:211, :795-803), walks every
// node's nominated pods per pod (:236-253) and re-sums all quotas per pod (elasticquota.go:48-59).  Hoisted:
//   computePodResourceRequest for pending and nominated pods          capacity_scheduling.go:865-883
//   Σ Used, Σ Min over all quotas                                    elasticquota.go:52-55
//   per namespace: nominated requests of OTHER namespaces whose quota is not over min   :248-250
// What stays per pod (the kernel): the same-namespace nominated pods with priority >= the pod's, and cmp2.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/spx.h"
#include "parallel.hpp"

namespace {

constexpr int S = SPX_QUOTA_SLOTS;

struct Vec {
  int64_t v[S] = {0};
  uint8_t present = 0;
};

inline int64_t wadd(int64_t a, int64_t b) { return static_cast<int64_t>(static_cast<uint64_t>(a) + static_cast<uint64_t>(b)); }

int slot_of(const spx_quota_objects* q, const spx_resource_classes* rc, int32_t res) {
  if (res == SPX_RES_CPU) return 0;
  if (res == SPX_RES_MEMORY) return 1;
  if (res == SPX_RES_EPHEMERAL) return 2;
  if (res == SPX_RES_PODS) return 3;
  if (!rc || res < 0 || res >= rc->n_res || !(rc->flags[res] & SPX_RC_SCALAR)) return -1;
  for (int s = 0; s < q->n_scalar_slots; ++s)
    if (q->scalar_res[s] == res) return 4 + s;
  return -2;
}

// framework.Resource.Add / SetMaxResource over one resource list; false when a scalar has no slot
bool apply(Vec& r, const spx_quota_objects* q, const spx_resource_classes* rc, const int32_t* res, const int64_t* qty, int32_t lo,
           int32_t hi, bool max_mode) {
  for (int32_t i = lo; i < hi; ++i) {
    const int s = slot_of(q, rc, res[i]);
    if (s == -2) return false;
    if (s < 0 || (max_mode && s == 3)) continue;
    if (max_mode) r.v[s] = qty[i] > r.v[s] ? qty[i] : r.v[s];
    else r.v[s] = wadd(r.v[s], qty[i]);
    if (s >= 4) r.present |= static_cast<uint8_t>(1u << s);
  }
  return true;
}

bool pod_request(const spx_pod_objects* p, const spx_quota_objects* q, const spx_resource_classes* rc, int64_t pod, Vec& out) {
  out = Vec{};
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c)
    if (p->ctr_kind[c] == SPX_CTR_APP && !apply(out, q, rc, p->req_res, p->req_qty, p->req_ptr[c], p->req_ptr[c + 1], false)) return false;
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c)
    if (p->ctr_kind[c] != SPX_CTR_APP && !apply(out, q, rc, p->req_res, p->req_qty, p->req_ptr[c], p->req_ptr[c + 1], true)) return false;
  if (p->ovh_ptr && !apply(out, q, rc, p->ovh_res, p->ovh_qty, p->ovh_ptr[pod], p->ovh_ptr[pod + 1], false)) return false;
  return true;
}

void add(Vec& r, const int64_t* v, uint8_t present) {
  for (int s = 0; s < S; ++s) r.v[s] = wadd(r.v[s], v[s]);
  r.present |= present;
}

bool cmp2(const int64_t* x1, uint8_t x1p, const int64_t* x2, const int64_t* y, uint8_t yp, int64_t bound) {
  for (int s = 0; s < 4; ++s)
    if (wadd(x1[s], x2[s]) > y[s]) return true;
  for (int s = 4; s < S; ++s) {
    if (!((x1p >> s) & 1)) continue;
    const int64_t yq = ((yp >> s) & 1) ? y[s] : bound;
    if (wadd(x1[s], x2[s]) > yq) return true;
  }
  return false;
}

}  // namespace

extern "C" int spx_flatten_quota(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* q,
                                 int32_t* pod_ns, int32_t* pod_priority, int64_t* pod_req, uint8_t* pod_req_present,
                                 int64_t* agg_used, uint8_t* agg_used_present, int64_t* agg_min, uint8_t* agg_min_present,
                                 int64_t* other_nominated, uint8_t* other_nominated_present, int32_t* nom_ptr,
                                 int32_t* nom_priority, int64_t* nom_pending_index, int64_t* nom_req, uint8_t* nom_req_present) {
  if (!pods || !q || !pod_ns || !pod_priority || !pod_req || !pod_req_present || !agg_used || !agg_used_present || !agg_min ||
      !agg_min_present || !other_nominated || !other_nominated_present || !nom_ptr || !nom_priority || !nom_pending_index ||
      !nom_req || !nom_req_present)
    return SPX_ERR_ARG;
  if (q->n_scalar_slots < 0 || q->n_scalar_slots > S - 4) return SPX_ERR_ARG;
  const int32_t NS = q->n_namespaces;
  static const int64_t zero[S] = {0};
  std::atomic<bool> bad{false};
  spx_host::parallel_rows(pods->n_pods, [&](int64_t row0, int64_t row1) {  // pods are independent (2-3 ms serial for 62.5k)
    for (int64_t p = row0; p < row1; ++p) {
      Vec r;
      if (!pod_request(pods, q, rc, p, r)) {
        bad = true;
        return;
      }
      pod_ns[p] = pods->ns[p];
      pod_priority[p] = pods->priority[p];
      std::memcpy(pod_req + p * S, r.v, sizeof r.v);
      pod_req_present[p] = r.present;
    }
  }, 4096);
  if (bad) return SPX_ERR_ARG;
  Vec used, mn;
  for (int32_t k = 0; k < NS; ++k) {
    if (!q->has_quota[k]) continue;
    add(used, q->used + static_cast<int64_t>(k) * S, q->used_present[k]);
    add(mn, q->min + static_cast<int64_t>(k) * S, q->min_present[k]);
  }
  std::memcpy(agg_used, used.v, sizeof used.v);
  *agg_used_present = used.present;
  std::memcpy(agg_min, mn.v, sizeof mn.v);
  *agg_min_present = mn.present;
  // nominated pods grouped by namespace (only those subject to a quota: info != nil)
  std::vector<Vec> reqs(static_cast<size_t>(q->n_nominated));
  std::vector<int32_t> count(static_cast<size_t>(NS) + 1, 0);
  for (int64_t j = 0; j < q->n_nominated; ++j) {
    if (!pod_request(q->nom_pods, q, rc, j, reqs[j])) return SPX_ERR_ARG;
    const int32_t ns = q->nom_ns[j];
    if (ns >= 0 && ns < NS && q->has_quota[ns]) ++count[ns + 1];
  }
  nom_ptr[0] = 0;
  for (int32_t k = 0; k < NS; ++k) nom_ptr[k + 1] = nom_ptr[k] + count[k + 1];
  std::vector<int32_t> fillp(nom_ptr, nom_ptr + NS);
  Vec all_not_over;  // Σ over namespaces whose quota is not over min
  std::vector<Vec> own(static_cast<size_t>(NS));
  for (int64_t j = 0; j < q->n_nominated; ++j) {
    const int32_t ns = q->nom_ns[j];
    if (ns < 0 || ns >= NS || !q->has_quota[ns]) continue;
    const int32_t at = fillp[ns]++;
    nom_priority[at] = q->nom_priority[j];
    nom_pending_index[at] = q->nom_pending_index[j];
    std::memcpy(nom_req + static_cast<int64_t>(at) * S, reqs[j].v, sizeof reqs[j].v);
    nom_req_present[at] = reqs[j].present;
    const bool over_min = cmp2(q->used + static_cast<int64_t>(ns) * S, q->used_present[ns], zero, q->min + static_cast<int64_t>(ns) * S,
                               q->min_present[ns], 0);
    if (!over_min) {
      add(all_not_over, reqs[j].v, reqs[j].present);
      add(own[ns], reqs[j].v, reqs[j].present);
    }
  }
  // NB: a pending pod that is itself nominated in ANOTHER namespace's list cannot exist (same pod, same namespace),
  // so the uid exclusion (capacity_scheduling.go:239-241) only ever concerns the same-namespace list handled on device.
  for (int32_t k = 0; k < NS; ++k) {
    Vec o;
    for (int s = 0; s < S; ++s) o.v[s] = static_cast<int64_t>(static_cast<uint64_t>(all_not_over.v[s]) - static_cast<uint64_t>(own[k].v[s]));
    // key presence of the union of the OTHER namespaces' scalar maps
    uint8_t pr = 0;
    for (int32_t m = 0; m < NS; ++m)
      if (m != k) pr |= own[m].present;
    o.present = pr;
    std::memcpy(other_nominated + static_cast<int64_t>(k) * S, o.v, sizeof o.v);
    other_nominated_present[k] = o.present;
  }
  return SPX_OK;
}
