// flatten_nrt.cc — object tables -> SoA for NodeResourceTopologyMatch (host side, once per snapshot).
//
// Hoisted out of the per-(pod,node) path (the reference redoes all of it inside every Filter/Score call):
//   NRT DeepCopy + OverReserve subtraction of assumed pods          cache/store.go:84-91, :315-356
//   TopologyManager config decode                                    nodeconfig/topologymanager.go:78-162
//   createNUMANodeList / extractResources / extractCosts             pluginhelpers.go:105-161
//   util.ResourceList(node allocatable) key set                      pkg/util/resource.go:30-44
//   per-size minimum average NUMA distance                           least_numa.go:102-138
//   pod QoS class, IncludeNonNative, GetPodEffectiveRequest          filter.go:183-186, pkg/util/resource.go:51-85
#include <algorithm>
#include <cstdint>
#include <climits>
#include <cstring>
#include <mutex>
#include <vector>

#include <atomic>

#include "../../include/spx.h"
#include "parallel.hpp"

namespace {

constexpr int Z = SPX_NRT_MAX_ZONES;
constexpr int RM = SPX_NRT_MAX_RES;
constexpr int CM = SPX_NRT_MAX_CTRS;

inline bool is_fixed_native(int32_t res) {
  return res == SPX_RES_CPU || res == SPX_RES_MEMORY || res == SPX_RES_EPHEMERAL || res == SPX_RES_PODS || res == SPX_RES_STORAGE;
}
inline bool rc_has(const spx_resource_classes* rc, int32_t res, int flag) {
  if (is_fixed_native(res)) return flag == SPX_RC_NATIVE;
  if (!rc || res < 0 || res >= rc->n_res) return false;
  return (rc->flags[res] & flag) != 0;
}

inline int slot_of(const spx_nrt_slots* s, int32_t res) {
  for (int i = 0; i < s->n_res; ++i)
    if (s->slot_res[i] == res) return i;
  return -1;
}

// v1qos.ComputePodQOS restated for the fields the hot path reads (cpu, memory; zero quantities ignored)
int pod_qos(const spx_pod_objects* p, int64_t pod) {
  int64_t req[2] = {0, 0}, lim[2] = {0, 0};
  bool has_req[2] = {false, false}, has_lim[2] = {false, false};
  bool guaranteed = true;
  for (int32_t c = p->ctr_ptr[pod]; c < p->ctr_ptr[pod + 1]; ++c) {
    for (int32_t i = p->req_ptr[c]; i < p->req_ptr[c + 1]; ++i) {
      const int32_t r = p->req_res[i];
      if ((r == SPX_RES_CPU || r == SPX_RES_MEMORY) && p->req_qty[i] > 0) {
        req[r] += p->req_qty[i];
        has_req[r] = true;
      }
    }
    bool found[2] = {false, false};
    for (int32_t i = p->lim_ptr[c]; i < p->lim_ptr[c + 1]; ++i) {
      const int32_t r = p->lim_res[i];
      if ((r == SPX_RES_CPU || r == SPX_RES_MEMORY) && p->lim_qty[i] > 0) {
        lim[r] += p->lim_qty[i];
        has_lim[r] = found[r] = true;
      }
    }
    if (!(found[0] && found[1])) guaranteed = false;
  }
  if (!has_req[0] && !has_req[1] && !has_lim[0] && !has_lim[1]) return SPX_QOS_BESTEFFORT;
  if (guaranteed)
    for (int r = 0; r < 2; ++r)
      if (has_req[r] && (!has_lim[r] || lim[r] != req[r])) guaranteed = false;
  if (guaranteed && (int(has_req[0]) + int(has_req[1])) == (int(has_lim[0]) + int(has_lim[1]))) return SPX_QOS_GUARANTEED;
  return SPX_QOS_BURSTABLE;
}

struct KV {
  int32_t res;
  int64_t qty;
};
inline KV* kv_find(std::vector<KV>& v, int32_t res) {
  for (auto& e : v)
    if (e.res == res) return &e;
  return nullptr;
}

}  // namespace

extern "C" int spx_flatten_nrt_slots(const spx_pod_objects* pods, const spx_nrt_objects* nrt, const spx_resource_classes* rc,
                                     const spx_nrt_params* p, int32_t* n_res_out, int32_t* slot_res, uint8_t* slot_flags,
                                     int64_t* slot_weight) {
  if (!pods || !nrt || !n_res_out || !slot_res || !slot_flags || !slot_weight) return SPX_ERR_ARG;
  // the distinct resource ids of three long arrays (every container request, every overhead entry, every zone resource: ~0.8 M entries
  // at config #5's share): scanned in pieces on the host threads, each piece's handful of ids merged under a lock
  std::vector<int32_t> ids;
  std::mutex ids_mu;
  auto scan = [&](const int32_t* res, int64_t n) {
    if (!res || n <= 0) return;
    spx_host::parallel_rows(n, [&](int64_t b, int64_t e) {
      int32_t mine[64];
      int n_mine = 0;
      bool spill = false;
      std::vector<int32_t> more;  // (more than 64 distinct ids in one piece: the call fails below anyway, but stay correct)
      for (int64_t i = b; i < e; ++i) {
        const int32_t r = res[i];
        bool seen = false;
        for (int k = 0; k < n_mine && !seen; ++k) seen = mine[k] == r;
        if (seen) continue;
        if (n_mine < 64) mine[n_mine++] = r;
        else if (std::find(more.begin(), more.end(), r) == more.end()) more.push_back(r), spill = true;
      }
      std::lock_guard<std::mutex> g(ids_mu);
      for (int k = 0; k < n_mine; ++k)
        if (std::find(ids.begin(), ids.end(), mine[k]) == ids.end()) ids.push_back(mine[k]);
      if (spill)
        for (const int32_t r : more)
          if (std::find(ids.begin(), ids.end(), r) == ids.end()) ids.push_back(r);
    }, 65536);
  };
  const int64_t n_ctr = pods->ctr_ptr[pods->n_pods];
  scan(pods->req_res, pods->req_ptr[n_ctr]);
  if (pods->ovh_ptr) scan(pods->ovh_res, pods->ovh_ptr[pods->n_pods]);
  const int32_t n_zones = nrt->zone_ptr[nrt->n_nodes];
  scan(nrt->zres_res, nrt->zres_ptr[n_zones]);
  std::sort(ids.begin(), ids.end());
  if (ids.size() > static_cast<size_t>(RM)) return SPX_ERR_ARG;  // more distinct resources than this build supports
  *n_res_out = static_cast<int32_t>(ids.size());
  for (size_t s = 0; s < ids.size(); ++s) {
    const int32_t r = ids[s];
    slot_res[s] = r;
    uint8_t f = 0;
    if (r == SPX_RES_CPU || r == SPX_RES_MEMORY || rc_has(rc, r, SPX_RC_HUGEPAGE)) f |= SPX_NRT_SLOT_AFFINE;
    if (r == SPX_RES_EPHEMERAL || r == SPX_RES_STORAGE || !rc_has(rc, r, SPX_RC_NATIVE)) f |= SPX_NRT_SLOT_HOST_LEVEL;
    if (r == SPX_RES_CPU) f |= SPX_NRT_SLOT_CPU;
    slot_flags[s] = f;
    int64_t w = 1;  // resourceToWeightMap.weight: missing or < 1 -> defaultWeight (score.go:49-60)
    if (p)
      for (int32_t k = 0; k < p->n_weights; ++k)
        if (p->weight_res[k] == r) w = p->weight[k] < 1 ? 1 : p->weight[k];
    slot_weight[s] = w;
  }
  return SPX_OK;
}

namespace {
// node i of the object tables -> row j of the SoA columns
int flatten_nrt_node(const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_nrt_slots* slots, int64_t i, int64_t j,
                     uint8_t* flags, int32_t* max_numa, uint8_t* n_zones, uint8_t* zone_id, uint8_t* zone_present, int64_t* zone_avail,
                     int32_t* zone_cost, float* min_avg_dist, uint8_t* node_present) {
  const int R = slots->n_res;
    // ---- TopologyManager config
    int scope = 0, policy = 0, mx = 8;
    const int lp = nrt->legacy_policy ? nrt->legacy_policy[i] : -1;
    if (lp >= 0) {
      policy = lp >> 1;
      scope = lp & 1;
    }
    if (nrt->attr_scope && nrt->attr_scope[i] >= 0) scope = nrt->attr_scope[i];
    if (nrt->attr_policy && nrt->attr_policy[i] >= 0) policy = nrt->attr_policy[i];
    if (nrt->attr_max_numa && nrt->attr_max_numa[i] > 1) mx = nrt->attr_max_numa[i] > 1024 ? 1024 : nrt->attr_max_numa[i];
    flags[j] = static_cast<uint8_t>((nrt->has_nrt[i] ? SPX_NRT_F_HAS_NRT : 0) | (nrt->fresh[i] ? SPX_NRT_F_FRESH : 0) |
                                    (policy == 3 ? SPX_NRT_F_SINGLE_NUMA : 0) | (scope == 1 ? SPX_NRT_F_POD_SCOPE : 0));
    max_numa[j] = mx;
    // ---- node-level key set of util.ResourceList(allocatable)
    uint8_t np = 0;
    for (int s = 0; s < R; ++s) {
      const int32_t r = slots->slot_res[s];
      bool has = r == SPX_RES_CPU || r == SPX_RES_MEMORY || r == SPX_RES_PODS || r == SPX_RES_EPHEMERAL;
      for (int32_t k = nodes->scalar_ptr[i]; !has && k < nodes->scalar_ptr[i + 1]; ++k) has = nodes->scalar_res[k] == r;
      if (has) np |= static_cast<uint8_t>(1u << s);
    }
    node_present[j] = np;
    // ---- NUMA node list (list order = zone order), with assumed pods subtracted from every zone
    int nz = 0;
    int32_t zsrc[Z];
    if (nrt->has_nrt[i]) {
      for (int32_t z = nrt->zone_ptr[i]; z < nrt->zone_ptr[i + 1]; ++z) {
        if (!nrt->zone_is_node[z]) continue;
        const int id = nrt->zone_numa_id[z];
        if (id < 0 || id > 64) continue;
        if (nz >= Z || id > 63) {
          return SPX_ERR_ARG;
        }  // beyond this build's limits (8 zones, ids 0..63)
        zsrc[nz] = z;
        zone_id[j * Z + nz] = static_cast<uint8_t>(id);
        uint8_t present = 0;
        for (int32_t k = nrt->zres_ptr[z]; k < nrt->zres_ptr[z + 1]; ++k) {
          const int s = slot_of(slots, nrt->zres_res[k]);
          if (s < 0) {
            return SPX_ERR_ARG;
          }
          int64_t avail = nrt->zres_avail[k];
          if (nrt->assumed_ptr)
            for (int32_t a = nrt->assumed_ptr[i]; a < nrt->assumed_ptr[i + 1]; ++a)
              for (int32_t q = nrt->arl_ptr[a]; q < nrt->arl_ptr[a + 1]; ++q)
                if (nrt->arl_res[q] == nrt->zres_res[k]) avail = avail < nrt->arl_qty[q] ? 0 : avail - nrt->arl_qty[q];
          present |= static_cast<uint8_t>(1u << s);
          zone_avail[(j * Z + nz) * R + s] = avail;
        }
        zone_present[j * Z + nz] = present;
        ++nz;
      }
    }
    n_zones[j] = static_cast<uint8_t>(nz);
    // ---- distance matrix by list position; 255 where Costs has no entry (least_numa.go:127-132)
    {
      // NUMA id -> list position (ids 0..63); an id carried by two zones keeps the slow walk below (every zone with that id takes
      // the entry).  With the map a zone's Costs list is walked once instead of once per column: the last entry for an id wins,
      // as in the column-by-column walk.
      int8_t pos_of[64];
      std::memset(pos_of, -1, sizeof pos_of);
      bool unique = true;
      for (int b = 0; b < nz; ++b) {
        const int id = zone_id[j * Z + b];
        unique &= pos_of[id] < 0;
        pos_of[id] = static_cast<int8_t>(b);
      }
      int32_t* row = zone_cost + j * Z * Z;
      for (int i = 0; i < Z * Z; ++i) row[i] = 255;
      if (nrt->zcost_ptr) {
        if (unique) {
          for (int a = 0; a < nz; ++a)
            for (int32_t k = nrt->zcost_ptr[zsrc[a]]; k < nrt->zcost_ptr[zsrc[a] + 1]; ++k) {
              const int32_t id = nrt->zcost_numa_id[k];
              if (id >= 0 && id < 64 && pos_of[id] >= 0) row[a * Z + pos_of[id]] = static_cast<int32_t>(nrt->zcost_value[k]);
            }
        } else {
          for (int a = 0; a < nz; ++a)
            for (int b = 0; b < nz; ++b) {
              const int want = zone_id[j * Z + b];
              for (int32_t k = nrt->zcost_ptr[zsrc[a]]; k < nrt->zcost_ptr[zsrc[a] + 1]; ++k)
                if (nrt->zcost_numa_id[k] == want) row[a * Z + b] = static_cast<int32_t>(nrt->zcost_value[k]);
            }
        }
      }
    }
    // ---- minAvgDistanceInCombinations for every subset size (float32 exactly as the reference)
    float best[Z];
    for (int k = 0; k < Z; ++k) best[k] = 255.0f;
    // every subset once, filed under its size; the sum over all ordered pairs of a subset grows from the subset without its
    // lowest member z: + cost[z][z] + sum over the others j of (cost[z][j] + cost[j][z])  (same integer as the double loop)
    int pair_sum[1 << Z], least[Z];
    pair_sum[0] = 0;
    for (int k = 0; k < Z; ++k) least[k] = INT32_MAX;
    const int32_t* c = zone_cost + j * Z * Z;
    for (unsigned m = 1; m < (1u << nz); ++m) {
      const int z = __builtin_ctz(m);
      const unsigned rest = m & (m - 1);
      int accu = pair_sum[rest] + c[z * Z + z];
      for (unsigned r = rest; r; r &= r - 1) {
        const int j = __builtin_ctz(r);
        accu += c[z * Z + j] + c[j * Z + z];
      }
      pair_sum[m] = accu;
      const int k = __builtin_popcount(m);
      if (accu < least[k - 1]) least[k - 1] = accu;
    }
    // float32(sum) / float32(k*k) is monotone in the sum: the minimum distance of a size is the distance of its smallest sum
    for (int k = 1; k <= nz; ++k) {
      const float d = static_cast<float>(least[k - 1]) / static_cast<float>(k * k);
      if (d < best[k - 1]) best[k - 1] = d;
    }
    for (int k = 0; k < Z; ++k) min_avg_dist[j * Z + k] = best[k];
  return SPX_OK;
}
}  // namespace

extern "C" int spx_flatten_nrt_nodes(const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_nrt_slots* slots,
                                     uint8_t* flags, int32_t* max_numa, uint8_t* n_zones, uint8_t* zone_id,
                                     uint8_t* zone_present, int64_t* zone_avail, int32_t* zone_cost, float* min_avg_dist,
                                     uint8_t* node_present) {
  if (!nodes || !nrt || !slots || !flags || !max_numa || !n_zones || !zone_id || !zone_present || !zone_avail || !zone_cost ||
      !min_avg_dist || !node_present)
    return SPX_ERR_ARG;
  const int64_t n = nodes->n_nodes;
  if (nrt->n_nodes != n) return SPX_ERR_ARG;
  const int R = slots->n_res;
  std::memset(zone_id, 0, static_cast<size_t>(n) * Z);
  std::memset(zone_present, 0, static_cast<size_t>(n) * Z);
  std::memset(zone_avail, 0, static_cast<size_t>(n) * Z * R * sizeof(int64_t));
  std::atomic<int> err{SPX_OK};
  // nodes are independent: split across host threads (20k nodes x 255 zone subsets for the distance minima alone)
  spx_host::parallel_rows(n, [&](int64_t row0, int64_t row1) {
    for (int64_t i = row0; i < row1; ++i) {
      const int rc = flatten_nrt_node(nodes, nrt, slots, i, i, flags, max_numa, n_zones, zone_id, zone_present, zone_avail, zone_cost, min_avg_dist, node_present);
      if (rc != SPX_OK) {
        err = rc;
        return;
      }
    }
  }, 256);
  return err.load();
}

// the same columns for the listed nodes only (a snapshot delta: spx_update_nrt_nodes takes these rows): row j describes node idx[j]
extern "C" int spx_flatten_nrt_node_rows(const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_nrt_slots* slots,
                                         const int64_t* idx, int64_t n_rows, uint8_t* flags, int32_t* max_numa, uint8_t* n_zones,
                                         uint8_t* zone_id, uint8_t* zone_present, int64_t* zone_avail, int32_t* zone_cost, float* min_avg_dist,
                                         uint8_t* node_present) {
  if (!nodes || !nrt || !slots || !idx || !flags || !max_numa || !n_zones || !zone_id || !zone_present || !zone_avail || !zone_cost ||
      !min_avg_dist || !node_present || n_rows < 0)
    return SPX_ERR_ARG;
  if (nrt->n_nodes != nodes->n_nodes) return SPX_ERR_ARG;
  const int R = slots->n_res;
  std::memset(zone_id, 0, static_cast<size_t>(n_rows) * Z);
  std::memset(zone_present, 0, static_cast<size_t>(n_rows) * Z);
  std::memset(zone_avail, 0, static_cast<size_t>(n_rows) * Z * R * sizeof(int64_t));
  for (int64_t j = 0; j < n_rows; ++j) {
    if (idx[j] < 0 || idx[j] >= nodes->n_nodes) return SPX_ERR_ARG;
    const int rc = flatten_nrt_node(nodes, nrt, slots, idx[j], j, flags, max_numa, n_zones, zone_id, zone_present, zone_avail, zone_cost, min_avg_dist, node_present);
    if (rc != SPX_OK) return rc;
  }
  return SPX_OK;
}

extern "C" int spx_flatten_nrt_pods(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_nrt_slots* slots,
                                    uint8_t* qos, uint8_t* non_native, uint8_t* n_ctr, uint8_t* ctr_kind, uint8_t* ctr_present,
                                    int64_t* ctr_req, uint8_t* pod_present, int64_t* pod_req) {
  if (!pods || !slots || !qos || !non_native || !n_ctr || !ctr_kind || !ctr_present || !ctr_req || !pod_present || !pod_req)
    return SPX_ERR_ARG;
  const int R = slots->n_res;
  const int64_t P = pods->n_pods;
  std::atomic<int> err{SPX_OK};
  spx_host::parallel_rows(P, [&](int64_t row0, int64_t row1) {
  std::vector<KV> init_res, res;
  for (int64_t i = row0; i < row1; ++i) {
    // the row's cells are cleared by the thread that fills them, row by row (one serial memset of the 16 MB request column cost
    // more than the whole parallel fill)
    std::memset(ctr_kind + i * CM, 0, CM);
    std::memset(ctr_present + i * CM, 0, CM);
    std::memset(ctr_req + i * CM * R, 0, static_cast<size_t>(CM) * R * sizeof(int64_t));
    std::memset(pod_req + i * R, 0, static_cast<size_t>(R) * sizeof(int64_t));
    const int32_t c0 = pods->ctr_ptr[i], c1 = pods->ctr_ptr[i + 1];
    if (c1 - c0 > CM) {
      err = SPX_ERR_ARG;
      return;
    }
    qos[i] = static_cast<uint8_t>(pod_qos(pods, i));
    bool nn = false;
    n_ctr[i] = static_cast<uint8_t>(c1 - c0);
    for (int32_t c = c0; c < c1; ++c) {
      const int64_t slot_base = (i * CM + (c - c0));
      ctr_kind[slot_base] = pods->ctr_kind[c];
      uint8_t present = 0;
      for (int32_t k = pods->req_ptr[c]; k < pods->req_ptr[c + 1]; ++k) {
        const int s = slot_of(slots, pods->req_res[k]);
        if (s < 0) {
          err = SPX_ERR_ARG;
          return;
        }
        present |= static_cast<uint8_t>(1u << s);
        ctr_req[slot_base * R + s] = pods->req_qty[k];
        if (!rc_has(rc, pods->req_res[k], SPX_RC_NATIVE)) nn = true;
      }
      ctr_present[slot_base] = present;
    }
    non_native[i] = nn ? 1 : 0;
    // GetPodEffectiveRequest (pkg/util/resource.go:51-85) with map key presence
    init_res.clear();
    res.clear();
    for (int32_t c = c0; c < c1; ++c) {
      if (pods->ctr_kind[c] == SPX_CTR_APP) continue;
      for (int32_t k = pods->req_ptr[c]; k < pods->req_ptr[c + 1]; ++k) {
        KV* e = kv_find(init_res, pods->req_res[k]);
        if (e && pods->req_qty[k] <= e->qty) continue;
        if (e) e->qty = pods->req_qty[k];
        else init_res.push_back({pods->req_res[k], pods->req_qty[k]});
      }
    }
    for (int32_t c = c0; c < c1; ++c) {
      if (pods->ctr_kind[c] != SPX_CTR_APP) continue;
      for (int32_t k = pods->req_ptr[c]; k < pods->req_ptr[c + 1]; ++k) {
        KV* e = kv_find(res, pods->req_res[k]);
        if (e) e->qty += pods->req_qty[k];
        else res.push_back({pods->req_res[k], pods->req_qty[k]});
      }
    }
    for (const KV& in : init_res) {
      KV* e = kv_find(res, in.res);
      if (e && in.qty <= e->qty) continue;
      if (e) e->qty = in.qty;
      else res.push_back(in);
    }
    if (pods->ovh_ptr)
      for (int32_t k = pods->ovh_ptr[i]; k < pods->ovh_ptr[i + 1]; ++k) {
        KV* e = kv_find(res, pods->ovh_res[k]);
        if (e) e->qty += pods->ovh_qty[k];
        else res.push_back({pods->ovh_res[k], pods->ovh_qty[k]});
      }
    uint8_t pp = 0;
    for (const KV& e : res) {
      const int s = slot_of(slots, e.res);
      if (s < 0) {
        err = SPX_ERR_ARG;
        return;
      }
      pp |= static_cast<uint8_t>(1u << s);
      pod_req[i * R + s] = e.qty;
    }
    pod_present[i] = pp;
  }
  }, 4096);
  return err.load();
}
