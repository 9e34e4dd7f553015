/*
 * orc_commit.c — the reference's one-pod-at-a-time scheduling cycle with its Reserve side effects, on the object tables
 * (TEST INFRASTRUCTURE, see spx_oracle.h).  The checker of spx_commit_sequential at full size.
 *
 * For each pending pod, in queue order, what upstream's scheduleOne does with the reference's plugins enabled:
 *   PreFilter   CapacityScheduling.PreFilter (capacity_scheduling.go:208-283)            -> orc_capacity_prefilter
 *               NetworkOverhead.PreFilter (networkoverhead.go:174-298)                    -> orc_net_prefilter_range
 *   Filter      TopologyMatch.Filter (noderesourcetopology/filter.go:179-245)             -> orc_nrt_filter
 *               NetworkOverhead.Filter (networkoverhead.go:326-359)
 *   Score       over the nodes that passed every Filter: Allocatable, TargetLoadPacking, LoadVariationRiskBalancing,
 *               TopologyMatch, NetworkOverhead — each followed by its NormalizeScore over that node list
 *   selectHost  highest sum of plugin weight x score; upstream draws one node of the tie set at random, so the tie set is
 *               reported as (lowest node index, size) and the cycle continues with the lowest index — the same convention
 *               spx_commit_sequential documents in include/spx.h
 *   Reserve / bind-time hooks, applied to THIS file's mutable copies of the caches the plugins read:
 *     TopologyMatch.Reserve (reserve.go:28-46) -> OverReserve.ReserveNodeResources (cache/overreserve.go:170-186) ->
 *       resourceStore.AddPod (cache/store.go:279-290): the node's store gains util.GetPodEffectiveRequest(pod); the store is
 *       applied at READ time by GetCachedNRTCopy -> UpdateNRT (overreserve.go:117-142, store.go:315-356) — subtracted from
 *       every zone, a zone with less drops to zero.  Here the store is the assumed_* CSR of spx_nrt_objects, which
 *       orc_nrt_filter / orc_nrt_score already subtract per call exactly like UpdateNRT: the entries are appended, never
 *       folded into the zone table (the device loop folds them; the two must agree);
 *     trimaran PodAssignEventHandler.updateCache (handler.go:131-139): (now, pod) is appended to ScheduledPodsCache[node];
 *       TargetLoadPacking.Score walks that slice (targetloadpacking.go:151-168)           -> orc_tlp_score_appended
 *     CapacityScheduling.Reserve (capacity_scheduling.go:350-364, elasticquota.go:89-98, :153-166) -> orc_capacity_reserve; a
 *       bound pod is no longer a nominated pod (the nominator drops it when it is assumed)
 *     NetworkOverhead: a bound pod has Spec.NodeName, so util.GetScheduledList (networkaware/util/util.go:215-232) lists
 *       it from the next PreFilter on (networkoverhead.go:209-224)
 *   LoadVariationRiskBalancing and Allocatable carry no state between cycles.
 *
 * Threads: one pod at a time; its node loop is cut into contiguous ranges, one per worker, joined per pod — upstream's
 * Parallelizer shape (targetloadpacking_test.go:386-405).  Everything that orders pods or nodes is serial.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spx_oracle.h"

enum { ARL_SLOTS = 16 }; /* room of one assumed resource list; unused slots carry resource id -1 (matches no zone resource) */

typedef struct commit_state {
  const orc_commit_args* a;
  int64_t n_nodes;
  uint32_t mask;
  /* --- mutable caches ---------------------------------------------------------------- */
  /* trimaran: entries appended per node since the snapshot (indexes into a->s->pods) */
  int64_t** app_ts;
  int32_t** app_pod;
  int32_t* app_n;
  int32_t* app_cap;
  /* NRT: the snapshot's table with an assumed_* CSR this file owns (per-node capacity, fixed-size lists, inert padding) */
  spx_nrt_objects nrt;
  int32_t* as_ptr;  /* [n_nodes + 1] */
  int32_t* as_used; /* live entries of node n: as_ptr[n] .. as_ptr[n] + as_used[n] */
  int32_t* arl_ptr; /* [total + 1], arl_ptr[k] = k * ARL_SLOTS */
  int32_t* arl_res;
  int64_t* arl_qty;
  int64_t as_total;
  /* AppGroups: placed_* rebuilt in place */
  spx_appgroup_objects ag;
  int32_t* placed_ptr;
  int32_t* placed_selector;
  int32_t* placed_node;
  /* ElasticQuota: Used + the nominated list */
  spx_quota_objects quota;
  int64_t* used;
  uint8_t* used_present;
  int32_t* nom_ns;
  /* --- per-pod scratch, [n_nodes] ------------------------------------------------------ */
  uint8_t* feasible;
  int64_t* sat;
  int64_t* vio;
  int64_t* cost;
  int64_t* sc[SPX_NUM_PLUGINS];
  int64_t* list;
  int64_t* idx;
  int64_t* total;
  /* --- worker pool ----------------------------------------------------------------------- */
  int threads;
  atomic_int phase;   /* generation counter: workers run one node range per increment */
  atomic_int pending; /* workers still inside the current generation */
  atomic_int quit;
  atomic_int failed;
  int64_t cur_pod;
  int cur_equally; /* NetworkOverhead: scoreEqually for the current pod */
  orc_snapshot snap; /* the caller's snapshot with nrt / appgroups pointing at this file's copies */
} commit_state;

typedef struct worker {
  commit_state* st;
  int id;
} worker;

/* ------------------------------------------------------------------ one node of one pod's cycle */
static void eval_node(commit_state* st, int64_t pod, int64_t node) {
  const orc_snapshot* s = &st->snap;
  const uint32_t m = st->mask;
  uint8_t ok = 1;
  if (m & (1u << SPX_PLUGIN_NRT)) {
    const int f = orc_nrt_filter(s->nodes, s->nrt, s->rc, s->pods, pod, node);
    if (f != 0) ok = 0;
  }
  if (ok && (m & (1u << SPX_PLUGIN_NETOVERHEAD)) && !st->cur_equally && st->vio[node] > st->sat[node]) ok = 0; /* networkoverhead.go:349-357 */
  st->feasible[node] = ok;
  if (!ok) return; /* Score runs over the nodes that passed every Filter */
  if (m & (1u << SPX_PLUGIN_ALLOCATABLE)) st->sc[SPX_PLUGIN_ALLOCATABLE][node] = orc_allocatable_score(s->nodes, s->rc, s->alloc_params, node);
  if (m & (1u << SPX_PLUGIN_TLP))
    st->sc[SPX_PLUGIN_TLP][node] = orc_tlp_score_appended(s->nodes, s->metrics, s->assigned, s->pods, s->tlp_params, pod, node, st->app_ts[node],
                                                          st->app_pod[node], st->app_n[node], s->pods);
  if (m & (1u << SPX_PLUGIN_LVRB)) st->sc[SPX_PLUGIN_LVRB][node] = orc_lvrb_score(s->nodes, s->metrics, s->pods, s->lvrb_params, pod, node);
  if (m & (1u << SPX_PLUGIN_NRT)) st->sc[SPX_PLUGIN_NRT][node] = orc_nrt_score(s->nrt, s->rc, s->pods, s->nrt_params, pod, node);
  if (m & (1u << SPX_PLUGIN_NETOVERHEAD)) st->sc[SPX_PLUGIN_NETOVERHEAD][node] = st->cur_equally ? 0 : st->cost[node]; /* Score :362-386 */
}

static void run_range(commit_state* st, int id) {
  const int64_t n = st->n_nodes;
  const int64_t b = n * id / st->threads, e = n * (id + 1) / st->threads;
  const orc_snapshot* s = &st->snap;
  const int64_t pod = st->cur_pod;
  if ((st->mask & (1u << SPX_PLUGIN_NETOVERHEAD)) && !st->cur_equally) {
    const int r = orc_net_prefilter_range(s->nodes, s->pods, s->appgroups, s->nettopo, pod, b, e, st->sat, st->vio, st->cost);
    if (r < 0) atomic_store(&st->failed, 1);
  }
  for (int64_t node = b; node < e; ++node) eval_node(st, pod, node);
}

static void* worker_main(void* arg) {
  worker* w = (worker*)arg;
  commit_state* st = w->st;
  int seen = 0;
  for (;;) {
    int spins = 0;
    while (atomic_load_explicit(&st->phase, memory_order_acquire) == seen) {
      if (atomic_load_explicit(&st->quit, memory_order_acquire)) return 0;
      if (++spins > 2000) {
        sched_yield();
        spins = 0;
      }
    }
    ++seen;
    run_range(st, w->id);
    atomic_fetch_sub_explicit(&st->pending, 1, memory_order_acq_rel);
  }
}

static void fan_out(commit_state* st) {
  if (st->threads == 1) {
    run_range(st, 0);
    return;
  }
  atomic_store_explicit(&st->pending, st->threads - 1, memory_order_release);
  atomic_fetch_add_explicit(&st->phase, 1, memory_order_acq_rel);
  run_range(st, 0);
  int spins = 0;
  while (atomic_load_explicit(&st->pending, memory_order_acquire) != 0)
    if (++spins > 2000) {
      sched_yield();
      spins = 0;
    }
}

/* ------------------------------------------------------------------ mutable caches */
static void* xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

/* lay the assumed store out with `cap[n]` entry slots for node n, keeping the live entries */
static int assumed_layout(commit_state* st, const int32_t* cap, const spx_nrt_objects* from, const int32_t* from_used) {
  const int64_t n = st->n_nodes;
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) total += cap[i];
  int32_t* ptr = (int32_t*)xcalloc((size_t)n + 1, sizeof(int32_t));
  int32_t* used = (int32_t*)xcalloc((size_t)n, sizeof(int32_t));
  int32_t* aptr = (int32_t*)xcalloc((size_t)total + 1, sizeof(int32_t));
  int32_t* ares = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total * ARL_SLOTS + 1));
  int64_t* aqty = (int64_t*)xcalloc((size_t)(total * ARL_SLOTS + 1), sizeof(int64_t));
  if (!ptr || !used || !aptr || !ares || !aqty) return -1;
  for (int64_t k = 0; k < total * ARL_SLOTS; ++k) ares[k] = -1;
  for (int64_t k = 0; k <= total; ++k) aptr[k] = (int32_t)(k * ARL_SLOTS);
  int64_t at = 0;
  for (int64_t i = 0; i < n; ++i) {
    ptr[i] = (int32_t)at;
    int32_t live = 0;
    if (from && from->assumed_ptr) {
      const int32_t lo = from->assumed_ptr[i];
      const int32_t hi = from_used ? lo + from_used[i] : from->assumed_ptr[i + 1];
      for (int32_t e = lo; e < hi; ++e, ++live) {
        int slot = 0;
        for (int32_t k = from->arl_ptr[e]; k < from->arl_ptr[e + 1]; ++k) {
          if (from->arl_res[k] < 0) continue;
          if (slot >= ARL_SLOTS) return -2;
          ares[(at + live) * ARL_SLOTS + slot] = from->arl_res[k];
          aqty[(at + live) * ARL_SLOTS + slot] = from->arl_qty[k];
          ++slot;
        }
      }
    }
    used[i] = live;
    at += cap[i];
  }
  ptr[n] = (int32_t)at;
  free(st->as_ptr);
  free(st->as_used);
  free(st->arl_ptr);
  free(st->arl_res);
  free(st->arl_qty);
  st->as_ptr = ptr;
  st->as_used = used;
  st->arl_ptr = aptr;
  st->arl_res = ares;
  st->arl_qty = aqty;
  st->as_total = total;
  st->nrt.assumed_ptr = ptr;
  st->nrt.arl_ptr = aptr;
  st->nrt.arl_res = ares;
  st->nrt.arl_qty = aqty;
  return 0;
}

/* OverReserve.ReserveNodeResources overreserve.go:170-186 + resourceStore.AddPod store.go:279-290 */
static int nrt_reserve(commit_state* st, int64_t pod, int64_t node) {
  const orc_snapshot* s = &st->snap;
  if (!st->nrt.has_nrt[node]) return 0; /* !ov.nrts.Contains(nodeName): "ignoring reserve" */
  int32_t res[ARL_SLOTS];
  int64_t qty[ARL_SLOTS];
  const int n = orc_pod_effective_request(s->pods, pod, res, qty, ARL_SLOTS); /* util.GetPodEffectiveRequest(pod) */
  if (n > ARL_SLOTS) return -2;
  if (st->as_used[node] == st->as_ptr[node + 1] - st->as_ptr[node]) { /* the node's slots are full: double them */
    int32_t* cap = (int32_t*)malloc(sizeof(int32_t) * (size_t)st->n_nodes);
    if (!cap) return -1;
    for (int64_t i = 0; i < st->n_nodes; ++i) cap[i] = st->as_ptr[i + 1] - st->as_ptr[i];
    cap[node] = cap[node] * 2 + 2;
    spx_nrt_objects old = st->nrt;
    int32_t* old_used = st->as_used;
    int32_t *p0 = st->as_ptr, *p1 = st->arl_ptr, *p2 = st->arl_res;
    int64_t* p3 = st->arl_qty;
    st->as_ptr = st->as_used = st->arl_ptr = st->arl_res = 0;
    st->arl_qty = 0;
    const int rc = assumed_layout(st, cap, &old, old_used);
    free(cap);
    free(p0);
    free(old_used);
    free(p1);
    free(p2);
    free(p3);
    if (rc) return rc;
  }
  const int64_t e = (int64_t)st->as_ptr[node] + st->as_used[node];
  for (int i = 0; i < n; ++i) {
    st->arl_res[e * ARL_SLOTS + i] = res[i];
    st->arl_qty[e * ARL_SLOTS + i] = qty[i];
  }
  st->as_used[node] += 1;
  return 0;
}

/* PodAssignEventHandler.updateCache handler.go:131-139 */
static int trimaran_bind(commit_state* st, int64_t pod, int64_t node, int64_t now) {
  if (st->app_n[node] == st->app_cap[node]) {
    const int32_t cap = st->app_cap[node] ? st->app_cap[node] * 2 : 4;
    int64_t* ts = (int64_t*)realloc(st->app_ts[node], sizeof(int64_t) * (size_t)cap);
    if (!ts) return -1;
    st->app_ts[node] = ts;
    int32_t* pp = (int32_t*)realloc(st->app_pod[node], sizeof(int32_t) * (size_t)cap);
    if (!pp) return -1;
    st->app_pod[node] = pp;
    st->app_cap[node] = cap;
  }
  st->app_ts[node][st->app_n[node]] = now;
  st->app_pod[node][st->app_n[node]] = (int32_t)pod;
  st->app_n[node] += 1;
  return 0;
}

/* the bound pod joins its AppGroup's scheduled list (util.GetScheduledList util.go:215-232) */
static void net_bind(commit_state* st, int64_t pod, int64_t node) {
  const spx_pod_objects* pods = st->snap.pods;
  const int32_t g = pods->appgroup[pod];
  if (g < 0 || g >= st->ag.n_groups) return;
  const int32_t at = st->placed_ptr[g + 1], end = st->placed_ptr[st->ag.n_groups];
  memmove(st->placed_selector + at + 1, st->placed_selector + at, sizeof(int32_t) * (size_t)(end - at));
  memmove(st->placed_node + at + 1, st->placed_node + at, sizeof(int32_t) * (size_t)(end - at));
  st->placed_selector[at] = pods->selector[pod];
  st->placed_node[at] = (int32_t)node;
  for (int32_t k = g + 1; k <= st->ag.n_groups; ++k) st->placed_ptr[k] += 1;
}

/* CapacityScheduling.Reserve + the nominator forgetting a pod that is now assumed.  The list entry is made inert instead of
 * removed (entry j's pod is row j of nom_pods): namespace -1 is "a nominated pod whose namespace has no ElasticQuota", which
 * PreFilter skips (capacity_scheduling.go:241-244, `info == nil`). */
static void quota_bind(commit_state* st, int64_t pod) {
  orc_capacity_reserve(st->snap.pods, st->snap.rc, &st->quota, pod, st->used, st->used_present);
  for (int64_t j = 0; j < st->quota.n_nominated; ++j)
    if (st->quota.nom_pending_index[j] == pod) st->nom_ns[j] = -1;
}

static void state_free(commit_state* st) {
  if (st->app_ts)
    for (int64_t i = 0; i < st->n_nodes; ++i) {
      free(st->app_ts[i]);
      free(st->app_pod[i]);
    }
  free(st->app_ts);
  free(st->app_pod);
  free(st->app_n);
  free(st->app_cap);
  free(st->as_ptr);
  free(st->as_used);
  free(st->arl_ptr);
  free(st->arl_res);
  free(st->arl_qty);
  free(st->placed_ptr);
  free(st->placed_selector);
  free(st->placed_node);
  free(st->used);
  free(st->used_present);
  free(st->nom_ns);
  free(st->feasible);
  free(st->sat);
  for (int p = 0; p < SPX_NUM_PLUGINS; ++p) free(st->sc[p]);
  free(st->list);
  free(st->idx);
  free(st->total);
}

int orc_commit_sequential(const orc_commit_args* a, int32_t* node_out, int64_t* score_out, int32_t* ties_out, uint8_t* verdict_out) {
  if (!a || !a->s || !a->s->nodes || !a->s->pods || !node_out || a->row_end < a->row_begin) return -1;
  const orc_snapshot* s0 = a->s;
  const int64_t n = s0->nodes->n_nodes;
  const uint32_t m = a->plugin_mask;
  const uint32_t known = (1u << SPX_PLUGIN_ALLOCATABLE) | (1u << SPX_PLUGIN_TLP) | (1u << SPX_PLUGIN_LVRB) | (1u << SPX_PLUGIN_NRT) |
                         (1u << SPX_PLUGIN_NETOVERHEAD) | (1u << SPX_PLUGIN_CAPACITY);
  if (m & ~known) return -1;
  if ((m & (1u << SPX_PLUGIN_NRT)) && (!s0->nrt || !s0->nrt_params)) return -1;
  if ((m & (1u << SPX_PLUGIN_NETOVERHEAD)) && (!s0->appgroups || !s0->nettopo)) return -1;
  if ((m & (1u << SPX_PLUGIN_CAPACITY)) && !a->quota) return -1;
  if ((m & ((1u << SPX_PLUGIN_TLP) | (1u << SPX_PLUGIN_LVRB))) && !s0->metrics) return -1;

  commit_state st;
  memset(&st, 0, sizeof st);
  st.a = a;
  st.n_nodes = n;
  st.mask = m;
  st.snap = *s0;
  int rc = -1;
  pthread_t* th = 0;
  worker* wk = 0;
  int started = 0;

  st.app_ts = (int64_t**)xcalloc((size_t)n, sizeof(int64_t*));
  st.app_pod = (int32_t**)xcalloc((size_t)n, sizeof(int32_t*));
  st.app_n = (int32_t*)xcalloc((size_t)n, sizeof(int32_t));
  st.app_cap = (int32_t*)xcalloc((size_t)n, sizeof(int32_t));
  st.feasible = (uint8_t*)xcalloc((size_t)n, 1);
  st.sat = (int64_t*)xcalloc((size_t)n * 3, sizeof(int64_t));
  st.vio = st.sat + n;
  st.cost = st.vio + n;
  for (int p = 0; p < SPX_NUM_PLUGINS; ++p) st.sc[p] = (int64_t*)xcalloc((size_t)n, sizeof(int64_t));
  st.list = (int64_t*)xcalloc((size_t)n, sizeof(int64_t));
  st.idx = (int64_t*)xcalloc((size_t)n, sizeof(int64_t));
  st.total = (int64_t*)xcalloc((size_t)n, sizeof(int64_t));
  if (!st.app_ts || !st.app_pod || !st.app_n || !st.app_cap || !st.feasible || !st.sat || !st.list || !st.idx || !st.total) goto done;

  if (m & (1u << SPX_PLUGIN_NRT)) {
    st.nrt = *s0->nrt;
    int32_t* cap = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    if (!cap) goto done;
    for (int64_t i = 0; i < n; ++i) cap[i] = (s0->nrt->assumed_ptr ? s0->nrt->assumed_ptr[i + 1] - s0->nrt->assumed_ptr[i] : 0) + 4;
    const int r = assumed_layout(&st, cap, s0->nrt, 0);
    free(cap);
    if (r) {
      rc = r;
      goto done;
    }
    st.snap.nrt = &st.nrt;
  }
  if (m & (1u << SPX_PLUGIN_NETOVERHEAD)) {
    st.ag = *s0->appgroups;
    const int32_t g = st.ag.n_groups;
    const int32_t placed = g > 0 ? s0->appgroups->placed_ptr[g] : 0;
    const size_t room = (size_t)placed + (size_t)(a->row_end - a->row_begin) + 1;
    st.placed_ptr = (int32_t*)xcalloc((size_t)g + 1, sizeof(int32_t));
    st.placed_selector = (int32_t*)xcalloc(room, sizeof(int32_t));
    st.placed_node = (int32_t*)xcalloc(room, sizeof(int32_t));
    if (!st.placed_ptr || !st.placed_selector || !st.placed_node) goto done;
    memcpy(st.placed_ptr, s0->appgroups->placed_ptr, sizeof(int32_t) * ((size_t)g + 1));
    memcpy(st.placed_selector, s0->appgroups->placed_selector, sizeof(int32_t) * (size_t)placed);
    memcpy(st.placed_node, s0->appgroups->placed_node, sizeof(int32_t) * (size_t)placed);
    st.ag.placed_ptr = st.placed_ptr;
    st.ag.placed_selector = st.placed_selector;
    st.ag.placed_node = st.placed_node;
    st.snap.appgroups = &st.ag;
  }
  if (m & (1u << SPX_PLUGIN_CAPACITY)) {
    st.quota = *a->quota;
    const size_t q = (size_t)st.quota.n_namespaces, nn = (size_t)st.quota.n_nominated;
    st.used = (int64_t*)xcalloc(q * SPX_QUOTA_SLOTS, sizeof(int64_t));
    st.used_present = (uint8_t*)xcalloc(q, 1);
    st.nom_ns = (int32_t*)xcalloc(nn, sizeof(int32_t));
    if (!st.used || !st.used_present || !st.nom_ns) goto done;
    memcpy(st.used, a->quota->used, sizeof(int64_t) * q * SPX_QUOTA_SLOTS);
    memcpy(st.used_present, a->quota->used_present, q);
    if (nn) memcpy(st.nom_ns, a->quota->nom_ns, sizeof(int32_t) * nn);
    st.quota.used = st.used;
    st.quota.used_present = st.used_present;
    st.quota.nom_ns = st.nom_ns;
  }

  st.threads = a->threads < 1 ? 1 : a->threads;
  if (st.threads > n) st.threads = (int)(n > 0 ? n : 1);
  atomic_init(&st.phase, 0);
  atomic_init(&st.pending, 0);
  atomic_init(&st.quit, 0);
  atomic_init(&st.failed, 0);
  if (st.threads > 1) {
    th = (pthread_t*)calloc((size_t)st.threads, sizeof(pthread_t));
    wk = (worker*)calloc((size_t)st.threads, sizeof(worker));
    if (!th || !wk) goto done;
    for (int t = 1; t < st.threads; ++t) {
      wk[t].st = &st;
      wk[t].id = t;
      if (pthread_create(&th[t], 0, worker_main, &wk[t])) {
        st.threads = t; /* run with what started */
        break;
      }
      started = t;
    }
  }

  for (int64_t pod = a->row_begin; pod < a->row_end; ++pod) {
    const int64_t o = pod - a->row_begin;
    node_out[o] = -1;
    if (score_out) score_out[o] = 0;
    if (ties_out) ties_out[o] = 0;
    if (verdict_out) verdict_out[o] = 0;
    /* PreFilter */
    if (m & (1u << SPX_PLUGIN_CAPACITY)) {
      const int pre = orc_capacity_prefilter(st.snap.pods, st.snap.rc, &st.quota, pod);
      if (pre != 0) {
        if (verdict_out) verdict_out[o] = (uint8_t)pre;
        continue; /* Unschedulable in PreFilter: nothing is reserved */
      }
    }
    st.cur_pod = pod;
    st.cur_equally = 0;
    if (m & (1u << SPX_PLUGIN_NETOVERHEAD)) {
      /* the early exits of PreFilter (:187-228) do not depend on the node: take them from an empty range */
      const int r = orc_net_prefilter_range(st.snap.nodes, st.snap.pods, st.snap.appgroups, st.snap.nettopo, pod, 0, 0, st.sat, st.vio, st.cost);
      if (r < 0) goto done;
      st.cur_equally = r;
    }
    /* Filter + Score over the node list */
    fan_out(&st);
    if (atomic_load(&st.failed)) goto done;
    /* NormalizeScore per plugin over the feasible list, then the weighted sum */
    int64_t k = 0;
    for (int64_t node = 0; node < n; ++node)
      if (st.feasible[node]) st.idx[k++] = node;
    if (k == 0) {
      if (verdict_out) verdict_out[o] = ORC_COMMIT_NO_FEASIBLE_NODE;
      continue;
    }
    for (int64_t i = 0; i < k; ++i) st.total[i] = 0;
    for (int p = 0; p < SPX_NUM_PLUGINS; ++p) {
      if (!(m & (1u << p)) || p == SPX_PLUGIN_CAPACITY) continue;
      for (int64_t i = 0; i < k; ++i) st.list[i] = st.sc[p][st.idx[i]];
      if (p == SPX_PLUGIN_ALLOCATABLE) orc_allocatable_normalize(st.list, k); /* allocatable.go:143-168 */
      if (p == SPX_PLUGIN_NETOVERHEAD) orc_net_normalize(st.list, k);         /* networkoverhead.go:389-418 */
      /* TLP, LVRB: no-op NormalizeScore; TopologyMatch: no ScoreExtensions (score.go:104-106) */
      const int64_t w = a->weights ? a->weights[p] : 1;
      for (int64_t i = 0; i < k; ++i) st.total[i] += w * st.list[i];
    }
    int64_t best = st.total[0];
    for (int64_t i = 1; i < k; ++i)
      if (st.total[i] > best) best = st.total[i];
    int64_t first = -1;
    int32_t ties = 0;
    for (int64_t i = 0; i < k; ++i)
      if (st.total[i] == best) {
        if (first < 0) first = st.idx[i];
        ++ties;
      }
    node_out[o] = (int32_t)first;
    if (score_out) score_out[o] = best;
    if (ties_out) ties_out[o] = ties;
    /* Reserve + bind */
    if (m & (1u << SPX_PLUGIN_NRT)) {
      const int r = nrt_reserve(&st, pod, first);
      if (r) {
        rc = r;
        goto done;
      }
    }
    if (m & (1u << SPX_PLUGIN_NETOVERHEAD)) net_bind(&st, pod, first);
    if (m & (1u << SPX_PLUGIN_CAPACITY)) quota_bind(&st, pod);
    if (m & (1u << SPX_PLUGIN_TLP))
      if (trimaran_bind(&st, pod, first, a->bind_ts)) goto done;
  }
  if (a->tlp_appended_out) /* per node: how many cache entries the cycle appended (a cheap state checksum for the caller) */
    for (int64_t i = 0; i < n; ++i) a->tlp_appended_out[i] = st.app_n[i];
  rc = 0;
done:
  atomic_store(&st.quit, 1);
  for (int t = 1; t <= started; ++t) pthread_join(th[t], 0);
  free(th);
  free(wk);
  state_free(&st);
  return rc;
}

/* sched_getaffinity ∩ cgroup v2 cpu.max (or v1 cfs quota): the number of threads that can actually run at once */
int orc_usable_cpus(void) {
  int n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  long long quota = -1, period = -1;
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[64];
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else {
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    if (f) {
      if (fscanf(f, "%lld", &quota) != 1) quota = -1;
      fclose(f);
      f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
      if (f) {
        if (fscanf(f, "%lld", &period) != 1) period = -1;
        fclose(f);
      }
    }
  }
  if (quota > 0 && period > 0) {
    int c = (int)((quota + period - 1) / period);
    if (c >= 1 && c < n) n = c;
  }
  return n < 1 ? 1 : n;
}
