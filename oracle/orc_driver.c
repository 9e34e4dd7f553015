/*
 * orc_driver.c — batch driver around the per-(pod,node) oracle functions (TEST INFRASTRUCTURE).
 *
 * Call structure restated from upstream RunScorePlugins as the reference's own tests and
 * benchmarks drive it (pkg/noderesources/allocatable_test.go:289-297,
 * pkg/trimaran/targetloadpacking/targetloadpacking_test.go:369-381): for each pod, Score() on every
 * (feasible) node, then the plugin's NormalizeScore() over that pod's node list.
 *
 * Threads split POD ROWS between workers.  Upstream instead fans one pod's node loop out over 16
 * goroutines (Parallelizer, copied at targetloadpacking_test.go:386-405) and joins per pod; the row
 * split used here has no per-pod join and is therefore the more favourable layout for the CPU.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#include "spx_oracle.h"

typedef struct job {
  const orc_snapshot* s;
  int plugin;
  int64_t row_begin, row_end, base;
  const uint8_t* mask;
  int64_t* out_raw;
  int64_t* out_norm;
  int rc;
} job;

static int64_t score_one(const orc_snapshot* s, int plugin, int64_t pod, int64_t node) {
  switch (plugin) {
    case SPX_PLUGIN_ALLOCATABLE: return orc_allocatable_score(s->nodes, s->rc, s->alloc_params, node);
    case SPX_PLUGIN_TLP: return orc_tlp_score(s->nodes, s->metrics, s->assigned, s->pods, s->tlp_params, pod, node);
    case SPX_PLUGIN_LVRB: return orc_lvrb_score(s->nodes, s->metrics, s->pods, s->lvrb_params, pod, node);
    case SPX_PLUGIN_LROC: /* NormalizeScore is a no-op: lowriskovercommitment.go:153-155 */
      return orc_lroc_score(s->nodes, s->node_pods, s->metrics, s->pods, s->lroc_params, pod, node);
    case SPX_PLUGIN_PEAKS: return orc_peaks_score(s->nodes, s->metrics, s->power_models, s->pods, pod, node);
    case SPX_PLUGIN_NRT: return orc_nrt_score(s->nrt, s->rc, s->pods, s->nrt_params, pod, node); /* no NormalizeScore: score.go:104-106 */
    default: return 0;
  }
}

static void* run(void* arg) {
  job* j = (job*)arg;
  const int64_t n = j->s->nodes->n_nodes;
  int64_t* list = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  if (!list || !idx) {
    j->rc = -1;
    free(list);
    free(idx);
    return 0;
  }
  int64_t* nsat = 0;
  int64_t* nvio = 0;
  int64_t* ncost = 0;
  if (j->plugin == SPX_PLUGIN_NETOVERHEAD) {
    nsat = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)(n > 0 ? n : 1));
    nvio = nsat + n;
    ncost = nvio + n;
  }
  for (int64_t pod = j->row_begin; pod < j->row_end; ++pod) {
    const uint8_t* m = j->mask ? j->mask + (size_t)pod * (size_t)n : 0;
    int64_t k = 0;
    int equally = 0;
    if (j->plugin == SPX_PLUGIN_NETOVERHEAD) /* PreFilter once per pod (networkoverhead.go:174) */
      equally = orc_net_prefilter(j->s->nodes, j->s->pods, j->s->appgroups, j->s->nettopo, pod, nsat, nvio, ncost);
    for (int64_t node = 0; node < n; ++node) {
      if (m && !m[node]) continue; /* upstream only scores nodes that passed Filter */
      if (j->plugin == SPX_PLUGIN_NETOVERHEAD) {
        if (!equally && nvio[node] > nsat[node]) continue; /* the plugin's own Filter (networkoverhead.go:349-357) */
        list[k] = equally ? 0 : ncost[node];               /* Score :362-386 */
        idx[k] = node;
        ++k;
        continue;
      }
      list[k] = score_one(j->s, j->plugin, pod, node);
      idx[k] = node;
      ++k;
    }
    size_t off = (size_t)(pod - j->base) * (size_t)n;
    if (j->out_raw) {
      memset(j->out_raw + off, 0, sizeof(int64_t) * (size_t)n);
      for (int64_t i = 0; i < k; ++i) j->out_raw[off + (size_t)idx[i]] = list[i];
    }
    /* NormalizeScore: Allocatable rescales (allocatable.go:143); TLP and LVRB are no-ops
     * (targetloadpacking.go:193-195, loadvariationriskbalancing.go:134-136) */
    if (j->plugin == SPX_PLUGIN_ALLOCATABLE) orc_allocatable_normalize(list, k);
    if (j->plugin == SPX_PLUGIN_NETOVERHEAD) orc_net_normalize(list, k);
    if (j->plugin == SPX_PLUGIN_PEAKS) orc_peaks_normalize(list, k); /* peaks.go:150-166 */
    if (j->out_norm) {
      memset(j->out_norm + off, 0, sizeof(int64_t) * (size_t)n);
      for (int64_t i = 0; i < k; ++i) j->out_norm[off + (size_t)idx[i]] = list[i];
    }
  }
  free(list);
  free(idx);
  free(nsat);
  return 0;
}

int orc_score_rows(const orc_snapshot* s, int plugin, int64_t row_begin, int64_t row_end,
                   const uint8_t* mask, int threads, int64_t* out_raw, int64_t* out_norm) {
  if (!s || row_end < row_begin) return -1;
  if (threads < 1) threads = 1;
  int64_t rows = row_end - row_begin;
  if (threads > rows) threads = (int)(rows > 0 ? rows : 1);
  job* jobs = (job*)calloc((size_t)threads, sizeof(job));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!jobs || !th) return -1;
  int rc = 0;
  for (int t = 0; t < threads; ++t) {
    jobs[t].s = s;
    jobs[t].plugin = plugin;
    jobs[t].base = row_begin;
    jobs[t].row_begin = row_begin + rows * t / threads;
    jobs[t].row_end = row_begin + rows * (t + 1) / threads;
    jobs[t].mask = mask;
    jobs[t].out_raw = out_raw;
    jobs[t].out_norm = out_norm;
    if (threads == 1)
      run(&jobs[t]);
    else
      pthread_create(&th[t], 0, run, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) {
    if (threads > 1) pthread_join(th[t], 0);
    if (jobs[t].rc) rc = jobs[t].rc;
  }
  free(jobs);
  free(th);
  return rc;
}

typedef struct fjob {
  const orc_snapshot* s;
  int plugin;
  int64_t row_begin, row_end, base;
  uint8_t* out;
} fjob;

static void* frun(void* arg) {
  fjob* j = (fjob*)arg;
  const int64_t n = j->s->nodes->n_nodes;
  if (j->plugin == SPX_PLUGIN_NETOVERHEAD) {
    int64_t* buf = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)(n > 0 ? n : 1));
    for (int64_t pod = j->row_begin; pod < j->row_end; ++pod) {
      int eq = orc_net_prefilter(j->s->nodes, j->s->pods, j->s->appgroups, j->s->nettopo, pod, buf, buf + n, buf + 2 * n);
      for (int64_t node = 0; node < n; ++node) {
        uint8_t st = 0;
        if (eq < 0) st = 255;
        else if (!eq && buf[n + node] > buf[node]) st = SPX_NET_ST_UNSCHEDULABLE;
        j->out[(size_t)(pod - j->base) * (size_t)n + (size_t)node] = st;
      }
    }
    free(buf);
    return 0;
  }
  for (int64_t pod = j->row_begin; pod < j->row_end; ++pod)
    for (int64_t node = 0; node < n; ++node) {
      int st = 0;
      if (j->plugin == SPX_PLUGIN_NRT) st = orc_nrt_filter(j->s->nodes, j->s->nrt, j->s->rc, j->s->pods, pod, node);
      j->out[(size_t)(pod - j->base) * (size_t)n + (size_t)node] = (uint8_t)(st < 0 ? 255 : st);
    }
  return 0;
}

int orc_filter_rows(const orc_snapshot* s, int plugin, int64_t row_begin, int64_t row_end, int threads, uint8_t* out_status) {
  if (!s || !out_status || row_end < row_begin) return -1;
  int64_t rows = row_end - row_begin;
  if (threads < 1) threads = 1;
  if (threads > rows) threads = (int)(rows > 0 ? rows : 1);
  fjob* jobs = (fjob*)calloc((size_t)threads, sizeof(fjob));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!jobs || !th) return -1;
  for (int t = 0; t < threads; ++t) {
    jobs[t].s = s;
    jobs[t].plugin = plugin;
    jobs[t].base = row_begin;
    jobs[t].row_begin = row_begin + rows * t / threads;
    jobs[t].row_end = row_begin + rows * (t + 1) / threads;
    jobs[t].out = out_status;
    if (threads == 1) frun(&jobs[t]);
    else pthread_create(&th[t], 0, frun, &jobs[t]);
  }
  if (threads > 1)
    for (int t = 0; t < threads; ++t) pthread_join(th[t], 0);
  free(jobs);
  free(th);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * The reference's own parallel structure: ONE pod at a time; that pod's node loop is cut into chunks which `workers`
 * goroutines claim from a shared counter (workqueue.ParallelizeUntil with chunkSizeFor, copied into the benchmark at
 * pkg/trimaran/targetloadpacking/targetloadpacking_test.go:386-405: parallelism = 16, chunk = min(sqrt(n), n/16 + 1));
 * the caller joins, then runs NormalizeScore serially (:369-381).  Restated with a persistent pthread pool and two
 * barriers per pod standing in for the WaitGroup.  Used by bench.py's cpu_baseline ("reference structure") leg.
 */
typedef struct cyc {
  const orc_snapshot* s;
  int plugin;
  int64_t n, chunk, pod;
  int64_t* list;
  atomic_llong next;
  int stop;
  pthread_barrier_t start, done;
} cyc;

static void cyc_work(cyc* c) {
  for (;;) {
    const int64_t k = (int64_t)atomic_fetch_add(&c->next, 1);
    const int64_t b = k * c->chunk;
    if (b >= c->n) return;
    const int64_t e = b + c->chunk < c->n ? b + c->chunk : c->n;
    for (int64_t node = b; node < e; ++node) c->list[node] = score_one(c->s, c->plugin, c->pod, node);
  }
}

static void* cyc_thread(void* arg) {
  cyc* c = (cyc*)arg;
  for (;;) {
    pthread_barrier_wait(&c->start);
    if (c->stop) return 0;
    cyc_work(c);
    pthread_barrier_wait(&c->done);
  }
}

int orc_cycle_rows(const orc_snapshot* s, int plugin, int64_t row_begin, int64_t row_end, int workers, int64_t* out_norm) {
  if (!s || row_end < row_begin || plugin == SPX_PLUGIN_NETOVERHEAD) return -1;
  if (workers < 1) workers = 1;
  cyc c;
  memset(&c, 0, sizeof c);
  c.s = s;
  c.plugin = plugin;
  c.n = s->nodes->n_nodes;
  int64_t sq = (int64_t)sqrt((double)c.n), r = c.n / workers + 1; /* chunkSizeFor, parallelism = workers */
  c.chunk = sq > r ? r : (sq < 1 ? 1 : sq);
  c.list = (int64_t*)malloc(sizeof(int64_t) * (size_t)(c.n > 0 ? c.n : 1));
  if (!c.list) return -1;
  pthread_t* th = 0;
  if (workers > 1) { /* the calling thread is one of the workers, as in ParallelizeUntil's goroutines + wg.Wait */
    th = (pthread_t*)calloc((size_t)(workers - 1), sizeof(pthread_t));
    pthread_barrier_init(&c.start, 0, (unsigned)workers);
    pthread_barrier_init(&c.done, 0, (unsigned)workers);
    for (int t = 0; t < workers - 1; ++t) pthread_create(&th[t], 0, cyc_thread, &c);
  }
  for (int64_t pod = row_begin; pod < row_end; ++pod) {
    c.pod = pod;
    atomic_store(&c.next, 0);
    if (workers > 1) pthread_barrier_wait(&c.start);
    cyc_work(&c);
    if (workers > 1) pthread_barrier_wait(&c.done);
    if (plugin == SPX_PLUGIN_ALLOCATABLE) orc_allocatable_normalize(c.list, c.n);
    if (plugin == SPX_PLUGIN_PEAKS) orc_peaks_normalize(c.list, c.n);
    if (out_norm) memcpy(out_norm + (size_t)(pod - row_begin) * (size_t)c.n, c.list, sizeof(int64_t) * (size_t)c.n);
  }
  if (workers > 1) {
    c.stop = 1;
    pthread_barrier_wait(&c.start);
    for (int t = 0; t < workers - 1; ++t) pthread_join(th[t], 0);
    pthread_barrier_destroy(&c.start);
    pthread_barrier_destroy(&c.done);
    free(th);
  }
  free(c.list);
  return 0;
}
