/*
 * spx_oracle.h — CPU restatement of the reference's Filter/Score algorithms (TEST INFRASTRUCTURE).
 *
 * This directory is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.  The product (libspx.so) must not link,
 * import or call anything here.
 *
 * Every function walks the same object tables (include/spx.h, spx_*_objects) the product's
 * flatteners consume, one (pod, node) pair at a time, in the reference's own call structure,
 * and cites the reference file:line it restates.  Arithmetic rules restated from Go:
 * int64 wraps on overflow and `/` truncates toward zero; float64 is IEEE double without
 * FMA contraction (build with -ffp-contract=off); math.Round is half-away-from-zero;
 * int64(float64) truncates toward zero.
 *
 * Parity pinning: the known-answer tables of the reference's own unit tests are transcribed
 * as data under tests/golden/ and checked by tests/test_oracle_golden.py.  The reference is Go
 * and no Go toolchain exists in this image, so it cannot be executed here.
 */
#ifndef SPX_ORACLE_H
#define SPX_ORACLE_H

#include "../include/spx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- noderesources.Allocatable (pkg/noderesources/allocatable.go, resource_allocation.go) */
int64_t orc_allocatable_score(const spx_node_objects* nodes, const spx_resource_classes* rc,
                              const spx_allocatable_params* p, int64_t node);
void orc_allocatable_normalize(int64_t* scores, int64_t n);

/* ---- trimaran.TargetLoadPacking (pkg/trimaran/targetloadpacking/targetloadpacking.go) */
int64_t orc_tlp_predict_utilisation(const spx_pod_objects* pods, int32_t ctr, const spx_tlp_params* p);
int64_t orc_tlp_score(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                      const spx_assigned_objects* assigned, const spx_pod_objects* pods,
                      const spx_tlp_params* p, int64_t pod, int64_t node);

/* the cache walk of :151-168 over one node's entries given as a slice, and Score with `n_more` entries appended to the snapshot's */
int64_t orc_tlp_missing_entries(const int64_t* e_ts_unix, const int32_t* e_pod, int32_t n_entries, const spx_pod_objects* entry_pods,
                                int64_t window_end, const spx_tlp_params* p);
int64_t orc_tlp_score_appended(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                               const spx_assigned_objects* assigned, const spx_pod_objects* pods,
                               const spx_tlp_params* p, int64_t pod, int64_t node,
                               const int64_t* more_ts_unix, const int32_t* more_pod, int32_t n_more, const spx_pod_objects* more_pods);

/* ---- trimaran.LoadVariationRiskBalancing (.../loadvariationriskbalancing/{loadvariationriskbalancing,analysis}.go,
 *      pkg/trimaran/resourcestats.go) */
typedef struct orc_resource_stats {
  double used_avg;
  double used_stdev;
  double req;
  double capacity;
} orc_resource_stats;
double orc_go_pow(double x, double y);
double orc_lvrb_compute_score(orc_resource_stats* rs, double margin, double sensitivity);
void orc_get_mu_sigma(const orc_resource_stats* rs, double* mu, double* sigma);
int orc_get_resource_data(const spx_metrics_objects* metrics, int64_t node, int type, double* avg, double* stdev);
void orc_get_resource_requested(const spx_pod_objects* pods, int64_t pod, int64_t* milli_cpu, int64_t* memory);
int64_t orc_lvrb_score(const spx_node_objects* nodes, const spx_metrics_objects* metrics,
                       const spx_pod_objects* pods, const spx_lvrb_params* p, int64_t pod, int64_t node);

/* helpers of orc_trimaran.c shared with orc_lroc.c / orc_peaks.c */
int orc_node_metrics(const spx_metrics_objects* m, int64_t node, int32_t* lo, int32_t* hi);
int orc_find_qty(const int32_t* res, const int64_t* qty, int32_t lo, int32_t hi, int32_t want, int64_t* out);
int orc_create_resource_stats(const spx_node_objects* nodes, const spx_metrics_objects* metrics, int64_t node,
                              int64_t req_cpu, int64_t req_mem, int type, orc_resource_stats* rs);

/* ---- trimaran.LowRiskOverCommitment (pkg/trimaran/lowriskovercommitment/{lowriskovercommitment,beta}.go,
 *      pkg/trimaran/resourcestats.go:109-232; gonum mathext.RegIncBeta restated from its published algorithm) */
typedef struct orc_node_requests_limits { /* trimaran.NodeRequestsAndLimits resourcestats.go:149-160 */
  int64_t req_cpu;
  int64_t req_mem;
  int64_t lim_cpu;
  int64_t lim_mem;
  int64_t req_minus_pod_cpu;
  int64_t req_minus_pod_mem;
  int64_t lim_minus_pod_cpu;
  int64_t lim_minus_pod_mem;
  int64_t cap_cpu;
  int64_t cap_mem;
} orc_node_requests_limits;
double orc_reg_inc_beta(double a, double b, double x);
double orc_beta_distribution_function(double alpha, double beta, double x);
double orc_beta_max_variance(double m1);
int orc_beta_match_moments(double m1, double m2, double* alpha, double* beta);
double orc_lroc_compute_probability(double mu, double sigma, double threshold, int* has_dist, double* alpha, double* beta);
void orc_get_resource_limits(const spx_pod_objects* pods, int64_t pod, int64_t* milli_cpu, int64_t* memory);
void orc_node_requests_and_limits(const spx_node_objects* nodes, const spx_node_pods_objects* node_pods, int64_t node,
                                  const int64_t* pod_rl, orc_node_requests_limits* out);
double orc_lroc_compute_risk(const spx_node_objects* nodes, const spx_metrics_objects* metrics, int64_t node, int type,
                             const orc_node_requests_limits* nrl, const spx_lroc_params* p);
int64_t orc_lroc_score(const spx_node_objects* nodes, const spx_node_pods_objects* node_pods, const spx_metrics_objects* metrics,
                       const spx_pod_objects* pods, const spx_lroc_params* p, int64_t pod, int64_t node);

/* ---- trimaran.Peaks (pkg/trimaran/peaks/peaks.go; k8s.io/kubernetes v1.35.7 resource.GetResourceRequestQuantity restated) */
int64_t orc_get_resource_request_quantity_cpu_milli(const spx_pod_objects* pods, int64_t pod);
int64_t orc_peaks_score(const spx_node_objects* nodes, const spx_metrics_objects* metrics, const spx_power_model_objects* models,
                        const spx_pod_objects* pods, int64_t pod, int64_t node);
void orc_peaks_normalize(int64_t* scores, int64_t n);

/* ---- NodeResourceTopologyMatch (pkg/noderesourcetopology/{filter,score,least_numa,...}.go) */
int orc_pod_qos(const spx_pod_objects* pods, int64_t pod);
int orc_include_non_native(const spx_pod_objects* pods, const spx_resource_classes* rc, int64_t pod);
/* 0 = pass, SPX_NRT_ST_* = Unschedulable with that message, -1 = Error("inconsistent resource accounting") */
int orc_nrt_filter(const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_resource_classes* rc,
                   const spx_pod_objects* pods, int64_t pod, int64_t node);
int64_t orc_nrt_score(const spx_nrt_objects* nrt, const spx_resource_classes* rc, const spx_pod_objects* pods,
                      const spx_nrt_params* p, int64_t pod, int64_t node);
int64_t orc_nrt_normalize_score(int numa_nodes_count, int is_min_avg_distance, int highest_numa_id);
/* hooks for the reference's helper-level tables (see the end of orc_nrt.c) */
int orc_nrt_is_host_level(const spx_resource_classes* rc, int32_t res);
int orc_nrt_is_numa_affine(const spx_resource_classes* rc, int32_t res);
int orc_pod_effective_request(const spx_pod_objects* pods, int64_t pod, int32_t* res_out, int64_t* qty_out, int cap);
void orc_nrt_conf(const spx_nrt_objects* nrt, int64_t node, int* policy, int* scope, int* max_numa);
int orc_nrt_only_non_numa(const spx_nrt_objects* nrt, int64_t node, const spx_pod_objects* pods, int64_t pod);
int orc_nrt_test_subtract_numa(const spx_nrt_objects* nrt, const spx_resource_classes* rc, int64_t node, int numa_id, int qos,
                               const spx_pod_objects* pods, int64_t pod, const int32_t* q_res, int n_q, int64_t* out);
void orc_nrt_test_subtract_numas(const spx_nrt_objects* nrt, int64_t node, const spx_pod_objects* pods, int64_t pod, uint64_t bits,
                                 const int32_t* q_res, int n_q, int64_t* out);
/* preemption.GetNRTPostPodsEviction preemption.go:39-157 (arguments as spx_nrt_post_eviction); returns the SPX_EVICT_* code */
int orc_nrt_post_eviction(const spx_nrt_objects* nrt, const spx_resource_classes* rc, int64_t node, const spx_pod_objects* victims,
                          const uint8_t* victim_qos, const int32_t* ctr_numa, int32_t placement_present, int32_t placement_containers,
                          int64_t* zres_avail_out);
int orc_nrt_numa_nodes_required(const spx_nrt_objects* nrt, const spx_resource_classes* rc, const spx_pod_objects* pods,
                                int64_t pod, int64_t node, int qos, uint64_t* bitmask, int* is_min_distance);

/* ---- networkaware NetworkOverhead + TopologicalSort (pkg/networkaware/...) */
int orc_net_prefilter(const spx_node_objects* nodes, const spx_pod_objects* pods, const spx_appgroup_objects* ag,
                      const spx_nettopo_objects* nt, int64_t pod, int64_t* sat, int64_t* vio, int64_t* cost);
int orc_net_prefilter_range(const spx_node_objects* nodes, const spx_pod_objects* pods, const spx_appgroup_objects* ag,
                            const spx_nettopo_objects* nt, int64_t pod, int64_t node_begin, int64_t node_end,
                            int64_t* sat, int64_t* vio, int64_t* cost);
void orc_net_normalize(int64_t* scores, int64_t n);
int32_t orc_find_pod_order(const spx_appgroup_objects* ag, int32_t g, int32_t selector);
int orc_toposort_less(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int64_t p1, int64_t p2);
/* adjacent pairs of the proposed queue order that satisfy neither Less nor a PrioritySort tie; -1: not a permutation */
int64_t orc_toposort_order_violations(const spx_pod_objects* pods, const spx_appgroup_objects* ag, const int32_t* perm, int64_t n);

/* ---- CapacityScheduling.PreFilter (pkg/capacityscheduling/{capacity_scheduling,elasticquota}.go) */
int orc_quota_cmp2(const int64_t* x1, uint8_t x1_present, const int64_t* x2, const int64_t* y, uint8_t y_present, int64_t bound);
int orc_capacity_prefilter(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* q, int64_t pod);
/* CapacityScheduling.Reserve -> reserveResource (capacity_scheduling.go:350-364, elasticquota.go:89-98) on a caller-owned Used table */
void orc_capacity_reserve(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* q, int64_t pod,
                          int64_t* used, uint8_t* used_present);

/* ---- batch drivers: for each pod row in [row_begin,row_end): for each node: Score(); then
 *      NormalizeScore() over that pod's node list (feasible nodes only when `mask` != NULL, as
 *      upstream RunScorePlugins does).  out_raw / out_norm are [rows][n_nodes] int64 (either may
 *      be NULL).  `threads` > 1 splits the node loop like upstream's Parallelizer
 *      (chunked parallel-for copied at targetloadpacking_test.go:386-405). */
typedef struct orc_snapshot {
  const spx_node_objects* nodes;
  const spx_resource_classes* rc;
  const spx_pod_objects* pods;
  const spx_metrics_objects* metrics;
  const spx_assigned_objects* assigned;
  const spx_allocatable_params* alloc_params;
  const spx_tlp_params* tlp_params;
  const spx_lvrb_params* lvrb_params;
  const spx_nrt_objects* nrt;
  const spx_nrt_params* nrt_params;
  const spx_appgroup_objects* appgroups;
  const spx_nettopo_objects* nettopo;
  const spx_node_pods_objects* node_pods; /* LowRiskOverCommitment */
  const spx_lroc_params* lroc_params;
  const spx_power_model_objects* power_models; /* Peaks */
} orc_snapshot;

int orc_score_rows(const orc_snapshot* s, int plugin, int64_t row_begin, int64_t row_end,
                   const uint8_t* mask, int threads, int64_t* out_raw, int64_t* out_norm);

/* Filter() of a filter plugin for pod rows [row_begin,row_end) x all nodes: out_status [rows][n_nodes],
 * 0 = pass, else the plugin's reason code (255 = fwk.Error) */
int orc_filter_rows(const orc_snapshot* s, int plugin, int64_t row_begin, int64_t row_end, int threads, uint8_t* out_status);
/* one pod at a time, that pod's node loop chunked over `workers` threads that join per pod, NormalizeScore serial: the
 * reference benchmark's structure (targetloadpacking_test.go:369-405, parallelism 16).  out_norm ([rows][n_nodes]) may be NULL. */
int orc_cycle_rows(const orc_snapshot* s, int plugin, int64_t row_begin, int64_t row_end, int workers, int64_t* out_norm);

/* ---- the one-pod-at-a-time cycle with its Reserve side effects (orc_commit.c): what spx_commit_sequential must reproduce.
 * Pod rows [row_begin,row_end) in queue order; per pod: PreFilter, Filter + Score over the node list (cut into `threads`
 * contiguous ranges joined per pod), NormalizeScore, the weighted sum (weights[plugin id]; NULL = all 1), the tie set as
 * (lowest node index, size), then Reserve / bind on this call's private copies of the caches (NRT assumed store, trimaran
 * ScheduledPodsCache with bind time `bind_ts`, ElasticQuota Used + nominated list, AppGroup scheduled list).  The caller's tables
 * are not modified.  plugin_mask is a subset of {ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY}; `quota` only with CAPACITY.
 * node_out[i] = -1 for a pod that was turned away: verdict_out[i] (may be NULL) = the CapacityScheduling PreFilter code
 * (SPX_QUOTA_ST_*) or ORC_COMMIT_NO_FEASIBLE_NODE; score_out / ties_out may be NULL.  tlp_appended_out (may be NULL, [n_nodes]) =
 * cache entries appended per node.  Returns 0, <0 on bad arguments / allocation failure / the reference's Error paths. */
#define ORC_COMMIT_NO_FEASIBLE_NODE 255
typedef struct orc_commit_args {
  const orc_snapshot* s;
  const spx_quota_objects* quota;
  uint32_t plugin_mask;
  const int64_t* weights;
  int64_t row_begin;
  int64_t row_end;
  int64_t bind_ts;
  int threads;
  int32_t* tlp_appended_out;
} orc_commit_args;
int orc_commit_sequential(const orc_commit_args* a, int32_t* node_out, int64_t* score_out, int32_t* ties_out, uint8_t* verdict_out);
/* CPUs this process may run on (sched_getaffinity) capped by the cgroup's cpu.max quota — what "all host cores" means here */
int orc_usable_cpus(void);

#ifdef __cplusplus
}
#endif
#endif
