/*
 * spx.h — C ABI of the MI355X batched Filter/Score engine (libspx.so).
 *
 * This is the drop-in boundary for the one hot path of kubernetes-sigs/scheduler-plugins:
 * the per-pod x per-node Filter()/Score()/NormalizeScore() loops.  A Go plugin keeps its
 * framework.FilterPlugin / framework.ScorePlugin method signatures and binds these entry
 * points through cgo (INTEGRATION.md shows the stub).  Reference interfaces replaced:
 *
 *   Allocatable.Score / NormalizeScore   pkg/noderesources/allocatable.go:63,143
 *   TargetLoadPacking.Score              pkg/trimaran/targetloadpacking/targetloadpacking.go:107
 *   LoadVariationRiskBalancing.Score     pkg/trimaran/loadvariationriskbalancing/loadvariationriskbalancing.go:84
 *   TopologyMatch.Filter / Score         pkg/noderesourcetopology/filter.go:179, score.go:62
 *   NetworkOverhead.PreFilter/Filter/Score/NormalizeScore
 *                                        pkg/networkaware/networkoverhead/networkoverhead.go:174,326,362,389
 *   TopologicalSort.Less                 pkg/networkaware/topologicalsort/topologicalsort.go:102
 *   CapacityScheduling.PreFilter         pkg/capacityscheduling/capacity_scheduling.go:208
 *
 * Rules of the boundary (cgo-safe):
 *   - every function returns int: 0 = ok, <0 = error (spx_last_error() gives the text);
 *   - all tables are flat arrays of fixed-width ints/doubles, caller-owned and only
 *     borrowed for the duration of the call (no pointer is retained, no pointer-to-pointer);
 *   - results live in device memory owned by the engine; rows are fetched into
 *     caller-provided buffers; fetches after spx_sync() are read-only and lock-free, so
 *     16 concurrent reader goroutines (upstream Parallelizer) may call spx_fetch_* at once;
 *   - no global state (unlike targetloadpacking.go:49-53): one engine per scheduler profile.
 *
 * Two table levels exist:
 *   "object tables"  (spx_*_objects): a lossless columnar/CSR image of the API objects the
 *                    reference reads (pods, nodes, watcher metrics, NRT zones, AppGroups ...);
 *                    this is what the Go shim marshals.
 *   "SoA tables"     (spx_*_soa): dense per-node / per-pod columns the kernels read from HBM.
 *   spx_flatten_*() (host C++, part of this library) turns the former into the latter.
 *
 * The header is kept in a regular "one field per line" form because the Python ctypes
 * binding (scheduler-plugins_amd/_abi.py) parses it — it is the single source of truth.
 */
#ifndef SPX_H
#define SPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ constants */

#define SPX_OK 0
#define SPX_ERR_ARG (-1)
#define SPX_ERR_HIP (-2)
#define SPX_ERR_STATE (-3)
#define SPX_ERR_NOGPU (-4)

/* plugin ids (bit i of a plugin mask = plugin i) */
#define SPX_PLUGIN_ALLOCATABLE 0
#define SPX_PLUGIN_TLP 1
#define SPX_PLUGIN_LVRB 2
#define SPX_PLUGIN_NRT 3
#define SPX_PLUGIN_NETOVERHEAD 4
#define SPX_PLUGIN_CAPACITY 5
#define SPX_PLUGIN_TOPOSORT 6
#define SPX_PLUGIN_LROC 7  /* trimaran LowRiskOverCommitment */
#define SPX_PLUGIN_PEAKS 8 /* trimaran Peaks */
#define SPX_NUM_PLUGINS 9

/* fwk.Status codes (k8s.io/kube-scheduler/framework) */
#define SPX_STATUS_SUCCESS 0
#define SPX_STATUS_ERROR 1
#define SPX_STATUS_UNSCHEDULABLE 2

/* canonical resource ids; quantities are int64 in canonical units:
 * cpu = millicores (Quantity.MilliValue()), everything else = Quantity.Value() */
#define SPX_RES_CPU 0
#define SPX_RES_MEMORY 1
#define SPX_RES_EPHEMERAL 2
#define SPX_RES_PODS 3
#define SPX_RES_STORAGE 4
#define SPX_RES_FIRST_DYNAMIC 8

/* resource class flags (spx_resource_classes.flags[id]) */
#define SPX_RC_HUGEPAGE 1   /* v1helper.IsHugePageResourceName  */
#define SPX_RC_NATIVE 2     /* v1helper.IsNativeResource        */
#define SPX_RC_SCALAR 4     /* schedutil.IsScalarResourceName   */

/* container kinds */
#define SPX_CTR_APP 0
#define SPX_CTR_INIT 1
#define SPX_CTR_SIDECAR 2 /* init container with restartPolicy Always (pkg/util/sidecar.go) */

/* watcher metric types / operators (paypal/load-watcher v0.2.4 constants, resolved by the shim) */
#define SPX_MT_CPU 0
#define SPX_MT_MEMORY 1
#define SPX_MT_OTHER 2
#define SPX_MO_AVG 0
#define SPX_MO_STD 1
#define SPX_MO_LATEST 2
#define SPX_MO_EMPTY 3 /* Operator == "" */
#define SPX_MO_OTHER 4

/* Allocatable modes (apis/config/types.go ModeType) */
#define SPX_MODE_LEAST 0
#define SPX_MODE_MOST 1

/* ------------------------------------------------------------------ object tables */

/* class flags per resource id (ids are interned by the caller once per snapshot) */
typedef struct spx_resource_classes {
  int32_t n_res;
  const uint8_t* flags;
} spx_resource_classes;

/* v1.Pod image.  Containers of pod i are ctr_ptr[i]..ctr_ptr[i+1]-1, init containers first
 * (spec order) then app containers (spec order) — the order
 * append(pod.Spec.InitContainers, pod.Spec.Containers...) the reference walks.
 * Resource lists are CSR: container c requests = req_res/req_qty[req_ptr[c]..req_ptr[c+1]). */
typedef struct spx_pod_objects {
  int64_t n_pods;
  const int32_t* ctr_ptr;
  const uint8_t* ctr_kind;
  const int32_t* req_ptr;
  const int32_t* req_res;
  const int64_t* req_qty;
  const int32_t* lim_ptr;
  const int32_t* lim_res;
  const int64_t* lim_qty;
  const int32_t* ovh_ptr;
  const int32_t* ovh_res;
  const int64_t* ovh_qty;
  const int32_t* priority;
  const int64_t* queue_ts;
  const int32_t* appgroup;
  const int32_t* selector;
  const int32_t* ns;
} spx_pod_objects;

/* framework.NodeInfo / v1.Node image */
typedef struct spx_node_objects {
  int64_t n_nodes;
  const int64_t* alloc_cpu_milli;
  const int64_t* alloc_mem;
  const int64_t* alloc_eph;
  const int64_t* alloc_pods;
  const int32_t* scalar_ptr;
  const int32_t* scalar_res;
  const int64_t* scalar_qty;
  const int64_t* cap_cpu_milli;
  const int32_t* region;
  const int32_t* zone;
} spx_node_objects;

/* watcher.WatcherMetrics image (pkg/trimaran/collector.go:110-123) */
typedef struct spx_metrics_objects {
  int32_t map_is_nil;
  int64_t window_end;
  const uint8_t* node_present;
  const uint8_t* node_metrics_nil;
  const int32_t* m_ptr;
  const uint8_t* m_type;
  const uint8_t* m_op;
  const double* m_value;
} spx_metrics_objects;

/* PodAssignEventHandler.ScheduledPodsCache image (pkg/trimaran/handler.go:47-58):
 * entries of node n are e_ptr[n]..e_ptr[n+1]-1; e_pod indexes `pods` */
typedef struct spx_assigned_objects {
  const int32_t* e_ptr;
  const int64_t* e_ts_unix;
  const int32_t* e_pod;
  const spx_pod_objects* pods;
} spx_assigned_objects;

/* framework.NodeInfo.GetPods() image (what trimaran.GetNodeRequestsAndLimits walks, pkg/trimaran/resourcestats.go:163-206):
 * the pods already on node n are p_pod[p_ptr[n]..p_ptr[n+1]), indexes into `pods` */
typedef struct spx_node_pods_objects {
  const int32_t* p_ptr;
  const int32_t* p_pod;
  const spx_pod_objects* pods;
} spx_node_pods_objects;

/* PeaksArgs.NodePowerModel image (apis/config/types.go:309-324): the power model of node n as getPowerModel resolves it by
 * node name (peaks.go:190-196); all three are 0 for a node without an entry */
typedef struct spx_power_model_objects {
  const double* k0;
  const double* k1;
  const double* k2;
} spx_power_model_objects;

/* NodeResourceTopology CR image per node plus the NRT cache's verdict for it
 * (pkg/noderesourcetopology/cache: GetCachedNRTCopy -> (nrt, CachedNRTInfo{Fresh})).
 * legacy_policy encodes TopologyPolicies[0] (nodeconfig/topologymanager.go:131-162) as
 * (policy << 1) | scope, -1 when absent/unknown; attr_* are the Attributes overrides (:98-115),
 * -1 when absent or invalid.  policy: 0 none, 1 best-effort, 2 restricted, 3 single-numa-node;
 * scope: 0 container, 1 pod.  zone_numa_id / zcost_numa_id = numanode.NameToID(name), -1 on error.
 * assumed_* is the OverReserve store (cache/store.go:315-356): one resource list per assumed pod. */
typedef struct spx_nrt_objects {
  int64_t n_nodes;
  const uint8_t* has_nrt;
  const uint8_t* fresh;
  const int8_t* legacy_policy;
  const int8_t* attr_scope;
  const int8_t* attr_policy;
  const int32_t* attr_max_numa;
  const int32_t* zone_ptr;
  const uint8_t* zone_is_node;
  const int32_t* zone_numa_id;
  const int32_t* zres_ptr;
  const int32_t* zres_res;
  const int64_t* zres_avail;
  const int32_t* zcost_ptr;
  const int32_t* zcost_numa_id;
  const int64_t* zcost_value;
  const int32_t* assumed_ptr;
  const int32_t* arl_ptr;
  const int32_t* arl_res;
  const int64_t* arl_qty;
  const int64_t* zres_allocatable; /* optional (may be NULL), same layout as zres_avail: ResourceInfo.Allocatable; only the eviction simulation reads it */
} spx_nrt_objects;

/* diktyo AppGroup CRs (appgroup.diktyo.x-k8s.io v1alpha1) as the network-aware plugins read them.
 * Group g: Spec.Workloads = wl_ptr[g]..wl_ptr[g+1]-1 with their Dependencies (dep_*), Status.TopologyOrder
 * (topo_*), and the scheduled list util.GetScheduledList builds from the pod lister (placed_*: pods labelled
 * with the group that already have a node name; placed_node = index of that node in the snapshot).
 * Selector ids MUST preserve the lexicographic order of the selector strings: util.FindPodOrder
 * (pkg/networkaware/util/util.go:138-153) binary-searches TopologyOrder with string comparisons. */
typedef struct spx_appgroup_objects {
  int32_t n_groups;
  const int32_t* wl_ptr;
  const int32_t* wl_selector;
  const int32_t* dep_ptr;
  const int32_t* dep_selector;
  const int64_t* dep_max_cost;
  const int32_t* topo_ptr;
  const int32_t* topo_selector;
  const int32_t* topo_index;
  const int32_t* placed_ptr;
  const int32_t* placed_selector;
  const int32_t* placed_node;
} spx_appgroup_objects;

/* NetworkTopology CR (networktopology.diktyo.x-k8s.io v1alpha1), the weights set the plugin was configured
 * with (no.weightsName), keyed by interned label values: region ids and zone ids are the same id spaces as
 * spx_node_objects.region / .zone.  Origin o's CostList is the slice [ptr[o], ptr[o+1]) of the dest and cost arrays; later entries
 * for the same destination override earlier ones (map assignment, networkoverhead.go:467-472). */
typedef struct spx_nettopo_objects {
  int32_t n_regions;
  int32_t n_zones;
  const int32_t* rc_ptr;
  const int32_t* rc_dest;
  const int64_t* rc_cost;
  const int32_t* zc_ptr;
  const int32_t* zc_dest;
  const int64_t* zc_cost;
} spx_nettopo_objects;

/* CapacityScheduling: ElasticQuotaInfos + nominated pods (pkg/capacityscheduling/elasticquota.go:62-68,
 * capacity_scheduling.go:231-253).  A framework.Resource is a vector of SPX_QUOTA_SLOTS int64:
 * [0] MilliCPU [1] Memory [2] EphemeralStorage [3] AllowedPodNumber [4..7] ScalarResources by slot
 * (quota_scalar_res gives the canonical resource id of each scalar slot); *_present bit s (4..7) = the
 * scalar key exists in that Resource's map.  Quotas are indexed by namespace id (spx_pod_objects.ns). */
#define SPX_QUOTA_SLOTS 8
#define SPX_QUOTA_ST_OVER_MAX 1 /* "...is rejected in PreFilter because ElasticQuota %v is more than Max"      capacity_scheduling.go:276 */
#define SPX_QUOTA_ST_OVER_MIN 2 /* "...is rejected in PreFilter because total ElasticQuota used is more than min" :280 */
typedef struct spx_quota_objects {
  int32_t n_namespaces;
  int32_t n_scalar_slots;
  const int32_t* scalar_res;
  const uint8_t* has_quota;
  const int64_t* min;
  const uint8_t* min_present;
  const int64_t* max;
  const uint8_t* max_present;
  const int64_t* used;
  const uint8_t* used_present;
  int64_t n_nominated;
  const int32_t* nom_ns;
  const int32_t* nom_priority;
  const int64_t* nom_pending_index;
  const spx_pod_objects* nom_pods;
} spx_quota_objects;

/* ------------------------------------------------------------------ plugin params */

typedef struct spx_allocatable_params {
  int32_t mode;
  int32_t n_res;
  const int32_t* res;
  const int64_t* weight;
} spx_allocatable_params;

typedef struct spx_tlp_params {
  int64_t target_utilization;
  int64_t default_requests_milli;
  double requests_multiplier;
} spx_tlp_params;

typedef struct spx_lvrb_params {
  double safe_variance_margin;
  double safe_variance_sensitivity;
} spx_lvrb_params;

/* LowRiskOverCommitmentArgs after defaulting (apis/config/v1/defaults.go:172-188): SmoothingWindowSize > 0,
 * RiskLimitWeights["cpu"/"memory"] (lowriskovercommitment.go:76-81) */
typedef struct spx_lroc_params {
  int64_t smoothing_window_size;
  double risk_limit_weight_cpu;
  double risk_limit_weight_mem;
} spx_lroc_params;

/* NodeResourceTopologyMatch scoring strategy (apis/config/types.go ScoringStrategyType) */
#define SPX_NRT_MOST_ALLOCATED 0
#define SPX_NRT_BALANCED_ALLOCATION 1
#define SPX_NRT_LEAST_ALLOCATED 2
#define SPX_NRT_LEAST_NUMA_NODES 3

typedef struct spx_nrt_params {
  int32_t strategy;
  int32_t n_weights;
  const int32_t* weight_res;
  const int64_t* weight;
} spx_nrt_params;

/* ------------------------------------------------------------------ SoA tables (what the kernels read) */

/* Allocatable: alloc is [n_res][n_nodes] resource-major, rows in spx_allocatable_params.res order */
typedef struct spx_alloc_nodes_soa {
  int64_t n_nodes;
  int32_t n_res;
  const int64_t* alloc;
} spx_alloc_nodes_soa;

/* trimaran node columns (TLP reads Capacity, LVRB reads Allocatable — SURVEY appendix B.6) */
typedef struct spx_trimaran_nodes_soa {
  int64_t n_nodes;
  const int64_t* cap_cpu_milli;
  const double* tlp_cpu_util;
  const int64_t* tlp_missing_milli;
  const uint8_t* tlp_valid;
  const int64_t* lv_alloc_cpu_milli;
  const int64_t* lv_alloc_mem;
  const double* lv_cpu_avg;
  const double* lv_cpu_std;
  const double* lv_mem_avg;
  const double* lv_mem_std;
  const uint8_t* lv_flags;
} spx_trimaran_nodes_soa;

#define SPX_LV_HAS_METRICS 1
#define SPX_LV_CPU_VALID 2
#define SPX_LV_MEM_VALID 4

typedef struct spx_trimaran_pods_soa {
  int64_t n_pods;
  const int64_t* tlp_pod_milli;
  const int64_t* lv_req_cpu_milli;
  const int64_t* lv_req_mem;
} spx_trimaran_pods_soa;

/* LowRiskOverCommitment.  Reads the LVRB columns of spx_trimaran_nodes_soa (allocatable, avg/std, flags:
 * the same CreateResourceStats / GetResourceData inputs) plus, per node, the sums GetNodeRequestsAndLimits accumulates over the
 * pods already on it (requests, and limits raised to the requests; resourcestats.go:184-206, before the capacity caps) */
typedef struct spx_lroc_nodes_soa {
  int64_t n_nodes;
  const int64_t* req_cpu_milli;
  const int64_t* req_mem;
  const int64_t* lim_cpu_milli;
  const int64_t* lim_mem;
} spx_lroc_nodes_soa;

/* PodResourcesStateData (lowriskovercommitment.go:259-275): requests, and limits raised to the requests */
typedef struct spx_lroc_pods_soa {
  int64_t n_pods;
  const int64_t* req_cpu_milli;
  const int64_t* req_mem;
  const int64_t* lim_cpu_milli;
  const int64_t* lim_mem;
} spx_lroc_pods_soa;

/* Peaks.  cpu_util is the FIRST cpu metric whose operator is AVG or Latest (peaks.go:117-126; TLP takes
 * the last one, LVRB prefers AVG — three different selections); valid = the node has metrics and such a
 * metric exists; cap_cpu_milli is node.Status.Capacity (:131) */
typedef struct spx_peaks_nodes_soa {
  int64_t n_nodes;
  const int64_t* cap_cpu_milli;
  const double* cpu_util;
  const uint8_t* valid;
  const double* k1;
  const double* k2;
} spx_peaks_nodes_soa;

/* resource.GetResourceRequestQuantity(pod, cpu).MilliValue() (peaks.go:114-115) */
typedef struct spx_peaks_pods_soa {
  int64_t n_pods;
  const int64_t* cpu_milli;
} spx_peaks_pods_soa;

/* NodeResourceTopologyMatch.  Resources are renumbered into dense "slots" 0..n_res-1 (the union of
 * what pods request and zones report; slot_res gives the canonical id).  Limits of this build:
 * n_res <= 8, NUMA zones per node <= 8, containers per pod <= 8 (flatten fails beyond them). */
#define SPX_NRT_MAX_RES 8
#define SPX_NRT_MAX_ZONES 8
#define SPX_NRT_MAX_CTRS 8
#define SPX_NRT_F_HAS_NRT 1
#define SPX_NRT_F_FRESH 2
#define SPX_NRT_F_SINGLE_NUMA 4
#define SPX_NRT_F_POD_SCOPE 8
#define SPX_NRT_SLOT_AFFINE 1     /* isNUMAAffineResource  numaresources.go:120-135 */
#define SPX_NRT_SLOT_HOST_LEVEL 2 /* isHostLevelResource   numaresources.go:105-118 */
#define SPX_NRT_SLOT_CPU 4        /* quantity is millicores: Value() = ceil(milli/1000) */
#define SPX_QOS_GUARANTEED 0
#define SPX_QOS_BURSTABLE 1
#define SPX_QOS_BESTEFFORT 2

/* filter status reason codes of the NRT status table (0 = pass) and their reference messages */
#define SPX_NRT_ST_INVALID_TOPOLOGY 1 /* "invalid node topology data"      filter.go:199 */
#define SPX_NRT_ST_INIT_CONTAINER 2   /* "cannot align init container"     filter.go:55  */
#define SPX_NRT_ST_SIDECAR_CONTAINER 3/* "cannot align sidecar container"  filter.go:55  */
#define SPX_NRT_ST_CONTAINER 4        /* "cannot align container"          filter.go:67  */
#define SPX_NRT_ST_POD 5              /* "cannot align pod"                filter.go:172 */

typedef struct spx_nrt_slots {
  int32_t n_res;
  const int32_t* slot_res;
  const uint8_t* slot_flags;
  const int64_t* slot_weight;
} spx_nrt_slots;

typedef struct spx_nrt_nodes_soa {
  int64_t n_nodes;
  int32_t n_res;
  const uint8_t* flags;
  const int32_t* max_numa;
  const uint8_t* n_zones;
  const uint8_t* zone_id;
  const uint8_t* zone_present;
  const int64_t* zone_avail;
  const int32_t* zone_cost;
  const float* min_avg_dist;
  const uint8_t* node_present;
} spx_nrt_nodes_soa;

typedef struct spx_nrt_pods_soa {
  int64_t n_pods;
  int32_t n_res;
  const uint8_t* qos;
  const uint8_t* non_native;
  const uint8_t* n_ctr;
  const uint8_t* ctr_kind;
  const uint8_t* ctr_present;
  const int64_t* ctr_req;
  const uint8_t* pod_present;
  const int64_t* pod_req;
} spx_nrt_pods_soa;

/* NetworkOverhead.  A pod's PreFilter state depends only on its (AppGroup, workload selector) "workload
 * key"; the matched (placed pod, dependency) pairs are flattened once per key. */
#define SPX_NET_ST_UNSCHEDULABLE 1 /* "Node %v does not meet several network requirements ..." networkoverhead.go:355-357 */
#define SPX_NET_RAW_COST 0
#define SPX_NET_RAW_SATISFIED 1
#define SPX_NET_RAW_VIOLATED 2
#define SPX_NET_SAME_ZONE 1   /* networkoverhead.go:60  */
#define SPX_NET_MAX_COST 100  /* networkoverhead.go:63  */

typedef struct spx_net_nodes_soa {
  int64_t n_nodes;
  const int32_t* region;
  const int32_t* zone;
} spx_net_nodes_soa;

typedef struct spx_net_topo_soa {
  int32_t n_regions;
  int32_t n_zones;
  const int32_t* region_cost;
  const int32_t* zone_cost;
} spx_net_topo_soa;

typedef struct spx_net_pods_soa {
  int64_t n_pods;
  int32_t n_keys;
  const int32_t* pod_key;
  const uint8_t* key_score_equally;
  const int32_t* pair_ptr;
  const int32_t* pair_node;
  const int64_t* pair_max_cost;
  const int32_t* topo_order;
} spx_net_pods_soa;

/* TopologicalSort: the four per-pod values TopologicalSort.Less reads (topologicalsort.go:102-132): pod.Spec.Priority and
 * QueuedPodInfo.Timestamp (upstream PrioritySort, :109-113), the AppGroup id (-1 = no label) and the pod's index in its
 * group's Status.TopologyOrder as util.FindPodOrder returns it (-1 = selector not listed) — the `topo_order` column
 * spx_flatten_net_keys produces. */
typedef struct spx_sort_keys_soa {
  int64_t n_pods;
  const int32_t* priority;
  const int64_t* queue_ts;
  const int32_t* appgroup;
  const int32_t* topo_order;
} spx_sort_keys_soa;

/* CapacityScheduling.PreFilter: per-pod request vectors and per-namespace nominated lists (CSR, any order) */
typedef struct spx_quota_soa {
  int64_t n_pods;
  int32_t n_namespaces;
  const int32_t* pod_ns;
  const int32_t* pod_priority;
  const int64_t* pod_req;
  const uint8_t* pod_req_present;
  const uint8_t* has_quota;
  const int64_t* used;
  const uint8_t* used_present;
  const int64_t* max;
  const uint8_t* max_present;
  const int64_t* agg_used;
  const uint8_t* agg_used_present;
  const int64_t* agg_min;
  const uint8_t* agg_min_present;
  const int64_t* other_nominated;
  const uint8_t* other_nominated_present;
  const int32_t* nom_ptr;
  const int32_t* nom_priority;
  const int64_t* nom_pending_index;
  const int64_t* nom_req;
  const uint8_t* nom_req_present;
  const int64_t* min;          /* optional (may be NULL), [NS][8] + min_present [NS]: ElasticQuotaInfo.Min per namespace.  Only the */
  const uint8_t* min_present;  /* sequential commit loop reads them (a bound pod can push its quota over min, elasticquota.go:166-191) */
} spx_quota_soa;

/* NetworkOverhead bookkeeping of the sequential commit loop: what binding pending pod p adds to the AppGroup scheduled lists the
 * later pods see.  Entries of pod p: eff_key / eff_max_cost [eff_ptr[p], eff_ptr[p+1]); eff_max_cost >= 0: workload key eff_key gains
 * the pair (p's node, that MaxNetworkCost); -1: the key merely stops scoring equally.  Built by spx_flatten_net_commit. */
typedef struct spx_net_commit_soa {
  int64_t n_pods;
  const int32_t* eff_ptr;
  const int32_t* eff_key;
  const int64_t* eff_max_cost;
} spx_net_commit_soa;

/* ------------------------------------------------------------------ engine */

/* What each entry point stands in for in the reference (sigs.k8s.io/scheduler-plugins; the Go method keeps its
 * signature and becomes an index into a fetched row — INTEGRATION.md shows the cgo side):
 *
 *   spx_eval + spx_fetch_scores(ALLOCATABLE)   Allocatable.Score + NormalizeScore      pkg/noderesources/allocatable.go:63-71, :143-168
 *   spx_fetch_raw(ALLOCATABLE)                 Allocatable.Score (raw int64)           pkg/noderesources/allocatable.go:117-140
 *   spx_eval + spx_fetch_scores(TLP)           TargetLoadPacking.Score                 pkg/trimaran/targetloadpacking/targetloadpacking.go:107-187
 *   spx_eval + spx_fetch_scores(LVRB)          LoadVariationRiskBalancing.Score        pkg/trimaran/loadvariationriskbalancing/loadvariationriskbalancing.go:84-122
 *   spx_eval + spx_fetch_scores(LROC)          LowRiskOverCommitment.PreScore + Score  pkg/trimaran/lowriskovercommitment/lowriskovercommitment.go:96-141, :158-255, beta.go:85-191
 *   spx_eval + spx_fetch_scores(PEAKS)         Peaks.Score + NormalizeScore            pkg/trimaran/peaks/peaks.go:103-144, :150-166, :186-188
 *   spx_fetch_raw(PEAKS)                       Peaks.Score (raw int64: power jump x 1e15)
 *   spx_eval + spx_fetch_status(NRT)           TopologyMatch.Filter                    pkg/noderesourcetopology/filter.go:179-245
 *   spx_eval + spx_fetch_scores(NRT)           TopologyMatch.Score                     pkg/noderesourcetopology/score.go:62-102
 *   spx_eval + spx_fetch_status(NETOVERHEAD)   NetworkOverhead.PreFilter + Filter      pkg/networkaware/networkoverhead/networkoverhead.go:174-298, :326-359
 *   spx_fetch_raw(NETOVERHEAD, cost/sat/vio)   NetworkOverhead.Score, PreFilterState   networkoverhead.go:362-386, :85-115
 *   spx_eval + spx_fetch_scores(NETOVERHEAD)   NetworkOverhead.NormalizeScore          networkoverhead.go:389-418
 *   spx_eval + spx_fetch_prefilter(CAPACITY)   CapacityScheduling.PreFilter            pkg/capacityscheduling/capacity_scheduling.go:208-283
 *   spx_flatten_net_keys + spx_toposort_less   TopologicalSort.Less, FindPodOrder      pkg/networkaware/topologicalsort/topologicalsort.go:102-132, util/util.go:138-153
 *   spx_upload_sort_keys + spx_sort_keys       the activeQ ordering TopologicalSort.Less induces, as one device sort
 *   spx_flatten_trimaran_*                     GetNodeMetrics / ScheduledPodsCache / PredictUtilisation / GetResourceRequested
 *                                              pkg/trimaran/collector.go:110-123, handler.go:47-58, targetloadpacking.go:198-205, resourcestats.go:45-146
 *   spx_flatten_nrt_*                          createNUMANodeList / TopologyManagerFromNodeResourceTopology / GetPodEffectiveRequest / OverReserve
 *                                              pkg/noderesourcetopology/pluginhelpers.go:105-173, nodeconfig/topologymanager.go:78-162, pkg/util/resource.go:51-85, cache/store.go:315-356
 *   spx_flatten_net_topo                       populateCostMap                         networkoverhead.go:448-497
 *   spx_flatten_quota                          ElasticQuotaInfos (used / min / max, nominated pods)   pkg/capacityscheduling/elasticquota.go:48-123
 *   spx_eval_best + spx_fetch_best             upstream prioritizeNodes + selectHost input (sum of weight x score over feasible nodes)
 *   spx_decide + spx_fetch_best                the same decision input without materialising the per-plugin tables
 *   spx_commit_sequential                      upstream scheduleOne repeated over the queue, with trimaran's bind-time bookkeeping
 *                                              pkg/trimaran/handler.go:131-139, targetloadpacking.go:151-168
 */

typedef struct spx_engine spx_engine;

/* create an engine on HIP device `device_id`; fails with SPX_ERR_NOGPU when no device exists
 * (there is no CPU fallback by design) */
int spx_create(int device_id, spx_engine** out);
int spx_destroy(spx_engine* e);
/* last error text: of engine `e`, or of the failed spx_create when e == NULL (thread-local) */
const char* spx_last_error(const spx_engine* e);
/* number of entry points this build exports, and the ABI version */
int spx_abi_version(void);
/* run on an externally owned hipStream_t (e.g. torch's current stream); NULL = engine's own */
int spx_set_stream(spx_engine* e, void* hip_stream);
/* the stream kernels are launched on (for HIP-event timing by the caller) */
int spx_get_stream(spx_engine* e, void** hip_stream);

int spx_set_allocatable_params(spx_engine* e, const spx_allocatable_params* p);
int spx_set_tlp_params(spx_engine* e, const spx_tlp_params* p);
int spx_set_lvrb_params(spx_engine* e, const spx_lvrb_params* p);

int spx_set_nrt_params(spx_engine* e, const spx_nrt_params* p);
int spx_set_lroc_params(spx_engine* e, const spx_lroc_params* p);

int spx_upload_alloc_nodes(spx_engine* e, const spx_alloc_nodes_soa* t);
int spx_upload_trimaran_nodes(spx_engine* e, const spx_trimaran_nodes_soa* t);
int spx_upload_trimaran_pods(spx_engine* e, const spx_trimaran_pods_soa* t);
/* LowRiskOverCommitment tables; the node table needs spx_upload_trimaran_nodes first (same node count) */
int spx_upload_lroc_nodes(spx_engine* e, const spx_lroc_nodes_soa* t);
int spx_upload_lroc_pods(spx_engine* e, const spx_lroc_pods_soa* t);
int spx_upload_peaks_nodes(spx_engine* e, const spx_peaks_nodes_soa* t);
int spx_upload_peaks_pods(spx_engine* e, const spx_peaks_pods_soa* t);
/* Snapshot deltas (SURVEY 8d "upload deltas"): `t` holds t->n_nodes ROWS in the layout of the full upload; row i replaces node
 * idx[i] of the table already on the device (spx_upload_* must have run once: it fixes the shape).  The changed rows travel as one
 * staged blob and are scattered into the device columns; for NRT the float64 formulation's derived columns are recomputed on the
 * device for those nodes (bit-identical to a full re-upload).  Every table evaluated from the old rows becomes stale.
 *   trimaran: collector refresh of a node's metrics (collector.go:139-150), the bind-time cache (handler.go:131-139)
 *   NRT: a republished NodeResourceTopology (pluginhelpers.go:105-161), an assumed pod charged to a node (overreserve.go:170-203) */
int spx_update_trimaran_nodes(spx_engine* e, const int64_t* idx, const spx_trimaran_nodes_soa* t);
int spx_update_nrt_nodes(spx_engine* e, const int64_t* idx, const spx_nrt_nodes_soa* t);
/* NetworkOverhead: pairs appended to the workload keys' lists (AppGroup scheduled lists grow between cycles, networkoverhead.go:654-694).
 * Entry i: key[i] gains (node[i], max_cost[i]); max_cost[i] == -1: the key merely stops scoring equally.  Built by spx_flatten_net_placed.
 * Equal to re-uploading spx_flatten_net_keys' tables of the grown AppGroups (tests/test_gpu_delta.py). */
int spx_update_net_placed(spx_engine* e, int64_t n, const int32_t* key, const int32_t* node, const int64_t* max_cost);
/* CapacityScheduling: rows of ElasticQuotaInfo.Used replaced (AddPod / DeletePod events, capacity_scheduling.go:679-803): namespace ns[i]
 * takes used[i][SPX_QUOTA_SLOTS] / used_present[i]; agg_used[SPX_QUOTA_SLOTS] / *agg_used_present: the new aggregate over all quotas. */
int spx_update_quota_used(spx_engine* e, int64_t n_rows, const int32_t* ns, const int64_t* used, const uint8_t* used_present, const int64_t* agg_used, const uint8_t* agg_used_present);
int spx_upload_nrt_slots(spx_engine* e, const spx_nrt_slots* t);
int spx_upload_nrt_nodes(spx_engine* e, const spx_nrt_nodes_soa* t);
int spx_upload_nrt_pods(spx_engine* e, const spx_nrt_pods_soa* t);
int spx_upload_net_nodes(spx_engine* e, const spx_net_nodes_soa* t);
int spx_upload_net_topo(spx_engine* e, const spx_net_topo_soa* t);
int spx_upload_net_pods(spx_engine* e, const spx_net_pods_soa* t);
int spx_upload_quota(spx_engine* e, const spx_quota_soa* t);
/* after spx_upload_net_pods; only spx_commit_sequential with NETOVERHEAD in its mask needs it */
int spx_upload_net_commit(spx_engine* e, const spx_net_commit_soa* t);
/* TopologicalSort as one batched sort of the pending queue (replaces the activeQ heap's pairwise Less calls).  Less is not a
 * strict weak order (same AppGroup: `orderP1 <= orderP2`; otherwise PrioritySort), but it is complete, so an order exists in
 * which EVERY ADJACENT PAIR (x, y) satisfies Less(x, y) — or ties under PrioritySort (equal priority and timestamp).
 * spx_sort_keys returns such an order: perm_out[i] = pod row at queue position i.  Construction (kernels_sort.hip): stable
 * radix sort by (priority descending, timestamp ascending), then every maximal run of consecutive pods of one AppGroup is
 * reordered by topology index.  The table is independent of the other pod tables (its n_pods is the queue length).
 * Synchronous; spx_last_eval_ms reports the device time. */
int spx_upload_sort_keys(spx_engine* e, const spx_sort_keys_soa* t);
int spx_sort_keys(spx_engine* e, int32_t* perm_out);
/* CapacityScheduling.PreFilter status per pod (n_pods bytes: 0 Success, SPX_QUOTA_ST_*); valid after spx_eval with the CAPACITY bit */
int spx_fetch_prefilter(spx_engine* e, int plugin, int64_t row_begin, int64_t row_end, uint8_t* out);

/* optional per-(pod,node) feasibility mask for normalizing score plugins: uint8 [n_pods][n_nodes],
 * non-zero = node passed Filter for that pod (upstream scores feasible nodes only).  NULL clears it. */
int spx_upload_feasible_mask(spx_engine* e, const uint8_t* mask, int64_t n_pods, int64_t n_nodes);

/* evaluate plugins in `plugin_mask` for pod rows [row_begin,row_end) against all nodes;
 * asynchronous on the engine stream; result tables stay in HBM */
int spx_eval(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end);
int spx_sync(spx_engine* e);

/* normalized score row of one pod (n_nodes bytes, 0..100) */
int spx_fetch_scores(spx_engine* e, int plugin, int64_t pod_row, uint8_t* out);
/* filter status row of one pod (n_nodes bytes; 0 = pass, else plugin-specific reason code) */
int spx_fetch_status(spx_engine* e, int plugin, int64_t pod_row, uint8_t* out);
/* rows [row_begin,row_end) of a score / status table in one strided copy: row r lands at out + (r - row_begin) * out_stride
 * (out_stride >= n_nodes bytes) — what a caller that drains a whole batch (or a parity harness) uses instead of row-by-row fetches */
int spx_fetch_score_rows(spx_engine* e, int plugin, int64_t row_begin, int64_t row_end, uint8_t* out, int64_t out_stride);
int spx_fetch_status_rows(spx_engine* e, int plugin, int64_t row_begin, int64_t row_end, uint8_t* out, int64_t out_stride);
/* exactness bookkeeping of the fast formulations: how many cells the float32 sweeps of TLP, LVRB and
 * LowRiskOverCommitment could not prove to round like the reference and re-evaluated with the reference's float64 sequence,
 * accumulated per plugin id since the last reset (SPX_NUM_PLUGINS entries; plugins without a fallback report 0).  Synchronises
 * the engine stream. */
int spx_fetch_stats(spx_engine* e, int64_t* reevaluated_cells, int reset);
/* raw int64 Score() row (before NormalizeScore) recomputed for one pod — the parity harness
 * and direct-call tests observe raw values (e.g. networkoverhead_test.go:803) */
int spx_fetch_raw(spx_engine* e, int plugin, int which, int64_t pod_row, int64_t* out);

/* device pointer + row stride (bytes) of a plugin's uint8 score table, for RCCL all-gather
 * by the caller (torch.distributed) or for zero-copy consumers */
int spx_score_table(spx_engine* e, int plugin, void** dptr, int64_t* row_stride, int64_t* n_rows);
/* make the engine write a plugin's score table into caller-owned device memory */
int spx_bind_score_table(spx_engine* e, int plugin, void* dptr, int64_t row_stride, int64_t n_rows);

/* per-pod weighted argmax over the evaluated plugins: best[k] node indices and their
 * sum_i weight[i]*score_i (upstream selectHost input); infeasible nodes are skipped.
 * `weights` holds SPX_NUM_PLUGINS entries indexed by plugin id */
int spx_set_plugin_weights(spx_engine* e, const int64_t* weights);
int spx_eval_best(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end);
/* per pod in [row_begin,row_end): best node (lowest index among ties, -1 when no node is feasible or the pod
 * failed CapacityScheduling.PreFilter), its weighted score, how many nodes tie for it, how many were feasible;
 * n_ties / n_feasible may be NULL */
int spx_fetch_best(spx_engine* e, int64_t row_begin, int64_t row_end, int32_t* node_idx, int64_t* weighted_score, int32_t* n_ties, int32_t* n_feasible);

/* Decisions without tables: what spx_eval + spx_eval_best would leave for spx_fetch_best, computed in one sweep that never
 * writes a score table (the per-row argmax is folded into the sweep: no 1 B/cell/plugin to HBM and back).  Fused for the
 * Filter-less profiles made of TLP, optionally ALLOCATABLE, and any of the Score-only plugins LVRB / LROC / PEAKS, with
 * non-negative plugin weights and no caller feasibility mask.  ALLOCATABLE's and TLP's tables are never written; the
 * Score-only plugins' tables ARE evaluated (spx_eval on those plugins, so they can be fetched afterwards) and read once by
 * the fused sweep in place of spx_eval_best's pass over every table.  Any other request is served by running spx_eval and
 * spx_eval_best.  Asynchronous on the engine stream like spx_eval. */
int spx_decide(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end);

/* Sequential scheduling of pod rows [row_begin,row_end), in row (= queue) order.  Unlike spx_eval's frozen
 * snapshot, every pod sees the commits of the pods before it:
 *   trimaran           a bound pod adds its predicted CPU utilisation to its node's missing utilisation
 *                      (pkg/trimaran/handler.go:131-139 feeding targetloadpacking.go:151-168);
 *   NRT                TopologyMatch.Reserve: the pod's effective request is subtracted from every zone of its node that reports the
 *                      resource (reserve.go:28-46, cache/overreserve.go:170-186, cache/store.go:315-356);
 *   CAPACITY           Reserve: the namespace's Used grows by the pod's request, a nominated pod that gets bound stops counting as
 *                      nominated (capacity_scheduling.go:350-364, elasticquota.go:89-98) — needs spx_quota_soa.min;
 *   NETOVERHEAD        the pod joins its AppGroup's scheduled list (networkoverhead.go:205-224) — needs spx_upload_net_commit.
 * plugin_mask is a subset of {ALLOCATABLE, TLP, LVRB, NRT, NETOVERHEAD, CAPACITY}.  Without a Filter plugin the whole chain runs in one
 * workgroup (about 3 us per pod); with one, every pod is one single-row spx_eval + spx_eval_best + a bookkeeping launch on the engine
 * stream (tens of us per pod), score / status tables end up holding each row as its pod saw it (except Allocatable's when Filter
 * plugins are in the mask: its feasibility-aware normalisation then happens inside the argmax kernel, as in spx_decide).  A pod that fails PreFilter or has no
 * feasible node gets node -1 and reserves nothing.  Per pod: the node with the highest
 * sum of plugin_weight x score (lowest index among ties; upstream's selectHost draws among them), that sum, and the size
 * of the tie set (NULL = not wanted).  tlp_missing_out (NULL = not wanted) receives the per-node missing utilisation after
 * the last commit.  The engine's uploaded tables are left untouched; with LVRB in the mask its score table is (re)evaluated
 * for the row range first (LVRB carries no commit state, so the loop reads those rows).  Synchronous. */
int spx_commit_sequential(spx_engine* e, uint32_t plugin_mask, int64_t row_begin, int64_t row_end, int32_t* node_idx, int64_t* weighted_score, int32_t* n_ties, int64_t* tlp_missing_out);

/* duration in ms of the last spx_eval's kernels measured with HIP events on the engine stream */
int spx_last_eval_ms(spx_engine* e, float* ms);

/* Per-engine options (no process-wide state: two profiles in one scheduler process may differ).  Unknown option or value
 * out of range -> SPX_ERR_ARG.  Options take effect at the next launch, except SPX_OPT_ROW_ALIGN (before the first upload).
 *   SPX_OPT_ROW_ALIGN          row padding of the result tables in bytes (multiple of 16; default 128)
 *   SPX_OPT_REFERENCE_KERNELS  bit mask of plugin ids whose sweep runs the reference-arithmetic ("generic") kernel instead of
 *                              the fast formulation: TLP and LVRB (one switch for both), NRT, NETOVERHEAD, LROC (int64 form)
 *   SPX_OPT_LROC_FLOAT64       1 = LowRiskOverCommitment in the float64 form (no float32 quotient; that form also runs, whatever the
 *                              option says, when a column holds a value from 2^47 or a limit below its request)
 *   SPX_OPT_DECIDE_UNFUSED     1 = spx_decide always runs spx_eval + spx_eval_best
 *   SPX_OPT_NRT_SINGLE_LAUNCH  1 = NRT Filter and Score in one launch (default: two for Least/MostAllocated)
 *   SPX_OPT_COMMIT_FROM_MEMORY 1 = spx_commit_sequential keeps node state in memory (any node count) instead of registers
 *   SPX_OPT_PEAKS_TILE         nodes per lane of Peaks' float64 (min/max pass, write pass) — SPX_OPT_PEAKS_ESTIMATE 0: 44 (default), 84, 48, 88
 *   SPX_OPT_NRT_POD_CLASSES    1 (default) = a whole-batch NRT sweep evaluates one representative row per class of pods whose
 *                              records agree in everything the sweep reads and copies it to the rest of the class; 0 = every row
 *   SPX_OPT_NRT_LN_LIST_PERMILLE  LeastNUMANodes, batch Score launch: room, in thousandths of the node count, of each per-(row, scope)
 *                              list of cells left to the complete subset search (default 375; 4 bytes per entry).  A list that
 *                              overflows sends the launch back to the complete sweep: same table, slower
 *   SPX_OPT_PEAKS_POD_CLASSES  1 (default) = a whole-batch Peaks sweep with no Filter table in play evaluates one row per distinct
 *                              pod cpu request (all Peaks.Score reads of the pod, peaks.go:134-138) and copies it; 0 = every row
 *   SPX_OPT_COMMIT_COOP        1 (default) = spx_commit_sequential with Filter plugins in the mask runs as ONE cooperative persistent launch
 *                              (node state in registers, two granule exchanges per pod) when the profile fits it; 0 = always the per-pod
 *                              single-row launches replayed from a graph
 *   SPX_OPT_NRT_RANK_FILTER    1 (default) = a whole-batch NRT sweep over pod classes runs its Filter launch in rank space (requests and
 *                              zone quantities as positions in the chunk's sorted request list: integer subtracts instead of float64
 *                              compares, no zone-table mutation); 0 = the float64 Filter launch.  Same status table either way
 *   SPX_OPT_ROW_WORKGROUP      1 = the per-row kernels of a profile with Filter plugins (Allocatable's feasibility-aware NormalizeScore,
 *                              spx_decide's argmax) give every row a whole workgroup in batch launches too; 0 (default) = four rows per
 *                              workgroup, and the whole-workgroup mapping only when four rows' feasibility bytes would not fit the LDS
 *                              (rows of more than about 65k nodes).  Same tables either way
 *   SPX_OPT_TLP_AMB_TABLE      1 (default) = a multi-row TargetLoadPacking sweep (tables or spx_decide; LoadVariationRiskBalancing's sweep likewise, per cpu / memory request) first lists, per pod value and node tile,
 *                              where a cell of the float32 formulation can be within its error bound of a rounding tie or of the branch point
 *                              (a property of the node alone: the score is piecewise linear in the pod's integer millicores), and only the rows
 *                              named there carry the per-cell exactness bookkeeping; 0 = every cell carries it.  Same tables either way
 *   SPX_OPT_NRT_PACKED_SCORE   1 (default) = NodeResourceTopologyMatch LeastAllocated's Score launch of a row range keeps the zone totals of two
 *                              zones per register and scores in packed float32 (4 instructions per zone pair and resource instead of 6) when
 *                              every weighted slot qualifies: capacities / 2^s <= 32768 with 2^s the power of two common to the slot's
 *                              capacities and requests (cpu in whole cores, devices, hugepages in pages), or — one slot, memory in bytes —
 *                              through a per-launch table of the requests for which the float32 form differs from the division (those pods
 *                              are recomputed in float64 for the node window concerned); spx_nrt_packed_score_slots reports which;
 *                              0 = float64 throughout.  Same tables either way
 *   SPX_OPT_NET_ALLOC_FUSED    1 (default) = when NetworkOverhead and NodeResourcesAllocatable are evaluated together over a row range with
 *                              Filter plugins in play, the NetworkOverhead table sweep also writes Allocatable's table (NormalizeScore over
 *                              each pod's feasible nodes — the set that sweep already walks) instead of a separate launch that reads every
 *                              status table again; 0 = separate launches.  Same tables either way
 *   SPX_OPT_NRT_RANK_NARROW    1 (default) = chunks of the rank-space Filter's stream (SPX_OPT_NRT_RANK_FILTER) whose lists all have at most 127
 *                              distinct quantities keep four zones' counts per register instead of two (half the subtract / and instructions
 *                              per comparison); 0 = two per register everywhere.  Read when pod rows are uploaded.  Same tables either way
 *   SPX_OPT_PEAKS_ESTIMATE     1 (default) = both passes of Peaks (row min / max, then NormalizeScore) first bound every cell's raw score by a
 *                              float32 interval proven to contain the float64 value (13 float32 instructions + one v_exp_f32) and run the
 *                              float64 sequence (division, exp) only for the cells the interval cannot decide — the candidates for a row's
 *                              extremes, the cells next to a step of floor(100 (raw - min) / span), nodes outside the interval's
 *                              preconditions — listed by the sweep and evaluated by a second launch with every lane busy; rows in which
 *                              the interval decides little (a pod that requests no cpu) take the float64 sequence for the whole tile;
 *                              8 = the same with 8 instead of 16 nodes per lane; 0 = the float64 sequence for every cell.  Same tables either way.
 *                              (The interval's bounds assume cpu requests >= 0: a batch that holds a negative one runs the float64 passes)
 *                              Scratch, allocated by the first Peaks evaluation with the option on and kept: the undecided cells' lists, 24 bytes
 *                              per (swept pod row, tile of 1024 nodes — 512 with the value 8) whatever the chunking — 24 MB for 100 000 rows x
 *                              10 000 nodes, 240 MB for 500 000 x 20 000 — plus 96 bytes per node; a list that fills up falls back to one
 *                              "whole tile" entry, so the size bounds memory, not correctness
 *   SPX_OPT_NRT_FUSED          1 (default) = a whole-batch NodeResourceTopologyMatch sweep runs Filter and Score in ONE launch
 *                              (kernels_nrt_fused.hip: rank-space Filter, float32 Score, pod records staged once) for the Least- and
 *                              MostAllocated strategies with unit weights under the preconditions of SPX_OPT_NRT_RANK_FILTER and
 *                              SPX_OPT_NRT_PACKED_SCORE, and for BalancedAllocation with up to four resource slots (its undecided cells are
 *                              recomputed in float64 by a second launch); 0 = the Filter launch and the Score launch.  Same tables either way
 */
#define SPX_OPT_ROW_ALIGN 0
#define SPX_OPT_REFERENCE_KERNELS 1
#define SPX_OPT_LROC_FLOAT64 2
#define SPX_OPT_DECIDE_UNFUSED 3
#define SPX_OPT_NRT_SINGLE_LAUNCH 4
#define SPX_OPT_COMMIT_FROM_MEMORY 5
#define SPX_OPT_PEAKS_TILE 6
#define SPX_OPT_NRT_POD_CLASSES 7
#define SPX_OPT_PEAKS_POD_CLASSES 8
#define SPX_OPT_NRT_LN_LIST_PERMILLE 9
#define SPX_OPT_COMMIT_COOP 10
#define SPX_OPT_NRT_RANK_FILTER 11
#define SPX_OPT_ROW_WORKGROUP 12
#define SPX_OPT_TLP_AMB_TABLE 13
#define SPX_OPT_NRT_PACKED_SCORE 14
#define SPX_OPT_NET_ALLOC_FUSED 15
#define SPX_OPT_NRT_RANK_NARROW 16
#define SPX_OPT_PEAKS_ESTIMATE 17
#define SPX_OPT_NRT_FUSED 18
#define SPX_NUM_OPTIONS 19
int spx_set_option(spx_engine* e, int option, int64_t value);
int spx_get_option(const spx_engine* e, int option, int64_t* value);

/* Pod equivalence classes of the uploaded NRT pod batch (spx_upload_nrt_pods): n_unique = rows that represent a class (or only
 * themselves), n_copies = rows that repeat an earlier row's NRT record in everything the sweep reads — Deployment replicas, and
 * pods whose verdict does not depend on quantities (not filtered: filter.go:186-190; non-Guaranteed: score.go:72-76,
 * numaresources.go:137-142).  n_unique + n_copies = n_pods.  A whole-batch spx_eval of NRT evaluates the unique rows and copies. */
int spx_nrt_pod_classes(const spx_engine* e, int64_t* n_unique, int64_t* n_copies);
/* The same for the uploaded Peaks pod batch (spx_upload_peaks_pods): Peaks.Score reads one number of the pod — the cpu request of
 * peaks.go:134 (GetResourceRequestQuantity) — so pods that request the same amount get the same raw row, and, when no Filter plugin
 * or feasibility mask narrows a pod's node list, the same NormalizeScore (peaks.go:150-166). */
int spx_peaks_pod_classes(const spx_engine* e, int64_t* n_unique, int64_t* n_copies);

/* Object tables -> SoA columns -> device in one call per plugin family: the host flatteners (spx_flatten_*, below) run with the
 * engine's current plugin parameters and their result is uploaded.  For callers that hold object tables — marshalled by the shim or
 * decoded by spx_ingest_* — and would rather not size and own the intermediate arrays (the cgo shim: shim/go/pkg/spx/snapshot.go).
 * spx_load_trimaran serves Allocatable + TargetLoadPacking + LoadVariationRiskBalancing (rc / assigned may be NULL);
 * spx_load_network also uploads the commit effects spx_commit_sequential needs. */
int spx_load_trimaran(spx_engine* e, const spx_node_objects* nodes, const spx_resource_classes* rc, const spx_pod_objects* pods, const spx_metrics_objects* metrics, const spx_assigned_objects* assigned);
/* a new pending batch only (node tables stay): the trimaran / Allocatable pod columns flattened straight into pinned staging, one pass */
int spx_load_trimaran_pods(spx_engine* e, const spx_pod_objects* pods);
int spx_load_nrt(spx_engine* e, const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_resource_classes* rc, const spx_pod_objects* pods, const spx_nrt_params* params);
/* wall time in ms of the six stages of the last spx_load_nrt on this engine: [0] spx_flatten_nrt_slots, [1] spx_flatten_nrt_nodes,
 * [2] spx_flatten_nrt_pods, [3] parameters + slot table, [4] spx_upload_nrt_nodes, [5] spx_upload_nrt_pods (bench.py reports them) */
int spx_last_load_nrt_ms(const spx_engine* e, double* ms6);
int spx_load_network(spx_engine* e, const spx_node_objects* nodes, const spx_pod_objects* pods, const spx_appgroup_objects* appgroups, const spx_nettopo_objects* nettopo);
int spx_load_quota(spx_engine* e, const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* quota);
/* The whole profile in one call: the loaders above run side by side on host threads of the library (they fill disjoint tables and share
 * the engine's stream; spx_load_nrt itself runs its node half and its pod half on two threads).  nodes and pods are required; a loader
 * whose members are NULL is skipped: metrics (+ rc, assigned) = spx_load_trimaran, nrt + nrt_params = spx_load_nrt, appgroups + nettopo =
 * spx_load_network, quota = spx_load_quota.  Returns the first failing loader's code.  No other call on the engine may run meanwhile. */
typedef struct spx_profile_objects {
  const spx_node_objects* nodes;
  const spx_resource_classes* rc;
  const spx_pod_objects* pods;
  const spx_metrics_objects* metrics;
  const spx_assigned_objects* assigned;
  const spx_nrt_objects* nrt;
  const spx_nrt_params* nrt_params;
  const spx_appgroup_objects* appgroups;
  const spx_nettopo_objects* nettopo;
  const spx_quota_objects* quota;
} spx_profile_objects;
int spx_load_profile(spx_engine* e, const spx_profile_objects* o);

/* Which form the last spx_commit_sequential ran: 1 = the one-workgroup chain of the Filter-less profile, 2 = per-pod single-row
 * launches (replayed from a graph), 3 = the cooperative persistent kernel; 0 = none yet */
int spx_commit_path(const spx_engine* e);
/* Which Filter launch the last NodeResourceTopologyMatch sweep ran: 1 = float64 compares (k_nrt_fast / the reference-arithmetic
 * kernel), 2 = rank space (SPX_OPT_NRT_RANK_FILTER: whole-batch sweeps), 3 = rank space inside the fused Filter + Score launch
 * (SPX_OPT_NRT_FUSED); 0 = none yet */
int spx_nrt_filter_path(const spx_engine* e);
/* SPX_OPT_NRT_PACKED_SCORE with the uploaded tables and parameters: 0 = the LeastAllocated Score launch keeps float64 (other strategy,
 * large weights, a slot that qualifies neither way, option off); else bit 24 set, bits 0..15 = the weighted slots (positions of the
 * uploaded slot table) that are packed unconditionally, bits 16..23 = 1 + the slot that goes through the per-launch table, 0 = none */
int spx_nrt_packed_score_slots(const spx_engine* e);

/* which formulation of a plugin's sweep the uploaded tables select: 0 = generic (reference arithmetic, operation for
 * operation), 1 = fast formulation (same results; see DESIGN.md for each kernel's preconditions); <0 on error.
 * Tests use it to make sure both formulations are exercised. */
int spx_kernel_path(const spx_engine* e, int plugin);

/* make the engine write a Filter plugin's status table (NRT, NETOVERHEAD) into caller-owned device memory; row_stride must be
 * the engine's (spx_score_table reports it), n_rows >= n_pods; dptr NULL unbinds */
int spx_bind_status_table(spx_engine* e, int plugin, void* dptr, int64_t row_stride, int64_t n_rows);

/* ------------------------------------------------------------------ several devices, one host process
 *
 * north_star: "the pods x nodes matrix shards by pod rows across the 8 GPUs of one node with a single RCCL all-gather over
 * xGMI to reassemble the global score/feasibility table", driven by ONE Go scheduler process.  spx_multi owns one engine per
 * device and one host thread per device (launches of a step are issued concurrently).  Pod rows shard in equal contiguous
 * ranges (spx_multi_shard); node tables are replicated: the caller uploads the node tables to every rank's engine and each
 * rank's slice of the pod columns to that rank's engine (spx_multi_engine + the spx_upload_* calls — slicing SoA pod columns is
 * pointer arithmetic).  The evaluation itself has no collective.  Afterwards:
 *   spx_multi_gather_best        all-gathers the per-pod decisions (20 B per pod) so that every device holds the global vector and
 *                                hands it to the host in batch order;
 *   spx_multi_bind_global_table  gives every device a [n_pods_total][row_stride] uint8 table (which: 0 score, 1 Filter status)
 *                                into whose own slice the rank's engine writes directly (no staging copy), and
 *   spx_multi_allgather_table    reassembles it on every device in place.
 * transport: SPX_MULTI_TRANSPORT_RCCL = ncclAllGather on each engine's stream (librccl.so.1 is dlopen'ed here, so a
 * single-GPU scheduler never maps it; needs distinct devices); SPX_MULTI_TRANSPORT_PEER_COPY = the same bytes with
 * hipMemcpyPeerAsync (also works with several ranks on one device: how the sharding logic is tested on a one-GPU box).
 * Calls on one spx_multi must come from one thread at a time. */
#define SPX_MULTI_TRANSPORT_RCCL 0
#define SPX_MULTI_TRANSPORT_PEER_COPY 1
typedef struct spx_multi spx_multi;
int spx_multi_create(const int* device_ids, int n_devices, int transport, spx_multi** out);
int spx_multi_destroy(spx_multi* m);
/* of `m`, or of the failed spx_multi_create when m == NULL (thread-local) */
const char* spx_multi_last_error(const spx_multi* m);
int spx_multi_size(const spx_multi* m);
int spx_multi_engine(spx_multi* m, int rank, spx_engine** out);
/* ranks of the RCCL communicator the exchange runs on, as RCCL reports it (ncclCommCount of rank 0's communicator after
 * ncclCommInitAll); 0 with the peer-copy transport.  bench.py prints it so that a scaling record shows how many ranks RCCL saw */
int spx_multi_rccl_ranks(const spx_multi* m);
/* rank's rows of a batch of n_pods_total pending pods: [rank * per, (rank + 1) * per) clipped to the batch, per = ceil(n / size) */
int spx_multi_shard(const spx_multi* m, int64_t n_pods_total, int rank, int64_t* row_begin, int64_t* row_end);
/* every rank: spx_eval / spx_eval_best / spx_decide over all of its local rows, issued concurrently, asynchronous */
int spx_multi_eval(spx_multi* m, uint32_t plugin_mask);
int spx_multi_eval_best(spx_multi* m, uint32_t plugin_mask);
int spx_multi_decide(spx_multi* m, uint32_t plugin_mask);
int spx_multi_sync(spx_multi* m);
/* outputs are indexed by batch row, n_pods_total entries each; n_ties / n_feasible may be NULL.  Synchronous. */
int spx_multi_gather_best(spx_multi* m, int64_t n_pods_total, int32_t* node_idx, int64_t* weighted_score, int32_t* n_ties, int32_t* n_feasible);
int spx_multi_bind_global_table(spx_multi* m, int plugin, int which, int64_t n_pods_total);
int spx_multi_allgather_table(spx_multi* m, int plugin, int which);
/* rank's copy of the global table (device pointer on that rank's device) */
int spx_multi_global_table(spx_multi* m, int plugin, int which, int rank, void** dptr, int64_t* row_stride, int64_t* n_rows);
/* batch rows [row_begin,row_end) of rank's copy of a gathered global table (any rank holds all rows) */
int spx_multi_fetch_global_rows(spx_multi* m, int plugin, int which, int rank, int64_t row_begin, int64_t row_end, uint8_t* out, int64_t out_stride);
/* region timing with HIP events on every rank's stream: spx_multi_mark(m, 0) ... work ... spx_multi_mark(m, 1), then the
 * elapsed time between the marks per rank (ms_per_rank: size entries, may be NULL) and its maximum */
int spx_multi_mark(spx_multi* m, int which);
int spx_multi_marked_ms(spx_multi* m, float* ms_max, float* ms_per_rank);
/* HIP-event durations, max over ranks: of the last spx_multi_eval / _decide, and of the last gather (either may be NULL) */
int spx_multi_last_ms(spx_multi* m, float* eval_ms, float* gather_ms);

/* ------------------------------------------------------------------ host flatteners (object -> SoA) */

/* output buffers are caller-allocated with the sizes noted */
int spx_flatten_alloc_nodes(const spx_node_objects* nodes, const spx_resource_classes* rc, const spx_allocatable_params* p, int64_t* alloc_out);
int spx_flatten_trimaran_nodes(const spx_node_objects* nodes, const spx_metrics_objects* metrics, const spx_assigned_objects* assigned, const spx_tlp_params* tlp, int64_t* cap_cpu_milli, double* tlp_cpu_util, int64_t* tlp_missing_milli, uint8_t* tlp_valid, int64_t* lv_alloc_cpu_milli, int64_t* lv_alloc_mem, double* lv_cpu_avg, double* lv_cpu_std, double* lv_mem_avg, double* lv_mem_std, uint8_t* lv_flags);
int spx_flatten_trimaran_pods(const spx_pod_objects* pods, const spx_tlp_params* tlp, int64_t* tlp_pod_milli, int64_t* lv_req_cpu_milli, int64_t* lv_req_mem);
/* the same columns for the listed nodes only (n_rows rows; row j = node idx[j]): the input of spx_update_trimaran_nodes */
int spx_flatten_trimaran_node_rows(const spx_node_objects* nodes, const spx_metrics_objects* metrics, const spx_assigned_objects* assigned, const spx_tlp_params* tlp, const int64_t* idx, int64_t n_rows, int64_t* cap_cpu_milli, double* tlp_cpu_util, int64_t* tlp_missing_milli, uint8_t* tlp_valid, int64_t* lv_alloc_cpu_milli, int64_t* lv_alloc_mem, double* lv_cpu_avg, double* lv_cpu_std, double* lv_mem_avg, double* lv_mem_std, uint8_t* lv_flags);
/* LowRiskOverCommitment: every output array has one entry per node / per pod */
int spx_flatten_lroc_nodes(const spx_node_objects* nodes, const spx_node_pods_objects* node_pods, int64_t* req_cpu_milli, int64_t* req_mem, int64_t* lim_cpu_milli, int64_t* lim_mem);
int spx_flatten_lroc_pods(const spx_pod_objects* pods, int64_t* req_cpu_milli, int64_t* req_mem, int64_t* lim_cpu_milli, int64_t* lim_mem);
/* Peaks: node arrays sized [N], pod array [P] */
int spx_flatten_peaks_nodes(const spx_node_objects* nodes, const spx_metrics_objects* metrics, const spx_power_model_objects* models, int64_t* cap_cpu_milli, double* cpu_util, uint8_t* valid, double* k1, double* k2);
int spx_flatten_peaks_pods(const spx_pod_objects* pods, int64_t* cpu_milli);

/* NRT: builds the dense slot numbering from every resource id pods request or zones report
 * (slot_res/slot_flags/slot_weight sized SPX_NRT_MAX_RES; *n_res_out receives the count) */
int spx_flatten_nrt_slots(const spx_pod_objects* pods, const spx_nrt_objects* nrt, const spx_resource_classes* rc, const spx_nrt_params* p, int32_t* n_res_out, int32_t* slot_res, uint8_t* slot_flags, int64_t* slot_weight);
/* node arrays sized: flags[N], max_numa[N], n_zones[N], zone_id[N*8], zone_present[N*8], zone_avail[N*8*n_res], zone_cost[N*8*8], min_avg_dist[N*8], node_present[N] */
int spx_flatten_nrt_nodes(const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_nrt_slots* slots, uint8_t* flags, int32_t* max_numa, uint8_t* n_zones, uint8_t* zone_id, uint8_t* zone_present, int64_t* zone_avail, int32_t* zone_cost, float* min_avg_dist, uint8_t* node_present);
/* the same columns for the listed nodes only (n_rows rows; row j = node idx[j]): the input of spx_update_nrt_nodes */
int spx_flatten_nrt_node_rows(const spx_node_objects* nodes, const spx_nrt_objects* nrt, const spx_nrt_slots* slots, const int64_t* idx, int64_t n_rows, uint8_t* flags, int32_t* max_numa, uint8_t* n_zones, uint8_t* zone_id, uint8_t* zone_present, int64_t* zone_avail, int32_t* zone_cost, float* min_avg_dist, uint8_t* node_present);
/* pod arrays sized: qos[P], non_native[P], n_ctr[P], ctr_kind[P*8], ctr_present[P*8], ctr_req[P*8*n_res], pod_present[P], pod_req[P*n_res] */
int spx_flatten_nrt_pods(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_nrt_slots* slots, uint8_t* qos, uint8_t* non_native, uint8_t* n_ctr, uint8_t* ctr_kind, uint8_t* ctr_present, int64_t* ctr_req, uint8_t* pod_present, int64_t* pod_req);

/* NetworkOverhead / TopologicalSort.  region_cost[n_regions*n_regions] and zone_cost[n_zones*n_zones]: -1 = no entry.
 * spx_flatten_net_keys sizes: *n_keys_out and *n_pairs_out first (pass NULL arrays), then fill
 * pod_key[P], topo_order[P], key_score_equally[n_keys], pair_ptr[n_keys+1], pair_node/pair_max_cost[n_pairs]. */
int spx_flatten_net_topo(const spx_nettopo_objects* nt, int32_t* region_cost, int32_t* zone_cost);
int spx_flatten_net_keys(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int32_t* n_keys_out, int64_t* n_pairs_out, int32_t* pod_key, int32_t* topo_order, uint8_t* key_score_equally, int32_t* pair_ptr, int32_t* pair_node, int64_t* pair_max_cost);
/* TopologicalSort.Less (topologicalsort.go:102-132) for n pairs of pod indices, from the flattened keys */
int spx_toposort_less(const spx_pod_objects* pods, const int32_t* topo_order, int64_t n_pairs, const int64_t* a, const int64_t* b, uint8_t* less_out);
/* spx_net_commit_soa columns: *n_entries_out first (NULL arrays), then eff_ptr[P+1], eff_key / eff_max_cost[n_entries] */
int spx_flatten_net_commit(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int64_t* n_entries_out, int32_t* eff_ptr, int32_t* eff_key, int64_t* eff_max_cost);
/* pods that joined AppGroup scheduled lists since the flatten: the entries spx_update_net_placed appends (sizes first: NULL arrays) */
int spx_flatten_net_placed(const spx_pod_objects* pods, const spx_appgroup_objects* ag, int64_t n_placed, const int32_t* group, const int32_t* selector, const int32_t* node, int64_t* n_entries_out, int32_t* key_out, int32_t* node_out, int64_t* cost_out);

/* NRT preemption flow: preemption.GetNRTPostPodsEviction (pkg/noderesourcetopology/preemption/preemption.go:39-157)
 * for node `node`.  The Filter of a preemption dry-run (filter.go:205-220) is the ordinary Filter on the zone table this call
 * produces, so a batch of candidate nodes is evaluated by uploading their post-eviction availabilities (spx_upload_nrt_nodes).
 * `victims` are the victim pods; victim_qos[v] = v1qos.GetPodQOS; ctr_numa has one entry per container row of `victims`
 * (CSR order): numaplacement.EncodedInfo.NUMAAffinity -> NUMA id >= 0, -1 = no affinity, SPX_EVICT_CTR_UNKNOWN = lookup error.
 * placement_present = 0 stands for a nil EncodedInfo; placement_containers = EncodedInfo.Containers().
 * zres_avail_out receives the node's zone-resource availabilities (its slice of nrt->zres_avail, same order) after the
 * simulation, or unchanged when *code_out != SPX_EVICT_OK — the reference hands the original NRT back on every error. */
#define SPX_EVICT_OK 0
#define SPX_EVICT_NO_NRT 1                /* "NRT not found, cannot process eviction simulation"                    :41 */
#define SPX_EVICT_NO_VICTIMS 2            /* "no victims found, cannot process eviction simulation"                  :45 */
#define SPX_EVICT_NO_PLACEMENT 3          /* "numa placement info not found, cannot process eviction simulation"    :49 */
#define SPX_EVICT_NO_CONTAINERS 4         /* "no containers found in numa placement info, cannot process ..."       :53 */
#define SPX_EVICT_NOTHING_TO_ADD 5        /* "no resources to add, cannot process eviction simulation"               :62 */
#define SPX_EVICT_EXCEEDS_ALLOCATABLE 6   /* "resource release request exceeds NUMA allocatable"                     :147 */
#define SPX_EVICT_CTR_UNKNOWN (-2)
int spx_nrt_post_eviction(const spx_nrt_objects* nrt, const spx_resource_classes* rc, int64_t node, const spx_pod_objects* victims, const uint8_t* victim_qos, const int32_t* ctr_numa, int32_t placement_present, int32_t placement_containers, int64_t* zres_avail_out, int32_t* code_out);

/* CapacityScheduling: sizes are pod_req[P*8], pod_req_present[P]; the per-namespace arrays [NS*8] / [NS];
 * agg_*[8] / [1]; other_nominated[NS*8]: nominated requests of OTHER namespaces whose quota is not over min
 * (capacity_scheduling.go:248-250), i.e. total minus the namespace's own share; nom_* are the nominated pods
 * grouped by namespace (nom_ptr[NS+1]); n_nominated entries. */
int spx_flatten_quota(const spx_pod_objects* pods, const spx_resource_classes* rc, const spx_quota_objects* q, int32_t* pod_ns, int32_t* pod_priority, int64_t* pod_req, uint8_t* pod_req_present, int64_t* agg_used, uint8_t* agg_used_present, int64_t* agg_min, uint8_t* agg_min_present, int64_t* other_nominated, uint8_t* other_nominated_present, int32_t* nom_ptr, int32_t* nom_priority, int64_t* nom_pending_index, int64_t* nom_req, uint8_t* nom_req_present);

/* ------------------------------------------------------------------ wire format -> object tables (first slice)
 *
 * NodeResourceTopology objects as the API server serves them (JSON of topology.node.k8s.io/v1alpha2; schema = the reference's
 * manifests/crds/topology.node.k8s.io_noderesourcetopologies.yaml) decoded into the spx_nrt_objects columns, replacing the
 * informer-cache walk of pluginhelpers.go:105-161 / nodeconfig/topologymanager.go:78-162 on the Go side.  The handle owns the
 * tables; pointers returned by the accessors stay valid until the next spx_ingest_nrt_json / spx_ingest_destroy.  node_names fixes
 * the node order of the snapshot (objects of other names are counted and skipped); resource_names pre-seeds the resource
 * interner (ids >= SPX_RES_FIRST_DYNAMIC in that order) so that pod tables built elsewhere share the id space.  `fresh` is 1 for
 * every node and the assumed-pod lists are empty: both are the NRT cache's verdict, which stays with the caller. */
typedef struct spx_ingest spx_ingest;
int spx_ingest_create(const char* const* node_names, int64_t n_nodes, const char* const* resource_names, int32_t n_resource_names, spx_ingest** out);
int spx_ingest_destroy(spx_ingest* h);
const char* spx_ingest_error(const spx_ingest* h);
/* json: one object, a List ("items") or an array; may be called repeatedly — a later object replaces an earlier one of the same name */
int spx_ingest_nrt_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out, int64_t* n_unknown_out);
const spx_nrt_objects* spx_ingest_nrt_objects(const spx_ingest* h);
const spx_resource_classes* spx_ingest_resource_classes(const spx_ingest* h);
int32_t spx_ingest_resource_id(const spx_ingest* h, const char* name);
/* v1.Node objects -> spx_node_objects in the handle's node order (by metadata.name): status.allocatable / capacity in canonical
 * units, the scalar resources framework.Resource.Add keeps, topology.kubernetes.io/region and /zone label values interned
 * (-1 when absent or empty).  v1.Pod objects -> spx_pod_objects, appended in document order until spx_ingest_pods_reset: init
 * containers first (restartPolicy Always = SPX_CTR_SIDECAR) then app containers, requests / limits / overhead lists in document
 * order, spec.priority, metadata.creationTimestamp as queue timestamp (microseconds), namespace and the two AppGroup labels
 * (appgroup.diktyo.x-k8s.io, appgroup.diktyo.x-k8s.io.workload) interned, -1 when absent.  Name id spaces (kind: 0 region, 1 zone,
 * 2 namespace, 3 AppGroup, 4 workload selector) grow in first-seen order; spx_ingest_seed_names fixes them beforehand (workload
 * selectors are the exception: see spx_ingest_appgroups_json). */
int spx_ingest_nodes_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out, int64_t* n_unknown_out);
int spx_ingest_pods_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out);
int spx_ingest_pods_reset(spx_ingest* h);
const spx_node_objects* spx_ingest_node_objects(const spx_ingest* h);
const spx_pod_objects* spx_ingest_pod_objects(const spx_ingest* h);
int spx_ingest_seed_names(spx_ingest* h, int32_t kind, const char* const* names, int32_t n);
int32_t spx_ingest_name_id(const spx_ingest* h, int32_t kind, const char* name);
/* AppGroup CRs (appgroup.diktyo.x-k8s.io/v1alpha1) -> spx_appgroup_objects, row g = the group whose name id (kind 3) is g — the id
 * the pod table's appgroup column carries, whatever order CRs and pods arrive in; a later CR of the same name replaces the earlier one: spec.workloads[].workload.selector, their dependencies[].{workload.selector, maxNetworkCost}, status.topologyOrder[] as
 * written.  Workload selector ids always preserve the lexicographic order of the selector strings: whenever a call (AppGroups or pods)
 * leaves the table out of order it is re-sorted and the pod table's selector column renumbered — query ids (spx_ingest_name_id)
 * after the last ingestion call, not before.  The scheduled-pods list is not part of the CR and stays empty.
 * One NetworkTopology CR (networktopology.diktyo.x-k8s.io/v1alpha1) -> spx_nettopo_objects for the weights set weights_name
 * (populateCostMap, networkoverhead.go:448-497); region / zone names share the id spaces of the node table (kinds 0 and 1). */
int spx_ingest_appgroups_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_objects_out);
const spx_appgroup_objects* spx_ingest_appgroup_objects(const spx_ingest* h);
int spx_ingest_nettopo_json(spx_ingest* h, const char* json, int64_t len, const char* weights_name);
const spx_nettopo_objects* spx_ingest_nettopo_objects(const spx_ingest* h);
/* ElasticQuota CRs (scheduling.x-k8s.io/v1alpha1) -> spx_quota_objects indexed by the given namespace list (which also seeds the
 * namespace ids of the pod table): spec.min / spec.max with newElasticQuotaInfo's replacements for nil lists (elasticquota.go:70-76),
 * `used` from status.used; quotas of other namespaces are counted and skipped.  The nominated-pod list (capacity_scheduling.go:231-253 walks
 * PodNominator.NominatedPodsForNode over the snapshot's nodes) is rebuilt from the handle's pod table on every pod / quota call: the
 * pods whose status.nominatedNodeName names a node of the snapshot, nom_pending_index = the pod's own row (the reference skips the pod
 * under evaluation by UID).
 * NULL from the accessor until the first successful call. */
int spx_ingest_quota_json(spx_ingest* h, const char* json, int64_t len, const char* const* namespaces, int32_t n_namespaces, int64_t* n_objects_out, int64_t* n_unknown_out);
const spx_quota_objects* spx_ingest_quota_objects(const spx_ingest* h);
/* The load-watcher response trimaran's Collector polls (collector.go:139-150) -> spx_metrics_objects in the handle's node order.
 * The struct it decodes into (watcher.WatcherMetrics, github.com/paypal/load-watcher v0.2.4) is not vendored in the reference and
 * the reference holds no golden document: field names are the struct's published json tags, decoding rules are encoding/json's
 * (case-insensitive member names, unknown members ignored) — PARITY UNPINNED.  A successful call replaces the whole
 * metrics snapshot; a failed one keeps the previous one, like the Collector.  n_nodes_out counts the entries of
 * data.NodeMetricsMap, n_unknown_out those whose name is not in the handle's node list.  NULL from the accessor until the first
 * successful call. */
int spx_ingest_metrics_json(spx_ingest* h, const char* json, int64_t len, int64_t* n_nodes_out, int64_t* n_unknown_out);
const spx_metrics_objects* spx_ingest_metrics_objects(const spx_ingest* h);
/* resource.Quantity text -> canonical int64: MilliValue() when milli != 0 (cpu), Value() otherwise; both round up */
int spx_ingest_quantity(const char* text, int32_t milli, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* SPX_H */
